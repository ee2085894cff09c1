"""One-rank RCCL worker for tests/test_dist_gpu.py: the default `nccl` backend (= RCCL on ROCm) on the box's one GPU, every
collective helper of dynamic_asr_eval_amd/dist.py executed through it (device tensors, barrier(device_ids=...))."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from dynamic_asr_eval_amd import dist as ddist  # noqa: E402

rank, local_rank, world = ddist.init(force=True)            # backend chosen by dist.init: nccl when a GPU is visible
assert dist.is_initialized() and dist.get_backend() == "nccl", dist.get_backend()
counts = ddist.all_reduce_counts((3, 1, 4, 159))
records = ddist.gather_records([{"index": 1, "id": "b"}, {"index": 0, "id": "a"}])
mx = ddist.max_over_ranks(2.5)
ddist.barrier()
# a payload of the size the whole-concat harness broadcasts (adapted weights, ~0.36 GB) would be `dist.broadcast`; here a small one
w = torch.arange(1 << 16, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
dist.broadcast(w, src=0)
dist.all_reduce(w)
torch.cuda.synchronize()
ok = bool((w == torch.arange(1 << 16, dtype=torch.float32, device=w.device)).all())
ddist.shutdown()
print("RESULT " + json.dumps({"backend": "nccl", "world": world, "counts": list(counts), "ids": [r["id"] for r in records], "max": mx,
                              "tensor_ok": ok, "rccl_version": list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None}), flush=True)
