import sys; sys.path.insert(0, "/root/repo")
import torch
from dynamic_asr_eval_amd import ops
dev = torch.device("cuda:0")
T, C = 2048, 4096
lp = torch.log_softmax(torch.randn(1, T, C, device=dev), -1)
for S in (1, 100, 450, 800, 1000):
    tg = torch.randint(0, C - 1, (1, S), device=dev, dtype=torch.int32)
    il = torch.full((1,), T, dtype=torch.int32, device=dev); tl = torch.full((1,), S, dtype=torch.int32, device=dev)
    for _ in range(3): ops.ctc_loss(lp, tg, il, tl, C - 1, reduction="sum", grad_scale=1.0 / T)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.ctc_loss(lp, tg, il, tl, C - 1, reduction="sum", grad_scale=1.0 / T)
    e1.record(); torch.cuda.synchronize()
    print(f"S={S:5d}: ctc_loss (loss + grad) {e0.elapsed_time(e1)/10:.3f} ms", flush=True)
