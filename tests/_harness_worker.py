"""Worker for tests/test_multirank_harness_gpu.py: runs one harness mirror (`full`, `cross` or `concat`) on the small synthetic
dataset with a tiny checkpoint, as a single process (world 1) or as one rank of `python -m torch.distributed.run --nproc-per-node 2`
(DYN_DIST_BACKEND=gloo: both ranks share the box's one GPU; on an 8-GPU node the same code runs one rank per GPU over RCCL).
Usage: _harness_worker.py <full|cross|concat> <checkpoint> <save_path>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamic_asr_eval_amd import lib  # noqa: E402

which, ckpt, save = sys.argv[1:4]
common = ["-c", ckpt, "-seq", "512", "-o", "256", "-ds", "-nv", "-epochs", "1", "-kwargs", "optim_lr=1e-5", "vocab_size=128", "quiet=True",
          "spec_augment_n_freq_masks=0"]      # no random masks: a recording's result must not depend on which rank draws first
if which == "full":
    from dynamic_asr_eval_amd import run_dynamic_eval_full as H
    H.main(lib.apply_args(H.build_parser(), ["-d", "synthetic_small", "-s", save] + common))
elif which == "cross":
    from dynamic_asr_eval_amd import run_cross_dataset_eval as X
    X.main(lib.apply_args(X.build_parser(), ["-d", "synthetic_small", "-d2", "synthetic_small", "-split", "dev", "-s", save] + common))
elif which == "concat":
    from dynamic_asr_eval_amd import run_whole_concat_eval as W
    W.main(lib.apply_args(W.build_parser(), ["-d", "synthetic_small", "-s", save] + common))
else:
    raise SystemExit(f"unknown harness {which}")
import torch.distributed as dist  # noqa: E402
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
