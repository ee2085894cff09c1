"""Reference lcasr/run.py:30-164 is the same per-recording flow as lcasr/run_dynamic_eval_full.py (load model, eval_fn per
recording, greedy decode, normalise, WER, -log line, `_{repeat}.pkl` pickle with `elapsed_times`); one harness serves both.
Run:  python -m dynamic_asr_eval_amd.run -d synthetic -epochs 1 -kwargs optim_lr=9e-5"""
from . import lib
from .run_dynamic_eval_full import build_parser, main  # noqa: F401

if __name__ == '__main__':
    main(lib.apply_args(build_parser()))
