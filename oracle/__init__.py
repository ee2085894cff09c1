"""ORACLE — test infrastructure only.

CPU restatements (plain PyTorch fp32 / numpy fp64) of the reference's algorithm for the dynamic-eval path, each function
citing the reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg import
this package, and only as the checker or the timed CPU baseline; the product path (`dynamic-asr-eval_amd/`) never does and
fails loudly when the HIP library is missing.

Pinning (details in DESIGN.md §4):
  pinned by the reference's own code run in the build container (ast extraction, nothing stubbed; tests/golden/make_golden.py,
  tests/golden/make_reference_pins.py -> tests/golden/prepare_chunks.json, reference_pins.{npz,json}):
      prepare_chunks (lcasr/lib.py:128-145); frame_shuffle / add_random_noise / cutout (:81-84,379-417); the four arg->config
      helpers (:102-125,419-428); the stitch statements (:615-629) and the outer stitch (run_seq_eval.py:130-144); the TEDLIUM STM
      text handling (tedlium/run.py:25-51); SoftDTW._euclidean_dist_func (wav2vec2/soft_dtw_cuda.py:319-329);
  pinned by the ops the reference itself calls: torch.nn.CTCLoss + autograd, torch.optim.Adam, clip_grad_norm_, log_softmax,
      layer_norm, conv / linear; the wav2vec2 model is the transformers class the reference loads;
  PARITY UNPINNED (no reference artefact exists, the dependency is un-vendored and absent): SCConformerXL internals, SpecAugment's
      mask rule, MADGRAD, torch_ema, the log-mel constants, the WER normaliser, soft-DTW values (numba code; the reference's only
      check is a CPU<->GPU allclose, re-run between oracle and HIP in tests/test_softdtw_gpu.py), the enc-dec decoder."""
