"""ORACLE — test infrastructure only.

CPU restatements (plain PyTorch fp32 / numpy fp64) of the reference's algorithm for the dynamic-eval path, each function
citing the reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg import
this package, and only as the checker or the timed CPU baseline; the product path (`dynamic-asr-eval_amd/`) never does and
fails loudly when the HIP library is missing.

Pinning (details in DESIGN.md §4): `prepare_chunks` is pinned by the reference's own function run in the build container;
CTC, Adam, gradient clipping, log-softmax, layer norm, convolutions are the torch CPU ops the reference itself calls; the
wav2vec2 model is the transformers class the reference loads.  PARITY UNPINNED (no reference artefact exists, the dependency is
un-vendored): SCConformerXL internals, SpecAugment's mask rule, MADGRAD, torch_ema, the log-mel constants, the WER normaliser."""
