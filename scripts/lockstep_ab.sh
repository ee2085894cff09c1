#!/bin/bash
# Lockstep A/B on ONE box (r04): chains vs lockstep groups, before and after tuning the GEMM table for the group's batched shapes.
#   scripts/lockstep_ab.sh out_dir           (run from the repository root on the GPU box)
out=${1:-gpurun_out/lockstep_ab}; mkdir -p "$out"
COMMON="--warmup 0 --prewarm_s 20 --no_cpu_baseline --side_steps 0 --side_workloads 0"
run() {  # name, bench args...
    name=$1; shift
    python bench.py $COMMON "$@" > "$out/$name.json" 2> "$out/$name.err" || { echo "$name FAILED"; tail -n 5 "$out/$name.err"; return; }
    python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["value"], "ms/step", d["ms_per_step"], "excl", d["roofline"]["achieved"], "job", d["roofline"]["job_gemm_tflops"], flush=True)
PY
}
python bench.py --steps 3 --prewarm_s 30 $COMMON > "$out/warm.json" 2>/dev/null
run chains3 --steps 6 --chains 3
run lock3x1 --steps 6 --chains 1 --lockstep 3
run lock3x2 --steps 6 --chains 2 --lockstep 3
run lock2x3 --steps 6 --chains 3 --lockstep 2
if [ "$2" = "tune" ]; then
    timeout -k 10 400 python scripts/tune_gemm.py "$out/gemm_tuned_r3.inc" 3 > "$out/tune_r3.log" 2>&1 || { echo "tuner failed"; tail -n 5 "$out/tune_r3.log"; exit 0; }
    cp "$out/gemm_tuned_r3.inc" dynamic-asr-eval_amd/csrc/gemm_tuned.inc
    make -C dynamic-asr-eval_amd/csrc -j8 > "$out/make.log" 2>&1 || { echo "make failed"; tail -n 5 "$out/make.log"; exit 0; }
    run chains3_t --steps 6 --chains 3
    run lock3x1_t --steps 6 --chains 1 --lockstep 3
    run lock3x2_t --steps 6 --chains 2 --lockstep 3
fi
