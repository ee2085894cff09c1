#!/bin/bash
# Second lockstep A/B (r04), driver-form runs on ONE box: which (chains, lockstep) becomes bench.py's default.
out=${1:-gpurun_out/lockstep_ab2}; mkdir -p "$out"
COMMON="--no_cpu_baseline --side_steps 0 --side_workloads 0"
run() {
    name=$1; shift
    python bench.py $COMMON "$@" > "$out/$name.json" 2> "$out/$name.err" || { echo "$name FAILED"; tail -n 5 "$out/$name.err"; return; }
    python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["value"], "ms/step", d["ms_per_step"], "chains", d["config"]["chains_per_gpu"], "excl", d["roofline"]["achieved"], flush=True)
PY
}
python bench.py --steps 3 --warmup 0 --prewarm_s 30 $COMMON > "$out/warm.json" 2>/dev/null
if [ "$2" = "tune2" ]; then
    timeout -k 10 400 python scripts/tune_gemm.py "$out/gemm_tuned_r2.inc" 2 > "$out/tune_r2.log" 2>&1 && cp "$out/gemm_tuned_r2.inc" dynamic-asr-eval_amd/csrc/gemm_tuned.inc \
        && make -C dynamic-asr-eval_amd/csrc -j8 > "$out/make.log" 2>&1 || { echo "tune2 failed"; tail -n 5 "$out/tune_r2.log"; }
fi
run d_chains3 --steps 20 --warmup 5 --prewarm_s 20
run d_lock2x3 --steps 20 --warmup 5 --prewarm_s 20 --lockstep 2 --chains 3
run d_lock3x2 --steps 20 --warmup 5 --prewarm_s 20 --lockstep 3 --chains 2
run d_lock2x4 --steps 20 --warmup 5 --prewarm_s 20 --lockstep 2 --chains 4
run d_lock4x2 --steps 20 --warmup 5 --prewarm_s 20 --lockstep 4 --chains 2
