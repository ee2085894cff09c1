#!/bin/bash
# A/B on ONE box: alternating bench.py runs under different environment switches, after a throw-away warm-up run.
# usage: scripts/ab_bench.sh out_dir "NAME1:ENV=..,ENV=.." "NAME2:..." ...   (ROUNDS=2 by default)
out=$1; shift
mkdir -p "$out"
ROUNDS=${ROUNDS:-2}
ARGS=${ARGS:---steps 3 --warmup 0 --prewarm_s 20 --no_cpu_baseline --side_steps 0 --side_workloads 0}
python bench.py --steps 3 --warmup 0 --prewarm_s 30 --no_cpu_baseline --side_steps 0 --side_workloads 0 > "$out/warm.json" 2>/dev/null
for r in $(seq 1 $ROUNDS); do
  for cfg in "$@"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    # BENCH_EXTRA=--chains@6@--steps@6 in a config adds bench.py arguments for that config (@ = space)
    ( for e in ${envs//,/ }; do export "$e"; done; python bench.py $ARGS ${BENCH_EXTRA//@/ } > "$out/${name}_$r.json" 2> "$out/${name}_$r.err" )
    python - "$out/${name}_$r.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["value"], "excl", d["roofline"]["achieved"], "job", d["roofline"]["job_gemm_tflops"], flush=True)
PY
  done
done
