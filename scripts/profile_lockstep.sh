#!/bin/bash
# rocprofv3 kernel stats of the bench in lockstep mode (run on the GPU box): usage scripts/profile_lockstep.sh TAG R CHAINS STEPS
tag=${1:-r04}; R=${2:-3}; CH=${3:-1}; ST=${4:-3}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
common="--warmup 0 --no_cpu_baseline --side_steps 0 --side_workloads 0"
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/l${R}c${CH}" -o p -- python3 "$root/bench.py" --steps $ST --prewarm_s 10 --chains $CH --lockstep $R $common > "$out/l${R}c${CH}.json" 2> "$out/l${R}c${CH}.err" && echo "profile done"
cd "$root"
mkdir -p "$out/summary"
f=$(find "$out/l${R}c${CH}" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/summary/${tag}_kernel_stats_lockstep${R}_chains${CH}.csv"
cp "$out/l${R}c${CH}.json" "$out/summary/${tag}_bench_line_lockstep${R}_chains${CH}_under_rocprof.json"
find "$out" -name "*.csv" -size +1M -not -path "*/summary/*" -delete
find "$out" -name "*.db" -delete; find "$out" -name "*.rocpd" -delete
ls -la "$out/summary"
