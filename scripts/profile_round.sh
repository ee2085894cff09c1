#!/bin/bash
# rocprofv3 passes of one round (run on the GPU box): kernel stats with 1 chain x 1 recording (the form every round has used: comparable) and
# with the bench's default (r04: 2 chains x lockstep groups of 4), then the PMC passes (each in its own run, kernel trace only) on a 300 s
# recording, eager, 1 chain x 1 recording.  usage: scripts/profile_round.sh r04
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$root/bench.py"
common="--warmup 0 --no_cpu_baseline --side_steps 0 --side_workloads 0"
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/c1" -o p -- python3 "$B" --steps 2 --prewarm_s 10 --chains 1 --lockstep 1 $common > "$out/c1.json" 2> "$out/c1.err" && echo "c1 done" \
&& rocprofv3 --output-format csv --kernel-trace --stats -d "$out/c3" -o p -- python3 "$B" --steps 8 --prewarm_s 10 $common > "$out/c3.json" 2> "$out/c3.err" && echo "c3 done" \
&& rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$out/fetch" -o p -- python3 "$B" --steps 1 --seconds 300 --prewarm_s 0 --graphs 0 --chains 1 --lockstep 1 $common > "$out/fetch.json" 2> "$out/fetch.err" && echo "fetch done" \
&& rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$out/write" -o p -- python3 "$B" --steps 1 --seconds 300 --prewarm_s 0 --graphs 0 --chains 1 --lockstep 1 $common > "$out/write.json" 2> "$out/write.err" && echo "write done" \
&& rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$out/mfma" -o p -- python3 "$B" --steps 1 --seconds 300 --prewarm_s 0 --graphs 0 --chains 1 --lockstep 1 $common > "$out/mfma.json" 2> "$out/mfma.err" && echo "mfma done"
cd "$root"
mkdir -p "$out/summary"
f=$(find "$out/c1" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/summary/${tag}_kernel_stats_chains1.csv"
f=$(find "$out/c3" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/summary/${tag}_kernel_stats_default.csv"
python3 scripts/trace_overlap.py "$out/c3" "$out/summary/${tag}_trace_overlap_default.json" > "$out/summary/overlap.log" 2>&1
python3 scripts/pmc_summary.py "$out/fetch" "$out/write" "$out/summary/${tag}_gemm_traffic.json" > "$out/summary/pmc_summary.log" 2>&1
python3 scripts/pmc_mfma_util.py "$(dirname $(find $out/mfma -name '*kernel_trace.csv' | head -1))" "$out/summary/${tag}_gemm_mfma_util.json" > "$out/summary/mfma.log" 2>&1
python3 scripts/pmc_hbm_kernels.py "$out/fetch" "$out/write" "$out/summary/${tag}_hbm_kernels.json" > "$out/summary/hbm.log" 2>&1
cp "$out/c1.json" "$out/summary/${tag}_bench_line_chains1_under_rocprof.json"; cp "$out/c3.json" "$out/summary/${tag}_bench_line_default_under_rocprof.json"
# keep the merge small: raw traces stay on the box
find "$out" -name "*.csv" -size +1M -not -path "*/summary/*" -delete
find "$out" -name "*.db" -delete; find "$out" -name "*.rocpd" -delete
ls -la "$out/summary"
