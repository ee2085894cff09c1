#!/bin/bash
# rocprofv3 kernel stats of the wav2vec2 per-utterance loop (scripts/bench_side.py wav2vec2_su).  usage: scripts/profile_wav2vec2.sh r04
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_w2v_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/su" -o p -- python3 "$root/scripts/bench_side.py" wav2vec2_su > "$out/su.json" 2> "$out/su.err" && echo "su done"
cd "$root"
f=$(find "$out/su" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats_wav2vec2_su.csv"
cp "$out/su.json" "$out/${tag}_bench_side_wav2vec2_su_under_rocprof.json"
find "$out" -name "*.csv" -size +1M -not -name "${tag}_*" -delete
find "$out" -name "*.db" -delete; find "$out" -name "*.rocpd" -delete
ls -la "$out"
