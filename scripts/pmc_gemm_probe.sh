#!/bin/bash
# SQ counters of the GEMM kernel per tile shape (one rocprofv3 --pmc pass over scripts/probe_gemm2.py): where the waves of the
# 64x64 / 64x128 / 128x128 kernels spend their cycles (waiting on memory / barrier, issue stalls, LDS conflicts, MFMA busy).
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_gemm
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
  -d "$out/run" -o p -- python3 "$root/scripts/probe_gemm2.py" 8192,4096,768,NT,128:128:1 4096,4096,768,NT,64:64:1 6144,4096,768,NT,64:128:1 8192,4096,3072,NT,128:128:1 4096,4096,3072,NT,64:64:1 \
  4096,768,3072,NT 4096,768,768,NT 2048,768,3072,NN 2048,3072,768,NN 768,3072,2048,TN,128:128:1 > "$out/probe.log" 2> "$out/probe.err"
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/run/**/*counter_collection.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:70], r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")))
    d = agg.setdefault(k, collections.defaultdict(float))
    d[r["Counter_Name"]] += float(r["Counter_Value"]); d["_n"] += 1
with open(out + "/summary.txt", "w") as g:
    for k, d in agg.items():
        if "gemm_f32" not in k[0]:
            continue
        n = d["_n"] / max(1, len([c for c in d if c != "_n"]))
        wc = d["SQ_WAVE_CYCLES"] or 1
        line = (f"{k[0]} grid={k[1]} launches={n:.0f}: wait_any {d['SQ_WAIT_ANY'] / wc:.3f}  wait_inst_any {d['SQ_WAIT_INST_ANY'] / wc:.3f}  wait_inst_lds {d['SQ_WAIT_INST_LDS'] / wc:.3f}  "
                f"active_inst {d['SQ_ACTIVE_INST_ANY'] / wc:.3f}  lds_conflict/lds_active {d['SQ_LDS_BANK_CONFLICT'] / (d['SQ_LDS_IDX_ACTIVE'] or 1):.3f}  mfma_busy_cycles/launch {d['SQ_VALU_MFMA_BUSY_CYCLES'] / n:.3e}  wave_cycles/launch {wc / n:.3e}")
        print(line); g.write(line + "\n")
PY
cat "$out/probe.log"
find "$out/run" -name "*.csv" -size +1M -delete; find "$out" -name "*.db" -delete
