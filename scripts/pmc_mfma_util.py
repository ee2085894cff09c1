"""MFMA utilisation of gemm_f32_kernel from a `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.
Usage: python scripts/pmc_mfma_util.py <dir with p_kernel_trace.csv + p_counter_collection.csv> profiles/<tag>_gemm_mfma_util.json"""
import collections, csv, glob, json, sys

d, out = sys.argv[1], sys.argv[2]
kt = {}
for r in csv.DictReader(open(glob.glob(f"{d}/*kernel_trace.csv")[0])):
    kt[r['Dispatch_Id']] = (r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
cc = collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob(f"{d}/*counter_collection.csv")[0])):
    cc[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
n = sat = 0
busy = gui = dur = 0.0
for k, (name, dn) in kt.items():
    c = cc.get(k)
    if ('gemm_f32_kernel' not in name and 'gemm_f32_grouped_kernel' not in name) or not c or 'SQ_VALU_MFMA_BUSY_CYCLES' not in c:
        continue
    if c['SQ_VALU_MFMA_BUSY_CYCLES'] >= 2 ** 32 - 1:   # the counter saturates on multi-millisecond launches
        sat += 1
        continue
    n += 1; busy += c['SQ_VALU_MFMA_BUSY_CYCLES']; gui += c['GRBM_GUI_ACTIVE']; dur += dn
res = {"kernel": "gemm_f32_kernel + gemm_f32_grouped_kernel", "launches": n, "saturated_counters_skipped": sat,
       "mfma_busy_cycles_per_launch": busy / n, "gui_active_cycles_per_launch_per_xcd": gui / 8 / n, "avg_duration_us": dur / n / 1e3,
       "mfma_util": busy / ((gui / 8) * 1024), "mfma_tflops_from_busy_cycles": busy / 64 * 4096 / (dur * 1e-9) / 1e12,
       "clock_ghz": (gui / 8) / dur,
       "method": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over `bench.py --steps 1 --warmup 0 --seconds 300 "
                 "--no_cpu_baseline --side_steps 0 --graphs 0 --chains 1`; SQ_VALU_MFMA_BUSY_CYCLES = 64 cycles per v_mfma_f32_32x32x2_f32 (4096 FLOP) summed over the "
                 "1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; util = busy / (gui / 8 * 1024)"}
json.dump(res, open(out, "w"), indent=1)
print(res)
