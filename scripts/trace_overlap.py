"""From a rocprofv3 --kernel-trace csv of the bench command: how much of the wall time has a GEMM-family kernel resident, how much has any
kernel resident, and the average number of kernels in flight (3 chains).  usage: python scripts/trace_overlap.py <dir> out.json"""
import csv, glob, json, sys

d, out = sys.argv[1], sys.argv[2]
f = (glob.glob(f"{d}/*kernel_trace.csv") + glob.glob(f"{d}/*/*kernel_trace.csv"))[0]
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "gemm_f32" in r["Kernel_Name"] or "attention_" in r["Kernel_Name"]))
ev.sort()
# analyse the last 60 % of the trace (steady state of the timed region)
t_lo = ev[0][0] + 0.4 * (ev[-1][1] - ev[0][0])
ev = [e for e in ev if e[0] >= t_lo]


def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


wall = max(e[1] for e in ev) - min(e[0] for e in ev)
any_busy = union([(s, e) for s, e, _ in ev])
mm_busy = union([(s, e) for s, e, g in ev if g])
ksum = sum(e - s for s, e, _ in ev)
msum = sum(e - s for s, e, g in ev if g)
res = {"wall_s": wall / 1e9, "any_kernel_resident_frac": any_busy / wall, "matrix_kernel_resident_frac": mm_busy / wall,
       "avg_kernels_in_flight": ksum / wall, "avg_matrix_kernels_in_flight": msum / wall, "kernels": len(ev),
       "what": "steady-state part (last 60 %) of a kernel trace of bench.py with 3 chains; matrix kernel = gemm_f32* / attention_*"}
json.dump(res, open(out, "w"), indent=1)
print(res)
