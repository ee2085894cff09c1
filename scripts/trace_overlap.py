"""From a rocprofv3 --kernel-trace csv of the bench command: how much of the wall time has a GEMM-family kernel resident, how much has any
kernel resident, and the average number of kernels in flight (3 chains).  usage: python scripts/trace_overlap.py <dir> out.json"""
import csv, glob, json, sys

d, out = sys.argv[1], sys.argv[2]
f = (glob.glob(f"{d}/*kernel_trace.csv") + glob.glob(f"{d}/*/*kernel_trace.csv"))[0]
ev, qid = [], []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "gemm_f32" in r["Kernel_Name"] or "attention_" in r["Kernel_Name"], r["Kernel_Name"]))
    qid.append(r.get("Stream_Id") or r.get("Queue_Id") or "0")
order = sorted(range(len(ev)), key=lambda i: ev[i])
ev, qid = [ev[i] for i in order], [qid[i] for i in order]
# analyse the last 60 % of the trace (steady state of the timed region)
t_lo = ev[0][0] + 0.4 * (ev[-1][1] - ev[0][0])
qid = [q for e, q in zip(ev, qid) if e[0] >= t_lo]
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
any_busy = union([(s, e) for s, e, _, _ in ev])
mm_busy = union([(s, e) for s, e, g, _ in ev if g])
ksum = sum(e - s for s, e, _, _ in ev)
msum = sum(e - s for s, e, g, _ in ev if g)
res = {"wall_s": wall / 1e9, "any_kernel_resident_frac": any_busy / wall, "matrix_kernel_resident_frac": mm_busy / wall,
       "avg_kernels_in_flight": ksum / wall, "avg_matrix_kernels_in_flight": msum / wall, "kernels": len(ev),
       "what": "steady-state part (last 60 %) of a kernel trace of bench.py with 3 chains; matrix kernel = gemm_f32* / attention_*"}


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("dyn::", "")
    return n.split("(")[0].split("<")[0]


# What is resident while NO matrix kernel is: sweep over the boundaries of all kernels; in every elementary interval without a matrix
# kernel, credit the interval to each non-matrix kernel resident in it (shares can add up to more than the uncovered time).
pts = []
for s_, e_, g, n in ev:
    pts.append((s_, 1, g, n))
    pts.append((e_, -1, g, n))
pts.sort(key=lambda x: (x[0], x[1]))
n_mm, live, share, last = 0, {}, {}, pts[0][0]
alone = {}
for t, dlt, g, n in pts:
    if t > last and n_mm == 0:
        names = [k for k, v in live.items() if v > 0]
        for k in names:
            share[k] = share.get(k, 0) + (t - last)
        if len(names) == 1 and sum(live.values()) == 1:
            alone[names[0]] = alone.get(names[0], 0) + (t - last)
    last = t
    if g:
        n_mm += dlt
    else:
        k = short(n)
        live[k] = live.get(k, 0) + dlt
res["no_matrix_kernel_frac"] = 1.0 - mm_busy / wall
res["resident_while_no_matrix_kernel_frac_of_wall"] = {k: round(v / wall, 4) for k, v in sorted(share.items(), key=lambda kv: -kv[1])[:14]}
res["alone_on_the_chip_frac_of_wall"] = {k: round(v / wall, 4) for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:10]}
# Per stream (chain): how much of the wall time it has a kernel resident, and where the rest goes: gaps between consecutive kernels of the
# stream shorter than 20 us (dispatch gaps inside a graph replay), 20 - 500 us, and longer (the chain waits: host round trip, or its kernel is
# queued behind the other chains' workgroups)
per = {}
for (s_, e_, _, _), q in zip(ev, qid):
    per.setdefault(q, []).append((s_, e_))
streams = {}
for q, iv in per.items():
    if len(iv) < 1000:
        continue
    iv.sort()
    busy = union(iv)
    g_short = g_mid = g_long = 0
    n_long = 0
    end = iv[0][1]
    for s_, e_ in iv[1:]:
        g = s_ - end
        if g > 0:
            if g < 20000:
                g_short += g
            elif g < 500000:
                g_mid += g
            else:
                g_long += g
                n_long += 1
        end = max(end, e_)
    streams[str(q)] = {"kernels": len(iv), "resident_frac": round(busy / wall, 4), "gaps_lt_20us_frac": round(g_short / wall, 4),
                       "gaps_20_500us_frac": round(g_mid / wall, 4), "gaps_gt_500us_frac": round(g_long / wall, 4), "gaps_gt_500us": n_long}
res["per_stream"] = streams
json.dump(res, open(out, "w"), indent=1)
print(res)
