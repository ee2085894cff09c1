"""HBM-side traffic and achieved bandwidth per kernel from three rocprofv3 passes of the same command: `--kernel-trace --pmc FETCH_SIZE`,
`--kernel-trace --pmc WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes) — durations come from the kernel trace of the
FETCH pass.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count correction).
Usage: python scripts/pmc_hbm_kernels.py <fetch_dir> <write_dir> profiles/<tag>_hbm_kernels.json"""
import collections, csv, glob, json, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def counters(d, counter):
    f = (glob.glob(f"{d}/*counter_collection.csv") + glob.glob(f"{d}/*/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def durations(d):
    f = (glob.glob(f"{d}/*kernel_trace.csv") + glob.glob(f"{d}/*/*kernel_trace.csv"))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg


fd, wd, out = sys.argv[1:4]
fe, wr, du = counters(fd, "FETCH_SIZE"), counters(wd, "WRITE_SIZE"), durations(fd)
rows = []
for k, (n, ns) in du.items():
    if k not in fe or k not in wr or n == 0:
        continue
    b = (2 * fe[k][1] / fe[k][0] + wr[k][1] / wr[k][0]) * 1024
    us = ns / n / 1e3
    rows.append({"kernel": k, "launches": n, "avg_us": round(us, 2), "hbm_mb_per_launch": round(b / 1e6, 2), "tb_per_s": round(b / (us * 1e-6) / 1e12, 3),
                 "share_of_kernel_time": ns})
tot = sum(r["share_of_kernel_time"] for r in rows)
for r in rows:
    r["share_of_kernel_time"] = round(r["share_of_kernel_time"] / tot, 4)
rows.sort(key=lambda r: -r["share_of_kernel_time"])
json.dump({"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 0 --seconds 300 "
                     "--prewarm_s 0 --no_cpu_baseline --side_steps 0 --graphs 0 --chains 1`; durations from the FETCH pass under the profiler "
                     "(counter collection serialises launches and lengthens short kernels; the --stats csv of the same round has the undisturbed durations)",
           "kernels": rows[:40]}, open(out, "w"), indent=1)
for r in rows[:25]:
    print(r)
