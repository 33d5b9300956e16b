"""Turns the rocprofv3 output of one round (kernel statistics of `bench.py`, and the FETCH_SIZE and WRITE_SIZE passes of
`scripts/pmc_workload.py cfg4 1.0 2`) into the files kept under profiles/: the kernel-statistics table, the per-kernel
traffic table, and pmc_traffic.json, which bench.py reads for roofline.traffic.

    python scripts/summarise_profiles.py r01e gpurun_out/prof_r01e gpurun_out/pmc_fetch gpurun_out/pmc_write
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d_stats, d_fetch, d_write = sys.argv[1:5]


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    assert f, pattern
    return max(f, key=os.path.getmtime)  # gpurun merges the runs of a round into one directory: take the latest


shutil.copy(one(os.path.join(d_stats, "**", "*kernel_stats.csv")), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_bench_cfg4.csv"))


def means(d, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(one(os.path.join(d, "**", "*counter_collection.csv")))):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


fetch, write = means(d_fetch, "FETCH_SIZE"), means(d_write, "WRITE_SIZE")
rows = sorted(((k, n, f, write.get(k, (0, 0.0))[1]) for k, (n, f) in fetch.items()), key=lambda r: -(r[2] + r[3]))
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_write_cfg4.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches", "mean_FETCH_SIZE_KB", "mean_WRITE_SIZE_KB"])
    for k, n, a, b in rows[:30]:
        w.writerow([k, n, round(a, 1), round(b, 1)])


def find(prefix):
    hits = [r for r in rows if r[0].replace("void ", "").startswith("ba::" + prefix)]
    assert hits, prefix
    return max(hits, key=lambda r: r[1])  # the variant the benchmark network uses is the one dispatched most


kernels = {}
for name, doubled in (("k_bsr_spmv", True), ("k_schur_blocks", False), ("k_point_elim", False), ("k_obs_rows", False)):
    k, n, fkb, wkb = find(name)
    kernels[name] = {"fetch_kb_raw": fkb, "write_kb": wkb, "fetch_doubled": doubled,
                     "bytes_per_launch": int(round(((2.0 if doubled else 1.0) * fkb + wkb) * 1024))}
kernels["linearisation (k_point_elim + k_obs_rows)"] = {
    "fetch_kb_raw": kernels["k_point_elim"]["fetch_kb_raw"] + kernels["k_obs_rows"]["fetch_kb_raw"],
    "write_kb": kernels["k_point_elim"]["write_kb"] + kernels["k_obs_rows"]["write_kb"], "fetch_doubled": False,
    "bytes_per_launch": kernels["k_point_elim"]["bytes_per_launch"] + kernels["k_obs_rows"]["bytes_per_launch"],
    "note": "k_point_elim + k_obs_rows"}
out = {"config": "cfg4",
       "source": f"profiles/{tag}_pmc_fetch_write_cfg4.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, scripts/pmc_workload.py cfg4 1.0 2)",
       "correction": "FETCH_SIZE doubled for kernels whose dominant reads are 16 B per lane (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported; KB = 1024 B; Infinity-Cache hits are included in FETCH_SIZE",
       "kernels": kernels}
with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
for k, v in kernels.items():
    print("%-45s %8.1f MB per launch" % (k, v["bytes_per_launch"] / 1e6))
