"""Timeline of one linearisation + solve from a rocprofv3 --kernel-trace run: start, end and duration (us) of every kernel
longer than a threshold between two launches of k_point_elim, with the queue it ran on.

    python scripts/timeline.py gpurun_out/prof_dir [which_linearisation=-1] [min_us=20]"""
import csv
import glob
import sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_point_elim" in r["Kernel_Name"]]
i0 = st[which]
i1 = st[which + 1] if which + 1 < len(st) and which != -1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
n_pcg = 0
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"]
    if any(k in name for k in ("k_bsr_spmv", "k_amg_update", "k_amg_restrict", "k_amg_prolong")):
        n_pcg += 1
        continue
    if e - s >= min_us:
        print("%9.1f %9.1f %8.1f  q%s %s" % (s, e, e - s, r.get("Queue_Id", "?"), name[:60]))
print("PCG kernels skipped:", n_pcg)
