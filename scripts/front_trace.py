"""Per-launch durations of the sparse factorisation's kernels from a rocprofv3 --kernel-trace CSV (one factorisation)."""
import csv
import glob
import sys

d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last factorisation: from the last k_sc_gather_blocks on
st = [i for i, r in enumerate(rows) if "k_sc_gather_blocks" in r["Kernel_Name"]]
i0 = st[-1]
t0 = int(rows[i0]["Start_Timestamp"])
tot = {}
for r in rows[i0:]:
    n = r["Kernel_Name"]
    if not any(k in n for k in ("k_sc_", "k_front_", "fillBuffer")):
        if "k_backsub" in n:
            break
        continue
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    short = n.split("(")[0].split("::")[-1]
    tot[short] = tot.get(short, 0.0) + e - s
    print("%9.1f %9.1f %8.1f  %s" % (s, e, e - s, short))
print(tot)
