"""Idle time of the device over a stretch of a kernel trace (between two launches of k_point_elim near the end of the run):
gaps between consecutive kernels on any queue.  10M network under bench.py: 0.6 ms idle in 61 ms, the largest gap 63 us."""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_point_elim" in r["Kernel_Name"]]
i0, i1 = st[-3], st[-2]
t0 = int(rows[i0]["Start_Timestamp"])
busy_end = t0
gaps = []
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        gaps.append(((s - busy_end) / 1e3, (busy_end - t0) / 1e3, r["Kernel_Name"][:50]))
    busy_end = max(busy_end, e)
total = (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3
print("stretch %.1f us, idle %.1f us in %d gaps" % (total, sum(g[0] for g in gaps), len(gaps)))
for g in sorted(gaps, reverse=True)[:15]:
    print("  gap %7.1f us at %8.1f before %s" % g)
