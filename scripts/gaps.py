"""Idle gaps of the device inside one LM iteration, from a rocprofv3 --kernel-trace run: the window between two launches of
k_point_elim is cut where no kernel runs for longer than a threshold; prints each gap with the kernels either side.

    python scripts/gaps.py gpurun_out/prof_dir [which_linearisation=-3; 0 = the window with the most kernels] [min_gap_us=3]

Kernels of the PCG iteration are not listed one by one: their gaps are summed."""
import csv
import glob
import sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "k_point_elim" in r["Kernel_Name"]]
if which == 0:
    which = max(range(len(st) - 1), key=lambda k: st[k + 1] - st[k])
i0, i1 = st[which], st[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
busy_end, total_gap, n, pcg_gap, n_pcg = 0.0, 0.0, 0, 0.0, 0
prev = ""
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    pcg = any(k in r["Kernel_Name"] for k in ("k_bsr_spmv", "k_amg_update", "k_amg_restrict", "k_amg_prolong")) and any(
        k in prev for k in ("k_bsr_spmv", "k_amg_update", "k_amg_restrict", "k_amg_prolong"))
    if busy_end and pcg:
        pcg_gap += max(s - busy_end, 0.0)
        n_pcg += 1
    elif busy_end and s - busy_end >= min_gap:
        print("%9.1f  gap %6.1f us  after %-40s before %s" % (busy_end, s - busy_end, prev[:40], r["Kernel_Name"][:40]))
        total_gap += s - busy_end
        n += 1
    if e > busy_end:
        busy_end, prev = e, r["Kernel_Name"]
print("inside the PCG: %d kernel boundaries, %.1f us idle in them" % (n_pcg, pcg_gap))
print("window %.1f us, %d kernels, %d gaps >= %.1f us, %.1f us idle in them" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, i1 - i0, n, min_gap, total_gap))
