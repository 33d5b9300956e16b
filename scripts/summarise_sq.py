"""Per-kernel means of the SQ counter pass of scripts/profile_round.sh (rocprofv3 --pmc SQ_... -- python3 scripts/pmc_workload.py)
as the table profiles/<round>_pmc_sq_cfg4.csv:  python scripts/summarise_sq.py r03 gpurun_out/r03d/pmc_sq"""
import csv
import glob
import os
import sys
from collections import defaultdict

COUNTERS = ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY",
            "SQ_WAIT_INST_ANY")
tag, src = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
assert files, "no counter_collection.csv under " + src
tot = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
rows = []
for k, c in tot.items():
    n = len(launches[k])
    m = {x: c.get(x, 0.0) / n for x in COUNTERS}
    wc = m["SQ_WAVE_CYCLES"] or 1.0
    rows.append((m["SQ_WAVE_CYCLES"] * n, k, n, m, m["SQ_ACTIVE_INST_VALU"] / wc, m["SQ_ACTIVE_INST_ANY"] / wc, m["SQ_WAIT_INST_ANY"] / wc,
                 m["SQ_INSTS_VALU"] / (m["SQ_WAVES"] or 1.0)))
rows.sort(reverse=True)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_pmc_sq_cfg4.csv")
with open(out, "w") as fo:
    fo.write("# rocprofv3 --pmc " + " ".join(COUNTERS) + " -- python3 scripts/pmc_workload.py cfg4 1.0 2 (means per launch; shares are of "
             "SQ_WAVE_CYCLES)\n")
    w = csv.writer(fo)
    w.writerow(["kernel", "launches"] + list(COUNTERS) + ["valu_active_share", "any_inst_active_share", "wait_inst_any_share", "valu_insts_per_wave"])
    for _, k, n, m, a, b, c, d in rows[:40]:
        w.writerow([k, n] + ["%.6g" % m[x] for x in COUNTERS] + ["%.3f" % a, "%.3f" % b, "%.3f" % c, "%.0f" % d])
print("wrote", out)
