"""Kernel durations and the idle gaps between consecutive kernels in the steady state of a rocprofv3 --kernel-trace csv
(e.g. the replayed decode graph of the AR prior).  usage: python tools/graph_gaps.py <kernel_trace.csv> [last_n=4000]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
tot_d = tot_g = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    k = re.sub(r"\(.*", "", k)[:60]
    gap = max(0, s - prev_end) if prev_end is not None else 0
    a = agg[k]
    a[0] += 1
    a[1] += e - s
    a[2] += gap
    tot_d += e - s
    tot_g += gap
    prev_end = max(e, prev_end or 0)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{len(rows)} kernels over {span / 1e3:.0f} us: busy {tot_d / 1e3:.0f} us ({100 * tot_d / span:.0f} %), gaps {tot_g / 1e3:.0f} us; mean kernel {tot_d / len(rows) / 1e3:.2f} us, mean gap before a kernel {tot_g / len(rows) / 1e3:.2f} us")
print("kernel,calls,avg_us,avg_gap_before_us")
for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{k},{c},{d / c / 1e3:.2f},{g / c / 1e3:.2f}")
