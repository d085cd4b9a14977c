"""Serial run (one batch at a time): where the wall time of a step goes -- kernel time by class and the idle gaps between
consecutive kernels (rocprofv3 --kernel-trace CSV of `bench.py --in-flight 1`)."""
import csv
import collections
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"msh::\(anonymous namespace\)::", "", r["Kernel_Name"]).split("<")[0].split("(")[0].replace("void ", "")[:40]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# steady window: the longest stretch of dec_cross kernels without a pause > 5 ms (= the serial timed steps), middle 80 %
t0, t1 = rows[0][0], rows[-1][1]
a, b = t0 + (t1 - t0) * 0.25, t0 + (t1 - t0) * 0.75
sel = [r for r in rows if r[0] >= a and r[1] <= b]
busy = collections.Counter()
gaps = collections.Counter()
gap_n = collections.Counter()
last_end, last_name = sel[0][1], sel[0][2]
tot_busy = sel[0][1] - sel[0][0]
for s, e, n in sel[1:]:
    if s > last_end:
        g = s - last_end
        if g < 2_000_000:
            gaps[last_name + " -> " + n] += g
            gap_n[last_name + " -> " + n] += 1
    busy[n] += e - s
    tot_busy += e - s
    last_end, last_name = max(last_end, e), n
wall = sel[-1][1] - sel[0][0]
print("window %.1f ms: kernels %.1f ms (%.1f %%), gaps %.1f ms" % (wall / 1e6, tot_busy / 1e6, 100.0 * tot_busy / wall, sum(gaps.values()) / 1e6))
print("kernel time:")
for k, v in busy.most_common(14):
    print("  %-40s %7.2f ms" % (k, v / 1e6))
print("largest gap classes (total ms, count, mean us):")
for k, v in gaps.most_common(16):
    print("  %-70s %6.2f %6d %6.2f" % (k, v / 1e6, gap_n[k], v / gap_n[k] / 1e3))
