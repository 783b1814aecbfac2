#!/usr/bin/env python3
"""Diagnostic: split a rocprofv3 kernel trace of bench.py into graph-replay launches and the eager timing pass
(the last launches_timed launches of the dominant kernel) and print the mean duration of each."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
sub = sys.argv[2]
n_timed = int(sys.argv[3])          # launches of that kernel in the eager timing pass (all of them, timed or not)
rows = []
for r in csv.DictReader(open(path)):
    if sub in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
d = [e - s for s, e in rows]
print("launches", len(d))
print("graph replay: mean %.1f ns" % (sum(d[:-n_timed]) / max(1, len(d) - n_timed)))
print("timing pass : mean %.1f ns" % (sum(d[-n_timed:]) / n_timed))
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
print("gap to next conv launch: graph %.1f ns, timing pass %.1f ns" % (
    sum(gaps[:-n_timed]) / max(1, len(gaps) - n_timed), sum(gaps[-n_timed + 1:]) / (n_timed - 1)))
