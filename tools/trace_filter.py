#!/usr/bin/env python3
"""Average duration of the dispatches of one kernel at the grid size that accounts for most of its time, from a
rocprofv3 kernel_trace.csv: in a run that launches the same kernel on several workloads (bench.py's
legs of the other configs) the --stats average mixes them; this picks the headline shape.
Usage: trace_filter.py <kernel_trace.csv> <kernel name substring> [grid size]"""
import collections
import csv
import sys


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
    grids = collections.Counter(r.get("Grid_Size", r.get("Grid_Size_X", "?")) for r in rows)
    if len(sys.argv) > 3:
        top = sys.argv[3]
    else:       # the grid size that accounts for most of the kernel's time: the headline batch
        tot = collections.Counter()
        for r in rows:
            tot[r.get("Grid_Size", r.get("Grid_Size_X", "?"))] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        top = tot.most_common(1)[0][0]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r.get("Grid_Size", r.get("Grid_Size_X", "?")) == top]
    print("%s: %d dispatches in all; grid size %s: %d dispatches, average %.1f ns, min %d, max %d"
          % (sys.argv[2], len(rows), top, len(d), sum(d) / len(d), min(d), max(d)))
    for g, c in grids.most_common():
        dd = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r.get("Grid_Size", r.get("Grid_Size_X", "?")) == g]
        print("   grid %s: %d dispatches, average %.1f ns" % (g, c, sum(dd) / len(dd)))


if __name__ == "__main__":
    main()
