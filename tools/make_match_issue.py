#!/usr/bin/env python3
"""profiles/match_issue.json from the rocprofv3 --pmc passes of tools/match_one_shape.py (8 x 8000 queries
against one 100 000-point map): mean counters per launch of the matrix-core sweep, from which bench.py derives
the share of the vector-issue slots, of the matrix pipe and of LDS bank conflicts for that shape.
Usage: make_match_issue.py <counter csv>... <out.json>"""
import collections
import csv
import json
import sys

KERNEL = "k_match_bruteforce_fp4"
SHAPE = "map_8x8000x100000"


def main():
    tot, n = collections.Counter(), collections.Counter()
    for path in sys.argv[1:-1]:
        for r in csv.DictReader(open(path)):
            if KERNEL in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
    res = {SHAPE: {c: round(tot[c] / n[c]) for c in sorted(tot)},
           "_note": "mean counters per launch of %s on the shape %s (tools/match_one_shape.py)" % (KERNEL, SHAPE)}
    json.dump(res, open(sys.argv[-1], "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
