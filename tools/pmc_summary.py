#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per
dispatch of each orbx kernel.  Usage: pmc_summary.py <csv> [<csv> ...]"""
import collections
import csv
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "orbx" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, " ".join("%s=%.6g(n=%d)" % (c, sum(x) / len(x), len(x)) for c, x in sorted(v.items())))


if __name__ == "__main__":
    main()
