#!/usr/bin/env python3
"""profiles/issue.json from a rocprofv3 --pmc pass of tools/stage_bench.py (256 frames per launch):
mean SQ instruction counts per launch of every stage, from which bench.py derives the share of the
vector-issue slots a kernel uses (a wave64 VALU instruction holds its SIMD for 4 cycles).
Usage: make_issue.py <counter csv> [out.json]"""
import collections
import csv
import json
import sys

STAGE_OF = {"k_pyr_level0": "pyr_level0", "k_resize_slide": "pyr_resize", "k_blur_slide": "blur",
            "k_fast_cells_wave": "fast_cells", "k_fast_blur": "fast_blur", "k_quadtree": "quadtree", "k_orient_desc": "orient_desc"}
COUNTERS = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVES")


def main():
    tot = collections.defaultdict(collections.Counter)
    n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(sys.argv[1])):
        if r["Counter_Name"] not in COUNTERS:
            continue
        for k, st in STAGE_OF.items():
            if k in r["Kernel_Name"]:
                tot[st][r["Counter_Name"]] += float(r["Counter_Value"])
                n[st][r["Counter_Name"]] += 1
    res = {}
    for st in tot:
        per = 7 if st == "pyr_resize" else 1      # the resize chain is 7 launches per step
        res[st] = {c: round(tot[st][c] / (n[st][c] / per)) for c in tot[st]}
    res["_note"] = "mean SQ counters per stage per 256-frame extraction step (pyr_resize: sum of its 7 launches)"
    out = sys.argv[2] if len(sys.argv) > 2 else "profiles/issue.json"
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
