#!/usr/bin/env python3
"""profiles/traffic.json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
tools/stage_bench.py (256 frames per launch).

HBM bytes per launch of a stage = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in
KiB, and FETCH_SIZE tallies 128-byte read requests at 64 bytes on gfx950 (hence the factor 2).
Infinity-Cache hits are included in both counters, so this is memory-side traffic of the L2,
an upper bound on what reaches HBM.
Usage: make_traffic.py <fetch counter csv> <write counter csv> [out.json]"""
import collections
import csv
import json
import sys

STAGE_OF = {"k_pyr_level0": "pyr_level0", "k_resize_slide": "pyr_resize", "k_blur_slide": "blur",
            "k_fast_cells_wave": "fast_cells", "k_fast_blur": "fast_blur", "k_quadtree": "quadtree", "k_orient_desc": "orient_desc"}


def per_step(path, counter):
    tot = collections.Counter()
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        for k, st in STAGE_OF.items():
            if k in name:
                tot[st] += float(r["Counter_Value"])
                n[st] += 1
    # launches per extraction step: 7 for the resize chain, 1 for every other stage
    out = {}
    for st in tot:
        per = 7 if st == "pyr_resize" else 1
        out[st] = tot[st] / (n[st] / per)
    return out


def main():
    fetch = per_step(sys.argv[1], "FETCH_SIZE")
    write = per_step(sys.argv[2], "WRITE_SIZE")
    res = {}
    for st in STAGE_OF.values():
        if st in fetch and st in write:
            res[st] = int((2.0 * fetch[st] + write[st]) * 1024)
    res["_note"] = ("HBM-side bytes per stage per 256-frame launch = (2*FETCH_SIZE + WRITE_SIZE)*1024; "
                    "pyr_resize is the sum of its 7 launches; raw KiB: fetch=%s write=%s" %
                    ({k: round(v) for k, v in fetch.items()}, {k: round(v) for k, v in write.items()}))
    out = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
