#!/usr/bin/env python3
"""Vector / scalar / LDS / memory instruction counts per basic block of one kernel in a hipcc -S listing.
Usage: asm_blocks.py <listing.s> <kernel name substring> [min vector instructions to print a block]"""
import sys


def main():
    text = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    start = next(i for i, ln in enumerate(text) if ln.startswith("_Z") and want in ln and ln.rstrip().split(":")[0].endswith(ln.split(":")[0]))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    blocks, cur = [], ["entry", 0, 0, 0, 0, start + 1]
    for n in range(start + 1, end):
        t = text[n].strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith(".LBB"):
            blocks.append(cur)
            cur = [t.split(":")[0], 0, 0, 0, 0, n + 1]
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        k = 1 if op.startswith("v_") else 2 if op.startswith("s_") else 3 if op.startswith("ds_") else \
            4 if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else 0
        if k:
            cur[k] += 1
    blocks.append(cur)
    print("%d blocks, %d vector instructions in all" % (len(blocks), sum(b[1] for b in blocks)))
    print("%-14s %6s %6s %6s %6s  line" % ("block", "valu", "salu", "lds", "vmem"))
    for b in blocks:
        if b[1] >= floor:
            print("%-14s %6d %6d %6d %6d  %d" % tuple(b))


if __name__ == "__main__":
    main()
