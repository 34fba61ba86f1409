#!/usr/bin/env python3
"""Extract the 256x4 rBRIEF sampling-pattern integers (data only, no comments)
from the reference's src/ORBextractor.cc:152-410 (`bit_pattern_31_`, originally
OpenCV BSD data) into a C include file of plain numbers.

Usage: python tools/gen_pattern.py /root/reference/src/ORBextractor.cc out.inc
The output is data that defines the descriptor; it is committed, so this script
only needs the reference when the table is regenerated.
"""
import re
import sys


def main(src, dst):
    text = open(src, "r", errors="replace").read()
    start = text.index("bit_pattern_31_[256 * 4]")
    body = text[text.index("{", start) + 1: text.index("};", start)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    nums = [int(t) for t in re.findall(r"-?\d+", body)]
    assert len(nums) == 1024, len(nums)
    assert all(-13 <= v <= 13 for v in nums)
    with open(dst, "w") as f:
        f.write("/* rBRIEF 31x31 sampling pattern: 256 rows of (x0, y0, x1, y1).\n"
                " * Data table only (see reference src/ORBextractor.cc:152-410). */\n")
        for i in range(0, 1024, 16):
            f.write(", ".join("%d" % v for v in nums[i:i + 16]) + ",\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
