#!/usr/bin/env python3
"""Converts a GMAT .cof(.gz) gravity model into the packed float64 fixture nyx_amd ships
(nyx_amd/data/jgm3_70x70.f64: [degree, order, C packed lower-triangular, S packed]).

The source is the reference's data file data/01_planetary/JGM3.cof.gz (the only real gravity model
present in the reference checkout; EGM2008 is a missing blob).  It cannot be
read on the GPU box (no /root/reference there), hence the committed fixture.  Parsing rules restate
GravityFieldData::from_cof (reference io/gravity.rs:150-367) independently of the C++ loader in
nyx_amd/csrc/gravity_io.cpp — tests/test_gravity_io.py checks the two against each other.

usage: tools/convert_cof.py /root/reference/data/01_planetary/JGM3.cof.gz 70 70 nyx_amd/data/jgm3_70x70.f64
"""
import gzip
import sys

import numpy as np


def parse_cof(path, degree, order):
    opener = gzip.open if path.endswith(".gz") else open
    text = opener(path, "rb").read().decode("utf8")
    n = (degree + 1) * (degree + 2) // 2
    c, s = np.zeros(n), np.zeros(n)
    max_d = max_o = 0
    for line in text.split("\n"):
        if not line or not line.startswith("R"):
            continue
        items = line.split()
        d, o = int(items[1]), int(items[2])
        item = items[3]
        cv = sv = 0.0
        if degree == 0:
            cv = float(item)
        else:
            minus = item.count("-")
            if (minus == 3 and not item.startswith("-")) or minus == 4:
                p = item.split("-")
                if len(p) == 5:
                    cv, sv = float("-" + p[1] + "-" + p[2]), float("-" + p[3] + "-" + p[4])
                else:
                    cv, sv = float(p[0] + "-" + p[1]), float("-" + p[2] + "-" + p[3])
            else:
                cv = float(item)
        if len(items) > 4:
            sv = float(items[4])
        if d > degree:
            break
        if o <= order and o <= d:
            c[d * (d + 1) // 2 + o] = cv
            s[d * (d + 1) // 2 + o] = sv
        max_o, max_d = max(max_o, o), max(max_d, d)
    return max_d, max_o, c, s


if __name__ == "__main__":
    src, deg, ordr, dst = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    d, o, c, s = parse_cof(src, deg, ordr)
    n = (d + 1) * (d + 2) // 2
    np.concatenate([[float(d), float(o)], c[:n], s[:n]]).astype("<f8").tofile(dst)
    print(f"wrote {dst}: degree {d} order {o}, C20 = {c[3]!r}")
