#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks from a build log: one line per kernel."""
import re
import sys


def parse(path):
    cur, d = None, {}
    for line in open(path):
        m = re.search(r"Name: (\S+)", line)
        if m:
            cur = m.group(1)
            d[cur] = {}
        for key, short in (("VGPRs", "vgpr"), ("AGPRs", "agpr"), (r"ScratchSize \[bytes/lane\]", "scratch"), ("SGPRs Spill", "sgpr_spill"),
                           ("VGPRs Spill", "vgpr_spill"), (r"Occupancy \[waves/SIMD\]", "occ")):
            m = re.search(r"\s" + key + r": (\d+)", line)
            if m and cur:
                d[cur][short] = int(m.group(1))
    return d


if __name__ == "__main__":
    for k, v in parse(sys.argv[1]).items():
        m = re.match(r"_ZN\d+[a-z0-9_]*?(\d+)([a-z_0-9]+?)I", k)
        name = k
        mm = re.search(r"_ZN\d+grid_[a-z0-9]+?(\d\d)([a-z_0-9]+)", k)
        if mm:
            name = mm.group(2)[:int(mm.group(1))]
        nargs = k.count("PKS1_") + k.count("PKT_")
        print("%-50s ptr-args %d  %s" % (name, nargs, v))
