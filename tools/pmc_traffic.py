"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` into profiles/pmc_traffic.json.

usage: python tools/pmc_traffic.py <round> <precision> <fetch counter_collection.csv> <write counter_collection.csv> [batch=16384]

Entries are keyed "<robot>:<batch>:<kernel name>" and stamped with the sha of the generated header they were measured on
(bench.py prints `traffic` only when that sha matches the header of the library it runs -- a stale figure cannot be printed).
bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the dispatches of the kernel (MI355X_MICROARCH.md, HBM section: the
counters are in KiB; FETCH_SIZE is calibrated for 16-B/lane streaming loads only -- these kernels load 4 B/lane, so the read
part may under-count by up to 2x; WRITE_SIZE is exact for streaming stores)."""
import collections
import csv
import hashlib
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from gridcodegenerator_amd import host  # noqa: E402


def averages(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    rnd, precision, fetch_csv, write_csv = sys.argv[1:5]
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 16384
    fetch, write = averages(fetch_csv, "FETCH_SIZE"), averages(write_csv, "WRITE_SIZE")
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    with open(path) as fh:
        data = json.load(fh)
    # which (robot, batch) a kernel name belongs to: the bench runs iiwa7 at 16384 and atlas30 at 16384 (secondary)
    for kname in sorted(set(fetch) & set(write)):
        m = re.search(r"grid_(\w+?)::((?:forward|inverse)_dynamics_gradient_kernel\w*)", kname)
        if not m:
            continue
        robot, kernel = m.group(1), m.group(2)
        with open(host.library_paths(robot, precision)["header"], "rb") as fh:
            sha = hashlib.sha256(fh.read()).hexdigest()[:16]
        f, nf = fetch[kname]
        w, nw = write[kname]
        key = "%s:%d:%s" % (robot, batch, kernel)
        data[key] = {"kernel": kernel, "fetch_kb": round(f, 1), "write_kb": round(w, 1), "bytes": int((f + w) * 1024), "dispatches": [nf, nw],
                     "round": rnd, "precision": precision, "header_sha": sha}
        print(key, data[key])
    with open(path, "w") as fh:
        json.dump(data, fh, indent=2)


if __name__ == "__main__":
    main()
