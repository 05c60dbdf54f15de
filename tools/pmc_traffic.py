#!/usr/bin/env python3
"""HBM bytes per launch per kernel from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as
/opt/skills/guides/MI355X_MICROARCH.md prescribes):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w > profiles/rNN_hbm_traffic_pmc.json

hbm_mb_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: the counters are in KiB, and on gfx950 FETCH_SIZE tallies wide coalesced
reads at half their bytes (guide, HBM section), so it is doubled; WRITE_SIZE is taken as is."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9]+(<[^(]*>)?)", name)
    return m.group(1) if m else name


def collect(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no *counter_collection.csv under {d}")
    tot, n = defaultdict(float), defaultdict(int)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                n[k] += 1
    return tot, n


def csrc_sha16():
    """Hash of the kernel sources the counters were taken with: bench.py marks the figure stale once the sources have changed."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "demo2_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    ft, fn = collect(fd, "FETCH_SIZE")
    wt, wn = collect(wd, "WRITE_SIZE")
    out = {}
    for k in sorted(ft, key=lambda k: -(2 * ft[k] + wt.get(k, 0.0))):
        if k.startswith("__amd") or fn[k] == 0:
            continue
        f_kib, w_kib = ft[k] / fn[k], (wt.get(k, 0.0) / wn[k] if wn.get(k) else 0.0)
        out[k] = {"launches": fn[k], "fetch_kib_raw": round(f_kib, 1), "write_kib": round(w_kib, 1),
                  "hbm_mb_per_launch": round((2 * f_kib + w_kib) * 1024 / 1e6, 1)}
    json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE (two separate passes) -- python3 bench.py "
                          "--steps 2 --warmup 1 --no-cpu-baseline",
               "note": "hbm_mb_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB.  FETCH_SIZE is doubled for gfx950 (MI355X_MICROARCH.md, "
                       "HBM section: wide coalesced reads are tallied at half their bytes); WRITE_SIZE is taken as is.  Averages over "
                       "all launches of a kernel in 4 steps (1 warmup + 2 timed + 1 instrumented).  tools/pmc_traffic.py",
               "csrc_sha16": csrc_sha16(), "kernels": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
