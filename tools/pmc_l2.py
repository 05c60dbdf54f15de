#!/usr/bin/env python3
"""L2 (TCC) counters per kernel from rocprofv3 counter passes: hit rate, the share of L2 reads that go out to the fabric, and the bytes
the L2s request from it.  Request sizes: TCC_BUBBLE was meant to resolve the 128-B reads that FETCH_SIZE tallies at 64 B (the reason
FETCH_SIZE is doubled on gfx950), but it reads ~0 on this pool, so the fill is priced like FETCH_SIZE's correction: 128 B per request
(32-B requests, counted separately, at 32 B); the raw request counts are kept beside it.

    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_BUBBLE_sum --output-format csv -d gpurun_out/pmc_l2a -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d gpurun_out/pmc_l2b -o t -- python3 bench.py ...
    python tools/pmc_l2.py gpurun_out/pmc_l2a gpurun_out/pmc_l2b > profiles/rNN_l2_pmc.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import csrc_sha16, short  # noqa: E402


def main():
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
    out = {}
    for k, c in tot.items():
        if k.startswith("__amd"):
            continue
        a = {n: c[n] / cnt[k][n] for n in c}                     # per launch
        e = {"launches": max(cnt[k].values())}
        for n, v in a.items():
            e[n] = round(v, 1)
        if "TCC_HIT_sum" in a and "TCC_MISS_sum" in a and a["TCC_HIT_sum"] + a["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = round(a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"]), 4)
        if "TCC_EA0_RDREQ_sum" in a:
            r32 = a.get("TCC_EA0_RDREQ_32B_sum", 0.0)
            e["l2_fill_mb_per_launch"] = round(((a["TCC_EA0_RDREQ_sum"] - r32) * 128 + r32 * 32) / 1e6, 1)
            if "TCC_READ_sum" in a and a["TCC_READ_sum"] > 0:
                e["l2_read_miss_share"] = round(a["TCC_EA0_RDREQ_sum"] / a["TCC_READ_sum"], 4)     # reads that leave the L2, of all L2 reads
        if "TCC_EA0_RDREQ_DRAM_sum" in a and "TCC_EA0_RDREQ_sum" in a:
            e["share_of_read_requests_destined_for_dram"] = round(a["TCC_EA0_RDREQ_DRAM_sum"] / max(a["TCC_EA0_RDREQ_sum"], 1.0), 3)
        out[k] = e
    order = sorted(out, key=lambda k: -out[k].get("l2_fill_mb_per_launch", 0.0) * out[k]["launches"])
    json.dump({"note": "per-launch averages of the TCC counters (sum over the L2 channels of all XCDs); l2_fill_mb_per_launch = bytes the L2s "
                       "request from the fabric (Infinity Cache or HBM: the TCC cannot tell them apart) at 128 B per request (= FETCH_SIZE doubled); "
                       "l2_hit_rate counts reads and writes, l2_read_miss_share = fabric read requests / L2 read requests; "
                       "tools/pmc_l2.py", "csrc_sha16": csrc_sha16(), "kernels": {k: out[k] for k in order}}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
