#!/usr/bin/env python3
"""MFMA-pipe and wave-state counters per kernel from a rocprofv3 counter pass over bench.py:

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
        SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq -o s \\
        -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_grbm -o g -- python3 bench.py ... (same)
    python tools/pmc_mfma.py gpurun_out/pmc_sq gpurun_out/pmc_grbm > profiles/rNN_mfma_pmc.json

mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x effective clock): the share of SIMD cycles in which the
matrix pipe is busy (the counter is summed over the chip's SIMDs; a 16x16x32 bf16 MFMA holds the pipe 16 cycles).
effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, 'DVFS give-back'; reads high on launches under 0.3 ms).
The wave-state counters are fractions of SQ_WAVE_CYCLES."""
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


def collect(d):
    tot, n, dur = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int)), defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return tot, n, dur


def main():
    sq, sqn, sdur = collect(sys.argv[1])
    gr, grn, gdur = collect(sys.argv[2])
    out = {}
    for k in sorted(sq, key=lambda k: -sum(sdur[k])):
        if k.startswith("__amd") or not sdur[k]:
            continue
        c = {name: sq[k][name] / max(sqn[k][name], 1) for name in sq[k]}
        d_ns = sum(sdur[k]) / len(sdur[k])
        clk = None
        if k in gr and gdur[k]:
            clk = (gr[k]["GRBM_GUI_ACTIVE"] / max(grn[k]["GRBM_GUI_ACTIVE"], 1)) / 8.0 / (sum(gdur[k]) / len(gdur[k]))   # cycles per ns = GHz
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        e = {"launches": len(sdur[k]), "avg_us_under_pmc": round(d_ns / 1e3, 1),
             "effective_clock_ghz": None if clk is None else round(clk, 2),
             "mfma_busy_cycles": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)),
             "mfma_busy_pct": None if clk is None else round(100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * d_ns * clk), 1),
             "wait_any_pct": round(100 * c.get("SQ_WAIT_ANY", 0.0) / wc, 1), "wait_inst_any_pct": round(100 * c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 1),
             "active_inst_any_pct": round(100 * c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 1),
             "lds_bank_conflict_pct_of_lds_active": round(100 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 2)}
        out[k] = e
        if len(out) >= 12:
            break
    json.dump({"command": "rocprofv3 --kernel-trace --pmc <8 SQ counters> / --pmc GRBM_GUI_ACTIVE (two passes) -- python3 bench.py --steps 2 "
                          "--warmup 1 --no-cpu-baseline ; tools/pmc_mfma.py", "kernels": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
