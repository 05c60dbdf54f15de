#!/bin/bash
# Eval-forward latency at small batches, fused two-launch ATMoE + cached HDM quantities on / off, one process each (same box).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/lat
for B in 1 8 64; do
  for f in 1 0; do
    DEMO_FUSED_ATMOE_EVAL=$f python bench.py --mode fwd --batch $B --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/lat/b${B}_f$f.json 2> gpurun_out/lat/b${B}_f$f.err
    python - <<P
import json
d=json.load(open("gpurun_out/lat/b${B}_f$f.json")); print("B=$B fused_atmoe=$f", d["ms_per_step"], "ms", d["value"], "img/s")
P
  done
done
