#!/bin/bash
# Per-kernel time of the eval forward at B=1 and B=64, fused ATMoE inference kernels on / off (rocprofv3 kernel stats only).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/lat
mkdir -p $O
for f in 1 0; do
  export DEMO_FUSED_ATMOE_EVAL=$f
  for B in 1 64; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b${B}_f$f -o p -- python3 bench.py --mode fwd --batch $B --steps 100 --warmup 10 --no-cpu-baseline > $O/prof_b${B}_f$f.log 2>&1 || { tail -5 $O/prof_b${B}_f$f.log; exit 1; }
    find $O/prof_b${B}_f$f -name '*kernel_trace.csv' -delete
  done
done
find $O -name '*kernel_stats.csv' | head
