#!/bin/bash
# Interleaved same-box A/B of library builds: tools/ab_interleaved.sh <variantA|default> <variantB|default> [rounds]  (bench.py, alternating A B A B ...)
cd "$(dirname "$0")/.."
A=$1; B=$2; N=${3:-3}
O=gpurun_out/abi_${A}_${B}
mkdir -p $O
for r in $(seq 1 $N); do
  for v in $A $B; do
    if [ $v = default ]; then unset DEMO_HIP_LIB; else export DEMO_HIP_LIB=$PWD/demo2_amd/lib/libdemo_hip_$v.so; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_${v}_$r.json 2> $O/bench_${v}_$r.err || { tail -5 $O/bench_${v}_$r.err; exit 1; }
    python - <<P
import json
d=json.load(open("$O/bench_${v}_$r.json")); print("$v round $r:", d["ms_per_step"], "ms/step", d["value"], "img/s", "dominant TF", d["roofline"]["achieved"])
P
  done
done
