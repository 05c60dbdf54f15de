#!/bin/bash
# A/B of a compile-time library variant against the default on one box: tools/ab_lib.sh <variant> (demo2_amd/lib/libdemo_hip_<variant>.so)
cd "$(dirname "$0")/.."
V=$1
O=gpurun_out/ab_$V
mkdir -p $O
for lib in default $V; do
  if [ $lib = default ]; then unset DEMO_HIP_LIB; else export DEMO_HIP_LIB=$PWD/demo2_amd/lib/libdemo_hip_$lib.so; fi
  timeout -k 10 300 python tools/gemm_cost.py > $O/gemm_cost_$lib.txt 2>&1 || { tail -5 $O/gemm_cost_$lib.txt; exit 1; }
  for rep in 1 2; do
    timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_${lib}_$rep.json 2> $O/bench_${lib}_$rep.err || { tail -5 $O/bench_${lib}_$rep.err; exit 1; }
    python - <<P
import json
d=json.load(open("$O/bench_${lib}_$rep.json")); print("$lib run $rep:", d["ms_per_step"], "ms/step", d["value"], "img/s", "dominant TF", d["roofline"]["achieved"])
P
  done
done
