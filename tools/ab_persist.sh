#!/bin/bash
# three-way same-box A/B: committed library (base), new library with one tile per workgroup, new library persistent
cd "$(dirname "$0")/.."
O=gpurun_out/ab_persist
mkdir -p $O
run() {  # name
  timeout -k 10 300 python tools/gemm_cost.py > $O/gemm_cost_$1.txt 2>&1 || { tail -5 $O/gemm_cost_$1.txt; exit 1; }
  for rep in 1 2; do
    timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_$1_$rep.json 2> $O/bench_$1_$rep.err || { tail -5 $O/bench_$1_$rep.err; exit 1; }
    python - <<P
import json
d=json.load(open("$O/bench_$1_$rep.json")); print("$1 run $rep:", d["ms_per_step"], "ms/step", d["value"], "img/s", "dominant TF", d["roofline"]["achieved"])
P
  done
}
DEMO_HIP_LIB=$PWD/demo2_amd/lib/libdemo_hip_base.so run base
DEMO_GEMM_PERSIST=0 run new_p0
DEMO_GEMM_PERSIST=1 run new_p1
