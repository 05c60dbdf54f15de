#!/bin/bash
# A/B of a run-time tuning hook on one box: tools/ab_env.sh VAR value_a value_b  (gemm_cost table + two default bench runs per value)
cd "$(dirname "$0")/.."
VAR=$1; shift
O=gpurun_out/ab_$VAR
mkdir -p $O
for val in "$@"; do
  export $VAR=$val
  timeout -k 10 300 python tools/gemm_cost.py > $O/gemm_cost_$val.txt 2>&1 || { tail -5 $O/gemm_cost_$val.txt; exit 1; }
  for rep in 1 2; do
    timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_${val}_$rep.json 2> $O/bench_${val}_$rep.err || { tail -5 $O/bench_${val}_$rep.err; exit 1; }
    python - <<P
import json
d=json.load(open("$O/bench_${val}_$rep.json")); print("$VAR=$val run $rep:", d["ms_per_step"], "ms/step", d["value"], "img/s", "dominant TF", d["roofline"]["achieved"])
P
  done
done
