#!/bin/bash
# A/B of whole training steps on ONE box: saved activation derivative on/off, epilogue store policy variants (compile-time builds selected
# with DEMO_HIP_LIB).  Usage (on the GPU box): bash tools/ab_store.sh [noderiv] [attn0] [<library variant> ...]
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/ab
mkdir -p $O
L=$PWD/demo2_amd/lib
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/$name.json 2> $O/$name.err || echo "FAILED $name"
  python - <<P
import json
d=json.load(open("$O/$name.json")); r=d["roofline"]
print("$name", d["ms_per_step"], "ms", d["value"], "img/s  dominant", r["achieved"], "TF", r["avg_launch_us"], "us;", "; ".join(f'{s["N"]}x{s["K"]} {s["epilogue"]} {s["tflops"]}' for s in r["by_shape"][:8]))
P
}
run base0 DEMO_X=0
for v in "$@"; do
  case $v in
    noderiv) run noderiv DEMO_SAVE_ACT_DERIV=0 ;;
    attn0) run attn0 DEMO_ATTN_VARIANT=0 ;;
    *) run $v DEMO_HIP_LIB=$L/libdemo_hip_$v.so ;;
  esac
done
run base1 DEMO_X=0
