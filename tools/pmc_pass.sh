#!/bin/bash
# The two HBM-traffic counter passes of bench.py (separate runs, as the guide prescribes) -> gpurun_out/pmc_<tag>_{f,w}; aggregate with
# tools/pmc_traffic.py.  Usage: bash tools/pmc_pass.sh <tag> [bench args...]   (DEMO_HIP_LIB etc. are taken from the environment)
set -o pipefail
cd "$(dirname "$0")/.."
tag=$1; shift
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_${tag}_${c:0:1}
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $d.log 2>&1 || { echo "pmc pass $c failed"; tail -5 $d.log; }
done
python tools/pmc_traffic.py gpurun_out/pmc_${tag}_F gpurun_out/pmc_${tag}_W > gpurun_out/pmc_${tag}_traffic.json && python - <<P
import json
d=json.load(open("gpurun_out/pmc_${tag}_traffic.json"))["kernels"]
for k,v in list(d.items())[:8]: print("${tag}", k[:60], v)
P
