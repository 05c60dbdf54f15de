cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab2
for v in off on off on; do
  n=$(ls gpurun_out/ab2 | wc -l)
  if [ $v = on ]; then export DEMO_BF16_RESIDUAL=1; else export DEMO_BF16_RESIDUAL=0; fi
  python bench.py --no-cpu-baseline > gpurun_out/ab2/${n}_$v.json 2>/dev/null
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab2/*.json")):
    d=json.load(open(f)); r=d["roofline"]
    print(f.split("/")[-1], d["ms_per_step"], "ms", d["value"], "img/s dominant", r["achieved"], "TF", r["gemm_ms_per_step"], "ms")
P
