#!/bin/bash
# Everything profiles/rNN_* is made from, in one pass on ONE box (run on the GPU box; outputs under gpurun_out/prof/):
#   bench lines (default, forward, side configurations), rocprofv3 kernel stats of the default command, the counter passes
#   (HBM traffic, L2, MFMA pipe / clock), and the micro-benchmarks quoted in DESIGN.md.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
step() { echo "== $*" | tee -a $O/log.txt; }
step pmc fetch;            rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_F -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_F.log 2>&1 || echo FAILED
step pmc write;            rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_W -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_W.log 2>&1 || echo FAILED
python tools/pmc_traffic.py $O/pmc_F $O/pmc_W > $O/hbm_traffic_pmc.json || echo FAILED
# the bench lines below read roofline.traffic from the newest profiles/r*_hbm_traffic_pmc.json: publish this pass's file first (ROUND=rNN)
cp $O/hbm_traffic_pmc.json profiles/${ROUND:-r03}_hbm_traffic_pmc.json
step bench default;        python bench.py > $O/bench_default.json 2> $O/bench_default.err || echo FAILED
step kernel stats;         rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o t -- python3 bench.py > $O/bench_default_under_rocprof.json 2> $O/stats.err || echo FAILED
step pmc l2 a;             rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_BUBBLE_sum --output-format csv -d $O/pmc_l2a -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_l2a.log 2>&1 || echo FAILED
step pmc l2 b;             rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/pmc_l2b -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_l2b.log 2>&1 || echo FAILED
python tools/pmc_l2.py $O/pmc_l2a $O/pmc_l2b > $O/l2_pmc.json || echo FAILED
step pmc sq;               rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1 || echo FAILED
step pmc grbm;             rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm -o g -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_grbm.log 2>&1 || echo FAILED
python tools/pmc_mfma.py $O/pmc_sq $O/pmc_grbm > $O/mfma_pmc.json || echo FAILED
step bench fwd b64;        python bench.py --mode fwd --no-cpu-baseline > $O/bench_fwd_b64.json 2> $O/bench_fwd.err || echo FAILED
step bench RGBNT100;       python bench.py --config RGBNT100 --no-cpu-baseline > $O/bench_rgbnt100.json 2> $O/bench_rgbnt100.err || echo FAILED
step bench vit;            python bench.py --config vit --no-cpu-baseline > $O/bench_vit.json 2> $O/bench_vit.err || echo FAILED
step bench dgaf;           python bench.py --fusion dgaf --no-cpu-baseline > $O/bench_dgaf.json 2> $O/bench_dgaf.err || echo FAILED
step fwd B=1;              python bench.py --mode fwd --batch 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_fwd_b1.json 2>/dev/null || echo FAILED
step gemm cost;            python tools/gemm_cost.py > $O/gemm_cost.txt 2>&1 || echo FAILED
step vendor blas;          python tools/bench_blas_ref.py > $O/vendor_blas_reference.txt 2>&1 || echo FAILED
step cu fill;              python tools/bench_cu_fill.py > $O/cu_fill.txt 2>&1 || echo FAILED
step d8 accuracy;          { python tools/probes/act_deriv_d8_accuracy.py; python tools/probes/act_deriv_d8_accuracy.py --no-triplet; } > $O/act_deriv_d8_accuracy.txt 2>&1 || echo FAILED
step tn;                   python tools/bench_gemm_tn.py > $O/bench_gemm_tn.txt 2>&1 || echo FAILED
step attention;            python tools/bench_attn.py > $O/bench_attn.txt 2>&1 || echo FAILED
step layernorm;            python tools/bench_ln.py > $O/bench_ln.txt 2>&1 || echo FAILED
[ -x tools/probes/store_probe ] || hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o tools/probes/store_probe
step store probe;          tools/probes/store_probe > $O/store_probe.txt 2>&1 || echo FAILED
step linearity;            { python tools/probes/linearity_by_param.py 8; echo "== triplet weight 0"; LIN_TRIPLET=0 python tools/probes/linearity_by_param.py 8; } 2>&1 | grep -v "amdgpu.ids\|Building DeMo" > $O/linearity_by_param.txt || echo FAILED
step eval latency;         bash tools/eval_latency.sh > $O/eval_latency.txt 2>&1 || echo FAILED
step eval kernel stats;    bash tools/eval_profile.sh > $O/eval_profile.log 2>&1 || echo FAILED
for f in b1_f1 b1_f0 b64_f1; do cp gpurun_out/lat/prof_$f/p_kernel_stats.csv $O/fwd_${f}_kernel_stats.csv 2>/dev/null; done
# keep only the summaries (the raw traces are tens of MB)
cp $O/stats/t_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/stats $O/pmc_F $O/pmc_W $O/pmc_l2a $O/pmc_l2b $O/pmc_sq $O/pmc_grbm
ls -la $O
