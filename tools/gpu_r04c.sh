#!/bin/bash
# Third GPU call of round 4: k_ldl_chain with the two payload-is-its-own-flag hand-overs (L[p+2,p] to the chain workgroup, Minv_p to the urgent row
# workgroup), plain launches by default; K1/K2 with the pose table staged ten loads at a time; BALM_GT_NT=1 (streaming stores for Gt) as an A/B;
# the HBM yardsticks of tools/ubench_f64; small windows with the chain kernel forced; the shipped window's step.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04c; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null; cut -c1-200 $OUT/bench.json; grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/bench.json
BALM_GT_NT=1 timeout 600 python bench.py --no-cpu > $OUT/bench_gt_nt.json 2> $OUT/bench_gt_nt.err < /dev/null; echo "GT_NT=1:"; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_gt_nt.json | head -1; grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/bench_gt_nt.json
for g in 256 768 1024; do
  BALM_FACTORS_GRID=$g timeout 600 python bench.py --no-cpu --no-strong-ref --steps 60 > $OUT/bench_grid$g.json 2>/dev/null < /dev/null; echo "FACTORS_GRID=$g:"; grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/bench_grid$g.json | cut -c1-120
done
timeout 600 python tools/bench_solve.py 40 100 177 200 256 300 500 800 > $OUT/solve_default.txt 2>&1 < /dev/null; cut -c1-30,100-250 $OUT/solve_default.txt
for nh in 144 176; do
  BALM_CHAIN_NH=$nh timeout 600 python tools/bench_solve.py 177 200 > $OUT/solve_nh$nh.txt 2>&1 < /dev/null; sed "s/^/nh$nh /" $OUT/solve_nh$nh.txt | cut -c1-36,106-256
done
timeout 300 python tools/bench_solve.py 12 20 24 32 > $OUT/solve_small_default.txt 2>&1 < /dev/null; cut -c1-30,100-250 $OUT/solve_small_default.txt
BALM_SOLVE=chain timeout 300 python tools/bench_solve.py 12 20 24 32 > $OUT/solve_small_chain.txt 2>&1 < /dev/null; sed "s/^/chain-forced /" $OUT/solve_small_chain.txt | cut -c1-43,113-263
BALM_SOLVE_TRACE=1 timeout 300 python tools/chain_check.py 200 > $OUT/chain_trace_n1200.txt 2>&1 < /dev/null; sed -n 4,14p $OUT/chain_trace_n1200.txt | cut -c1-150
timeout 300 python tools/bench_realshape.py > $OUT/realshape.txt 2>&1 < /dev/null; grep "shipped\|default" $OUT/realshape.txt | cut -c1-220
timeout 300 python tools/bench_small.py > $OUT/small.txt 2>&1 < /dev/null; tail -5 $OUT/small.txt
BALM_BENCH_LOOPBACK=1 BALM_SOLVE_DEBUG=1 timeout 600 python bench.py --gpus 2 --no-cpu --no-accept --steps 20 > $OUT/bench_loopback2.json 2> $OUT/bench_loopback2.err < /dev/null; grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/bench_loopback2.json; grep "balm_hip: solve" $OUT/bench_loopback2.err | sort | uniq -c | head -3
timeout 200 tools/bin/ubench_f64 > $OUT/ubench_f64.txt 2>&1 < /dev/null; grep -i "copy\|read-only\|write-only" $OUT/ubench_f64.txt
