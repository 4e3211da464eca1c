#!/bin/bash
# Second GPU call of round 4: the tree after the five switches were settled (rows+lower build and the one-product back-substitution are
# the defaults, the losers are gone), the multi-device context with its persistent solve, bench.py's launch handling; then
#   * the solve by window: default, BALM_COOP=0 (plain launch of the persistent kernels), fewer helpers (BALM_CHAIN_NH)
#   * where a one-rank RCCL communicator's 0.7 ms per step comes from: tools/exp_dist_overhead.py over the knobs VERDICT r3 item 4 lists
REPO=$(pwd); OUT=$REPO/gpurun_out/r04b; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null; cut -c1-900 $OUT/bench.json
WS="40 100 177 200 256 500"
timeout 600 python tools/bench_solve.py $WS > $OUT/solve_default.txt 2>&1 < /dev/null; cut -c1-30,100-250 $OUT/solve_default.txt
BALM_COOP=0 timeout 600 python tools/bench_solve.py $WS > $OUT/solve_coop0.txt 2>&1 < /dev/null; sed "s/^/coop0 /" $OUT/solve_coop0.txt | cut -c1-36,106-256
for nh in 48 96 144; do
  BALM_CHAIN_NH=$nh timeout 600 python tools/bench_solve.py 100 177 200 > $OUT/solve_nh$nh.txt 2>&1 < /dev/null; sed "s/^/nh$nh /" $OUT/solve_nh$nh.txt | cut -c1-36,106-256
done
BALM_COOP=0 BALM_CHAIN_NH=96 timeout 600 python tools/bench_solve.py 177 200 > $OUT/solve_coop0_nh96.txt 2>&1 < /dev/null; sed "s/^/coop0+nh96 /" $OUT/solve_coop0_nh96.txt | cut -c1-40,110-260
BALM_COOP=0 timeout 600 python bench.py --no-cpu > $OUT/bench_coop0.json 2> $OUT/bench_coop0.err < /dev/null; cut -c1-200 $OUT/bench_coop0.json; grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/bench_coop0.json
# ---- the RCCL communicator tax
D=$OUT/dist_overhead.txt; : > $D
run() { # label, mode, env...
  local label=$1 mode=$2; shift 2
  ( env "$@" timeout 200 python tools/exp_dist_overhead.py $mode "$label" 2>&1 < /dev/null | grep -v "^$" | tail -3 ) >> $D
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -2 | tr '\n' ' ' >> $D; echo >> $D
}
run "-" plain A=1
run "-" rccl_only A=1
run "-" rccl_other_ctx A=1
run "-" rccl_destroyed A=1
run "GPU_MAX_HW_QUEUES=2" rccl_only GPU_MAX_HW_QUEUES=2
run "GPU_MAX_HW_QUEUES=8" rccl_only GPU_MAX_HW_QUEUES=8
run "NCCL_MIN/MAX_NCHANNELS=1" rccl_only NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1
run "NCCL_MIN/MAX_NCHANNELS=4" rccl_only NCCL_MAX_NCHANNELS=4 NCCL_MIN_NCHANNELS=4
run "RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0" rccl_only RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0
run "HSA_ENABLE_SDMA=0" rccl_only HSA_ENABLE_SDMA=0
run "HSA_ENABLE_INTERRUPT=0" rccl_only HSA_ENABLE_INTERRUPT=0
run "HSA_ENABLE_INTERRUPT=0 (plain)" plain HSA_ENABLE_INTERRUPT=0
run "BALM_STREAM_PRIORITY=high" rccl_only BALM_STREAM_PRIORITY=high
run "BALM_COOP=0" rccl_only BALM_COOP=0
run "BALM_COOP=0 (plain)" plain BALM_COOP=0
run "NCCL_IGNORE_CPU_AFFINITY=1" rccl_only NCCL_IGNORE_CPU_AFFINITY=1
run "rccl_affinity (restore the thread's CPU mask)" rccl_affinity A=1
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,ENV,GRAPH timeout 200 python tools/exp_dist_overhead.py rccl_only "NCCL_DEBUG=INFO" > $OUT/dist_nccl_debug.txt 2>&1 < /dev/null
grep -v "^$" $D | cut -c1-330
# every kernel of a short rccl_only run as the profiler's database records it (is anything of RCCL's running?)
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof_rccl
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_rccl -o s -- python $REPO/tools/exp_dist_overhead.py rccl_only prof > /dev/null 2>&1 < /dev/null
timeout 60 python $REPO/tools/rocprof_kernels.py $OUT/prof_rccl "" > $OUT/dist_kernels_rccl_only.txt 2>&1 < /dev/null; tail -25 $OUT/dist_kernels_rccl_only.txt
find $OUT/prof_rccl -name "*.db" -size +8M -delete
