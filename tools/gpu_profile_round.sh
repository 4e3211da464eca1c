#!/bin/bash
# Round artifacts: bench line (with CPU leg), rocprofv3 kernel trace stats, PMC passes -> pmc_traffic.json, f64 / sync
# microbenchmarks.  Usage: gpu_profile_round.sh r02f     (copy gpurun_out/<tag>/* into profiles/ afterwards)
TAG=${1:-rXX}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 900 python bench.py 2>gpurun_out/$TAG/bench.err | grep "^{" > gpurun_out/$TAG/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/$TAG/trace -o bench -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | grep "^{" > $REPO/gpurun_out/$TAG/bench_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" ; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/$TAG/pmc/$tag -o p -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
done
cd $REPO
python tools/rocpd_stats.py gpurun_out/$TAG/trace/bench_results.db > gpurun_out/$TAG/kernel_stats.csv
python tools/pmc_summary.py gpurun_out/$TAG/pmc > gpurun_out/$TAG/pmc_summary.csv
python tools/make_pmc_traffic.py gpurun_out/$TAG/pmc_summary.csv $TAG && cp profiles/pmc_traffic.json gpurun_out/$TAG/pmc_traffic.json
rm -rf gpurun_out/$TAG/trace gpurun_out/$TAG/pmc
timeout 120 tools/bin/ubench_f64 > gpurun_out/$TAG/ubench_f64.txt 2>&1
head -10 gpurun_out/$TAG/kernel_stats.csv; head -4 gpurun_out/$TAG/pmc_summary.csv | cut -c1-160; cut -c1-600 gpurun_out/$TAG/bench.json; cat gpurun_out/$TAG/ubench_f64.txt
