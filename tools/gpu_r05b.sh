#!/bin/bash
# Round 5, second GPU call: the switch clean-up (persistent solve: live-context share, retry after a timed-out wait), the link's
# yardsticks, and the per-dispatch kernel tables the first call missed (shipped window end to end; covariance at W=200/F=50 000 + PMC).
REPO=$(pwd); OUT=$REPO/gpurun_out/r05b; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_multi.py tests/test_gpu_graph.py -q -m gpu -x > $OUT/pytest_solve.txt 2>&1 < /dev/null; echo "pytest solve/multi/graph rc=$?"; tail -4 $OUT/pytest_solve.txt
timeout 300 tools/bin/ubench_h2d > $OUT/ubench_h2d.txt 2>&1; cat $OUT/ubench_h2d.txt
timeout 600 python tools/bench_upload.py > $OUT/uploads.txt 2>&1 < /dev/null; echo "bench_upload rc=$?"; cut -c1-200 $OUT/uploads.txt | grep -v shipped
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_rw -o rw -- python -m balm_amd.realworld --npz $REPO/datasets/realworld_w177.npz > $OUT/realworld_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_cov -o cov -- python $REPO/tools/bench_cov.py 200 50000 > $OUT/cov_under_rocprof.txt 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_cov/$C -o p -- python $REPO/tools/bench_cov.py 200 50000 > /dev/null 2>&1
done
cd $REPO
python tools/rocprof_kernels.py $OUT/trace_rw > $OUT/realworld_dispatches.txt 2>&1
python tools/rocprof_kernels.py $OUT/trace_cov > $OUT/cov_dispatches.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_cov > $OUT/cov_pmc_summary.csv 2>&1
rm -rf $OUT/trace_rw $OUT/trace_cov $OUT/pmc_cov
sed -n '/# averages/,$p' $OUT/realworld_dispatches.txt | head -60; sed -n '/# averages/,$p' $OUT/cov_dispatches.txt | head -30; head -12 $OUT/cov_pmc_summary.csv | cut -c1-200
timeout 1200 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest all rc=$?"; tail -3 $OUT/pytest_gpu.txt
