#!/bin/bash
REPO=$(pwd); TAG=${TAG:-r06p}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 900 python -m pytest tests/test_gpu_cov.py -x -q > $OUT/pytest_cov.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_cov.txt
timeout 300 python tools/bench_cov.py 2>&1 | grep -v amdgpu.ids | tee $OUT/cov.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_cov -o cov -- python $REPO/tools/bench_cov.py 200 50000 > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_cov/$C -o p -- python $REPO/tools/bench_cov.py 200 50000 > /dev/null 2>&1
done
cd $REPO
python tools/rocprof_kernels.py $OUT/trace_cov | sed -n '/# averages/,$p' > $OUT/cov_kernels_w200_f50000.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_cov > $OUT/cov_pmc_summary.csv 2>&1
rm -rf $OUT/trace_cov $OUT/pmc_cov
head -12 $OUT/cov_kernels_w200_f50000.txt; head -8 $OUT/cov_pmc_summary.csv | cut -c1-220
