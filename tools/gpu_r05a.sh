#!/bin/bash
# Round 5, first GPU call: the pinned-ring uploads (new tests + tools/bench_upload.py), then the kernel tables round 4 left stale:
# the shipped window end to end (association + LM) and balm_pose_covariance at W=200/F=50 000 with FETCH/WRITE counters.
REPO=$(pwd); OUT=$REPO/gpurun_out/r05a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_north_star.py -q -m gpu -x -k "fill_callback or timing_slots or full_size or build_clusters_matches or realworld or cpp_shim" > $OUT/pytest_new.txt 2>&1 < /dev/null; echo "pytest new rc=$?"; tail -4 $OUT/pytest_new.txt
timeout 600 python tools/bench_upload.py > $OUT/uploads.txt 2>&1 < /dev/null; echo "bench_upload rc=$?"; cut -c1-400 $OUT/uploads.txt
timeout 1200 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest all rc=$?"; tail -3 $OUT/pytest_gpu.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_rw -o rw -- python -m balm_amd.realworld --npz $REPO/datasets/realworld_w177.npz > $OUT/realworld_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_cov -o cov -- python $REPO/tools/bench_cov.py > $OUT/cov_under_rocprof.txt 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_cov/$C -o p -- python $REPO/tools/bench_cov.py > /dev/null 2>&1
done
cd $REPO
python tools/rocpd_stats.py $OUT/trace_rw/rw_results.db > $OUT/realworld_kernel_stats.csv 2>&1
python tools/rocpd_stats.py $OUT/trace_cov/cov_results.db > $OUT/cov_kernel_stats.csv 2>&1
python tools/pmc_summary.py $OUT/pmc_cov > $OUT/cov_pmc_summary.csv 2>&1
rm -rf $OUT/trace_rw $OUT/trace_cov $OUT/pmc_cov
timeout 300 python tools/bench_cov.py > $OUT/cov.txt 2>&1; cat $OUT/cov.txt
head -30 $OUT/realworld_kernel_stats.csv | cut -c1-200; head -14 $OUT/cov_kernel_stats.csv | cut -c1-200; head -12 $OUT/cov_pmc_summary.csv | cut -c1-200
