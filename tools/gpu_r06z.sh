#!/bin/bash
# Round 6's evidence on the tree the round stops at (one gpurun call): the whole GPU suite, smoke(), tools/gpu_profile_round.sh (default
# bench line with the CPU, real-data -- Python AND C++ -- legs; rocprofv3 kernel table; the three PMC passes -> pmc_traffic.json; ubench_f64),
# the loopback-2 bench line, then the kernel tables of the paths beside the headline: the shipped window end to end under rocprofv3, the
# covariance at W = 200 / F = 50 000 with FETCH / WRITE counters, the window map, the uploads (one context and eight shards), the C++ leg
# three times cold and warm, the cold call's breakdown; last, tools/gpu_i8prof.sh: everything about BALM_SYRK=int8 (accuracy, kernel tables with and
# without the switch, counters, the whole suite with the switch exported) and the covariance stage under it.  Every rocprofv3 run sits under `timeout`.
REPO=$(pwd); TAG=${TAG:-r06z}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 < /dev/null; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 2400 bash tools/gpu_profile_round.sh $TAG < /dev/null
BALM_BENCH_LOOPBACK=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu 2>/dev/null | grep "^{" > $OUT/bench_loopback2.json; echo "loopback bench rc=$?"
timeout 300 python tools/bench_upload.py 2>&1 | grep -v amdgpu.ids > $OUT/uploads.txt; cut -c1-200 $OUT/uploads.txt | grep -v shipped
timeout 600 python tools/bench_upload.py --shards 8 2>&1 | grep -v amdgpu.ids | tee $OUT/upload_shards8.txt
timeout 300 python tools/bench_voxel.py --real --no-cpu 2>&1 | grep -v amdgpu.ids | tee $OUT/voxel.txt
timeout 300 python tools/bench_window.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/window.txt
timeout 300 python tools/bench_cov.py 2>&1 | grep -v amdgpu.ids | tee $OUT/cov.txt
python -c "
from balm_amd import realworld as rw
rw.write_window_bin(rw.SHIPPED_WINDOW_NPZ, '/tmp/window.bin')"
for i in 1 2 3; do timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 5 2>&1 | grep -v amdgpu.ids; timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 1 - late 2>&1 | grep -v amdgpu.ids; done | tee $OUT/cpp_end_to_end.txt
[ -f $REPO/balm_amd/lib/ab/libbalm_hip_cold.so ] && LD_PRELOAD=$REPO/balm_amd/lib/ab/libbalm_hip_cold.so timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 1 2>&1 | grep -v amdgpu.ids > $OUT/cold_call.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_rw -o rw -- python -m balm_amd.realworld --npz $REPO/datasets/realworld_w177.npz > $OUT/realworld_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_cov -o cov -- python $REPO/tools/bench_cov.py 200 50000 > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_cov/$C -o p -- python $REPO/tools/bench_cov.py 200 50000 > /dev/null 2>&1
done
cd $REPO
python tools/rocprof_kernels.py $OUT/trace_rw | sed -n '/# averages/,$p' > $OUT/realworld_kernels.txt 2>&1
python tools/rocprof_kernels.py $OUT/trace_cov | sed -n '/# averages/,$p' > $OUT/cov_kernels_w200_f50000.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_cov > $OUT/cov_pmc_summary.csv 2>&1
rm -rf $OUT/trace_rw $OUT/trace_cov $OUT/pmc_cov
head -45 $OUT/realworld_kernels.txt; head -14 $OUT/cov_kernels_w200_f50000.txt; head -6 $OUT/cov_pmc_summary.csv | cut -c1-200
TAG=$TAG timeout 1800 bash tools/gpu_i8prof.sh < /dev/null
BALM_SYRK=int8 timeout 300 python tools/bench_cov.py 2>&1 | grep -v amdgpu.ids | tee $OUT/int8/cov_under_int8.txt
