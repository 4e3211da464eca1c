#!/bin/bash
REPO=$(pwd); TAG=${TAG:-r06c}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 600 python -m pytest tests/test_gpu_strided.py tests/test_gpu_voxel.py tests/test_gpu_window.py -x -q > $OUT/pytest_strided.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_strided.txt
for v in tools/bin/ubench_gather_*; do echo "== $v"; timeout 120 $v 2>&1 | grep -v amdgpu.ids | grep -v "rep 0"; done > $OUT/ubench_gather.txt; grep -E "==|rep 3" $OUT/ubench_gather.txt
python -c "
from balm_amd import realworld as rw
rw.write_window_bin(rw.SHIPPED_WINDOW_NPZ, '/tmp/window.bin')"
for i in 1 2 3; do timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 5 2>&1 | grep -v amdgpu.ids; timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 1 - late 2>&1 | grep -v amdgpu.ids; done | tee $OUT/cpp_e2e.txt
for m in first late; do LD_PRELOAD=$REPO/balm_amd/lib/ab/libbalm_hip_cold.so timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 1 - $m 2>&1 | grep -v amdgpu.ids | head -24; done | tee $OUT/cold_trace.txt
timeout 300 python -m balm_amd.realworld --npz datasets/realworld_w177.npz 2>/dev/null | tee $OUT/realworld.json | cut -c1-1500
timeout 300 python tools/bench_upload.py 2>&1 | grep -v amdgpu.ids > $OUT/uploads.txt; cut -c1-200 $OUT/uploads.txt | grep -v shipped | tail -12
