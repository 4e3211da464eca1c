#!/bin/bash
REPO=$(pwd); TAG=${TAG:-r06l}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 900 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_strided.py tests/test_north_star.py tests/test_gpu_parity.py -x -q -k "assoc or voxel or strided or realworld or real or shipped or scans or window" > $OUT/pytest_assoc.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_assoc.txt
timeout 300 python tools/bench_voxel.py --real --no-cpu 2>&1 | grep -v amdgpu.ids | tee $OUT/voxel.txt
python -c "
from balm_amd import realworld as rw
rw.write_window_bin(rw.SHIPPED_WINDOW_NPZ, '/tmp/window.bin')"
LD_PRELOAD=$REPO/balm_amd/lib/ab/libbalm_hip_cold.so timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 2 2>&1 | grep -v amdgpu.ids | tail -32 | tee $OUT/warm_trace.txt
for i in 1 2 3; do timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 5 2>&1 | grep -v amdgpu.ids; done | tee $OUT/cpp_e2e.txt
timeout 300 python -m balm_amd.realworld --npz datasets/realworld_w177.npz 2>/dev/null | tee $OUT/realworld.json | cut -c1-900
