#!/bin/bash
REPO=$(pwd); TAG=${TAG:-r06b}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 600 python -m pytest tests/test_gpu_strided.py -x -q > $OUT/pytest_strided.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -5 $OUT/pytest_strided.txt
for v in 16_0 24_0 32_0 48_0 16_512 32_512 32_1024; do echo "== ubench_gather_$v"; timeout 120 tools/bin/ubench_gather_$v 2>&1 | grep -v amdgpu.ids; done | tee $OUT/ubench_gather.txt
python -c "
from balm_amd import realworld as rw
rw.write_window_bin(rw.SHIPPED_WINDOW_NPZ, '/tmp/window.bin')"
for i in 1 2; do LD_PRELOAD=$REPO/balm_amd/lib/ab/libbalm_hip_cold.so timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 1 2>&1 | grep -v amdgpu.ids | head -24; done | tee $OUT/cold_trace.txt
