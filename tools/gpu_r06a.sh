#!/bin/bash
# round 6, first call: the strided point-container entries (parity), the C++ end-to-end leg warm / cold, the cold call's breakdown
REPO=$(pwd); TAG=${TAG:-r06a}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
nproc > $OUT/nproc.txt; lscpu | head -20 >> $OUT/nproc.txt
timeout 900 python -m pytest tests/test_gpu_strided.py tests/test_gpu_voxel.py tests/test_gpu_window.py -x -q > $OUT/pytest_strided.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -5 $OUT/pytest_strided.txt
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_north_star.py -x -q -k "shim or driver" > $OUT/pytest_shims.txt 2>&1 < /dev/null; echo "pytest shims rc=$?"; tail -3 $OUT/pytest_shims.txt
timeout 300 python -m balm_amd.realworld --npz datasets/realworld_w177.npz 2>/dev/null | tee $OUT/realworld.json | cut -c1-3000
python - <<'P' > $OUT/window.bin.log 2>&1
from balm_amd import realworld as rw
rw.write_window_bin(rw.SHIPPED_WINDOW_NPZ, "/tmp/window.bin")
P
for i in 1 2 3; do timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 5 2>&1 | grep -v amdgpu.ids | tee -a $OUT/cpp_e2e.txt; done
LD_PRELOAD=$REPO/balm_amd/lib/ab/libbalm_hip_cold.so timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 1 2>&1 | grep -v amdgpu.ids | tee $OUT/cold_trace.txt | head -60
