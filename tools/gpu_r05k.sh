#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/${TAG:-r05k}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 200 tools/bin/ubench_h2d 2>&1 | grep 'host_stage.h pipeline\|fresh pageable' | tee $OUT/ubench_h2d.txt
timeout 300 python tools/exp_upload_cadence.py 2>&1 | grep -v amdgpu.ids | tee $OUT/cadence.txt
timeout 300 python -m balm_amd.realworld --npz $REPO/datasets/realworld_w177.npz 2>/dev/null | cut -c330-1300 | tee $OUT/realworld.txt
timeout 300 python tools/bench_upload.py 2>&1 | grep -v 'shipped\|amdgpu.ids' | cut -c1-200 | tee $OUT/uploads.txt
timeout 900 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest all rc=$?"; tail -3 $OUT/pytest_gpu.txt
