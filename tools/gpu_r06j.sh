#!/bin/bash
REPO=$(pwd); TAG=${TAG:-r06j}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 1200 python -m pytest tests/test_gpu_multi.py -x -q -s > $OUT/pytest_multi.txt 2>&1 < /dev/null; echo "pytest rc=$?"; grep -E "shards:|passed|failed|Error|assert" $OUT/pytest_multi.txt | tail -30
timeout 600 python tools/bench_upload.py --shards 8 2>&1 | grep -v amdgpu.ids | tee $OUT/upload_shards8.txt
