#!/bin/bash
# Round 5, GPU call: association with lane / persistent-wave segment kernels, mailbox read-backs, arena outputs.
REPO=$(pwd); OUT=$REPO/gpurun_out/${TAG:-r05e}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 900 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_window.py -q -m gpu -x > $OUT/pytest_voxel.txt 2>&1 < /dev/null; echo "pytest voxel/window rc=$?"; tail -15 $OUT/pytest_voxel.txt
timeout 300 python tools/bench_voxel.py --real --no-cpu 2>&1 | grep -v amdgpu.ids | tee $OUT/voxel.txt
timeout 300 python -m balm_amd.realworld --npz $REPO/datasets/realworld_w177.npz > $OUT/realworld.json 2>/dev/null; cut -c330-1300 $OUT/realworld.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_rw -o rw -- python -m balm_amd.realworld --npz $REPO/datasets/realworld_w177.npz > $OUT/realworld_under_rocprof.json 2>/dev/null
cd $REPO
python tools/rocprof_kernels.py $OUT/trace_rw > $OUT/realworld_dispatches.txt 2>&1; rm -rf $OUT/trace_rw
sed -n '/# averages/,$p' $OUT/realworld_dispatches.txt | head -44
