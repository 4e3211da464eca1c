#!/bin/bash
REPO=$(pwd); TAG=${TAG:-r06e}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 900 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_strided.py tests/test_north_star.py tests/test_gpu_parity.py -x -q -k "assoc or voxel or strided or realworld or real or shipped or scans or window" > $OUT/pytest_assoc.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -5 $OUT/pytest_assoc.txt
timeout 300 python tools/bench_voxel.py --real --no-cpu 2>&1 | grep -v amdgpu.ids | tee $OUT/voxel.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_rw -o rw -- python -m balm_amd.realworld --npz $REPO/datasets/realworld_w177.npz > $OUT/realworld_under_rocprof.json 2>/dev/null
cd $REPO
python tools/rocprof_kernels.py $OUT/trace_rw | sed -n '/# averages/,$p' > $OUT/realworld_kernels.txt 2>&1
rm -rf $OUT/trace_rw
head -50 $OUT/realworld_kernels.txt
