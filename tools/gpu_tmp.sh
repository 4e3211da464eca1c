REPO=$(pwd); OUT=$REPO/gpurun_out/r05w5; mkdir -p $OUT; export PYTHONPATH=$REPO TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_voxel.py -q -x 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python tools/bench_voxel.py --real --no-cpu 2>&1 | grep -v amdgpu.ids | tee -a $OUT/voxel.txt; done
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_rw -o rw -- python -m balm_amd.realworld --npz $REPO/datasets/realworld_w177.npz > $OUT/realworld_under_rocprof.json 2>/dev/null; cd $REPO
python tools/rocprof_kernels.py $OUT/trace_rw > $OUT/realworld_dispatches.txt 2>&1; sed -n '/# averages/,$p' $OUT/realworld_dispatches.txt > $OUT/realworld_kernels.txt; rm -rf $OUT/trace_rw
grep -E "k_scan_heads" $OUT/realworld_kernels.txt | cut -c1-120 | head -3
