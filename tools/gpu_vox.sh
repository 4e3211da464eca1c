#!/bin/bash
# N3 iteration: association parity tests + kernel profile on the shipped window (if its scans travel) and on the synthetic one
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_voxel.py -q -x -s 2>&1 | tail -8
mkdir -p gpurun_out/vox; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/vox -o vox -- python $R/tools/bench_voxel.py --no-cpu --real 2>&1 | grep points
cd $R; python tools/rocpd_stats.py gpurun_out/vox/vox_results.db | sed -e 's/void rocprim::ROCPRIM_400200_NS::detail::trampoline_kernel<rocprim::ROCPRIM_400200_NS::detail::wrapped_/rocprim::/' | cut -c1-110 | head -${1:-22}
python tools/bench_voxel.py --no-cpu | grep points
