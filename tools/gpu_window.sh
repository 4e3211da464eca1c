#!/bin/bash
# sliding-window map: parity tests, timing, kernel profile
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_gpu_window.py -m gpu -q -x 2>&1 | tail -5
python tools/bench_window.py
mkdir -p gpurun_out/win; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/win -o win -- python $R/tools/bench_window.py > /dev/null 2>&1
cd $R; python tools/rocpd_stats.py gpurun_out/win/win_results.db | sed -e 's/void rocprim::ROCPRIM_400200_NS::detail::trampoline_kernel<rocprim::ROCPRIM_400200_NS::detail::wrapped_/rocprim::/' | cut -c1-150 | head -${1:-30}
