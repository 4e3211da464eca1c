#!/bin/bash
# N1 A/B matrix: lane mapping x block size at 40-point and 6-point runs
mkdir -p gpurun_out/r03c
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "build_clusters" 2>&1 | tail -3
{
for T in 0 1; do for BP in 256 320 384 448 512; do
  echo "# 40-point runs TERMS=$T BP=$BP"; BALM_BUILD_TERMS=$T BALM_BUILD_BP=$BP timeout 300 python tools/bench_cluster_build.py --features 3000 --pts 40 --reps 8
done; done
for BP in 256 384 512; do
  echo "# 6-point runs TERMS=0 BP=$BP"; BALM_BUILD_TERMS=0 BALM_BUILD_BP=$BP timeout 300 python tools/bench_cluster_build.py --features 20000 --pts 6 --reps 8
done
for T in 0 1; do
  echo "# 100-point runs TERMS=$T"; BALM_BUILD_TERMS=$T timeout 300 python tools/bench_cluster_build.py --features 1200 --pts 100
done
echo "# defaults: 6, 40, 1000"
timeout 300 python tools/bench_cluster_build.py --features 20000 --pts 6
timeout 300 python tools/bench_cluster_build.py --features 3000 --pts 40
timeout 300 python tools/bench_cluster_build.py --features 120 --pts 1000
} > gpurun_out/r03c/cluster_build_ab.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r03c/cluster_build_ab.txt'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   %.4f ms  frac %.3f  %.1f Gpt/s err %g'%(d['kernel_ms'],d['frac_of_8TBps'],d['points_per_s']/1e9,d['max_rel_err_vs_host_push']))
    else: print(l)
PY
