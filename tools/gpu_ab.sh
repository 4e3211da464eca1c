#!/bin/bash
# A/B of library builds on one box: bench.py --no-cpu with BALM_HIP_LIB pointing at each variant, interleaved
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for rep in 1 2; do
  for lib in balm_amd/lib/libbalm_hip.so balm_amd/lib/ab/*.so; do
    BALM_HIP_LIB=$REPO/$lib timeout 300 python bench.py --no-cpu --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$lib', 'ms/step %.3f' % d['ms_per_step'], 'syrk %.3f' % k['syrk'], 'frac %.3f' % d['roofline']['frac'], 'factors %.3f solve %.3f moments %.3f' % (k['factors'], k['solve'], k['moments']))"
  done
done
