#!/bin/bash
# round 5, the chain workgroup without the hop on its critical loop: A/B of kernels_solve.hip built with -DCH_STAGE=0 / 2
REPO=$(pwd); OUT=$REPO/gpurun_out/${TAG:-r05s}; mkdir -p $OUT; export PYTHONPATH=$REPO
for L in ${LIBS:-base st2}; do
  if [ $L = base ]; then unset BALM_HIP_LIB; else export BALM_HIP_LIB=$REPO/balm_amd/lib/ab/libbalm_hip_$L.so; fi
  echo "=== $L" | tee -a $OUT/chain.txt
  BALM_SOLVE_TRACE=1 timeout 300 python tools/chain_check.py 8 9 16 17 24 33 48 64 100 144 177 200 2>&1 | grep -v amdgpu.ids >> $OUT/chain.txt
  grep -E "^W= (100|177|200)|failures" $OUT/chain.txt | tail -7
  timeout 300 python tools/bench_solve.py 100 177 200 256 300 500 700 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench_solve_$L.txt | cut -c1-200
done
for L in ${LIBS:-base st2}; do
  [ $L = base ] && continue
  export BALM_HIP_LIB=$REPO/balm_amd/lib/ab/libbalm_hip_$L.so
  timeout 600 python -m pytest tests/test_gpu_solve.py -q -x 2>&1 | tail -3 | tee -a $OUT/pytest_$L.txt
done
