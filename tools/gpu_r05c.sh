#!/bin/bash
# Round 5, third GPU call: why the pinned-ring pipeline sits at 35-40 GB/s (its pieces by thread count / chunk size / ring depth, fresh
# vs warm buffers, against the runtime's own pageable path), k_cov_factors at one wave per SIMD vs its register-pressure variants, and the
# shipped window's kernel table with the association's kernels named.
REPO=$(pwd); OUT=$REPO/gpurun_out/r05c; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
for b in ubench_h2d_t16_n3_c16 ubench_h2d_t8_n3_c16 ubench_h2d_t32_n3_c16 ubench_h2d_t16_n4_c8 ubench_h2d_t16_n3_c32 ubench_h2d_t16_n4_c4; do
  echo "== $b" >> $OUT/ubench_h2d.txt; timeout 200 tools/bin/$b 2>&1 | grep -v '^HIP\|^ROCm\|^Host\|^Librccl' >> $OUT/ubench_h2d.txt
done
grep -v 'kernel pulls\|host memcpy\|one stream' $OUT/ubench_h2d.txt
for v in "" covlast covw2 covlastw2; do
  lib=""; [ -n "$v" ] && lib=$REPO/balm_amd/lib/ab/libbalm_hip_$v.so
  echo "== cov variant '$v'" | tee -a $OUT/cov_ab.txt
  BALM_HIP_LIB=$lib timeout 300 python tools/bench_cov.py 200 50000 2>&1 | tail -1 | tee -a $OUT/cov_ab.txt
  BALM_HIP_LIB=$lib timeout 300 python tools/bench_cov.py 100 2000 2>&1 | tail -1 | tee -a $OUT/cov_ab.txt
  BALM_HIP_LIB=$lib timeout 600 python -m pytest tests/test_gpu_cov.py -q -m gpu -x 2>&1 | tail -1 | tee -a $OUT/cov_ab.txt
done
cd /tmp
for v in "" covlastw2; do
  lib=""; [ -n "$v" ] && lib=$REPO/balm_amd/lib/ab/libbalm_hip_$v.so
  BALM_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_cov$v -o cov -- python $REPO/tools/bench_cov.py 200 50000 > /dev/null 2>&1
  python $REPO/tools/rocprof_kernels.py $OUT/trace_cov$v > $OUT/cov_dispatches_$v.txt 2>&1; rm -rf $OUT/trace_cov$v
  echo "== cov kernels '$v'"; sed -n '/# averages/,$p' $OUT/cov_dispatches_$v.txt | head -12
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_rw -o rw -- python -m balm_amd.realworld --npz $REPO/datasets/realworld_w177.npz > $OUT/realworld_under_rocprof.json 2>/dev/null
cd $REPO
python tools/rocprof_kernels.py $OUT/trace_rw > $OUT/realworld_dispatches.txt 2>&1; rm -rf $OUT/trace_rw
sed -n '/# averages/,$p' $OUT/realworld_dispatches.txt | head -70
