#!/bin/bash
# Sixteenth GPU call of round 4: k_world_moments with its passes over the window unrolled and all their loads issued first (BALM_MOMENTS_UNROLL=0: the rolled loop).
REPO=$(pwd); OUT=$REPO/gpurun_out/r04p; mkdir -p $OUT
: > $OUT/moments_ab.txt
for rep in 1 2; do
  for u in 1 0; do
    BALM_MOMENTS_UNROLL=$u timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>/dev/null < /dev/null > $OUT/b.json
    echo "rep $rep  unrolled=$u  $(grep -o '"ms_per_step": [0-9.]*' $OUT/b.json | head -1)  $(grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/b.json | cut -c1-150)  $(grep -o '"moments": {[^}]*}' $OUT/b.json | cut -c1-160)" | tee -a $OUT/moments_ab.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_north_star.py -q -m gpu -x > $OUT/pytest.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $OUT/pytest.txt
timeout 300 python tools/bench_small.py > $OUT/small.txt 2>&1 < /dev/null; tail -4 $OUT/small.txt
