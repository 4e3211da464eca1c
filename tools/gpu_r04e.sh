#!/bin/bash
# Fifth GPU call of round 4: the overlapped evaluation (K2's slabs beside K3's rounds) on the GPU for the first time -- its parity test, the whole
# suite with it on (default), and bench.py with it on / off, alternating.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "overlapped or bench_size" > $OUT/pytest_overlap.txt 2>&1 < /dev/null; echo "pytest overlap rc=$?"; tail -15 $OUT/pytest_overlap.txt
for rep in 1 2; do
  for ovl in 1 0; do
    BALM_OVERLAP=$ovl timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>$OUT/b.err < /dev/null > $OUT/b_$ovl.json
    echo "rep $rep overlap=$ovl  $(grep -o '"ms_per_step": [0-9.]*' $OUT/b_$ovl.json | head -1)  $(grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/b_$ovl.json)" | tee -a $OUT/overlap_ab.txt
  done
done
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null; cut -c1-300 $OUT/bench.json; grep -o '"roofline": {[^}]*}' $OUT/bench.json
