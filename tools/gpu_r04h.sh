#!/bin/bash
# Eighth GPU call of round 4: k_solve_small with the parallel ranking, the poses requested up front and the trailing update's lookahead; the LM iteration's
# scalars mailed by k_feature_eigen's last workgroup; K4 (reduce + assemble) as one launch on contexts without a transport.  Whole suite, small windows, bench.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04h; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_solve.py -q -m gpu -x -k "small" > $OUT/pytest_small.txt 2>&1 < /dev/null; echo "pytest small rc=$?"; tail -3 $OUT/pytest_small.txt
timeout 300 python tools/bench_solve.py 4 8 12 16 20 24 > $OUT/solve_small.txt 2>&1 < /dev/null; cut -c1-30,100-250 $OUT/solve_small.txt
timeout 300 python tools/bench_small.py > $OUT/small.txt 2>&1 < /dev/null; tail -5 $OUT/small.txt
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt
timeout 300 python bench.py --no-cpu --no-strong-ref --steps 100 2>$OUT/b.err < /dev/null > $OUT/bench_nocpu.json; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_nocpu.json | head -1; grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/bench_nocpu.json
timeout 300 python tools/bench_realshape.py > $OUT/realshape.txt 2>&1 < /dev/null; grep "shipped\|default" $OUT/realshape.txt | cut -c1-220
