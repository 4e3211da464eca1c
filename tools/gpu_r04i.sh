#!/bin/bash
# Ninth GPU call of round 4: the phase timeline of k_solve_small; the LM iteration of small windows with the one-workgroup eigen kernel mailing the scalars;
# bench after the ticket fusions' removal.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04i; mkdir -p $OUT
timeout 300 python tools/small_trace.py 8 16 20 24 > $OUT/small_trace.txt 2>&1 < /dev/null; cat $OUT/small_trace.txt | cut -c1-400
timeout 300 python tools/bench_small.py > $OUT/small.txt 2>&1 < /dev/null; tail -5 $OUT/small.txt
timeout 600 python -m pytest tests/test_gpu_solve.py tests/test_gpu_graph.py tests/test_gpu_parity.py -q -m gpu -x > $OUT/pytest_part.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_part.txt
timeout 300 python bench.py --no-cpu --no-strong-ref --steps 100 2>$OUT/b.err < /dev/null > $OUT/bench_nocpu.json; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_nocpu.json | head -1; grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/bench_nocpu.json
