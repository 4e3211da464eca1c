#!/bin/bash
# Thirteenth GPU call of round 4: the tree with the coalesced Gt stores as the default -- the whole suite, the default bench line (with the CPU leg), the other
# BASELINE sizes, the shipped window's step, the sliding-window map.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04m; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null; cut -c1-250 $OUT/bench.json; grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/bench.json; grep -o '"roofline": {[^}]*}' $OUT/bench.json | cut -c1-400; grep -o '"factors": {[^}]*}' $OUT/bench.json
timeout 300 python tools/bench_realshape.py > $OUT/realshape.txt 2>&1 < /dev/null; grep "shipped\|default" $OUT/realshape.txt | cut -c1-220
timeout 600 bash tools/gpu_configs.sh > $OUT/other_configs.txt 2>&1 < /dev/null; tail -12 $OUT/other_configs.txt | cut -c1-250
timeout 600 python tools/bench_window.py > $OUT/window.txt 2>&1 < /dev/null; tail -6 $OUT/window.txt | cut -c1-300
