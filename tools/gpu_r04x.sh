#!/bin/bash
# Round 4: plan_syrk lets small windows split K into shorter waves while one round of wave slots is not full -- the whole suite, small windows, other sizes.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04x; mkdir -p $OUT
timeout 300 python tools/bench_small.py > $OUT/small.txt 2>&1 < /dev/null; tail -4 $OUT/small.txt
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.txt
timeout 900 bash tools/gpu_configs.sh > $OUT/other_configs.txt 2>&1 < /dev/null; cat $OUT/other_configs.txt | cut -c1-260
timeout 300 python tools/bench_realshape.py > $OUT/realshape.txt 2>&1 < /dev/null; grep "solve default" $OUT/realshape.txt | cut -c1-200
timeout 600 python tools/bench_window.py > $OUT/window.txt 2>&1 < /dev/null; tail -2 $OUT/window.txt | cut -c1-300
timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>/dev/null < /dev/null > $OUT/b.json; grep -o '"ms_per_step": [0-9.]*' $OUT/b.json | head -1
