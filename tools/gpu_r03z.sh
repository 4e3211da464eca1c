#!/bin/bash
# round 3, late: the helpers' operand staging (global_load_lds) and the 2x2 macro-tile helpers -- tests, the solve by window, helper stats
timeout 600 python -m pytest tests/test_gpu_solve.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/bench_solve.py 256 300 350 400 500 600 700 800 2>&1 | cut -c1-30,100-250
BALM_CHAIN_MACRO=0 timeout 600 python tools/bench_solve.py 256 300 350 400 500 600 700 800 2>&1 | cut -c1-30,100-250 | sed 's/^/macro=0 /'
timeout 300 python tools/chain_helpers.py 2>&1 | tail -6
BALM_CHAIN_MACRO=0 timeout 300 python tools/chain_helpers.py 2>&1 | tail -6 | sed 's/^/macro=0 /'
