#!/bin/bash
# Where the INT8 opt-in pays: bench.py's timed loop at other sizes, default and with BALM_SYRK=int8 (-> profiles/r06_int8_by_size.txt)
for cfg in "64 5000 6" "100 20000 6" "200 10000 6" "200 25000 6" "200 50000 6" "200 100000 6" "200 200000 6" "300 20000 6" "480 20000 6"; do
  set -- $cfg
  for mode in fp64 int8; do
    if [ $mode = int8 ]; then export BALM_SYRK=int8; else unset BALM_SYRK; fi
    echo -n "W=$1 F=$2 $mode: "
    timeout 900 python bench.py --win $1 --features $2 --pts $3 --steps 20 --warmup 3 --no-cpu 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.1f it/s' % (d['ms_per_step'], d['value']), 'syrk %.3f factors %.3f solve %.3f' % (d['kernel_ms_per_step']['syrk'], d['kernel_ms_per_step']['factors'], d['kernel_ms_per_step']['solve']), 'final residual %.6f' % d['final_residual'])"
  done
done
unset BALM_SYRK
