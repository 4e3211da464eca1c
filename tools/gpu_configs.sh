#!/bin/bash
# other BASELINE configs on one GPU (bench.py with explicit sizes), one JSON line each
for cfg in "20 20 40" "64 5000 6" "200 25000 6" "177 2281 6" "200 200000 6" "480 2000 6" "500 5000 6"; do
  set -- $cfg
  echo -n "W=$1 F=$2: "
  timeout 900 python bench.py --win $1 --features $2 --pts $3 --steps 10 --warmup 2 --no-cpu 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.1f it/s' % (d['ms_per_step'], d['value']), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, 'syrk frac %.3f' % d['roofline']['frac'])"
done
