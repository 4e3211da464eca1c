#!/bin/bash
REPO=$(pwd); TAG=${TAG:-r06d}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
(lscpu | grep -i numa; cat /sys/bus/pci/devices/*/local_cpulist 2>/dev/null | sort | uniq -c | sort -rn | head -5; ls /sys/devices/system/node/ | head; for n in /sys/devices/system/node/node*; do echo "$n: $(cat $n/cpulist)"; done) > $OUT/numa.txt 2>&1; cat $OUT/numa.txt
for v in tools/bin/ubench_gather_*; do echo "== $v"; timeout 120 $v 2>&1 | grep -v amdgpu.ids | grep -v "rep 0"; done > $OUT/ubench_gather.txt; grep -E "==|rep 3|NUMA" $OUT/ubench_gather.txt
