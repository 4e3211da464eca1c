#!/bin/bash
# The C++ end-to-end leg (tools/bin/shim_realworld_e2e) six times as the scheduler places its reader thread, then with the process pinned to each
# NUMA node's CPUs in turn; every line says on which nodes the 177 clouds ended up (clouds_on_nodes=n0/n1/..) -> profiles/r06_cpp_leg_numa.txt
REPO=$(pwd); export PYTHONPATH=$REPO
python -c "
from balm_amd import realworld as rw
rw.write_window_bin(rw.SHIPPED_WINDOW_NPZ, '/tmp/window.bin')" 2>/dev/null
B=$(rocm-smi --showbus 2>/dev/null | grep -o "0000:[0-9A-Fa-f:.]*" | head -1 | tr A-F a-f)
echo "gpu $B on node $(cat /sys/bus/pci/devices/$B/numa_node), cpus $(cat /sys/bus/pci/devices/$B/local_cpulist)"
cut_() { grep SHIM_E2E | sed 's/.*clouds_on_nodes=/clouds_on_nodes=/; s/scans=.*cold_lm=[0-9.]* //; s/max_rot.*//'; }
for rep in 1 2 3 4 5 6; do echo -n "as scheduled:       "; timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 5 2>&1 | cut_; done
for rep in 1 2 3; do
  for n in $(ls /sys/devices/system/node | grep -o "node[0-9]*" | sed s/node//); do
    echo -n "taskset to node $n:  "; taskset -c $(cat /sys/devices/system/node/node$n/cpulist) timeout 120 tools/bin/shim_realworld_e2e /tmp/window.bin 5 2>&1 | cut_
  done
done
