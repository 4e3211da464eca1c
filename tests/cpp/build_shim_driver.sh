#!/bin/bash
# Builds tests/cpp/shim_driver.cpp against the reference's own headers (where they lie) + the stand-in
# Eigen/PCL/ROS headers, linking libbalm_hip.so.  Output: oracle/_ref/shim_driver (git-ignored; travels).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
REF=${BALM_REFERENCE_ROOT:-/root/reference}
[ -f "$REF/src/benchmark/bavoxel.hpp" ] || { echo "build_shim_driver: $REF not present"; exit 0; }
mkdir -p "$ROOT/oracle/_ref"
g++ -std=c++14 -O2 -w -pthread -I"$ROOT/oracle/compat" -I"$REF/include" -I"$REF/src/benchmark" -I"$ROOT/include" \
    -o "$ROOT/oracle/_ref/shim_driver" "$ROOT/tests/cpp/shim_driver.cpp" \
    -L"$ROOT/balm_amd/lib" -lbalm_hip -ldl -Wl,-rpath,'$ORIGIN/../../balm_amd/lib'
echo "built $ROOT/oracle/_ref/shim_driver"
# the consistency driver's interface (src/simulation headers re-declare the same class names: separate binary)
g++ -std=c++14 -O2 -w -pthread -I"$ROOT/oracle/compat" -I"$REF/src/simulation" -I"$ROOT/include" \
    -o "$ROOT/oracle/_ref/shim_sim_driver" "$ROOT/tests/cpp/shim_sim_driver.cpp" \
    -L"$ROOT/balm_amd/lib" -lbalm_hip -ldl -Wl,-rpath,'$ORIGIN/../../balm_amd/lib'
echo "built $ROOT/oracle/_ref/shim_sim_driver"
# the virtual benchmark's translation unit + include/balm_shim_virtual.hpp (its own class BALM2: separate binary)
g++ -std=c++14 -O3 -w -pthread -I"$ROOT/oracle/compat" -I"$REF/include" -I"$REF/src/benchmark" -I"$ROOT/include" \
    -o "$ROOT/oracle/_ref/shim_virtual_driver" "$ROOT/tests/cpp/shim_virtual_driver.cpp" \
    -L"$ROOT/balm_amd/lib" -lbalm_hip -ldl -Wl,-rpath,'$ORIGIN/../../balm_amd/lib'
echo "built $ROOT/oracle/_ref/shim_virtual_driver"
# the shipped window through BALM2_HIP::associate + damping_iter, timed (bench.py's realworld_end_to_end.cpp_shim leg): a driver of
# the PRODUCT path, so it does not live under oracle/ -- tools/bin/ is git-ignored and travels like the other built files
mkdir -p "$ROOT/tools/bin"
g++ -std=c++14 -O2 -w -pthread -I"$ROOT/oracle/compat" -I"$REF/include" -I"$REF/src/benchmark" -I"$ROOT/include" \
    -o "$ROOT/tools/bin/shim_realworld_e2e" "$ROOT/tests/cpp/shim_realworld_e2e.cpp" \
    -L"$ROOT/balm_amd/lib" -lbalm_hip -ldl -Wl,-rpath,'$ORIGIN/../../balm_amd/lib'
echo "built $ROOT/tools/bin/shim_realworld_e2e"
