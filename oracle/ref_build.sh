#!/bin/bash
# Builds oracle/_ref/libbalm_ref.so from the reference's own sources where they lie under
# /root/reference (read-only; nothing is copied), against the stand-in headers in oracle/compat/.
# Flags follow the reference's CMakeLists.txt:8-9 (-std=c++14 -O3).  No-op when /root/reference is
# absent (the GPU box uses the prebuilt .so that travels with the repo snapshot).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${BALM_REFERENCE_ROOT:-/root/reference}
if [ ! -f "$REF/src/benchmark/bavoxel.hpp" ]; then
  echo "ref_build: $REF not present; keeping any prebuilt oracle/_ref"; exit 0
fi
mkdir -p "$HERE/_ref"
OUT="$HERE/_ref/libbalm_ref.so"
if ! { [ "$OUT" -nt "$HERE/ref_driver.cpp" ] && [ "$OUT" -nt "$HERE/compat/Eigen/Core" ] && [ -z "$BALM_FORCE_BUILD" ]; }; then
  g++ -std=c++14 -O3 -fPIC -pthread -shared -w \
      -I"$HERE/compat" -I"$REF/include" -I"$REF/src/benchmark" \
      -o "$OUT" "$HERE/ref_driver.cpp"
  echo "ref_build: built $OUT"
fi
# the same translation unit with the host's vector ISA switched on (SURVEY 8d's optional extra column of the CPU baseline: the reference's
# CMakeLists.txt:8-9 only says -O3).  -march=x86-64-v3 (AVX2 + FMA + BMI2), not -march=native: the library is built in one container and
# timed on another host, and v3 is what every x86 server of the last decade runs; bench.py checks /proc/cpuinfo before loading it.
OUT="$HERE/_ref/libbalm_ref_v3.so"
if ! { [ "$OUT" -nt "$HERE/ref_driver.cpp" ] && [ "$OUT" -nt "$HERE/compat/Eigen/Core" ] && [ -z "$BALM_FORCE_BUILD" ]; }; then
  g++ -std=c++14 -O3 -march=x86-64-v3 -fPIC -pthread -shared -w -Wl,-Bsymbolic \
      -I"$HERE/compat" -I"$REF/include" -I"$REF/src/benchmark" \
      -o "$OUT" "$HERE/ref_driver.cpp"
  echo "ref_build: built $OUT"
fi
# the consistency / covariance sources (src/simulation) re-declare the same class names: separate object,
# symbols bound locally
OUT="$HERE/_ref/libbalm_ref_sim.so"
if ! { [ "$OUT" -nt "$HERE/ref_sim_driver.cpp" ] && [ "$OUT" -nt "$HERE/compat/Eigen/Core" ] && [ -z "$BALM_FORCE_BUILD" ]; }; then
  g++ -std=c++14 -O3 -fPIC -pthread -shared -w -Wl,-Bsymbolic \
      -I"$HERE/compat" -I"$REF/src/simulation" \
      -o "$OUT" "$HERE/ref_sim_driver.cpp"
  echo "ref_build: built $OUT"
fi
# the virtual benchmark's translation unit (its own class BALM2, the copy BASELINE configs[0..3] name): separate
# object, its main() renamed, symbols bound locally
OUT="$HERE/_ref/libbalm_ref_virtual.so"
if ! { [ "$OUT" -nt "$HERE/ref_virtual_driver.cpp" ] && [ "$OUT" -nt "$HERE/compat/Eigen/Core" ] && [ -z "$BALM_FORCE_BUILD" ]; }; then
  g++ -std=c++14 -O3 -fPIC -pthread -shared -w -Wl,-Bsymbolic \
      -I"$HERE/compat" -I"$REF/include" -I"$REF/src/benchmark" \
      -o "$OUT" "$HERE/ref_virtual_driver.cpp"
  echo "ref_build: built $OUT"
fi
