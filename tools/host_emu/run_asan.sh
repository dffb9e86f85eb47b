#!/bin/bash
# Builds the kernel sources for the host (tools/host_emu/hip/hip_runtime.h stands in for the HIP runtime) with
# AddressSanitizer and runs a few problems of every non-LEGACY kernel family at 1 and 4 emulated lanes per group
# (LANES="1 4 8" adds the 8-lane kernels, which read the multipliers of a roll-out step from the emulated LDS).
# The LEGACY line search reads the state of other groups of its wave; groups run one after the other here, so the
# LEGACY families are not emulated.  Usage:  bash tools/host_emu/run_asan.sh [problems per set]
# SAN=undefined runs the same under UndefinedBehaviorSanitizer (shifts, array bounds, signed overflow) instead.
set -e
cd "$(dirname "$0")/../.."
B=${1:-6}
OUT=${TMPDIR:-/tmp}/alilqr_emu
mkdir -p "$OUT"
SAN=${SAN:-address}
g++ -std=c++17 -O1 -g -fsanitize=$SAN -fno-sanitize-recover=all -fno-omit-frame-pointer -ffp-contract=off -I tools/host_emu -I include \
    -x c++ tools/host_emu/driver.cpp -o "$OUT/driver" -lpthread
for n in new_box new_box_active new_lat kin_rect lat_rect circle6 lat_circle fast_box; do
  python3 tools/host_emu/dump_problem.py $n $B "$OUT/$n.bin"
  for L in ${LANES:-1 4}; do
    ALILQR_LANES=$L "$OUT/driver" "$OUT/$n.bin" "$OUT/${n}_$L.out" > "$OUT/${n}_$L.log" 2>&1 || { echo "$n lanes $L: FAILED"; tail -20 "$OUT/${n}_$L.log"; exit 1; }
  done
  first=""; for L in ${LANES:-1 4}; do [ -z "$first" ] && first=$L; cmp "$OUT/${n}_${first}.out" "$OUT/${n}_$L.out"; done && echo "$n: clean under -fsanitize=$SAN, lanes ${LANES:-1 4} give identical bytes"
done
