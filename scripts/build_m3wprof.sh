#!/bin/bash
# scratch build of libtapenv with the MACS 3D wave kernel's phase clocks (-DM3W_PROF, tap_macs3_wave.h); see scripts/m3w_phases.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build_prof/m3w
(cd tap-net_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -fno-fast-math -ffp-contract=off -I../../include -I. -DM3W_PROF -c macs3_big.hip -o ../../build_prof/m3w/macs3_big.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $(ls tap-net_amd/csrc/build/*.o | grep -v /macs3_big.o) build_prof/m3w/macs3_big.o -o build_prof/libtapenv_m3wprof.so
