#!/bin/bash
# The -DTAP_PROF (clock stamps) and -DTAP_PROF_SWITCH (decomposition switches) builds of transition.hip that
# scripts/decompose_step.py loads: build_prof/libtapenv_PROF.so, build_prof/libtapenv_PROF_SWITCH.so.
# Needs the product objects (make -C tap-net_amd/csrc) for the other translation units.
set -e
cd "$(dirname "$0")/../tap-net_amd/csrc"
mkdir -p ../../build_prof
for v in PROF PROF_SWITCH; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -fno-fast-math \
      -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=12 -I../../include -I. -DTAP_$v -c transition.hip -o ../../build_prof/transition_$v.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $(ls build/*.o | grep -v /transition.o) ../../build_prof/transition_$v.o \
      -o ../../build_prof/libtapenv_$v.so ) &
done
wait
ls -la ../../build_prof/libtapenv_PROF*.so
