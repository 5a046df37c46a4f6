#!/bin/bash
# A/B builds of the streaming kernels (masks.hip, transition.hip, transition_macs.hip, rolling.hip): the round-3 form of
# the stream wave (TAP_STREAM_R3), its load order (TAP_STREAM_G2) and store pacing (TAP_STREAM_PACE), the rolling waves'
# late waits (TAP_ROLL_LATEWAIT), the store flavour of the fp32 expansion (TAP_STORE_MODE, tap_masks.h), stream waves
# per workgroup (TAP_TRANS_SW), and the kernarg preload (a variant whose flags contain NOPRELOAD is built without the
# Makefile's -amdgpu-kernarg-preload-count).  Build here (no GPU needed), run on the GPU box:
#   scripts/ab_transition.sh build                -> build_prof/libtapenv_ab_<name>.so
#   scripts/ab_transition.sh run "c2 c3 ..."      -> one bench line per variant and config (value, kernel_us)
# AB_VARIANTS="name:flags ..." overrides the list.
set -e
cd "$(dirname "$0")/.."
VARIANTS=${AB_VARIANTS:-"base: nopreload:-DNOPRELOAD r3:-DTAP_STREAM_R3 r3nopreload:-DTAP_STREAM_R3,-DNOPRELOAD g2:-DTAP_STREAM_G2 pace:-DTAP_STREAM_PACE"}
TUS="masks transition transition_macs rolling"
if [ "$1" = build ]; then
  mkdir -p build_prof
  for v in $VARIANTS; do
    name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }     # commas stand for spaces inside one variant's flags
    mkdir -p build_prof/ab_$name
    for tu in $TUS; do
      pl=""                                               # the Makefile's per-file kernarg preload counts
      case "$flags" in *NOPRELOAD*) ;; *) case $tu in transition|masks) pl="-mllvm -amdgpu-kernarg-preload-count=12";; rolling) pl="-mllvm -amdgpu-kernarg-preload-count=8";; esac;; esac
      ( cd tap-net_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed \
          -fno-fast-math -ffp-contract=off $pl -I../../include -I. $flags -c $tu.hip -o ../../build_prof/ab_$name/$tu.o ) &
    done
  done
  wait
  for v in $VARIANTS; do
    name=${v%%:*}
    others=$(ls tap-net_amd/csrc/build/*.o | grep -v -E "/(masks|transition|transition_macs|rolling)\.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $others build_prof/ab_$name/*.o -o build_prof/libtapenv_ab_$name.so
  done
  ls -la build_prof/*_ab_*.so
else
  for cfg in ${2:-c2}; do
  for rep in 1 2; do
  for v in $VARIANTS; do
    name=${v%%:*}
    TAP_LIB_PATH=$PWD/build_prof/libtapenv_ab_$name.so python bench.py --config $cfg ${AB_FLAGS} --steps ${AB_STEPS:-100} --warmup 5 --no-variants --no-cpu-baseline 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %s  %8.1f M env-steps/s  %s %.3f us  verified %s' % ('$name', '$cfg', d['value']/1e6, d['roofline']['kernel'], d['roofline']['kernel_us'], d['verified']))"
  done; done; done
fi
