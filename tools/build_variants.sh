#!/bin/bash
# Diagnostic builds of libfbsmi with other compile-time parameters of fbs_amd/csrc/fbsmi_em.hip, for A/B timing on ONE
# GPU box (devices differ by ~10 %: never compare timings from two gpurun calls):
#   FBSMI_LIB=tools/variants/<name>.so python tools/bench_em.py ...
# Usage: tools/build_variants.sh name1 "-DFLAG=..." name2 "-D..." ...   (pairs of name and extra hipcc flags)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-pass-failed"
SRC=${FBSMI_VARIANT_SRC:-fbsmi_em}     # the source file the variants rebuild (FBSMI_VARIANT_SRC=fbsmi_nn for the network kernels)
for f in fbsmi_prims fbsmi_lg fbsmi_sde fbsmi_nn fbsmi_em; do
  if [ ! -f tools/variants/obj/$f.o ] || [ fbs_amd/csrc/$f.hip -nt tools/variants/obj/$f.o ] || [ fbs_amd/csrc/fbsmi_device.h -nt tools/variants/obj/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c -o tools/variants/obj/$f.o fbs_amd/csrc/$f.hip &
  fi
done
wait
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  others=""
  for f in fbsmi_prims fbsmi_lg fbsmi_sde fbsmi_nn fbsmi_em; do [ $f = $SRC ] || others="$others tools/variants/obj/$f.o"; done
  ( /opt/rocm/bin/hipcc $FLAGS $extra -c -o tools/variants/obj/v_$name.o fbs_amd/csrc/$SRC.hip && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/$name.so tools/variants/obj/v_$name.o $others ) &
done
wait
ls tools/variants/*.so
