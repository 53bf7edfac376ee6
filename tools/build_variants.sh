#!/bin/bash
# Diagnostic builds of libfbsmi with other compile-time parameters of fbs_amd/csrc/fbsmi_em.hip, for A/B timing on ONE
# GPU box (devices differ by ~10 %: never compare timings from two gpurun calls):
#   FBSMI_LIB=tools/variants/<name>.so python tools/bench_em.py ...
# Usage: tools/build_variants.sh name1 "-DFLAG=..." name2 "-D..." ...   (pairs of name and extra hipcc flags)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-pass-failed"
for f in fbsmi_prims fbsmi_lg fbsmi_sde fbsmi_nn; do
  if [ ! -f tools/variants/obj/$f.o ] || [ fbs_amd/csrc/$f.hip -nt tools/variants/obj/$f.o ] || [ fbs_amd/csrc/fbsmi_device.h -nt tools/variants/obj/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c -o tools/variants/obj/$f.o fbs_amd/csrc/$f.hip &
  fi
done
wait
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $extra -c -o tools/variants/obj/em_$name.o fbs_amd/csrc/fbsmi_em.hip && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/$name.so tools/variants/obj/em_$name.o \
      tools/variants/obj/fbsmi_prims.o tools/variants/obj/fbsmi_lg.o tools/variants/obj/fbsmi_sde.o tools/variants/obj/fbsmi_nn.o ) &
done
wait
ls tools/variants/*.so
