#!/bin/bash
# Diagnostic builds of libfbsmi with other compile-time tile parameters, for A/B timing on the GPU box
# (FBSMI_LIB=tools/variants/<name>.so python tools/bench_em.py ...).  Usage: tools/build_variants.sh "B W" ["B W" ...]
# B = FBSMI_EM_SEGBATCH, W = FBSMI_EM_WAVES (fbs_amd/csrc/fbsmi_em.hip).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
for cfg in "$@"; do
  set -- $cfg
  out=tools/variants/em_b$1_w$2.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -Wno-pass-failed \
    -DFBSMI_EM_SEGBATCH=$1 -DFBSMI_EM_WAVES=$2 -o $out fbs_amd/csrc/fbsmi_prims.hip fbs_amd/csrc/fbsmi_lg.hip \
    fbs_amd/csrc/fbsmi_sde.hip fbs_amd/csrc/fbsmi_nn.hip fbs_amd/csrc/fbsmi_em.hip &
done
wait
ls -la tools/variants
