#!/bin/bash
# Counter passes of the d = 100 toy at 10 000 particles (the fat drift kernel and the small launches around it), one rocprofv3
# run per counter set.  Run through gpurun from the repo root; prints tools/summarise_pmc.py's JSON for the wide kernels.
export TMPDIR=/tmp
export GP100_SWEEPS=1
out=$PWD/gpurun_out/gp100_pmc
rm -rf $out; mkdir -p $out
python3 -c "from fbs_amd import _lib; _lib.build()"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o run -- python3 tools/bench_gp100.py 10000 > $out/p$i.log 2>&1 || echo "set $i failed: $set"
done
python3 tools/summarise_pmc.py $out > $out/summary.json
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/gp100_pmc/summary.json'))
for k, v in d.items():
    if any(s in k for s in ('k_lgw_gemm_fat', 'k_lgw_anc', 'k_lgw_lse', 'k_lg_norm<1, 0, false>', 'k_lg_cdf<1, 0>')) and v.get('dispatches', 0) > 100:
        print(k, json.dumps({c: round(x, 1) for c, x in v.items()}))
PY
find $out -name "*counter_collection.csv" -delete
