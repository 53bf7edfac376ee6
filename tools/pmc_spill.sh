#!/bin/bash
# Counter passes of the spill workload (tools/spill_run.py), one rocprofv3 run per counter set.  Run through gpurun from the repo root.
export TMPDIR=/tmp
out=$PWD/gpurun_out/spill_pmc
mkdir -p $out
python3 -c "from fbs_amd import _lib; _lib.build()"
rocprofv3 -L > $out/avail.txt 2>&1 || true
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o run -- python3 tools/spill_run.py --sweeps 1 > $out/p$i.log 2>&1 || echo "set $i failed: $set"
done
python3 - <<'PY'
import csv, glob, collections, json, os
out = {}
for f in sorted(glob.glob('gpurun_out/spill_pmc/p*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if any(s in k for s in ('k_lg_propQ', 'k_lg_norm<16, 0', 'k_lg_cdf<16, 0', 'k_lg_heaps')):
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    for k, v in acc.items():
        out.setdefault(k, {}).update({c: x / n[k][c] for c, x in v.items()})
        out[k]['launches_counted'] = max(n[k].values())
json.dump(out, open('gpurun_out/spill_pmc/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find $out -name "*counter_collection.csv" -delete
