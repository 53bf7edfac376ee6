set -e
export TMPDIR=/tmp
python3 -c "from fbs_amd import _lib; _lib.build(); import oracle; oracle.build()"   # build before any profiler preload exists
out=$PWD/gpurun_out
rm -rf $out/q_prof $out/q_valu
rocprofv3 --kernel-trace --stats --output-format csv -d $out/q_prof -o run -- python3 bench.py --no-cpu-baseline --no-single-chain --batch-scan "" > $out/q_prof.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $out/q_valu -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-chain --batch-scan "" > $out/q_valu.log 2>&1
find $out/q_prof -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/q_prof/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]: print(r['Name'][:60], r['Calls'], r['AverageNs'])
f=glob.glob('gpurun_out/q_valu/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:50]
    if 'k_lg_norm' in k or 'k_lg_prop' in k or 'k_lg_cdf' in k:
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    w=v['SQ_WAVES']
    print(k, {c: round(x/w,1) for c,x in v.items() if c!='SQ_WAVES'})
PY
find $out/q_valu -name "*.csv" -delete
