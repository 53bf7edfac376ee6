for cfg in "4194304 16 1" "4194304 16 4" "1048576 32 4"; do set -- $cfg
python bench.py --nparticles $1 --nsteps $2 --nchains $3 --no-cpu-baseline --no-single-chain --batch-scan "" --steps 5 --warmup 2 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('N=$1 T=$2 C=$3 ms/sweep', round(j['ms_per_step'],3), 'value %.3g'%j['value'], {k:round(v,2) for k,v in r['kernels_us'].items()}, 'prop GB/s', round(r['achieved'],1), 'sweep GB/s', round(r['whole_sweep_GBps'],1))
    elif 'rror' in l: print(l.strip())
"
done
