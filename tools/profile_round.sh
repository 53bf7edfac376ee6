#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  bash tools/profile_round.sh r01
# Produces gpurun_out/<tag>_*: the bench line, the rocprofv3 kernel-trace statistics of the same
# command, and two separate PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass).
# tools/summarise_profiles.py then condenses them into profiles/.
set -e
export GP100_SWEEPS=3   # keeps the dispatch count of the counter passes small
tag=${1:-r01}
out=$PWD/gpurun_out
export TMPDIR=/tmp
python3 -c "from fbs_amd import _lib; _lib.build(); import oracle; oracle.build()"   # build before any profiler preload exists
python3 bench.py > $out/${tag}_bench_stdout.json 2> $out/${tag}_bench_stderr.log
LEAN="--steps 1 --warmup 1 --no-cpu-baseline --no-single-chain --batch-scan '' --image-steps 0 --sharded-steps 0 --sharded-lg-steps 0"
tail -c 600 $out/${tag}_bench_stdout.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o run -- python3 bench.py --no-cpu-baseline --no-single-chain --batch-scan "" --image-steps 0 --sharded-steps 0 --sharded-lg-steps 0 > $out/${tag}_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-chain --batch-scan "" --image-steps 0 --sharded-steps 0 --sharded-lg-steps 0 > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-chain --batch-scan "" --image-steps 0 --sharded-steps 0 --sharded-lg-steps 0 > $out/${tag}_pmc_write.log 2>&1
# VALU / SALU / memory instruction counts per wave (what actually bounds the step kernels once a few chains share a CU)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $out/${tag}_pmc_valu -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-chain --batch-scan "" --image-steps 0 --sharded-steps 0 --sharded-lg-steps 0 > $out/${tag}_pmc_valu.log 2>&1
# LDS bank-conflict counters on the scan kernels (and on the MFMA drift kernel of the d = 100 toy)
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $out/${tag}_pmc_lds -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-chain --batch-scan "" --image-steps 0 --sharded-steps 0 --sharded-lg-steps 0 > $out/${tag}_pmc_lds.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $out/${tag}_pmc_lds_gp100 -o run -- python3 tools/bench_gp100.py 100 > $out/${tag}_pmc_lds_gp100.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_gp100 -o run -- python3 tools/bench_gp100.py 100 10000 > $out/${tag}_prof_gp100.log 2>&1
find $out/${tag}_prof_gp100 -name "*kernel_trace.csv" -delete
find $out/${tag}_prof $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_lds -name "*.csv" | head -20
# FETCH_SIZE calibration on a known byte count (4 and 16 bytes per lane)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch_calib -o run -- tools/fetch_calib > $out/${tag}_fetch_calib.log 2>&1
# the fused score-network step kernels at config 5's per-GPU shape: kernel trace + counters (tools/pmc_em.sh)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_em -o run -- python3 tools/bench_em.py --shapes c5_shard --dtype f32 --sliced 0 --only finish,concat > $out/${tag}_prof_em.log 2>&1
find $out/${tag}_prof_em -name "*kernel_trace.csv" -delete
tools/pmc_em.sh > $out/${tag}_pmc_em.json 2> $out/${tag}_pmc_em.log
# keep the merge-back small: the raw traces are large
find $out/${tag}_prof -name "*kernel_trace.csv" -delete
