#!/bin/bash
# PMC counters of the fused NN-score step kernels at the config-5 shard shape (run on the GPU box from the repo root):
# one rocprofv3 --pmc pass per counter group (never combined with tracing), CSVs under gpurun_out/pmc_em/<group>/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "from fbs_amd import _lib; _lib.build()"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_em/$name -- python3 $ROOT/tools/bench_em.py --shapes ${SHAPES:-c5_shard} --dtype f32 --sliced ${SLICED:-0} --iters 3 > /dev/null 2>&1
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $ROOT
python3 tools/summarise_pmc.py gpurun_out/pmc_em
