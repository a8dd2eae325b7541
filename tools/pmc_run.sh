#!/bin/bash
# rocprofv3 PMC passes for one bench.py workload (counters in their own runs, kernel-trace only).
# usage: tools/pmc_run.sh <outdir> <bench args...>
set -e
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
run() { # name, counters
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --no-cpu --steps 2 --warmup 1 > $out/$name.log 2>&1 || echo "pass $name failed"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD
run grbm GRBM_GUI_ACTIVE
