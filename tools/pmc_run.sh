#!/bin/bash
# rocprofv3 PMC passes for one bench.py workload (counters in their own runs, kernel-trace only).
# usage: tools/pmc_run.sh <outdir> "<passes>" <bench args...>     passes: subset of "fetch write tcc sq1 sq2 grbm"
set -e
out=$1; passes=$2; shift; shift
BENCH_ARGS="$@"
export TMPDIR=/tmp
mkdir -p $out
run() { # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --no-cpu --steps 2 --warmup 1 $BENCH_ARGS > $out/$name.log 2>&1 || echo "pass $name failed"
}
for p in $passes; do
  case $p in
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
    tcc)   run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum ;;
    sq1)   run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS ;;
    sq2)   run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM ;;
    sq3)   run sq3 SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN SQ_WAIT_INST_LDS SQ_INSTS_SENDMSG SQ_IFETCH SQ_IFETCH_LEVEL ;;
    grbm)  run grbm GRBM_GUI_ACTIVE ;;
    ta)    run ta TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum ;;
    tcp)   run tcp TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum ;;
    tcp2)  run tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum ;;
    list)  rocprofv3 -L > $out/counters.txt 2>&1 || true ;;
  esac
done
