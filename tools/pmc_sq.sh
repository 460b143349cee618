#!/bin/bash
# SQ / TCP / TCC / TA counter passes of the headline step: one rocprofv3 --pmc run
# per counter group (8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE cannot share a pass), never
# together with a trace domain other than --kernel-trace.  Usage: tools/pmc_sq.sh OUTDIR [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/${1:-gpurun_out/pmc}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $OUT/g$i -- \
      python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-build --brief "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed" >> $OUT/errors.log
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_LDS_DATA_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL
TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr
FETCH_SIZE
WRITE_SIZE
GROUPS
cd $R
python3 tools/pmc_table.py $OUT > $OUT/summary.csv 2>> $OUT/errors.log
