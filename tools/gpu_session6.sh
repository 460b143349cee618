#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s6; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $R/$O/pmc5/g$i -- \
      python3 $R/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu --no-build --brief > $R/$O/pmc5_g$i.log 2>&1 || echo "group $i failed"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL
FETCH_SIZE
WRITE_SIZE
GROUPS
cd $R
python3 tools/pmc_table.py $O/pmc5 > $O/pmc5_summary.csv
grep -i "bilateral2\|row_kernel" $O/pmc5_summary.csv | head -60
