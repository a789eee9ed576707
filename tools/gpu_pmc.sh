#!/bin/bash
# rocprofv3 kernel stats + PMC passes of the bench command; summaries under gpurun_out/r2/ (copy into profiles/)
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 -L > $out/counters_available.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $BENCH > $out/prof_stats_bench.json 2> $out/prof_stats.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS_F32" "SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU2 SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/prof_pmc$i -- $BENCH > /dev/null 2> $out/prof_pmc$i.err || echo "pmc set $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/prof_pmc1 /tmp/prof_pmc2 /tmp/prof_pmc3 /tmp/prof_pmc4 /tmp/prof_pmc5 /tmp/prof_pmc6 --filter tiled > $out/pmc_tiled.md
head -60 $out/pmc_tiled.md
