#!/bin/bash
# Round-3 evidence on the GPU box (summaries land in gpurun_out/r3/final/; copy what is judged into profiles/):
#   1 kernel stats + PMC passes of the bench command   2 the same for the PCISPH configuration
#   3 bench lines of the other configurations          4 the default bench line
set -o pipefail
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r3/final
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
pmc_sets=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
          "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
          "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS_F32"
          "SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU"
          "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE")
profile() {  # profile <tag> <steps> <bench args...>   (bench.py's exact / developed segments are switched off here)
  local tag=$1 steps=$2; shift 2
  local cmd="python3 $R/bench.py --no-cpu-baseline --developed-steps 0 --exact-steps 0 --drift-steps 0 $*"
  rm -rf /tmp/prof_${tag}_stats
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_stats -- $cmd --steps 20 --warmup 5 > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err
  cp $(find /tmp/prof_${tag}_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
  echo "$tag kernel stats done"
  local dirs="" i=0
  for set in "${pmc_sets[@]}"; do
    i=$((i+1)); rm -rf /tmp/prof_${tag}_pmc$i
    rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/prof_${tag}_pmc$i -- $cmd --steps $steps --warmup 1 > /dev/null 2> $out/${tag}_pmc$i.err || echo "$tag pmc set $i failed"
    dirs="$dirs /tmp/prof_${tag}_pmc$i"
  done
  python3 $R/tools/pmc_summary.py $dirs --filter tiled,k_density_pair > $out/${tag}_pmc.md
  echo "$tag pmc done"
}
profile wcsph16m 3
profile pcisph4m 3 --method pcisph --n3 160
cd $R
timeout -k 10 400 python bench.py --method pcisph --n3 160 --steps 20 --warmup 5 --no-cpu-baseline > $out/pcisph_4m_bench.json 2> $out/pcisph_4m.err; echo "pcisph 4m rc=$?"
timeout -k 10 600 python bench.py --method pcisph --n3 400 --extra-terms --steps 10 --warmup 3 --no-cpu-baseline --drift-steps 200 > $out/pcisph_64m_xsph_cohesion_bench.json 2> $out/pcisph_64m.err; echo "pcisph 64m rc=$?"
# the 4M PCISPH scene over 1500 steps (the predictor drifts away from the particles; the queries get bins of their own)
timeout -k 10 400 python tools/pci_long_run.py 160 1500 100 > $out/pcisph_4m_long_run.jsonl 2> $out/pcisph_4m_long_run.err; echo "pcisph long run rc=$?"
# kernel trace of 20 binned steps from the state after 400 steps
timeout -k 10 300 python tools/pci_drifted_state.py save 160 400 /tmp/pci400.npz > /dev/null
(cd /tmp && rm -rf /tmp/prof_pcidrift && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pcidrift -- python3 $R/tools/pci_drifted_state.py run /tmp/pci400.npz 20 1 > $out/pcisph_4m_drifted_step400_run.json 2> $out/pcisph_4m_drifted.err; cp $(find /tmp/prof_pcidrift -name "*kernel_stats.csv" | head -1) $out/pcisph_4m_drifted_step400_kernel_stats.csv); echo "pcisph drifted trace rc=$?"
timeout -k 10 300 python bench.py --n3 100 --steps 50 --warmup 10 --no-cpu-baseline --developed-steps 0 > $out/wcsph_1m_bench.json 2>/dev/null; echo "1m rc=$?"
timeout -k 10 300 python bench.py --n3 400 --steps 10 --warmup 3 --no-cpu-baseline --developed-steps 0 > $out/wcsph_64m_bench.json 2>/dev/null; echo "64m rc=$?"
timeout -k 10 300 python bench.py --math exact --steps 5 --warmup 2 --no-cpu-baseline --developed-steps 0 > $out/wcsph_16m_exact_bench.json 2>/dev/null; echo "exact rc=$?"
: > $out/slab_runs.jsonl
for mode in "--native --nccl --no-timing" "--native --nccl --no-timing --no-overlap"; do
  timeout -k 10 200 python tools/slab_periodic_bench.py $mode --steps 200 --warmup 20 2>> $out/slab_runs.err | grep '^{' >> $out/slab_runs.jsonl
done
timeout -k 10 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default bench rc=$?"
python - <<'PY'
import json, glob, os
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r3/final'
for f in sorted(glob.glob(out + '/*bench*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['value'], j['ms_per_step'], j.get('kernels_ms'), (j.get('developed') or {}).get('value'))
    except Exception as e:
        print(os.path.basename(f), 'unreadable', e)
PY
