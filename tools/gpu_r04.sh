#!/bin/bash
# round-4 GPU runs, one parametrised script: tools/gpu_r04.sh <what> [args...]; output under gpurun_out/r4/
set -o pipefail
mkdir -p gpurun_out/r4
what=$1; shift
case "$what" in
  some_tests)   f=$1; shift; timeout -k 10 1100 python -m pytest $f -x -q -m gpu "$@" > gpurun_out/r4/some_tests.log 2>&1; rc=$?; tail -25 gpurun_out/r4/some_tests.log; exit $rc ;;
  skin_tests)   timeout -k 10 900 python -m pytest tests/test_gpu_skin.py -x -q -m gpu "$@" > gpurun_out/r4/skin_tests.log 2>&1; rc=$?; tail -15 gpurun_out/r4/skin_tests.log; exit $rc ;;
  tests)        timeout -k 10 1100 python -m pytest tests -x -q -m gpu "$@" > gpurun_out/r4/tests.log 2>&1; rc=$?; tail -15 gpurun_out/r4/tests.log; exit $rc ;;
  bench)        tag=$1; shift; timeout -k 10 900 python bench.py "$@" > gpurun_out/r4/bench_$tag.json 2> gpurun_out/r4/bench_$tag.err; rc=$?; tail -3 gpurun_out/r4/bench_$tag.err; cat gpurun_out/r4/bench_$tag.json; exit $rc ;;
  prof)         # tools/gpu_r04.sh prof <tag> <kernel-filter> <bench args...>: rocprofv3 kernel stats + PMC passes of one bench command
    tag=$1; flt=$2; shift 2
    out=$GRAFT_REPO_ROOT/gpurun_out/r4/prof_$tag; mkdir -p $out
    export TMPDIR=/tmp; cd /tmp
    BENCH="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --developed-steps 0 --exact-steps 0 $*"
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_$tag -- $BENCH > $out/bench.json 2> $out/stats.err || exit 1
    cp $(find /tmp/prof_stats_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
    python3 - $(find /tmp/prof_stats_$tag -name "*kernel_trace.csv" | head -1) > $out/long_launches.txt <<'PYEOF'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    big = [round(x, 1) for x in v if x > 30.0]
    print(f"{k}: n={len(v)} total_us={sum(v):.0f} launches>30us: {big[:40]}")
PYEOF
    i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
               "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
               "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS_F32" \
               "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/prof_pmc_${tag}_$i -- $BENCH > /dev/null 2> $out/pmc$i.err || echo "pmc set $i failed"
    done
    python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/prof_pmc_${tag}_1 /tmp/prof_pmc_${tag}_2 /tmp/prof_pmc_${tag}_3 /tmp/prof_pmc_${tag}_4 /tmp/prof_pmc_${tag}_5 --filter "$flt" > $out/pmc.md
    python3 $GRAFT_REPO_ROOT/tools/make_traffic.py $out/pmc.md $out/traffic.json --particles 16003008 --source "profiles/r04_${tag}_pmc.md: rocprofv3 --pmc passes of: bench.py --no-cpu-baseline --developed-steps 0 --exact-steps 0 $*" > /dev/null
    head -30 $out/kernel_stats.csv; exit 0 ;;
  soak)         # tools/gpu_r04.sh soak <n3> <steps> <runs>: repeated long runs must end in the same bits
    n3=$1; steps=$2; runs=$3
    for k in $(seq 1 $runs); do timeout -k 10 400 python tools/soak_developed.py $n3 $steps run$k 2>/dev/null | grep '^{' >> gpurun_out/r4/soak_${n3}.jsonl || exit 1; tail -1 gpurun_out/r4/soak_${n3}.jsonl | cut -c1-400; done
    python3 -c "
import json,sys
r=[json.loads(l) for l in open('gpurun_out/r4/soak_${n3}.jsonl')][-$runs:]
print('digests', [x['state_sha1'] for x in r]); sys.exit(0 if len({x['state_sha1'] for x in r})==1 and all(x['bad_at'] is None for x in r) else 1)"; exit $? ;;
  stats)        # tools/gpu_r04.sh stats <tag> <bench args...>: rocprofv3 kernel stats of one bench command, launches > 30 us only
    tag=$1; shift
    out=$GRAFT_REPO_ROOT/gpurun_out/r4/stats_$tag; mkdir -p $out
    export TMPDIR=/tmp; cd /tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --developed-steps 0 --exact-steps 0 "$@" > $out/bench.json 2> $out/stats.err || exit 1
    cp $(find /tmp/stats_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
    python3 - $(find /tmp/stats_$tag -name "*kernel_trace.csv" | head -1) > $out/long_launches.txt <<'PYEOF'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    big = [x for x in v if x > 30.0]
    print(f"{k}: n={len(v)} total_us={sum(v):.0f} long={len(big)} mean_long_us={(sum(big)/len(big) if big else 0):.1f} min_long={(min(big) if big else 0):.1f}")
PYEOF
    cat $out/long_launches.txt | head -14; exit 0 ;;
  ab)           # tools/gpu_r04.sh ab <tag> "<variant names>" <reps> [bench args]: builds of tools/build_variant.sh, interleaved, in one call
    tag=$1; names=$2; reps=$3; shift 3
    : > gpurun_out/r4/ab_$tag.jsonl
    for r in $(seq 1 $reps); do for v in $names; do
      DSL_LIB=$GRAFT_REPO_ROOT/variants/libdslsph_$v.so timeout -k 10 300 python bench.py --no-cpu-baseline --developed-steps 0 --exact-steps 0 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=r.get('skin') or {}
print(json.dumps({'variant': '$v', 'value': r['value'], 'ms_per_step': r['ms_per_step'], 'rebuilds': k.get('rebuilds_in_timed_region'), 'density_ms': r['kernels_ms'].get('density'), 'force_ms': r['kernels_ms'].get('force_integrate'), 'developed': (r.get('developed') or {}).get('value'), 'drifted': (r.get('drifted') or {}).get('value')}))" >> gpurun_out/r4/ab_$tag.jsonl || exit 1
      tail -1 gpurun_out/r4/ab_$tag.jsonl
    done; done; exit 0 ;;
  opt_ab)       # tools/gpu_r04.sh opt_ab <tag> "<opt settings, e.g. grid_oversub=1 grid_oversub=4>" <reps> [bench args]: one build, library options interleaved
    tag=$1; opts=$2; reps=$3; shift 3
    : > gpurun_out/r4/opt_ab_$tag.jsonl
    for r in $(seq 1 $reps); do for o in $opts; do
      timeout -k 10 500 python bench.py --no-cpu-baseline --exact-steps 0 --opt $o "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=r.get('skin') or {}; d=r.get('developed') or {}
print(json.dumps({'opt': '$o', 'value': r['value'], 'ms_per_step': r['ms_per_step'], 'rebuilds': k.get('rebuilds_in_timed_region'), 'density_ms': r['kernels_ms'].get('density'), 'force_ms': r['kernels_ms'].get('force_integrate'), 'developed': d.get('value'), 'developed_kernels': d.get('kernels_ms'), 'drifted': (r.get('drifted') or {}).get('value')}))" >> gpurun_out/r4/opt_ab_$tag.jsonl || exit 1
      tail -1 gpurun_out/r4/opt_ab_$tag.jsonl
    done; done; exit 0 ;;
  skin_sweep)   # tools/gpu_r04.sh skin_sweep <tag> "<s values>" "<predict values>" [bench args]: the bench line per (s, predict)
    tag=$1; ss=$2; ps=$3; shift 3
    : > gpurun_out/r4/skin_sweep_$tag.jsonl
    for s in $ss; do for p in $ps; do
      timeout -k 10 300 python bench.py --no-cpu-baseline --developed-steps 0 --exact-steps 0 --skin $s --opt skin_predict=$p "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=r['skin']
print(json.dumps({'s': $s, 'predict': $p, 'value': r['value'], 'ms_per_step': r['ms_per_step'], 'rebuilds': k['rebuilds_in_timed_region'], 'tau_steps': k['tau_steps_last_rebuild'], 'fields_walked': k['fields_walked_per_particle'], 'density_ms': r['kernels_ms']['density'], 'force_ms': r['kernels_ms']['force_integrate'], 'suspensions': k['suspensions']}))" >> gpurun_out/r4/skin_sweep_$tag.jsonl || exit 1
      tail -1 gpurun_out/r4/skin_sweep_$tag.jsonl
    done; done; exit 0 ;;
  final)        # tools/gpu_r04.sh final <tag>: the round's evidence on one box -- default bench line, kernel stats + PMC of the
                # bench command, the other configurations' lines; summaries under gpurun_out/r4/final_<tag>/
    tag=$1; out=gpurun_out/r4/final_$tag; mkdir -p $out
    timeout -k 10 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default bench rc=$?"
    $0 prof $tag "k_force_list,k_density_list,k_list_build,k_density_pair,k_scatter,k_cell_rank" --steps 100 --warmup 20 > $out/prof.log 2>&1; echo "prof rc=$?"
    cp -r gpurun_out/r4/prof_$tag/* $out/ 2>/dev/null
    timeout -k 10 300 python bench.py --skin 0 --no-cpu-baseline --developed-steps 0 --exact-steps 0 > $out/wcsph_16m_plain_bench.json 2>/dev/null; echo "plain rc=$?"
    timeout -k 10 300 python bench.py --n3 100 --steps 50 --warmup 10 --no-cpu-baseline --developed-steps 0 --exact-steps 0 > $out/wcsph_1m_bench.json 2>/dev/null; echo "1m rc=$?"
    timeout -k 10 300 python bench.py --n3 400 --steps 20 --warmup 5 --no-cpu-baseline --developed-steps 0 --exact-steps 0 > $out/wcsph_64m_bench.json 2>/dev/null; echo "64m rc=$?"
    timeout -k 10 400 python bench.py --method pcisph --n3 160 --steps 20 --warmup 5 --no-cpu-baseline > $out/pcisph_4m_bench.json 2>/dev/null; echo "pcisph 4m rc=$?"
    timeout -k 10 600 python bench.py --method pcisph --n3 400 --extra-terms --steps 10 --warmup 3 --no-cpu-baseline --drift-steps 200 > $out/pcisph_64m_xsph_cohesion_bench.json 2>/dev/null; echo "pcisph 64m rc=$?"
    for mode in "--native --nccl --no-timing" "--native --nccl --no-timing --no-overlap"; do
      timeout -k 10 200 python tools/slab_periodic_bench.py $mode --steps 200 --warmup 20 2>/dev/null | grep "^{" >> $out/slab_runs.jsonl
    done; echo "slab rc=$?"
    python3 tools/benchline.py $out/*.json 2>/dev/null | cut -c1-220; exit 0 ;;
  *) echo "unknown: $what"; exit 2 ;;
esac
