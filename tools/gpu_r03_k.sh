#!/bin/bash
# r03 call K: soak statistics per build variant: which component makes long developed runs nondeterministic?
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $out
: > $out/soak_k.jsonl
soak() { tag=$1; lib=$2; DSL_LIB=$PWD/dieselfluid_amd/lib/$lib timeout -k 10 150 python tools/soak_developed.py 252 10500 $tag 2>> $out/soak_k.err | grep '^{' >> $out/soak_k.jsonl; tail -1 $out/soak_k.jsonl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['tag'], j['steps'], j['bad_at'], j['last'][-1])"; }
for rep in 1 2 3 4; do
  soak base_$rep libdslsph.so
  soak sb_$rep libdslsph_sb.so
  soak oldahead_$rep libdslsph_oldahead.so
  soak notwo_$rep libdslsph_notwo.so
  soak sboldnotwo_$rep libdslsph_sb_oldahead_notwo.so
done
