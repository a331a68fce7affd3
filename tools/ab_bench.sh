#!/bin/bash
# Same-box A/B of library builds: every variant under variants/libeccx_<name>.so (and the tree's own
# build as "head") runs the given workloads in rotation, ROUNDS times; prints the median kernel time
# per (workload, variant).  Boxes of the pool differ by +-3 %, so only same-box numbers compare.
#   usage (inside gpurun): bash tools/ab_bench.sh "p256r1_var_2^20 p384r1_var_2^19" "head coz even" [rounds]
#   BENCH_ARGS (environment) is appended to every bench.py call, e.g. BENCH_ARGS="--variant ct".
#   A variant's stderr is kept in gpurun_out/ab_<pid>.<variant>.err: a library that does not load and a kernel
#   that faults then look different.
WORKLOADS=${1:-"p256r1_var_2^20"}; VARIANTS=${2:-"head"}; ROUNDS=${3:-3}
OUT=gpurun_out/ab_$$.jsonl; : > $OUT
for r in $(seq 1 $ROUNDS); do
  for v in $VARIANTS; do
    lib=variants/libeccx_$v.so; [ "$v" = head ] && lib=eccoxide_amd/libeccx.so
    for w in $WORKLOADS; do
      [ -f "$lib" ] || { echo "missing $lib" >&2; exit 1; }
      ECCX_LIB_PATH=$(pwd)/$lib timeout -k 10 300 python bench.py --steps 8 --warmup 2 --workload "$w" --no-cpu-baseline $BENCH_ARGS 2>> gpurun_out/ab_$$.$v.err | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'w': d['config']['workload'], 'v': '$v', 'ms': d['roofline']['kernel_ms'], 'ok': d['parity_sample_ok']}))" >> $OUT
    done
  done
done
python - $OUT <<'PY'
import json, sys, collections
runs = collections.defaultdict(list)
for l in open(sys.argv[1]):
    d = json.loads(l); runs[(d['w'], d['v'])].append(d['ms']); assert d['ok']
for (w, v), ms in sorted(runs.items()):
    ms.sort(); print(f"{w:26s} {v:12s} median {ms[len(ms)//2]:8.3f} ms   min {ms[0]:8.3f}   runs {len(ms)}")
PY
