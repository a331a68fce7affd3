#!/bin/bash
# Run every bench workload once on the GPU box and collect the JSON lines.
# usage (inside gpurun): bash tools/bench_all.sh gpurun_out/bench_all.jsonl
OUT=${1:-gpurun_out/bench_all.jsonl}
: > $OUT
for w in "p256r1_var_2^20" "p256r1_base_2^20" "ed25519_base_2^20" "ed25519_var_2^20" "x25519_2^20" "p256r1_verify_2^20" "p384r1_var_2^19" "p521r1_var_2^19" "bls12_381_g1_var_2^20"; do
  timeout -k 10 400 python bench.py --steps 5 --warmup 1 --workload "$w" --cpu-sample 16384 2>/dev/null | tail -1 >> $OUT || exit 1
done
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --variant mirror --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --workload "bls12_381_g1_var_2^20" --variant glv --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --workload "ed25519_base_2^20" --variant lds --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --workload "p256r1_verify_2^20" --variant xonly --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    cb = d.get("cpu_baseline") or {}
    print(d["config"]["workload"], "%.3g/s" % d["value"], "%.2f ms" % d["roofline"]["kernel_ms"], "valu %.2f" % ((d.get("valu") or {}).get("frac", 0)),
          "cpu %.3g/s x%d" % (cb.get("value", 0), cb.get("cores", 0)), d["parity_sample_ok"])
PY
