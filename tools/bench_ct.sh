#!/bin/bash
# One bench.py line per secret-scalar workload (ECCX_CT_SCAN: --variant ct; ECCX_CT_GATHER: --variant ctg).
# usage (inside gpurun): bash tools/bench_ct.sh gpurun_out/bench_ct.jsonl
OUT=${1:-gpurun_out/bench_ct.jsonl}
: > $OUT
for w in "p256r1_var_2^20" "p384r1_var_2^19" "p521r1_var_2^19" "bls12_381_g1_var_2^20" "ed25519_var_2^20" \
         "p256r1_base_2^20" "p384r1_base_2^19" "p521r1_base_2^19" "bls12_381_g1_base_2^20" "ed25519_base_2^20"; do
  timeout -k 10 400 python bench.py --steps 5 --warmup 1 --workload "$w" --variant ct --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT || exit 1
done
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --workload "bls12_381_g1_var_2^20" --variant ctsub --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT || exit 1
for w in "p256r1_base_2^20" "p384r1_base_2^19" "p521r1_base_2^19" "bls12_381_g1_base_2^20" "ed25519_base_2^20"; do
  timeout -k 10 400 python bench.py --steps 5 --warmup 1 --workload "$w" --variant ctg --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT || exit 1
done
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    hp = d.get("host_path") or {}
    print(d["config"]["workload"], d["config"].get("variant", ""), "%.3g/s" % d["value"], "%.2f ms" % d["roofline"]["kernel_ms"], "host %.2f ms" % hp.get("ms", 0), d["parity_sample_ok"])
PY
