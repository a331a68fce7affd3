#!/bin/bash
# one-off: what do the rocpd databases of rocprofv3 look like on this image? (schema of the views prof_summary.py reads)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/schema
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES -d $OUT/pmc -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload "p256r1_base_2^20" > $OUT/run.log 2>&1
DB=$(find $OUT/pmc -name "*.db" | head -1)
python3 - "$DB" > $OUT/schema.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for (name, typ) in db.execute("select name, type from sqlite_master where type in ('table','view') order by name"):
    print(typ, name)
for v in ("kernels", "counters_collection"):
    print("==", v)
    for r in db.execute(f"pragma table_info({v})"): print("  ", r)
    for r in db.execute(f"select * from {v} limit 3"): print("  row", r)
PY
rm -rf $OUT/pmc
cat $OUT/schema.txt | head -80
