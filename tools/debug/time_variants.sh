# kernel time of a workload/variant under several library builds, parity not required (timing experiments):
#   bash tools/debug/time_variants.sh "p256r1_base_2^20" ct "head noinf"
W=$1; V=$2; LIBS=${3:-"head"}
for r in 1 2 3; do for v in $LIBS; do
  lib=variants/libeccx_$v.so; [ "$v" = head ] && lib=eccoxide_amd/libeccx.so
  ECCX_LIB_PATH=$(pwd)/$lib timeout -k 10 300 python bench.py --steps 8 --warmup 2 --workload "$W" --variant $V --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $V $v', '%.3f ms' % d['roofline']['kernel_ms'], d['parity_sample_ok'])"
done; done
