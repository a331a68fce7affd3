for v in head dbgent2 nopk; do
  lib=variants/libeccx_$v.so; [ "$v" = head ] && lib=eccoxide_amd/libeccx.so
  for r in 1 2; do
  ECCX_LIB_PATH=$(pwd)/$lib timeout -k 10 300 python bench.py --steps 8 --warmup 2 --workload "p256r1_base_2^20" --variant ct --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['roofline']['kernel_ms'], d['parity_sample_ok'])"
  done
done
