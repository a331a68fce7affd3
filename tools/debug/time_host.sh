# host-buffer path of a workload/variant under two library builds: bash tools/debug/time_host.sh "p256r1_base_2^20" ct "head prev"
W=$1; V=$2; LIBS=${3:-"head prev"}
for r in 1 2 3; do for v in $LIBS; do
  lib=variants/libeccx_$v.so; [ "$v" = head ] && lib=eccoxide_amd/libeccx.so
  ECCX_LIB_PATH=$(pwd)/$lib timeout -k 10 300 python bench.py --steps 5 --warmup 1 --workload "$W" --variant $V --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_path']; print('$W $V $v', 'host %.2f ms' % h['ms'], h['matches_device_run'], 'kernel %.2f' % d['roofline']['kernel_ms'])"
done; done
