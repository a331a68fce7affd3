#!/bin/bash
# Build libeccx.so of another commit (or of this tree with extra compiler flags) into variants/,
# for same-box A/B runs: tools/ab_bench.sh loads them through ECCX_LIB_PATH.
#   bash tools/build_variant.sh <name> <commit|WORKTREE> [EXTRA="-D..."]
set -e
NAME=$1; REV=$2; EXTRA=${3:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=/tmp/eccx_variant_$NAME
rm -rf $TMP; mkdir -p $TMP/r
if [ "$REV" = "WORKTREE" ]; then
  mkdir -p $TMP/r/eccoxide_amd $TMP/r/include
  cp -r $ROOT/eccoxide_amd/csrc $TMP/r/eccoxide_amd/csrc && cp $ROOT/include/eccx.h $TMP/r/include/
  rm -f $TMP/r/eccoxide_amd/csrc/*.o
else
  (cd $ROOT && git archive $REV eccoxide_amd/csrc include/eccx.h) | tar -x -C $TMP/r
fi
make -C $TMP/r/eccoxide_amd/csrc -j6 ARCH=gfx950 EXTRA="$EXTRA" > $TMP/build.log 2>&1 || { tail -20 $TMP/build.log; exit 1; }
mkdir -p $ROOT/variants && cp $TMP/r/eccoxide_amd/libeccx.so $ROOT/variants/libeccx_$NAME.so
ls -la $ROOT/variants/libeccx_$NAME.so
