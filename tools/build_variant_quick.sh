#!/bin/bash
# Like build_variant.sh for THIS tree, but recompiles only the named translation units with the extra flags
# and links them against the tree's other objects (minutes instead of a full build):
#   bash tools/build_variant_quick.sh <name> "<EXTRA flags>" k_p256 k_ed25519 ...
set -e
NAME=$1; EXTRA=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=/tmp/eccx_variantq_$NAME
rm -rf $TMP; mkdir -p $TMP/r/eccoxide_amd $TMP/r/include
cp -r $ROOT/eccoxide_amd/csrc $TMP/r/eccoxide_amd/csrc && cp $ROOT/include/eccx.h $TMP/r/include/
for u in "$@"; do rm -f $TMP/r/eccoxide_amd/csrc/$u.o; done
make -C $TMP/r/eccoxide_amd/csrc -j4 ARCH=gfx950 EXTRA="$EXTRA" > $TMP/build.log 2>&1 || { tail -20 $TMP/build.log; exit 1; }
mkdir -p $ROOT/variants && cp $TMP/r/eccoxide_amd/libeccx.so $ROOT/variants/libeccx_$NAME.so
ls -la $ROOT/variants/libeccx_$NAME.so
