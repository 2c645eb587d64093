#!/bin/bash
# tools/build_variant.sh <name> <sed-expr> <file> [<sed-expr> <file> ...] — builds ab/lib_<name>.so from a patched
# copy of csrc/ (experiments that should not touch the tree); compare with tools/ab.sh on one box.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=opencl-development-real-time-image-processing_amd
NAME=$1; shift
W=/tmp/var/$NAME; rm -rf $W; mkdir -p $W/$PKG $ROOT/ab
cp -r $ROOT/include $W/include; cp -r $ROOT/$PKG/csrc $W/$PKG/csrc
while [ $# -ge 2 ]; do sed -i -E "$1" $W/$PKG/csrc/$2; shift 2; done
if [ -n "$VARIANT_TUNE" ]; then   # VARIANT_TUNE=1: the tuning build of the patched copy (abx.py: T@ab/lib_<name>.so:KEY=V)
  mkdir -p $W/tools
  make -s -j8 -C $W/$PKG/csrc tune 2>&1 | grep -v hip-link | grep -v "^$" | tail -5 || true
  cp $W/tools/lib/libmi355_imgfilter_tune.so $ROOT/ab/lib_$NAME.so
else
  make -s -j8 -C $W/$PKG/csrc 2>&1 | grep -v hip-link | grep -v "^$" | tail -5 || true
  cp $W/$PKG/lib/libmi355_imgfilter.so $ROOT/ab/lib_$NAME.so
fi
(cd $ROOT && diff -r $PKG/csrc $W/$PKG/csrc | grep '^[<>]' | head -20)
