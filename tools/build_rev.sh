#!/bin/bash
# tools/build_rev.sh <git-rev> [name] — builds ab/lib_<name>.so from the csrc/ of a git revision (same-box A/B against
# an earlier state of the kernels: tools/abx.py --libs B,ab/lib_<name>.so)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=opencl-development-real-time-image-processing_amd
REV=$1; NAME=${2:-$1}
W=/tmp/rev/$NAME; rm -rf $W; mkdir -p $W $ROOT/ab
(cd $ROOT && git archive $REV include $PKG/csrc) | tar -x -C $W
make -s -j8 -C $W/$PKG/csrc all 2>&1 | grep -v hip-link | grep -v "^$" | tail -5 || true
cp $W/$PKG/lib/libmi355_imgfilter.so $ROOT/ab/lib_$NAME.so
ls -la $ROOT/ab/lib_$NAME.so
