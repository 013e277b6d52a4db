#!/bin/bash
# dev: build a -DMG_ABLATE variant of the library (the timing-ablation branches are not in the production kernels)
# into a scratch copy and print the path to put in MODGPU_LIB.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=${TMPDIR:-/tmp}/modgpu_ablate
rm -rf $B && mkdir -p $B/modimizer_amd && cp -r $R/include $B/ && cp -r $R/modimizer_amd/csrc $B/modimizer_amd/
rm -f $B/modimizer_amd/csrc/*.o
make -C $B/modimizer_amd/csrc -j8 -s EXTRA=-DMG_ABLATE >&2
echo $B/modimizer_amd/libmodgpu.so
