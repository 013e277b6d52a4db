#!/bin/bash
# dev: build a variant of the library into tools/variants/<name>/libmodgpu.so (git-ignored; it travels to the GPU box with
# gpurun).  usage: tools/build_variant.sh <name> [-DMG_PART_THREADS=512 ...]   from the working tree, or
#        REV=<commit> tools/build_variant.sh <name> [...]                       from a commit.   Use with MODGPU_LIB=.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
B=${TMPDIR:-/tmp}/modgpu_variant_$name
rm -rf $B && mkdir -p $B/modimizer_amd
if [ -n "$REV" ]; then
  (cd $R && git archive $REV include modimizer_amd/csrc) | tar -x -C $B
else
  cp -r $R/include $B/ && cp -r $R/modimizer_amd/csrc $B/modimizer_amd/
fi
rm -f $B/modimizer_amd/csrc/*.o
make -C $B/modimizer_amd/csrc -j8 -s EXTRA="$*" >&2
mkdir -p $R/tools/variants/$name && cp $B/modimizer_amd/libmodgpu.so $R/tools/variants/$name/
echo $R/tools/variants/$name/libmodgpu.so
