#!/bin/bash
# Build a variant of the engine under build/variants/NAME.so with extra compiler flags (e.g. -DZGPU_WALK_STATS).
# Usage: scripts/build_variant.sh NAME FLAGS...   -- select it with ZAMD_GPU_LIB=$PWD/build/variants/NAME.so
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/variants; OBJ=$ROOT/build/variants/obj_$NAME
mkdir -p $OBJ
for f in zgpu_engine zgpu_lz_serial zgpu_lz_parallel zgpu_lz_sorted zgpu_lz_fastwin zgpu_lz_parse zgpu_cont zgpu_huffman zgpu_stitch zgpu_inflate zgpu_comm; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value -Wno-unused-result "$@" -c $ROOT/zlib_amd/csrc/$f.hip -o $OBJ/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OBJ/*.o -ldl
rm -rf $OBJ
echo built $OUT/$NAME.so
