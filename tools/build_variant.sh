#!/bin/bash
# A/B builds of the library: tools/build_variant.sh <name> "<extra hipcc flags>" [source.hip ...]
# Recompiles the named sources of isle_amd/csrc (default: gram_lds.hip) with the extra flags and links them with the other objects of the
# regular build into tools/variants/libisle_<name>.so; ISLE_HIP_LIB=<that path> makes the Python binding load it (probes only).
set -e
NAME=$1; EXTRA=$2; shift 2 || true
SRCS=${@:-gram_lds.hip}
HERE=$(cd "$(dirname "$0")" && pwd); CS=$HERE/../isle_amd/csrc; OUT=$HERE/variants; mkdir -p $OUT/obj_$NAME
make -s -C $CS ../libisle_hip.so
OBJS=""
for o in api api_ks api_kmeans api_stages spmm gram_lds evd_tridiag dense kmeans threshold post ingest infer; do
  src=""; for s in $SRCS; do [ "${s%.*}" = "$o" ] && src=$s; done
  if [ -n "$src" ]; then
    x=""; [ "${src##*.}" = "cpp" ] && x="-x hip"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -Wno-unused-result -Wno-pass-failed $EXTRA $x -c -o $OUT/obj_$NAME/$o.o $CS/$src
    OBJS="$OBJS $OUT/obj_$NAME/$o.o"
  else OBJS="$OBJS $CS/$o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $OUT/libisle_$NAME.so $OBJS
echo built $OUT/libisle_$NAME.so
