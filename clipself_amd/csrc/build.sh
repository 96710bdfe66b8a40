#!/bin/bash
# Builds the C-ABI shared library of HIP kernels for gfx950 (MI355X).  No GPU needed to compile.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -munsafe-fp-atomics $CS_EXTRA_FLAGS"
OBJS=""
for f in gemm gemm_stream attention norm elementwise roialign_loss adamw preprocess runtime; do
  if [ ! -f "_obj_$f.o" ] || [ "$f.hip" -nt "_obj_$f.o" ] || [ cs_common.h -nt "_obj_$f.o" ] || [ gemm_common.h -nt "_obj_$f.o" ]; then
    $HIPCC $FLAGS -c "$f.hip" -o "_obj_$f.o" &
  fi
  OBJS="$OBJS _obj_$f.o"
done
wait
g++ -O2 -fPIC -std=c++17 -c errors.cpp -o _obj_errors.o
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS _obj_errors.o -o libclipself_hip.so
echo "built $(pwd)/libclipself_hip.so"
