#!/bin/bash
# A/B builds of the library: bash tools/build_variant.sh <name> "<extra hipcc flags>"  -> clipself_amd/csrc/ab/libclipself_hip_<name>.so
# (only the translation units that read the flags are recompiled: gemm_stream.hip by default, VARIANT_UNITS to override)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../clipself_amd/csrc"
mkdir -p ab
units=${VARIANT_UNITS:-gemm_stream}
objs=""
for f in gemm gemm_stream attention norm elementwise roialign_loss adamw preprocess runtime; do
  if [[ " $units " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -munsafe-fp-atomics $flags -c $f.hip -o ab/_obj_${f}_$name.o &
    objs="$objs ab/_obj_${f}_$name.o"
  else
    objs="$objs _obj_$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _obj_errors.o -o ab/libclipself_hip_$name.so
rm -f ab/_obj_*_$name.o
echo "built ab/libclipself_hip_$name.so"
