#!/bin/bash
# usage: tools/build_variant.sh NAME "EXTRA FLAGS" [files...]  ->  skele_raytracer_amd/lib/var/libskr_NAME.so (load it with SKR_LIBRARY=...; travels to the GPU box, unlike build/)
# Only the translation units named (default: render_nodes) are recompiled with the flags; the rest are the product's objects.
set -e -o pipefail
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-render_nodes}
make -j4 lib > /dev/null
mkdir -p build/var_$NAME skele_raytracer_amd/lib/var
OBJS=""
for o in render_kernel render_wave render_nodes render_generic accumulate api scene_host multi_gpu; do
  if echo " $FILES " | grep -q " $o "; then
    src=skele_raytracer_amd/csrc/$o.hip; x=""
    [ -f $src ] || { src=skele_raytracer_amd/csrc/$o.cpp; x="-x hip"; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function -Wno-pass-failed -Iinclude $FLAGS -c -o build/var_$NAME/$o.o $x $src &
    OBJS="$OBJS build/var_$NAME/$o.o"
  else
    OBJS="$OBJS build/obj/$o.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o skele_raytracer_amd/lib/var/libskr_$NAME.so $OBJS -ldl -lpthread
echo "built skele_raytracer_amd/lib/var/libskr_$NAME.so ($FLAGS; $FILES)"
