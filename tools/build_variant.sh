#!/bin/bash
# Builds an experimental variant of the library with extra -D flags: tools/build_variant.sh NAME "-DMRT_X=1 ..."
# -> metal-raytracing_amd/variants/libmrt_hip_NAME.so (git-ignored; select it with MRT_LIB_PATH).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/metal-raytracing_amd/csrc; V=$ROOT/metal-raytracing_amd/variants/$1
mkdir -p $V
FL="-O3 -std=c++17 -fPIC -ffp-contract=off $2"
for f in bvh_build two_level renderer calibrate group; do /opt/rocm/bin/hipcc --offload-arch=gfx950 $FL -c -o $V/$f.o $C/$f.hip & done
for f in api host_geometry bvh_host_sah; do /opt/rocm/bin/hipcc $FL -c -o $V/$f.o $C/$f.cpp & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $LINK_EXTRA -o $ROOT/metal-raytracing_amd/variants/libmrt_hip_$1.so $V/*.o -ldl -lpthread
echo built $ROOT/metal-raytracing_amd/variants/libmrt_hip_$1.so
