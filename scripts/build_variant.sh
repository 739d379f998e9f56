#!/bin/bash
# Builds a second copy of the library with extra compiler flags into build/<name>/libpbr355.so (git-ignored), e.g.
#   scripts/build_variant.sh trace -DPBR_SURFACE_TRACE
# and PBR355_LIBRARY=build/<name>/libpbr355.so selects it.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=build/$name; mkdir -p $out
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -fno-slp-vectorize"
pids=()
for f in pbr355 ibl_kernels raster_kernels resolve_kernels; do
  extra=""; case $f in raster_kernels|pbr355) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc $flags $extra "$@" -c wgpu-renderer_amd/csrc/$f.hip -o $out/$f.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libpbr355.so $out/*.o
echo built $out/libpbr355.so
