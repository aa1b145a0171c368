#!/bin/bash
# tools/build_variant.sh NAME "EXTRA FLAGS"  -> gpurun_in/liblcr_NAME.so (A/B timing of kernel variants: LCR_LIB=... on the GPU box)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p gpurun_in/obj_$name
objs=""
for f in k0_ops k1_pileup k2_candidates k3_fragments k4_phase k4_enum k4_stage k4_post k4_grid k5_regions lcr_api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $@ -c longcallr_amd/csrc/$f.hip -o gpurun_in/obj_$name/$f.o &
  objs="$objs gpurun_in/obj_$name/$f.o"
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC $@ -c longcallr_amd/csrc/lcr_bam.cpp -o gpurun_in/obj_$name/lcr_bam.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o gpurun_in/liblcr_$name.so $objs gpurun_in/obj_$name/lcr_bam.o -lz
rm -rf gpurun_in/obj_$name
echo gpurun_in/liblcr_$name.so
