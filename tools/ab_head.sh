#!/bin/bash
# tools/ab_head.sh [rev]  -> gpurun_in/liblcr_head.so = liblcr built from the sources of <rev> (default HEAD), for a same-box A/B of the
# working tree against it: on the GPU box, alternate `LCR_LIB=gpurun_in/liblcr_head.so python bench.py --quick ...` and the default library
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
rm -rf gpurun_in/old_src; mkdir -p gpurun_in/old_src
git archive $rev longcallr_amd/csrc include | tar -x -C gpurun_in/old_src
cd gpurun_in/old_src/longcallr_amd/csrc; mkdir -p obj
for f in k0_ops k1_pileup k2_candidates k3_fragments k4_phase k4_enum k4_stage k4_post k4_grid k5_regions lcr_api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-value -c $f.hip -o obj/$f.o 2>/dev/null &
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -c lcr_bam.cpp -o obj/lcr_bam.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../../../liblcr_head.so obj/*.o -lz
cd ../../../..; rm -rf gpurun_in/old_src
ls -la gpurun_in/liblcr_head.so
