#!/bin/bash
# prints the register footprint of the traversal kernels for a set of -D flags: tools/vgpr.sh "-DRSX_LEAF_BATCH=2 ..."
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fopenmp -Wno-unused-value $1 \
    -Rpass-analysis=kernel-resource-usage -c "$R/source_amd/csrc/rsx_device.hip" -o /tmp/vgpr_probe.o 2>&1 \
  | grep remark | sed 's/.*remark: //; s/ \[-Rpass.*//' | grep -E "Function Name|VGPRs|AGPRs|Scratch|Occupancy" | paste - - - - - - \
  | grep -E "${2:-k_render_traceILb0|k_hit_batchILb0}" | sed 's/Function Name: //; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/' | cut -c1-200
