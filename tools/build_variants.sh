#!/bin/bash
# Builds kernel variants of librsx for A/B runs: tools/build_variants.sh name "-DFLAG=.. -DFLAG2=.." [name2 "flags" ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/source_amd/lib/variants"
while [ $# -gt 1 ]; do
    name=$1; flags=$2; shift 2
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fopenmp -Wno-unused-value $flags \
        "$R/source_amd/csrc/rsx_host.cpp" "$R/source_amd/csrc/rsx_hostwalk.cpp" "$R/source_amd/csrc/rsx_device.hip" -o "$R/source_amd/lib/variants/librsx_$name.so" &
done
wait
ls -la "$R/source_amd/lib/variants/"
