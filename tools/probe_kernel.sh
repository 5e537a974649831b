#!/bin/bash
# Compiles ONE instantiation of a device kernel by itself (seconds instead of the minute the whole translation unit takes) and prints
# its register footprint and where its scratch (spill) instructions lie, by source line:
#   tools/probe_kernel.sh 'k_render_trace<false, 0, 1, 1, true>(DScene, RenderParams, Sample *, unsigned long long *, FuseParams)' ["-DFLAG .."]
# Output files: /tmp/probe/probe.s (assembly with .loc line tables).
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/probe
n=$(grep -n '#include "dev_selftest.hpp"' "$R/source_amd/csrc/rsx_device.hip" | cut -d: -f1)
head -n "$n" "$R/source_amd/csrc/rsx_device.hip" | sed "s#\"../../include/rsx.h\"#\"$R/include/rsx.h\"#; s#\"rsx_internal.h\"#\"$R/source_amd/csrc/rsx_internal.h\"#; s#\"dev_\(.*\)\"#\"$R/source_amd/csrc/dev_\1\"#" > /tmp/probe/probe.hip
echo "template __global__ void $1;" >> /tmp/probe/probe.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value --cuda-device-only -gline-tables-only $2 \
    -Rpass-analysis=kernel-resource-usage -S /tmp/probe/probe.hip -o /tmp/probe/probe.s 2>&1 \
  | tee /tmp/probe/compile.log | grep -E 'error|remark' | sed 's/.*remark: //; s/ \[-Rpass.*//' | grep -E "error|Function Name|VGPRs|AGPRs|Scratch|Occupancy" | paste - - - - - - \
  | grep -E "${3:-k_render_trace|k_accumulate|fused_flush}" | sed 's/Function Name: //; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/' | cut -c1-260
grep -E ' error: ' /tmp/probe/compile.log | head -5
python3 - <<'PY'
import re, collections
cur = None; files = {}; out = collections.Counter(); fn = None
for line in open('/tmp/probe/probe.s'):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]; continue
    m = re.match(r'^(_Z\w+):', line)
    if m: fn = m.group(1)[:40]
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m: cur = (files.get(int(m.group(1)), '?'), int(m.group(2))); continue
    m = re.match(r'\s*(scratch_(?:store|load)_dword(?:x(\d))?)', line)
    if m: out[(fn, m.group(1).split('_')[1], cur)] += int(m.group(2) or 1)
tot = collections.Counter()
for (fn, kind, loc), n in sorted(out.items(), key=lambda kv: (kv[0][0], kv[0][2] or ('', 0), kv[0][1])):
    print("  %-40s %-5s %3d dwords at %s:%s" % (fn, kind, n, loc[0] if loc else '?', loc[1] if loc else '?')); tot[(fn, kind)] += n
print(dict(tot))
# device functions left out of line (a call passes struct arguments through scratch: world_trace_wave once went 34 -> 61 ms that way)
import subprocess
funcs = [l.split()[1].split(',')[0] for l in open('/tmp/probe/probe.s') if l.strip().startswith('.type') and '@function' in l]
calls = [f for f in funcs if not re.match(r'_Z\d+k_', f)]
if calls:
    out = subprocess.run(['c++filt'] + calls, capture_output=True, text=True).stdout.split('\n')
    print("out of line:", "; ".join(o.split('(')[0] for o in out if o))
PY
