#!/usr/bin/env python3
"""Kernel-resource table of the gfx950 code object inside a built librsx: registers, spills, scratch, LDS and the waves per SIMD they allow.
python tools/kernel_resources.py [lib] > profiles/<tag>_kernel_resources.txt   (needs only the ROCm LLVM tools: runs anywhere)."""
import os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "source_amd", "lib", "librsx.so")
LLVM = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + d + "/fat.bin", lib])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + d + "/fat.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + d + "/k.co"])
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", d + "/k.co"], capture_output=True, text=True).stdout
    size = os.path.getsize(d + "/k.co")
kernels, cur = [], {}
for line in notes.split("\n"):
    m = re.match(r"\s+\.(\w+):\s+(.*)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip().strip("'")
    if k == "agpr_count" and cur.get("name"):
        pass
    if k in ("name",):
        cur["name"] = v
    elif k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count", "max_flat_workgroup_size"):
        cur[k] = int(v)
    if k == "wavefront_size":
        kernels.append(cur); cur = {}
names = subprocess.run(["c++filt"] + [k["name"] for k in kernels], capture_output=True, text=True).stdout.strip().split("\n")
print("# %s  (gfx950 code object: %d bytes, %d kernels)" % (os.path.relpath(lib, R), size, len(kernels)))
print("# waves/SIMD = min(8, 512 // ceil8(vgpr + agpr)); scratch = private segment bytes per lane; spills are counts of registers")
print("%-92s %5s %5s %6s %6s %8s %6s" % ("kernel", "vgpr", "sgpr", "vspill", "sspill", "scratch", "waves"))
for k, n in sorted(zip(kernels, names), key=lambda kn: kn[1]):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    regs = k.get("vgpr_count", 0) + k.get("agpr_count", 0)
    waves = min(8, 512 // max(8, (regs + 7) // 8 * 8))
    print("%-92s %5d %5d %6d %6d %8d %6d" % (n[:92], k.get("vgpr_count", 0), k.get("sgpr_count", 0), k.get("vgpr_spill_count", 0), k.get("sgpr_spill_count", 0),
                                          k.get("private_segment_fixed_size", 0), waves))
