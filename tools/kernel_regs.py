#!/usr/bin/env python3
"""Register / spill / LDS / scratch numbers of every kernel in libprt_hip.so (from the code object's metadata notes).

    python tools/kernel_regs.py [substring ...]      only kernels whose demangled name contains every substring
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.environ.get("PRT_SO", os.path.join(ROOT, "par_raytracer_amd", "libprt_hip.so"))
with tempfile.TemporaryDirectory() as tmp:
    subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + so,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + tmp + "/k.co"], check=False, stderr=subprocess.DEVNULL)
    co = tmp + "/k.co"
    if not os.path.exists(co) or os.path.getsize(co) == 0:
        # the fat binary lives in the .hip_fatbin section
        subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, tmp + "/fat.bin"], check=True)
        subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + tmp + "/fat.bin",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], stdout=subprocess.PIPE, check=True).stdout.decode()
rows = []
for blk in notes.split("- .agpr_count:")[1:]:
    def g(key):
        m = re.search(r"\.%s:\s*(\S+)" % key, blk)
        return m.group(1) if m else "?"
    name = g("name")
    dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE).stdout.decode().strip()
    rows.append((dem, g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
want = sys.argv[1:]
print("%-6s %-6s %-7s %-7s %-8s %-6s  kernel" % ("vgpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
for r in sorted(rows):
    if all(w in r[0] for w in want):
        print("%-6s %-6s %-7s %-7s %-8s %-6s  %s" % (r[1], r[2], r[3], r[4], r[5], r[6], re.sub(r"^void prt::", "", r[0])[:150]))
