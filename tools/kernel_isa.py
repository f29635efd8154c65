#!/usr/bin/env python3
"""Disassemble one kernel of libprt_hip.so and summarise it per basic block: instruction counts, scratch (spill) traffic
and where the node step (v_cvt_f32_ubyte*) and the triangle test (v_rcp / v_div) live.

    python tools/kernel_isa.py "k_pool<256, 5, false, true, false, false, false, 0, false>" [--dump out.s]
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.environ.get("PRT_SO", os.path.join(ROOT, "par_raytracer_amd", "libprt_hip.so"))
want = sys.argv[1]
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
with tempfile.TemporaryDirectory() as tmp:
    subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, tmp + "/fat.bin"], check=True)
    subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + tmp + "/fat.bin",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + tmp + "/k.co"], check=True)
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--demangle", "--no-show-raw-insn", tmp + "/k.co"], stdout=subprocess.PIPE, check=True).stdout.decode()
# split into functions
funcs = re.split(r"\n(?=[0-9a-f]{16} <)", dis)
body = None
for f in funcs:
    head = f.split("\n", 1)[0]
    if want in head:
        body = f
        break
if body is None:
    sys.exit("kernel not found: " + want)
if dump:
    open(dump, "w").write(body)
lines = body.split("\n")[1:]
# instructions with their addresses; branch targets (given as +0xOFFSET in the comment) start new blocks
ins_list, targets = [], set()
base = int(body.split(" ", 1)[0], 16)
for ln in lines:
    m = re.match(r"^\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
    if not m:
        continue
    ins, addr = m.group(1).strip(), int(m.group(2), 16)
    ins_list.append((addr, ins))
    if ins.startswith(("s_cbranch", "s_branch")):
        t = re.search(r"\+0x([0-9a-fA-F]+)>\s*$", ln)
        if t:
            targets.add(base + int(t.group(1), 16))
blocks, cur, name = [], [], "%x" % 0
prev_branch = False
for addr, ins in ins_list:
    if cur and (addr in targets or prev_branch):
        blocks.append((name, cur))
        name, cur = "%x" % (addr - base), []
    cur.append(ins)
    prev_branch = ins.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc"))
if cur: blocks.append((name, cur))
tot = sum(len(b) for _, b in blocks)
print("%d instructions in %d blocks" % (tot, len(blocks)))
print("%-10s %5s %5s %5s %5s %5s %5s %5s  notes" % ("block", "ins", "valu", "salu", "vmem", "lds", "scrL", "scrS"))
for i, (n, b) in enumerate(blocks):
    valu = sum(1 for x in b if x.startswith("v_"))
    salu = sum(1 for x in b if x.startswith("s_"))
    vmem = sum(1 for x in b if x.startswith(("global_", "buffer_", "flat_")))
    lds = sum(1 for x in b if x.startswith("ds_"))
    sl = sum(1 for x in b if x.startswith("scratch_load"))
    ss = sum(1 for x in b if x.startswith("scratch_store"))
    notes = []
    cvt = sum(1 for x in b if x.startswith("v_cvt_f32_ubyte"))
    if cvt: notes.append("node-step(cvt %d)" % cvt)
    if any(x.startswith(("v_rcp_f32", "v_div_scale")) for x in b): notes.append("div")
    rl = sum(1 for x in b if x.startswith(("v_readlane", "v_writelane")))
    if rl: notes.append("lane-spill %d" % rl)
    if len(b) >= 20 or sl or ss or notes:
        print("%-10s %5d %5d %5d %5d %5d %5d %5d  %s" % (n, len(b), valu, salu, vmem, lds, sl, ss, " ".join(notes)))
