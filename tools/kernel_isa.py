"""Static instruction mix and an issue-cycle estimate of the kernels of one built source, per loop.
    python tools/kernel_isa.py <object-substring> [kernel-substring] [top-N opcodes]
Disassembles the gfx950 code object inside csrc/_obj/<...>.o (after plainrenderer_amd.build.build()). Cycle classes are the ones
tools/valu_rates.hip measured on MI355X (cycles per wave64 instruction per SIMD): full rate 2.9, half rate 4.2 (min / max / med3, conversions,
compares, selects, integer multiplies, bit-field ops, left shifts, DPP, packed and mixed-precision ops, and ANY VALU op reading an SGPR),
transcendentals 8.2."""
import os, re, subprocess, sys, tempfile, glob
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj = [o for o in glob.glob(os.path.join(ROOT, "plainrenderer_amd/csrc/_obj/*.o")) if sys.argv[1] in os.path.basename(o)][0]
tmp = tempfile.mkdtemp()
co = os.path.join(tmp, "dev.co")
subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, os.path.join(tmp, "fat.bin")])
subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + os.path.join(tmp, "fat.bin"), "--output=" + co])
asm = subprocess.check_output([LLVM + "/llvm-objdump", "-d", co]).decode()
meta = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co]).decode()
want = sys.argv[2] if len(sys.argv) > 2 else ""
top = int(sys.argv[3]) if len(sys.argv) > 3 else 0

FULL = re.compile(r"v_(fma|fmac|mul|add|sub|subrev|mac|mad|fmaak|fmamk)_f32|v_(add|sub|subrev)_u32|v_(and|or|xor)_b32|v_mov_b32|v_lshrrev_b32|v_add_co|v_addc_co|v_ashrrev_i32")
TRANS = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_")
SGPR = re.compile(r"[ ,]s\d+\b|[ ,]s\[|[ ,]vcc")


def cost(line):
    op = line.split()[0]
    if not op.startswith("v_"): return 0.0, "other"
    if TRANS.match(op): return 8.2, "trans"
    args = " " + line.split(None, 1)[1] if len(line.split(None, 1)) > 1 else ""
    if "dpp" in op or "dpp" in args or "quad_perm" in args or "row_" in args or "wave_" in args: return 4.2, "half"
    if FULL.match(op) and not op.startswith(("v_cmp", "v_cndmask")):
        if SGPR.search(args.split("//")[0]): return 4.2, "sgpr"
        return 2.9, "full"
    return 4.2, "half"


def summary(lines):
    c = Counter(); cyc = 0.0
    for l in lines:
        k, cls = cost(l); cyc += k; c[cls] += 1
    vm = sum(1 for l in lines if l.startswith(("global_", "buffer_", "flat_", "scratch_")))
    lds = sum(1 for l in lines if l.startswith("ds_"))
    return "valu %4d (full %d, half %d, sgpr-operand %d, transcendental %d) ~%6.0f issue cycles, vmem %d, lds %d, salu %d" % (
        c["full"] + c["half"] + c["sgpr"] + c["trans"], c["full"], c["half"], c["sgpr"], c["trans"], cyc, vm, lds, sum(1 for l in lines if l.startswith("s_")))


for m in re.finditer(r'^([0-9a-f]+) <(\S+)>:\n(.*?)(?=^\n|\Z)', asm, re.M | re.S):
    base, name, body = int(m.group(1), 16), m.group(2), m.group(3)
    dem = subprocess.check_output(["c++filt", name]).decode().strip()
    if want not in dem: continue
    ins = []  # (address, text)
    for l in body.split("\n"):
        mm = re.match(r"\s+(\S.*?)\s*// ([0-9A-F]+):.*?(<\S+\+0x[0-9a-f]+>)?$", l)
        if mm: ins.append((int(mm.group(2), 16), mm.group(1) + (" " + mm.group(3) if mm.group(3) else "")))
    if not ins: continue
    blk = [b for b in meta.split("- .agpr_count") if re.search(r"\.name:\s+" + re.escape(name) + r"\s", b)]
    regs = ""
    if blk:
        f = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk[0])
        regs = "vgpr %s sgpr %s scratch %s lds %s" % tuple((f(k).group(1) if f(k) else "?") for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"))
    print(dem[:150])
    print("   whole kernel: %d instructions, %s; %s" % (len(ins), summary([t for _, t in ins]), regs))
    addr_index = {a: i for i, (a, _) in enumerate(ins)}
    for i, (a, t) in enumerate(ins):
        mm = re.match(r"s_cbranch_\w+ .*<\S+\+0x([0-9a-f]+)>", t) or re.match(r"s_branch .*<\S+\+0x([0-9a-f]+)>", t)
        if mm:
            target = base + int(mm.group(1), 16)
            if target <= a and target in addr_index:
                loop = [x for _, x in ins[addr_index[target]:i + 1]]
                print("   loop at +0x%x (%d instructions): %s" % (target - base, len(loop), summary(loop)))
                if os.environ.get("DUMP_LOOP") == "%x" % (target - base):
                    for x in loop: print("        %-5s %s" % (cost(x)[1], x))
    if top:
        for k, v in Counter(t.split()[0] for _, t in ins).most_common(top): print("    %5d %s" % (v, k))
