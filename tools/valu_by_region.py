"""Static VALU instructions of one kernel BY SOURCE REGION: every instruction is attributed through its inlining chain (DWARF, llvm-symbolizer --inlining) to the
innermost function of a caller-supplied list it was inlined from - or, failing that, to the kernel body - so that the cost of `calcShadow`, `froxelLookup`,
`shadeIndirect` ... can be read next to the operation count the shader's math asks for (profiles/r06_instruction_budget.txt).
    python tools/valu_by_region.py <source.hip> <kernel-name-substring (demangled)> <region function>[,<region function>...]
Quarter-rate instructions (transcendentals, 32-bit integer multiplies) are listed separately. Counts are STATIC (every instruction once); the dynamic count per wave
(SQ_INSTS_VALU / SQ_WAVES) is lower where code is skipped (out-of-range pack path, cascade loop bodies) and higher where loops repeat (the spatial filter's samples)."""
import os, re, subprocess, sys, tempfile
from collections import Counter, OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plainrenderer_amd import build as b
LLVM = "/opt/rocm/lib/llvm/bin"


def main(src, pat, regions):
    flags = b.FLAGS
    if os.sep + "kernels_fast" + os.sep in os.path.abspath(src):
        flags = [b.FAST_FLAGS_REPLACE.get(f, f) for f in b.FLAGS] + b.FAST_FLAGS_EXTRA
    with open(src) as fh:
        for line in fh:
            if line.startswith("// PLR_BUILD_FLAGS:"):
                flags = flags + line.split(":", 1)[1].split()
    tmp = tempfile.mkdtemp()
    obj = os.path.join(tmp, "dev.o")
    subprocess.run([b.HIPCC, "-x", "hip"] + [f for f in flags if f != "-fPIC"] + ["-g", "--cuda-device-only", "-c", "-o", obj, src], check=True, capture_output=True)
    co = os.path.join(tmp, "dev.co")  # (the device-only object is an offload bundle)
    if subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + obj, "--output=" + co], capture_output=True).returncode == 0 and os.path.getsize(co) > 0:
        obj = co
    dis = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--demangle", obj]).decode()
    blocks = re.split(r"\n(?=[0-9a-f]{16} <)", dis)
    body = None
    for blk in blocks:
        head = blk.split("\n", 1)[0]
        if pat in head and "<" in head:
            body = blk
            name = head[head.index("<") + 1:]
            break
    if body is None:
        raise SystemExit("no kernel matches " + pat)
    ins = []
    for l in body.split("\n")[1:]:
        m = re.match(r"\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1)))
    addrs = "\n".join("0x%x" % a for a, _ in ins)
    sym = subprocess.run([LLVM + "/llvm-symbolizer", "--obj=" + obj, "--inlining", "--functions=short", "--output-style=GNU"], input=addrs.encode(), capture_output=True).stdout.decode()
    # GNU style: per address a list of "function\nfile:line" pairs (innermost first); addresses separated by nothing - so count by pairs per address with --inlining
    # is ambiguous; use the LLVM style instead, which separates addresses by an empty line
    sym = subprocess.run([LLVM + "/llvm-symbolizer", "--obj=" + obj, "--inlining", "--functions=short"], input=addrs.encode(), capture_output=True).stdout.decode()
    chains = [c for c in sym.strip().split("\n\n")]
    assert len(chains) == len(ins), (len(chains), len(ins))
    Q = ("v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp", "v_sin", "v_cos")
    M = ("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64")
    valu, trans, imul, salu, vmem = Counter(), Counter(), Counter(), Counter(), Counter()
    for (addr, op), chain in zip(ins, chains):
        lines = chain.split("\n")
        funcs = lines[0::2]  # innermost first
        region = "(kernel body)"
        for f in funcs:  # the innermost region function of the chain
            short = f.split("(")[0].split("::")[-1].split("<")[0]
            if short in regions:
                region = short
                break
        if op.startswith("v_"):
            valu[region] += 1
            if op.startswith(Q):
                trans[region] += 1
            if op.startswith(M):
                imul[region] += 1
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")):
            salu[region] += 1
        elif op.startswith(("global_", "buffer_", "ds_", "scratch_")):
            vmem[region] += 1
    print(name[:150])
    print("%-34s %6s %6s %6s %6s %6s" % ("region", "VALU", "trans", "imul", "SALU", "mem"))
    for r, n in valu.most_common():
        print("%-34s %6d %6d %6d %6d %6d" % (r, n, trans[r], imul[r], salu[r], vmem[r]))
    for r in salu:
        if r not in valu:
            print("%-34s %6d %6d %6d %6d %6d" % (r, 0, 0, 0, salu[r], vmem[r]))
    print("%-34s %6d %6d %6d %6d %6d" % ("total", sum(valu.values()), sum(trans.values()), sum(imul.values()), sum(salu.values()), sum(vmem.values())))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], set(sys.argv[3].split(",")))
