"""Instruction mix of the big kernels of the current build, by issue class, priced with the measured rate table and set beside the
hardware's instruction counters.

    python tools/isa_mix.py [profiles/<tag>_sq_counters.csv] > profiles/<tag>_isa_mix.txt

Static: the gfx950 code objects inside plainrenderer_amd/csrc/_obj/*.o (after plainrenderer_amd.build.build()) are disassembled and every
VALU instruction of a kernel is put into one class of the rate table tools/valu_rates.hip measured on MI355X (profiles/r04_valu_rates.txt,
cycles per wave64 instruction per SIMD, nominal 2.4 GHz):
    full      2.4 - 3.0   v_fma / fmac / mul / add / sub _f32 (also with neg / abs modifiers, inline constants, literals), v_add / sub _u32,
                          v_and / or / xor / not _b32, v_mov_b32, v_lshrrev_b32, v_ashrrev_i32, v_mul_f16
    half      4.1 - 4.7   min / max / med3, every conversion, floor / fract / trunc / rndne, compares, v_cndmask, integer multiplies, three-operand
                          integer ops, bit-field ops, left shifts, v_lshl_add_u64, v_fma_mix_f32, packed ops, v_readfirstlane
    sgpr      4.1         a full-rate opcode with an SGPR / VCC source operand
    dpp       4.2 - 4.4   any DPP-modified instruction
    quarter   8.1 - 8.6   v_rcp / rsq / sqrt / exp / log / sin / cos _f32, v_fma_f16, v_fma_mixlo_f16, v_permlane32_swap
Dynamic: SQ_INSTS_VALU / SQ_WAVES of the same build (rocprofv3 --pmc, tools/profile_round.sh) = VALU instructions a wave really issues; the
static mix is scaled to it (a kernel's rare paths - sky pixels, off-screen discs, out-of-range encoder - are in the static count only).
"""
import glob, os, re, subprocess, sys, tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
RATE = {"full": 2.8, "half": 4.3, "sgpr": 4.1, "dpp": 4.3, "quarter": 8.3}
FULL = re.compile(r"v_(fma|fmac|mul|add|sub|subrev|mac|mad|fmaak|fmamk|mul_legacy)_f32|v_(add|sub|subrev)_u32|v_(and|or|xor|not)_b32|v_mov_b32|v_lshrrev_b32|v_ashrrev_i32|v_mul_f16")
QUARTER = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_|v_rcp_iflag|v_fma_f16|v_fma_mixlo|v_fma_mixhi|v_permlane32_swap")
SGPR = re.compile(r"[ ,]s\d+\b|[ ,]s\[|[ ,]vcc|[ ,]exec")

KERNELS = [  # (object substring, demangled-name substring, label, waves per launch at 4K are read from the counters)
    ("shading_fast", "upscaleAndShadeKernel<2, 0, true>", "indirectLightUpscale + deferred shade"),
    ("taa_fast", "temporalFilterStripKernel<true, true, 4, true, false>", "temporalFilter (TAA)"),
    ("gi_spatial_fast", "spatialFilterFastKernel<3, 64, true, true, false>", "filterIndirectDiffuseSpatial (x2 per frame)"),
    ("sdf_trace_fast", "sdfDiffuseTraceFastKernel<true, false, 3, false, false>", "sdfDiffuseTrace"),
    ("stream_fast", "temporalGiFilterFastKernel<3, true>", "filterIndirectDiffuseTemporal (packed texels only)"),
    ("stream_fast", "applyBloomTonemapKernel<true>", "applyBloom + tonemapping"),
]


def classify(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        return None
    args = " " + line.split(None, 1)[1] if len(line.split(None, 1)) > 1 else ""
    args = args.split("//")[0]
    if QUARTER.match(op):
        return "quarter"
    if "dpp" in op or "quad_perm" in args or "row_" in args or "wave_" in args:
        return "dpp"
    if FULL.match(op) and not op.startswith(("v_cmp", "v_cndmask")):
        return "sgpr" if SGPR.search(args) else "full"
    return "half"


def disassemble(obj_sub):
    obj = [o for o in glob.glob(os.path.join(ROOT, "plainrenderer_amd/csrc/_obj/*.o")) if obj_sub in os.path.basename(o)][0]
    tmp = tempfile.mkdtemp()
    co = os.path.join(tmp, "dev.co")
    subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, os.path.join(tmp, "fat.bin")])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + os.path.join(tmp, "fat.bin"), "--output=" + co])
    asm = subprocess.check_output([LLVM + "/llvm-objdump", "-d", co]).decode()
    meta = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co]).decode()
    return asm, meta


def kernel_body(asm, meta, want):
    for m in re.finditer(r'^([0-9a-f]+) <(\S+)>:\n(.*?)(?=^\n|\Z)', asm, re.M | re.S):
        name, body = m.group(2), m.group(3)
        dem = subprocess.check_output(["c++filt", name]).decode().strip()
        if want not in dem:
            continue
        ins = []
        for l in body.split("\n"):
            mm = re.match(r"\s+(\S.*?)\s*// ([0-9A-F]+):", l)
            if mm:
                ins.append(mm.group(1))
        blk = [b for b in meta.split("- .agpr_count") if re.search(r"\.name:\s+" + re.escape(name) + r"\s", b)]
        regs = {}
        if blk:
            for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                f = re.search(r"\." + k + r":\s+(\d+)", blk[0])
                regs[k] = int(f.group(1)) if f else -1
        return ins, regs
    return None, None


def read_counters(path):
    out, header, digest = {}, None, None
    if not path or not os.path.exists(path):
        return out, digest
    for line in open(path):
        line = line.strip()
        if line.startswith("# kernel source digest:"):
            digest = line.split(":", 1)[1].strip()
        elif line.startswith("kernel,"):
            header = line.split(",")
        elif header and line and not line.startswith("#"):
            cols = line.split(",")
            out[cols[0]] = {k: float(v) for k, v in zip(header[2:], cols[2:]) if v not in ("", "nan")}
    return out, digest


def main():
    counters_path = sys.argv[1] if len(sys.argv) > 1 else (sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.csv"))) or [None])[-1]
    counters, digest = read_counters(counters_path)
    sys.path.insert(0, ROOT)
    import bench
    print("# instruction mix by issue class of the build with kernel source digest %s (tools/isa_mix.py)" % bench.kernel_source_digest())
    print("# counters: %s (digest %s)" % (os.path.basename(counters_path) if counters_path else "none", digest))
    print("# rates (cycles per wave64 instruction per SIMD, profiles/r04_valu_rates.txt): " + ", ".join("%s %.1f" % kv for kv in RATE.items()))
    cache = {}
    for obj_sub, want, label in KERNELS:
        if obj_sub not in cache:
            cache[obj_sub] = disassemble(obj_sub)
        ins, regs = kernel_body(*cache[obj_sub], want)
        if ins is None:
            print("\n%s: kernel %s not found" % (label, want))
            continue
        c = Counter(filter(None, (classify(l) for l in ins)))
        static_valu = sum(c.values())
        static_cycles = sum(RATE[k] * v for k, v in c.items())
        vmem = sum(1 for l in ins if l.startswith(("global_", "buffer_", "flat_", "scratch_")))
        lds = sum(1 for l in ins if l.startswith("ds_"))
        salu = sum(1 for l in ins if l.startswith("s_"))
        print("\n%s\n  %s" % (label, want))
        print("  registers: vgpr %(vgpr_count)d sgpr %(sgpr_count)d scratch %(private_segment_fixed_size)d B lds %(group_segment_fixed_size)d B" % regs)
        print("  static: %d instructions: VALU %d = full %d + half %d + sgpr-sourced %d + dpp %d + quarter %d; VMEM %d, LDS %d, SALU %d" % (
            len(ins), static_valu, c["full"], c["half"], c["sgpr"], c["dpp"], c["quarter"], vmem, lds, salu))
        print("  static class shares: " + ", ".join("%s %.1f %%" % (k, 100.0 * c[k] / static_valu) for k in ("full", "half", "sgpr", "dpp", "quarter")) +
              "; priced: %.0f cycles = %.2f cycles per VALU instruction" % (static_cycles, static_cycles / static_valu))
        key = [k for k in counters if want.replace(", ", "; ") in k]
        if key:
            d = counters[key[0]]
            waves, valu = d.get("SQ_WAVES"), d.get("SQ_INSTS_VALU")
            if waves and valu:
                per_wave = valu / waves
                scale = per_wave / static_valu
                est_cycles = static_cycles * scale  # per wave
                waves_per_simd = waves / 1024.0
                t_est = est_cycles * waves_per_simd / 2.4e9 * 1e6
                t_floor = per_wave * 2.0 * waves_per_simd / 2.4e9 * 1e6
                line = "  dynamic: SQ_INSTS_VALU %.0f / SQ_WAVES %.0f = %.1f VALU per wave (%.0f %% of the static count)" % (valu, waves, per_wave, 100 * scale)
                print(line)
                print("  per wave by class (static shares x dynamic count): " + ", ".join("%s %.0f" % (k, c[k] * scale) for k in ("full", "half", "sgpr", "dpp", "quarter")))
                print("  issue estimate: %.0f cycles per wave x %.1f waves per SIMD / 2.4 GHz = %.1f us; at the 2-cycle peak rate the same instructions take %.1f us" % (
                    est_cycles, waves_per_simd, t_est, t_floor))
                if d.get("GRBM_GUI_ACTIVE"):
                    t_meas = d["GRBM_GUI_ACTIVE"] / 8.0 / 2.4e9 * 1e6
                    print("  measured (GRBM_GUI_ACTIVE / 8 XCDs at 2.4 GHz, kernels serialised by the profiler): %.1f us -> issue estimate / measured = %.2f, VALU roofline fraction %.2f" % (
                        t_meas, t_est / t_meas, t_floor / t_meas))
                if d.get("TCP_TOTAL_CACHE_ACCESSES_sum") and d.get("GRBM_GUI_ACTIVE"):
                    acc = d["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0
                    print("  L1: %.0f cache-line accesses per CU = %.2f per cycle; VMEM reads per wave %.1f" % (acc, acc / (d["GRBM_GUI_ACTIVE"] / 8.0), d.get("SQ_INSTS_VMEM_RD", 0) / waves))
        top = Counter(l.split()[0] for l in ins if classify(l) in ("half", "sgpr", "dpp", "quarter")).most_common(12)
        print("  slow-class opcodes: " + ", ".join("%s %d" % kv for kv in top))


if __name__ == "__main__":
    main()
