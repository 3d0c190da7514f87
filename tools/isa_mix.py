"""Instruction mix of the big kernels of the current build, priced with the round-5 issue model and set beside the hardware's counters.

    python tools/isa_mix.py [profiles/<tag>_sq_counters.csv] > profiles/<tag>_isa_mix.txt

Static: the gfx950 code objects inside plainrenderer_amd/csrc/_obj/*.o (after plainrenderer_amd.build.build()) are disassembled and classified.
The model (profiles/r05_valu_rates.txt, measured with tools/valu_issue_probe.hip in SHADER cycles - no clock assumption):
    every VALU instruction            2.2 cycles of its wave's SIMD issue (the "half-rate" class - min / max / med3, conversions, compares, v_cndmask, shifts, bit-field
                                      ops - is a second issue resource that overlaps with full-rate work: it only bounds a stream through 4.4 x its own count)
    surcharges beyond the 2.2         transcendental (v_rcp / rsq / sqrt / exp / log / sin / cos) + 9.8 (10 - 14 cycles each among other work), v_fma_mix* / v_dot2* / v_pk_* + 2.1,
                                      DPP + 2.9, v_mul_lo / hi / mad_u64 + 1.8, an SGPR / VCC source on a full-rate opcode + 0.9, v_readfirstlane + 4.4
    every scalar instruction          3.0 cycles (s_waitcnt / s_nop: 0) - the CU's scalar unit serves four SIMDs
Dynamic: SQ_INSTS_VALU, SQ_INSTS_SALU / SQ_WAVES of the same build (rocprofv3 --pmc, tools/profile_round.sh) = what a wave really issues; the static shares are
scaled to them. Measured = GRBM_GUI_ACTIVE / 8 XCDs: cycles the kernel occupies the chip. estimate / measured says how much of the kernel's time its own instruction
stream explains; the rest is waiting (memory, barriers, launch ramp).
"""
import glob, os, re, subprocess, sys, tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
BASE = 2.2  # cycles per VALU instruction of any class
SURCHARGE = {"full": 0.0, "half": 0.0, "sgpr": 0.9, "dpp": 2.9, "mix": 2.1, "mul": 1.8, "lane": 4.4, "quarter": 9.8}
HALF_BOUND = 4.4  # a stream cannot issue its half-rate class faster than this
SALU_COST = 3.0
MIX = re.compile(r"v_fma_mix|v_dot2|v_pk_")
MUL = re.compile(r"v_mul_lo_|v_mul_hi_|v_mad_u64|v_mad_i64")
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
    if MIX.match(op):
        return "mix"
    if MUL.match(op):
        return "mul"
    if op.startswith("v_readfirstlane") or op.startswith("v_readlane"):
        return "lane"
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
    print("# model (profiles/r05_valu_rates.txt, shader cycles): %.1f per VALU instruction + surcharges %s; half-rate class bound %.1f x its count; %.1f per scalar instruction "
          "(s_waitcnt / s_nop free)" % (BASE, ", ".join("%s +%.1f" % kv for kv in SURCHARGE.items() if kv[1]), HALF_BOUND, SALU_COST))
    cache = {}
    order = ("full", "half", "sgpr", "dpp", "mix", "mul", "lane", "quarter")
    for obj_sub, want, label in KERNELS:
        if obj_sub not in cache:
            cache[obj_sub] = disassemble(obj_sub)
        ins, regs = kernel_body(*cache[obj_sub], want)
        if ins is None:
            print("\n%s: kernel %s not found" % (label, want))
            continue
        c = Counter(filter(None, (classify(l) for l in ins)))
        static_valu = sum(c.values())
        vmem = sum(1 for l in ins if l.startswith(("global_", "buffer_", "flat_", "scratch_")))
        lds = sum(1 for l in ins if l.startswith("ds_"))
        salu = sum(1 for l in ins if l.startswith("s_"))
        salu_priced = sum(1 for l in ins if l.startswith("s_") and not l.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_code_end")))
        static_valu_cycles = BASE * static_valu + sum(SURCHARGE[k] * v for k, v in c.items())
        print("\n%s\n  %s" % (label, want))
        print("  registers: vgpr %(vgpr_count)d sgpr %(sgpr_count)d scratch %(private_segment_fixed_size)d B lds %(group_segment_fixed_size)d B" % regs)
        print("  static: %d instructions: VALU %d (%s); VMEM %d, LDS %d, SALU %d of which %d priced" % (
            len(ins), static_valu, ", ".join("%s %d" % (k, c[k]) for k in order if c[k]), vmem, lds, salu, salu_priced))
        print("  static: VALU stream %.0f cycles = %.2f per VALU instruction; half-rate class %.1f %% of the VALU instructions (bounds the stream only above 50 %%); scalar %.0f cycles" % (
            static_valu_cycles, static_valu_cycles / static_valu, 100.0 * c["half"] / static_valu, SALU_COST * salu_priced))
        key = [k for k in counters if want.replace(", ", "; ") in k]
        if key:
            d = counters[key[0]]
            waves, valu = d.get("SQ_WAVES"), d.get("SQ_INSTS_VALU")
            if waves and valu:
                per_wave = valu / waves
                scale = per_wave / static_valu
                salu_dyn = d.get("SQ_INSTS_SALU", 0.0) / waves
                salu_dyn_priced = salu_dyn * (salu_priced / max(salu, 1))
                valu_cycles = max(static_valu_cycles * scale, HALF_BOUND * c["half"] * scale)
                est_cycles = valu_cycles + SALU_COST * salu_dyn_priced  # per wave
                waves_per_simd = waves / 1024.0
                print("  dynamic: SQ_INSTS_VALU / SQ_WAVES = %.1f VALU per wave (%.0f %% of the static count), SQ_INSTS_SALU / SQ_WAVES = %.1f (%.0f priced)" % (per_wave, 100 * scale, salu_dyn, salu_dyn_priced))
                print("  per wave by class (static shares x dynamic count): " + ", ".join("%s %.0f" % (k, c[k] * scale) for k in order if c[k]))
                print("  issue estimate: %.0f (VALU) + %.0f (scalar) = %.0f cycles per wave x %.1f waves per SIMD = %.0f cycles; at 2 cycles per VALU instruction and nothing else: %.0f" % (
                    valu_cycles, SALU_COST * salu_dyn_priced, est_cycles, waves_per_simd, est_cycles * waves_per_simd, per_wave * 2.0 * waves_per_simd))
                if d.get("GRBM_GUI_ACTIVE"):
                    meas = d["GRBM_GUI_ACTIVE"] / 8.0
                    print("  measured: GRBM_GUI_ACTIVE / 8 XCDs = %.0f cycles (kernels serialised by the profiler; %.1f us at 2.4 GHz) -> issue estimate / measured = %.2f (VALU stream alone %.2f), "
                          "VALU roofline fraction (2 cycles per instruction) %.2f" % (meas, meas / 2400.0, est_cycles * waves_per_simd / meas, valu_cycles * waves_per_simd / meas, per_wave * 2.0 * waves_per_simd / meas))
                if d.get("TCP_TOTAL_CACHE_ACCESSES_sum") and d.get("GRBM_GUI_ACTIVE"):
                    acc = d["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0
                    print("  L1: %.0f cache-line accesses per CU = %.2f per cycle; VMEM reads per wave %.1f" % (acc, acc / (d["GRBM_GUI_ACTIVE"] / 8.0), d.get("SQ_INSTS_VMEM_RD", 0) / waves))
        top = Counter(l.split()[0] for l in ins if classify(l) in ("sgpr", "dpp", "mix", "mul", "lane", "quarter")).most_common(12)
        print("  opcodes that carry a surcharge: " + ", ".join("%s %d" % kv for kv in top))


if __name__ == "__main__":
    main()
