"""Generates tools/valu_issue_probe.hip: the VALU issue-rate probe of round 5.

tools/valu_rates.hip (rounds 2-4) read 2.4 - 3.0 "cycles" for the full-rate instructions where MI355X_MICROARCH.md says 2 (v_fma_f32, wave64 on a SIMD-32). Its
loop body was 8 instructions sharing two source registers, it ran on every SIMD of the chip at once and converted wall time at an ASSUMED 2.4 GHz. This probe
separates the candidates:
  * cycles are SHADER cycles: every wave brackets its loop with s_memtime (tick = shader cycle, MI355X_MICROARCH.md "Per-instruction cycle constants");
  * loop bodies of 64 instructions (and, for the comparison, of 8): the share of the loop's s_sub / s_cmp / s_cbranch;
  * registers are named explicitly in the asm, so the VGPR BANK (register index mod 4) of every operand is chosen: one / two / three VGPR sources, all in one
    bank or spread over banks, shared between the copies or distinct per copy;
  * each body runs (a) in ONE workgroup on an otherwise idle chip at 1, 2 and 4 waves per SIMD - nothing for the power management to throttle - and (b) on every
    CU at 4 waves per SIMD, where the wall time of the launch over the cycles of its longest wave gives the clock the chip actually sustained.
    python tools/gen_valu_issue_probe.py && hipcc --offload-arch=gfx950 -O3 tools/valu_issue_probe.hip -o /tmp/valu_issue_probe && /tmp/valu_issue_probe
"""
import os

ACC0 = 32          # accumulators v32 .. v95
SRC0 = 100         # sources v100 .. v123
BODY = 64


def acc(i, bank=None, count=64):
    """accumulator register of copy i; bank given: only registers of that bank (16 of them, reused four times per body)"""
    if bank is None:
        return ACC0 + (i % count)
    return ACC0 + 4 * (i % 16) + bank


def body(line_fn, n=BODY):
    return "\\n".join(line_fn(i) for i in range(n))


VARIANTS = []


def variant(name, note, line_fn, n=BODY):
    VARIANTS.append((name, note, body(line_fn, n), n))


S = SRC0  # v100: bank 0, v101: bank 1, v102: bank 2, v103: bank 3, v104: bank 0 ...
variant("fma_3src_shared", "v_fma_f32 acc, acc, b, c - b, c shared by all copies (banks 0, 1), 64 accumulators over all banks",
        lambda i: "v_fma_f32 v%d, v%d, v%d, v%d" % (acc(i), acc(i), S, S + 1))
variant("fma_3src_shared_body8", "the same, loop body of 8 instructions (the old probe's shape)",
        lambda i: "v_fma_f32 v%d, v%d, v%d, v%d" % (acc(i, count=8), acc(i, count=8), S, S + 1), n=8)
variant("fma_3src_same_bank", "three VGPR sources all in bank 0 (acc = v32 + 4k, b = v100, c = v104)",
        lambda i: "v_fma_f32 v%d, v%d, v%d, v%d" % (acc(i, 0), acc(i, 0), S, S + 4))
variant("fma_3src_three_banks", "three VGPR sources in banks 0, 1, 2 (acc = v32 + 4k, b = v101, c = v102)",
        lambda i: "v_fma_f32 v%d, v%d, v%d, v%d" % (acc(i, 0), acc(i, 0), S + 1, S + 2))
variant("fma_3src_two_in_one_bank", "acc and b in bank 0, c in bank 1",
        lambda i: "v_fma_f32 v%d, v%d, v%d, v%d" % (acc(i, 0), acc(i, 0), S, S + 1))
variant("fma_3src_distinct", "b and c distinct per copy (16 + 8 registers over all banks)",
        lambda i: "v_fma_f32 v%d, v%d, v%d, v%d" % (acc(i), acc(i), S + (i % 16), S + 16 + (i % 8)))
variant("fmac_2src", "v_fmac_f32 acc, b, c: the accumulator is the destination's own read (VOP2)",
        lambda i: "v_fmac_f32 v%d, v%d, v%d" % (acc(i), S, S + 1))
variant("mul_2src_two_banks", "v_mul_f32 acc(bank 0), acc, b(bank 1)",
        lambda i: "v_mul_f32 v%d, v%d, v%d" % (acc(i, 0), acc(i, 0), S + 1))
variant("mul_2src_same_bank", "v_mul_f32 acc(bank 0), acc, b(bank 0)",
        lambda i: "v_mul_f32 v%d, v%d, v%d" % (acc(i, 0), acc(i, 0), S))
variant("mul_1src_inline", "v_mul_f32 acc, 2.0, acc: one VGPR source",
        lambda i: "v_mul_f32 v%d, 2.0, v%d" % (acc(i), acc(i)))
variant("mov_1src", "v_mov_b32 acc, b",
        lambda i: "v_mov_b32 v%d, v%d" % (acc(i), S))
variant("fma_1src_inline", "v_fma_f32 acc, acc, 2.0, 1.0: one VGPR source, VOP3",
        lambda i: "v_fma_f32 v%d, v%d, 2.0, 1.0" % (acc(i), acc(i)))
variant("add_u32", "v_add_u32 acc, acc, b",
        lambda i: "v_add_u32 v%d, v%d, v%d" % (acc(i), acc(i), S))
variant("max_f32", "v_max_f32 acc, acc, b (the table's half-rate class)",
        lambda i: "v_max_f32 v%d, v%d, v%d" % (acc(i), acc(i), S))
variant("cndmask", "v_cndmask_b32 acc, acc, b, vcc",
        lambda i: "v_cndmask_b32 v%d, v%d, v%d, vcc" % (acc(i), acc(i), S))
variant("fma_sgpr_src", "v_fma_f32 acc, acc, s6, c: one SGPR source",
        lambda i: "v_fma_f32 v%d, v%d, s6, v%d" % (acc(i), acc(i), S + 1))
variant("fma_mix_f32", "v_fma_mix_f32 acc, b(f16 lo), c(f16 lo), acc - what the spatial filter accumulates with (twelve per two samples)",
        lambda i: "v_fma_mix_f32 v%d, v%d, v%d, v%d op_sel_hi:[1,1,0]" % (acc(i), S, S + 1, acc(i)))
variant("dot2_f32_f16", "v_dot2_f32_f16 acc, b, c, acc - two f16 products and the accumulate in one instruction (VERDICT r04 item 4)",
        lambda i: "v_dot2_f32_f16 v%d, v%d, v%d, v%d" % (acc(i), S, S + 1, acc(i)))
variant("perm_b32", "v_perm_b32 acc, acc, b, c - the byte shuffle that would pair two samples' halves for v_dot2",
        lambda i: "v_perm_b32 v%d, v%d, v%d, v%d" % (acc(i), acc(i), S, S + 1))
variant("pk_fma_f32", "v_pk_fma_f32 on register pairs (32 pair accumulators)",
        lambda i: "v_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d]" % (ACC0 + 2 * (i % 32), ACC0 + 2 * (i % 32) + 1, ACC0 + 2 * (i % 32), ACC0 + 2 * (i % 32) + 1, S, S + 1, S + 2, S + 3))
variant("rcp_f32", "v_rcp_f32 acc, acc (transcendental)",
        lambda i: "v_rcp_f32 v%d, v%d" % (acc(i), acc(i)))
variant("cvt_f32_f16", "v_cvt_f32_f16 acc, b",
        lambda i: "v_cvt_f32_f16 v%d, v%d" % (acc(i), S))
variant("fma_then_max_alternating", "v_fma_f32 / v_max_f32 alternating: does a half-rate neighbour cost the full-rate one anything?",
        lambda i: ("v_fma_f32 v%d, v%d, v%d, v%d" % (acc(i), acc(i), S, S + 1)) if i % 2 == 0 else ("v_max_f32 v%d, v%d, v%d" % (acc(i), acc(i), S)))

# ---- mixes: which classes overlap? (the alternating fma / max body above runs at the full rate: two units, or an issue rule?)
FMA = lambda i: "v_fma_f32 v%d, v%d, v%d, v%d" % (acc(i), acc(i), S, S + 1)
MAXF = lambda i: "v_max_f32 v%d, v%d, v%d" % (acc(i), acc(i), S)
CVT = lambda i: "v_cvt_f32_f16 v%d, v%d" % (acc(i), S)
MIX = lambda i: "v_fma_mix_f32 v%d, v%d, v%d, v%d op_sel_hi:[1,1,0]" % (acc(i), S, S + 1, acc(i))
PERM = lambda i: "v_perm_b32 v%d, v%d, v%d, v%d" % (acc(i), acc(i), S, S + 1)
FMAS = lambda i: "v_fma_f32 v%d, v%d, s6, v%d" % (acc(i), acc(i), S + 1)
RCP = lambda i: "v_rcp_f32 v%d, v%d" % (acc(i), acc(i))
CMP = lambda i: "v_cmp_lt_f32 vcc, v%d, v%d" % (acc(i), S)
CND64 = lambda i: "v_cndmask_b32_e64 v%d, v%d, v%d, s[10:11]" % (acc(i), acc(i), S)
LSHL = lambda i: "v_lshlrev_b32 v%d, 1, v%d" % (acc(i), acc(i))
DPP = lambda i: "v_mov_b32_dpp v%d, v%d row_shr:1 row_mask:0xf bank_mask:0xf" % (acc(i), acc(i))
MULLO = lambda i: "v_mul_lo_u32 v%d, v%d, v%d" % (acc(i), acc(i), S)
MED3 = lambda i: "v_med3_f32 v%d, v%d, v%d, v%d" % (acc(i), acc(i), S, S + 1)
DOT2 = lambda i: "v_dot2_f32_f16 v%d, v%d, v%d, v%d" % (acc(i), S, S + 1, acc(i))


def pattern(seq):
    """seq: list of line functions, repeated over the 64 slots"""
    return lambda i: seq[i % len(seq)](i)


def blocks(a, b, run):
    """run instructions of a, then run of b, ..."""
    return lambda i: (a if (i // run) % 2 == 0 else b)(i)


variant("mix_max_max_fma_fma", "pairs: max max fma fma", pattern([MAXF, MAXF, FMA, FMA]))
variant("mix_max3_fma1", "three v_max_f32 per v_fma_f32", pattern([MAXF, MAXF, MAXF, FMA]))
variant("mix_max1_fma3", "one v_max_f32 per three v_fma_f32", pattern([MAXF, FMA, FMA, FMA]))
variant("mix_blocks32_max_fma", "32 v_max_f32 then 32 v_fma_f32 per body: only different WAVES can mix the classes", blocks(MAXF, FMA, 32))
variant("mix_blocks8_max_fma", "runs of 8", blocks(MAXF, FMA, 8))
variant("mix_max_cvt", "v_max_f32 / v_cvt_f32_f16 alternating: two half-rate classes", pattern([MAXF, CVT]))
variant("mix_fma_cvt", "v_fma_f32 / v_cvt_f32_f16 alternating", pattern([FMA, CVT]))
variant("mix_fma_fmamix", "v_fma_f32 / v_fma_mix_f32 alternating", pattern([FMA, MIX]))
variant("mix_fma_dot2", "v_fma_f32 / v_dot2_f32_f16 alternating", pattern([FMA, DOT2]))
variant("mix_fma_perm", "v_fma_f32 / v_perm_b32 alternating", pattern([FMA, PERM]))
variant("mix_fma_fmasgpr", "v_fma_f32 / v_fma_f32 with an SGPR source alternating", pattern([FMA, FMAS]))
variant("mix_fma_lshl", "v_fma_f32 / v_lshlrev_b32 alternating", pattern([FMA, LSHL]))
variant("mix_fma_dpp", "v_fma_f32 / v_mov_b32_dpp row_shr:1 alternating", pattern([FMA, DPP]))
variant("mix_fma_mullo", "v_fma_f32 / v_mul_lo_u32 alternating", pattern([FMA, MULLO]))
variant("mix_fma_med3", "v_fma_f32 / v_med3_f32 alternating", pattern([FMA, MED3]))
variant("mix_fma_rcp", "v_fma_f32 x 3 / v_rcp_f32 x 1", pattern([FMA, FMA, FMA, RCP]))
variant("mix_max_rcp", "v_max_f32 x 3 / v_rcp_f32 x 1", pattern([MAXF, MAXF, MAXF, RCP]))
variant("cmp_vcc", "v_cmp_lt_f32 vcc, acc, b", CMP)
variant("mix_fma_cmp", "v_fma_f32 / v_cmp_lt_f32 alternating", pattern([FMA, CMP]))
variant("cndmask_e64_sgprpair", "v_cndmask_b32_e64 acc, acc, b, s[10:11]", CND64)
variant("mix_fma_cndmask64", "v_fma_f32 / v_cndmask_b32_e64 alternating", pattern([FMA, CND64]))
variant("mix_cmp_cndmask_fma_fma", "v_cmp, v_cndmask (vcc), v_fma, v_fma: the select idiom between full-rate work", pattern([CMP, lambda i: "v_cndmask_b32 v%d, v%d, v%d, vcc" % (acc(i), acc(i), S), FMA, FMA]))
variant("mix_half_four_kinds", "max, cvt, lshl, perm round robin: half-rate classes only", pattern([MAXF, CVT, LSHL, PERM]))
variant("mix_realistic", "fma fma max fma cvt fma fma cndmask64: a big kernel's class mix (5 full : 3 half)", pattern([FMA, FMA, MAXF, FMA, CVT, FMA, FMA, CND64]))

# ---- transcendentals among other work: v_rcp_f32 alone streams at 8.1 cycles, one per three v_fma_f32 cost the group 21 cycles. Grouping? Spacing?
variant("mix_fma12_rcp4_grouped", "twelve v_fma_f32 then four v_rcp_f32 back to back (the 3 : 1 ratio, grouped)", lambda i: (RCP if i % 16 >= 12 else FMA)(i))
variant("mix_fma24_rcp8_grouped", "twenty-four v_fma_f32 then eight v_rcp_f32", lambda i: (RCP if i % 32 >= 24 else FMA)(i))
variant("mix_fma7_rcp1", "seven v_fma_f32 per v_rcp_f32", pattern([FMA] * 7 + [RCP]))
variant("mix_fma15_rcp1", "fifteen v_fma_f32 per v_rcp_f32", pattern([FMA] * 15 + [RCP]))
variant("mix_mul3_rcp1", "three v_mul_f32 (VOP2) per v_rcp_f32", pattern([lambda i: "v_mul_f32 v%d, v%d, v%d" % (acc(i), acc(i), S)] * 3 + [RCP]))
variant("mix_fma3_rsq1", "three v_fma_f32 per v_rsq_f32", pattern([FMA, FMA, FMA, lambda i: "v_rsq_f32 v%d, v%d" % (acc(i), acc(i))]))
variant("mix_fma3_exp1", "three v_fma_f32 per v_exp_f32", pattern([FMA, FMA, FMA, lambda i: "v_exp_f32 v%d, v%d" % (acc(i), acc(i))]))
variant("mix_max15_rcp1", "fifteen v_max_f32 per v_rcp_f32", pattern([MAXF] * 15 + [RCP]))

# ---- other instruction kinds between VALU work: do they take the wave's VALU issue time?
MULV = lambda i: "v_mul_f32 v%d, v%d, v%d" % (acc(i), acc(i), S)
variant("mix_mul_salu", "v_mul_f32 / s_add_u32 s12, s12, s13 alternating", pattern([MULV, lambda i: "s_add_u32 s12, s12, s13"]))
variant("mix_mul_salu3", "v_mul_f32 / three SALU (s_add_u32, s_and_b32, s_lshl_b32)", pattern([MULV, lambda i: "s_add_u32 s12, s12, s13", lambda i: "s_and_b32 s14, s12, s13", lambda i: "s_lshl_b32 s15, s13, 2"]))
variant("mix_mul_waitcnt", "v_mul_f32 / s_waitcnt vmcnt(0) lgkmcnt(0) alternating (nothing outstanding)", pattern([MULV, lambda i: "s_waitcnt vmcnt(0) lgkmcnt(0)"]))
variant("mix_mul_snop", "v_mul_f32 / s_nop 0 alternating", pattern([MULV, lambda i: "s_nop 0"]))
variant("mix_mul_readfirstlane", "v_mul_f32 / v_readfirstlane_b32 s12, acc alternating", pattern([MULV, lambda i: "v_readfirstlane_b32 s12, v%d" % acc(i)]))
variant("mix_mul_saveexec", "v_mul_f32 x 3 / s_and_saveexec_b64 + s_mov exec restore (a branch's mask dance)", pattern([MULV, lambda i: "s_or_saveexec_b64 s[16:17], vcc", MULV, lambda i: "s_mov_b64 exec, s[16:17]"]))

CLOBBERS = ", ".join('"v%d"' % r for r in range(ACC0, 128)) + ', "vcc", "s6", "s7", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "memory"'

HEADER = r'''// GENERATED by tools/gen_valu_issue_probe.py - do not edit. VALU issue rates on gfx950 in SHADER cycles (s_memtime), with explicit registers (VGPR banks),
// 64-instruction loop bodies, one-workgroup and whole-chip runs. See the generator's docstring.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Stamp { unsigned long long t0, t1; unsigned hwid, pad; };

#define INIT_REGS "INIT_BODY"
'''

KERNEL = r'''
__global__ __launch_bounds__(1024) void k_%(name)s(Stamp* stamps, float seed, int iters) {
    const float x = seed * (float)(threadIdx.x + 1);
    unsigned long long t0, t1;
    unsigned hwid;
    asm volatile(INIT_REGS :: "v"(x) : %(clob)s);
    asm volatile("s_mov_b32 s6, 0x3f800100\ns_mov_b32 s10, 0x55555555\ns_mov_b32 s11, 0x0f0f0f0f\ns_mov_b64 vcc, s[10:11]\ns_mov_b32 s12, 0\ns_mov_b32 s13, 4\ns_getreg_b32 %%0, hwreg(HW_REG_HW_ID)" : "=s"(hwid) :: "s6", "s10", "s11", "vcc");
    __syncthreads();
    asm volatile("s_memtime %%0\ns_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; i++) asm volatile("%(body)s" ::: %(clob)s);
    asm volatile("s_memtime %%0\ns_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float r;
    asm volatile("v_add_f32 %%0, v32, v33" : "=v"(r));
    if (r == 123.456f) stamps[0].pad = 1u;
    if ((threadIdx.x & 63u) == 0u) {
        Stamp s; s.t0 = t0; s.t1 = t1; s.hwid = hwid; s.pad = 0u;
        stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
    }
}
'''

MAIN = r'''
typedef void (*kern_t)(Stamp*, float, int);
struct Test { const char* name; const char* note; kern_t k; int bodyLen; };

// cycles of the longest wave, and over all waves: first start .. last end
static void stats(const std::vector<Stamp>& s, double* longest, double* span) {
    unsigned long long lo = ~0ull, hi = 0, mx = 0;
    for (const Stamp& w : s) { lo = std::min(lo, w.t0); hi = std::max(hi, w.t1); mx = std::max(mx, w.t1 - w.t0); }
    *longest = (double)mx; *span = (double)(hi - lo);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz (nominal)\n", prop.gcnArchName, cus, prop.clockRate);
    printf("cycles = s_memtime ticks (shader cycles) of the longest wave / (instructions per wave x waves per SIMD): cycles per wave64 instruction per SIMD\n");
    Stamp* dev;
    const int maxWaves = cus * 16;
    CHECK(hipMalloc(&dev, sizeof(Stamp) * maxWaves));
    std::vector<Test> tests = {
TESTS
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-28s | %8s %8s %8s %8s | %10s %10s %10s | %s\n", "variant", "1wg:1w", "1wg:1/S", "1wg:2/S", "1wg:4/S", "chip:4/S", "chip ms", "clock GHz", "what");
    for (const Test& t : tests) {
        double one[4];
        const int threads[4] = {64, 256, 512, 1024};
        const int wavesPerSimd[4] = {1, 1, 2, 4};
        for (int c = 0; c < 4; c++) {
            const int iters = 20000;
            t.k<<<1, threads[c]>>>(dev, 1.0001f, 2000); // warm: code in the instruction cache
            t.k<<<1, threads[c]>>>(dev, 1.0001f, iters);
            CHECK(hipDeviceSynchronize());
            std::vector<Stamp> s(threads[c] / 64);
            CHECK(hipMemcpy(s.data(), dev, sizeof(Stamp) * s.size(), hipMemcpyDeviceToHost));
            double longest, span;
            stats(s, &longest, &span);
            one[c] = longest / ((double)iters * t.bodyLen * wavesPerSimd[c]);
        }
        // every CU, 4 waves per SIMD (128 VGPRs per lane: one 1024-thread workgroup fills a CU's register file)
        const int iters = 40000;
        t.k<<<cus, 1024>>>(dev, 1.0001f, 2000);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        t.k<<<cus, 1024>>>(dev, 1.0001f, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<Stamp> s((size_t)cus * 16);
        CHECK(hipMemcpy(s.data(), dev, sizeof(Stamp) * s.size(), hipMemcpyDeviceToHost));
        double longest, span;
        stats(s, &longest, &span);
        const double chip = longest / ((double)iters * t.bodyLen * 4);
        // the cycles of the longest wave over the launch's wall time = the clock the chip sustained under this load (s_memtime counters of different XCDs are
        // not synchronised: no first-start .. last-end span across waves)
        const double ghz = longest / (ms * 1e-3) / 1e9;
        printf("%-28s | %8.2f %8.2f %8.2f %8.2f | %10.2f %10.3f %10.3f | %s\n", t.name, one[0], one[1], one[2], one[3], chip, ms, ghz, t.note);
    }
    return 0;
}
'''


def main():
    init = "\\n".join("v_mov_b32 v%d, %%0" % r for r in range(ACC0, 128))
    out = HEADER.replace("INIT_BODY", init)
    tests = []
    for name, note, text, n in VARIANTS:
        out += KERNEL % {"name": name, "body": text, "clob": CLOBBERS}
        tests.append('        {"%s", "%s", k_%s, %d},' % (name, note.replace('"', "'"), name, n))
    out += MAIN.replace("TESTS", "\n".join(tests))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_issue_probe.hip")
    with open(path, "w") as fh:
        fh.write(out)
    print(path)


if __name__ == "__main__":
    main()
