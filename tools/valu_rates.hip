// Instruction issue rates on gfx950 (MI355X): cycles per wave64 instruction per SIMD for the VALU / LDS-crossbar operations the
// VALU-bound kernels of this pipeline are made of. Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Each test runs ITER x 8 independent copies of one instruction per wave, 8 waves per SIMD on every CU; the wall time of the launch
// divided by the instructions one SIMD issued gives seconds per instruction, reported as cycles at the measured shader clock
// (s_memtime delta / wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 4096;

#define BODY8(ASM, ...)                                                                                               \
    for (int i = 0; i < ITER; i++) {                                                                                  \
        asm volatile(ASM(0) "\n" ASM(1) "\n" ASM(2) "\n" ASM(3) "\n" ASM(4) "\n" ASM(5) "\n" ASM(6) "\n" ASM(7)     \
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                     : "v"(b), "v"(c) : "vcc", "s4", "s5", "s6");                                                     \
    }

#define KERNEL(NAME, ASM)                                                                                             \
    __global__ __launch_bounds__(256) void NAME(float* out, unsigned long long* clk, float bIn, float cIn) {         \
        float r[8];                                                                                                   \
        for (int k = 0; k < 8; k++) r[k] = bIn * (float)(threadIdx.x + k + 1);                                        \
        float b = bIn, c = cIn;                                                                                       \
        unsigned long long t0 = __builtin_readcyclecounter();                                                         \
        BODY8(ASM)                                                                                                    \
        unsigned long long t1 = __builtin_readcyclecounter();                                                         \
        float s = 0.f;                                                                                                \
        for (int k = 0; k < 8; k++) s += r[k];                                                                        \
        if (s == 123.456f) out[0] = s;                                                                                \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;                                                    \
    }

#define PK_KERNEL(NAME, ASM)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(float* out, unsigned long long* clk, float bIn, float cIn) {         \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                         \
        f2 r[8];                                                                                                      \
        for (int k = 0; k < 8; k++) r[k] = f2{bIn * (float)(threadIdx.x + k + 1), bIn};                               \
        f2 b{bIn, bIn * 2.f}, c{cIn, cIn * 3.f};                                                                      \
        unsigned long long t0 = __builtin_readcyclecounter();                                                         \
        BODY8(ASM)                                                                                                    \
        unsigned long long t1 = __builtin_readcyclecounter();                                                         \
        float s = 0.f;                                                                                                \
        for (int k = 0; k < 8; k++) s += r[k].x + r[k].y;                                                             \
        if (s == 123.456f) out[0] = s;                                                                                \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;                                                    \
    }

#define A_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9"
#define A_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8"
#define A_ADD(k) "v_add_f32 %" #k ", %" #k ", %8"
#define A_MAC(k) "v_fmac_f32 %" #k ", %8, %9"
#define A_MAX(k) "v_max_f32 %" #k ", %" #k ", %8"
#define A_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9"
#define A_ADDU(k) "v_add_u32 %" #k ", %" #k ", %8"
#define A_AND(k) "v_and_b32 %" #k ", %" #k ", %8"
#define A_LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k
#define A_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 1, %8"
#define A_MADU24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9"
#define A_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8"
#define A_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 9"
#define A_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9"
#define A_MOV(k) "v_mov_b32 %" #k ", %8"
#define A_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc"
#define A_CMP(k) "v_cmp_lt_f32 vcc, %" #k ", %8"
#define A_CVTFU(k) "v_cvt_f32_u32 %" #k ", %" #k
#define A_CVTUF(k) "v_cvt_u32_f32 %" #k ", %" #k
#define A_CVTF16(k) "v_cvt_f16_f32 %" #k ", %" #k
#define A_CVTF32H(k) "v_cvt_f32_f16 %" #k ", %" #k
#define A_FLOOR(k) "v_floor_f32 %" #k ", %" #k
#define A_FRACT(k) "v_fract_f32 %" #k ", %" #k
#define A_RCP(k) "v_rcp_f32 %" #k ", %" #k
#define A_RSQ(k) "v_rsq_f32 %" #k ", %" #k
#define A_SQRT(k) "v_sqrt_f32 %" #k ", %" #k
#define A_EXP(k) "v_exp_f32 %" #k ", %" #k
#define A_LOG(k) "v_log_f32 %" #k ", %" #k
#define A_SIN(k) "v_sin_f32 %" #k ", %" #k
#define A_DPP_SHR(k) "v_mov_b32_dpp %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf"
#define A_DPP_WSHR(k) "v_mov_b32_dpp %" #k ", %" #k " wave_shr:1 row_mask:0xf bank_mask:0xf"
#define A_DPP_WSHL(k) "v_mov_b32_dpp %" #k ", %" #k " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define A_DPP_FMAC(k) "v_fmac_f32_dpp %" #k ", %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf"
#define A_DPP_ADD(k) "v_add_f32_dpp %" #k ", %" #k ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define A_DPP_BCAST(k) "v_mov_b32_dpp %" #k ", %" #k " row_bcast:15 row_mask:0xa bank_mask:0xf"
#define A_BPERM(k) "ds_bpermute_b32 %" #k ", %8, %" #k "\ns_waitcnt lgkmcnt(0)"
#define A_SWIZ(k) "ds_swizzle_b32 %" #k ", %" #k " offset:0x041F\ns_waitcnt lgkmcnt(0)"
#define A_PERMLANE(k) "v_permlane32_swap_b32 %" #k ", %" #k
#define A_FMAMIX(k) "v_fma_mix_f32 %" #k ", %" #k ", %8, %9"
#define A_PKFMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9"
#define A_PKMUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %8"
#define A_PKADD(k) "v_pk_add_f32 %" #k ", %" #k ", %8"
#define A_PKMOV(k) "v_pk_mov_b32 %" #k ", %8, %9"
#define A_PKFMA16(k) "v_pk_fma_f16 %" #k ", %" #k ", %8, %9"
#define A_PKADD16(k) "v_pk_add_f16 %" #k ", %" #k ", %8"
#define A_PKMAX16(k) "v_pk_max_f16 %" #k ", %" #k ", %8"
#define A_CVTPKRTZ(k) "v_cvt_pkrtz_f16_f32 %" #k ", %" #k ", %8"
#define A_MIN3(k) "v_min3_f32 %" #k ", %" #k ", %8, %9"
#define A_LDEXP(k) "v_ldexp_f32 %" #k ", %" #k ", 1"
#define A_ALIGNBIT(k) "v_alignbit_b32 %" #k ", %" #k ", %8, 7"
#define A_BFI(k) "v_bfi_b32 %" #k ", %" #k ", %8, %9"
#define A_XOR3(k) "v_xad_u32 %" #k ", %" #k ", %8, %9"
#define A_AND_OR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9"
#define A_OR(k) "v_or_b32 %" #k ", %" #k ", %8"
#define A_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8"
#define A_SUBF(k) "v_sub_f32 %" #k ", %" #k ", %8"
#define A_MIN(k) "v_min_f32 %" #k ", %" #k ", %8"
#define A_MAXI(k) "v_max_i32 %" #k ", %" #k ", %8"
#define A_MED3I(k) "v_med3_i32 %" #k ", %" #k ", %8, %9"
#define A_DOT2(k) "v_dot2_f32_f16 %" #k ", %8, %9, %" #k
#define A_CVTI(k) "v_cvt_i32_f32 %" #k ", %" #k
#define A_CVTUB(k) "v_cvt_f32_ubyte1 %" #k ", %" #k
#define A_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9"
#define A_LSHRREV(k) "v_lshrrev_b32 %" #k ", 16, %" #k
#define A_CND64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[4:5]"
#define A_FMAK(k) "v_fmaak_f32 %" #k ", %" #k ", %8, 0x3f000000"
#define A_CMPCND(k) "v_cmp_lt_f32 vcc, %" #k ", %8\nv_cndmask_b32 %" #k ", %" #k ", %9, vcc"
#define A_CMPCND64(k) "v_cmp_lt_f32_e64 s[4:5], %" #k ", %8\nv_cndmask_b32_e64 %" #k ", %" #k ", %9, s[4:5]"
#define A_CND_VCCSET(k) "v_cndmask_b32 %" #k ", %8, %9, vcc"
#define A_MULSGPR(k) "v_mul_f32 %" #k ", s6, %" #k
#define A_FMA2S(k) "v_fma_f32 %" #k ", %" #k ", s6, %8"

KERNEL(k_fma, A_FMA) KERNEL(k_mul, A_MUL) KERNEL(k_add, A_ADD) KERNEL(k_mac, A_MAC) KERNEL(k_max, A_MAX) KERNEL(k_med3, A_MED3)
KERNEL(k_addu, A_ADDU) KERNEL(k_and, A_AND) KERNEL(k_lshl, A_LSHL) KERNEL(k_lshladd, A_LSHLADD) KERNEL(k_madu24, A_MADU24) KERNEL(k_mullo, A_MULLO)
KERNEL(k_bfe, A_BFE) KERNEL(k_perm, A_PERM) KERNEL(k_mov, A_MOV) KERNEL(k_cndmask, A_CNDMASK) KERNEL(k_cmp, A_CMP)
KERNEL(k_cvtfu, A_CVTFU) KERNEL(k_cvtuf, A_CVTUF) KERNEL(k_cvtf16, A_CVTF16) KERNEL(k_cvtf32h, A_CVTF32H) KERNEL(k_floor, A_FLOOR) KERNEL(k_fract, A_FRACT)
KERNEL(k_rcp, A_RCP) KERNEL(k_rsq, A_RSQ) KERNEL(k_sqrt, A_SQRT) KERNEL(k_exp, A_EXP) KERNEL(k_log, A_LOG) KERNEL(k_sin, A_SIN)
KERNEL(k_dpp_wshr, A_DPP_WSHR) KERNEL(k_dpp_wshl, A_DPP_WSHL) KERNEL(k_dpp_fmac, A_DPP_FMAC) KERNEL(k_dpp_shr, A_DPP_SHR) KERNEL(k_dpp_add, A_DPP_ADD) KERNEL(k_dpp_bcast, A_DPP_BCAST) KERNEL(k_bperm, A_BPERM) KERNEL(k_swiz, A_SWIZ)
KERNEL(k_permlane, A_PERMLANE) KERNEL(k_fmamix, A_FMAMIX) KERNEL(k_pkfma16, A_PKFMA16) KERNEL(k_pkadd16, A_PKADD16) KERNEL(k_pkmax16, A_PKMAX16)
KERNEL(k_cvtpkrtz, A_CVTPKRTZ) KERNEL(k_min3, A_MIN3) KERNEL(k_ldexp, A_LDEXP) KERNEL(k_alignbit, A_ALIGNBIT) KERNEL(k_bfi, A_BFI) KERNEL(k_xad, A_XOR3)
KERNEL(k_and_or, A_AND_OR)
KERNEL(k_or, A_OR) KERNEL(k_xor, A_XOR) KERNEL(k_subf, A_SUBF) KERNEL(k_min, A_MIN) KERNEL(k_maxi, A_MAXI) KERNEL(k_med3i, A_MED3I) KERNEL(k_dot2, A_DOT2)
KERNEL(k_cvti, A_CVTI) KERNEL(k_cvtub, A_CVTUB) KERNEL(k_add3, A_ADD3) KERNEL(k_lshrrev, A_LSHRREV) KERNEL(k_cnd64, A_CND64) KERNEL(k_fmak, A_FMAK)
KERNEL(k_mulsgpr, A_MULSGPR) KERNEL(k_fma2s, A_FMA2S) KERNEL(k_cmpcnd, A_CMPCND) KERNEL(k_cmpcnd64, A_CMPCND64) KERNEL(k_cndvccset, A_CND_VCCSET)
PK_KERNEL(k_pkfma, A_PKFMA) PK_KERNEL(k_pkmul, A_PKMUL) PK_KERNEL(k_pkadd, A_PKADD) PK_KERNEL(k_pkmov, A_PKMOV)

// ---- round 4 additions: 64-bit address arithmetic, the conversions / integer forms the big kernels actually use, VOP3 encodings with modifiers, f16 scalar ops
#define A_LSHLADD64(k) "v_lshl_add_u64 %" #k ", %" #k ", 2, %8"
#define A_MAD64B(k) "v_mad_u64_u32 %" #k ", s[4:5], %9, %9, %" #k
#define A_MAD64(k) "v_mad_u64_u32 %" #k ", s[4:5], %9, %9, %" #k
#define A_CVTFLR(k) "v_cvt_flr_i32_f32 %" #k ", %" #k
#define A_MULU24(k) "v_mul_u32_u24 %" #k ", %" #k ", %8"
#define A_ASHR(k) "v_ashrrev_i32 %" #k ", 8, %" #k
#define A_SUBU(k) "v_sub_u32 %" #k ", %" #k ", %8"
#define A_CVTFI(k) "v_cvt_f32_i32 %" #k ", %" #k
#define A_CVTUB0(k) "v_cvt_f32_ubyte0 %" #k ", %" #k
#define A_FMAMIXHI(k) "v_fma_mix_f32 %" #k ", %8, %9, %" #k " op_sel:[1,0,0] op_sel_hi:[1,0,0]"
#define A_FMAMIXLO(k) "v_fma_mixlo_f16 %" #k ", %" #k ", %8, %9"
#define A_CMPU(k) "v_cmp_ge_u32 vcc, %" #k ", %8"
#define A_MULABS(k) "v_mul_f32_e64 %" #k ", |%" #k "|, %8"
#define A_FMANEG(k) "v_fma_f32 %" #k ", -%" #k ", %8, %9"
#define A_FMAINL(k) "v_fma_f32 %" #k ", %" #k ", %8, 1.0"
#define A_FMALIT(k) "v_fma_f32 %" #k ", %" #k ", %8, 0x40490fdb"
#define A_MULLIT(k) "v_mul_f32 %" #k ", 0x40490fdb, %" #k
#define A_ADDSGPR(k) "v_add_f32 %" #k ", s6, %" #k
#define A_ANDSGPR(k) "v_and_b32 %" #k ", s6, %" #k
#define A_FMAC_S(k) "v_fmac_f32 %" #k ", s6, %8"
#define A_MULF16(k) "v_mul_f16 %" #k ", %" #k ", %8"
#define A_FMAF16(k) "v_fma_f16 %" #k ", %" #k ", %8, %9"
#define A_PKMULF16(k) "v_pk_mul_f16 %" #k ", %" #k ", %8"
#define A_SDWA_CVT(k) "v_cvt_f32_f16_sdwa %" #k ", %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
#define A_SDWA_MUL(k) "v_mul_f32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"
#define A_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9"
#define A_SUBREV(k) "v_subrev_f32 %" #k ", %8, %" #k
#define A_MULLEG(k) "v_mul_legacy_f32 %" #k ", %" #k ", %8"
#define A_ADDCO(k) "v_add_co_u32 %" #k ", vcc, %" #k ", %8"
#define A_RNDNE(k) "v_rndne_f32 %" #k ", %" #k
#define A_TRUNC(k) "v_trunc_f32 %" #k ", %" #k
#define A_CVTPKU8(k) "v_cvt_pk_u8_f32 %" #k ", %" #k ", 1, %8"
#define A_NOT(k) "v_not_b32 %" #k ", %" #k
#define A_BCNT(k) "v_bcnt_u32_b32 %" #k ", %" #k ", %8"
#define A_MBCNT(k) "v_mbcnt_lo_u32_b32 %" #k ", %" #k ", %8"
#define A_READLANE(k) "v_readfirstlane_b32 s4, %" #k
#define A_LSHLREV16(k) "v_lshlrev_b32 %" #k ", 16, %" #k
#define A_MULPOW2(k) "v_mul_u32_u24 %" #k ", 16, %" #k
#define A_DSREAD32(k) "ds_read_b32 %" #k ", %8\ns_waitcnt lgkmcnt(0)"
#define A_DSREAD64x(k) "ds_read_b32 %" #k ", %8 offset:" #k "00"
KERNEL(k_cvtflr, A_CVTFLR) KERNEL(k_mulu24, A_MULU24) KERNEL(k_ashr, A_ASHR) KERNEL(k_subu, A_SUBU) KERNEL(k_cvtfi, A_CVTFI)
KERNEL(k_cvtub0, A_CVTUB0) KERNEL(k_fmamixhi, A_FMAMIXHI) KERNEL(k_fmamixlo, A_FMAMIXLO) KERNEL(k_cmpu, A_CMPU) KERNEL(k_mulabs, A_MULABS) KERNEL(k_fmaneg, A_FMANEG)
KERNEL(k_fmainl, A_FMAINL) KERNEL(k_mullit, A_MULLIT) KERNEL(k_addsgpr, A_ADDSGPR) KERNEL(k_andsgpr, A_ANDSGPR) KERNEL(k_fmacs, A_FMAC_S)
KERNEL(k_mulf16, A_MULF16) KERNEL(k_fmaf16, A_FMAF16) KERNEL(k_pkmulf16, A_PKMULF16) KERNEL(k_sdwacvt, A_SDWA_CVT) KERNEL(k_sdwamul, A_SDWA_MUL) KERNEL(k_max3, A_MAX3)
KERNEL(k_subrev, A_SUBREV) KERNEL(k_mulleg, A_MULLEG) KERNEL(k_addco, A_ADDCO) KERNEL(k_rndne, A_RNDNE) KERNEL(k_trunc, A_TRUNC) KERNEL(k_cvtpku8, A_CVTPKU8)
KERNEL(k_not, A_NOT) KERNEL(k_bcnt, A_BCNT) KERNEL(k_mbcnt, A_MBCNT) KERNEL(k_readlane, A_READLANE) KERNEL(k_lshl16, A_LSHLREV16) KERNEL(k_mulpow2, A_MULPOW2)

#define U64_KERNEL(NAME, ASM)                                                                                         \
    __global__ __launch_bounds__(256) void NAME(float* out, unsigned long long* clk, float bIn, float cIn) {         \
        unsigned long long r[8];                                                                                      \
        for (int k = 0; k < 8; k++) r[k] = (unsigned long long)(bIn * (float)(threadIdx.x + k + 1));                  \
        unsigned long long b = (unsigned long long)(bIn * 3.f); unsigned c = (unsigned)(cIn * 8.f);                   \
        unsigned long long t0 = __builtin_readcyclecounter();                                                         \
        BODY8(ASM)                                                                                                    \
        unsigned long long t1 = __builtin_readcyclecounter();                                                         \
        unsigned long long s = 0;                                                                                     \
        for (int k = 0; k < 8; k++) s += r[k];                                                                        \
        if (s == 123456ull) out[0] = 1.f;                                                                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;                                                    \
    }
U64_KERNEL(k_lshladd64, A_LSHLADD64) U64_KERNEL(k_mad64, A_MAD64B)

typedef void (*kern_t)(float*, unsigned long long*, float, float);
struct Test { const char* name; kern_t k; int instrPerSlot; };

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&clk, 64));
    std::vector<Test> tests = {
        {"v_fma_f32", k_fma, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1}, {"v_fmac_f32", k_mac, 1}, {"v_max_f32", k_max, 1}, {"v_med3_f32", k_med3, 1},
        {"v_min3_f32", k_min3, 1}, {"v_pk_fma_f32", k_pkfma, 1}, {"v_pk_mul_f32", k_pkmul, 1}, {"v_pk_add_f32", k_pkadd, 1}, {"v_pk_mov_b32", k_pkmov, 1},
        {"v_pk_fma_f16", k_pkfma16, 1}, {"v_pk_add_f16", k_pkadd16, 1}, {"v_pk_max_f16", k_pkmax16, 1}, {"v_fma_mix_f32", k_fmamix, 1},
        {"v_add_u32", k_addu, 1}, {"v_and_b32", k_and, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_lshl_add_u32", k_lshladd, 1}, {"v_mad_u32_u24", k_madu24, 1},
        {"v_mul_lo_u32", k_mullo, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_bfi_b32", k_bfi, 1}, {"v_perm_b32", k_perm, 1}, {"v_alignbit_b32", k_alignbit, 1},
        {"v_xad_u32", k_xad, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_mov_b32", k_mov, 1}, {"v_cndmask_b32", k_cndmask, 1}, {"v_cmp_lt_f32", k_cmp, 1},
        {"v_cvt_f32_u32", k_cvtfu, 1}, {"v_cvt_u32_f32", k_cvtuf, 1}, {"v_cvt_f16_f32", k_cvtf16, 1}, {"v_cvt_f32_f16", k_cvtf32h, 1}, {"v_cvt_pkrtz_f16_f32", k_cvtpkrtz, 1},
        {"v_floor_f32", k_floor, 1}, {"v_fract_f32", k_fract, 1}, {"v_ldexp_f32", k_ldexp, 1},
        {"v_rcp_f32", k_rcp, 1}, {"v_rsq_f32", k_rsq, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_exp_f32", k_exp, 1}, {"v_log_f32", k_log, 1}, {"v_sin_f32", k_sin, 1},
        {"v_or_b32", k_or, 1}, {"v_xor_b32", k_xor, 1}, {"v_sub_f32", k_subf, 1}, {"v_min_f32", k_min, 1}, {"v_max_i32", k_maxi, 1}, {"v_med3_i32", k_med3i, 1},
        {"v_dot2_f32_f16", k_dot2, 1}, {"v_cvt_i32_f32", k_cvti, 1}, {"v_cvt_f32_ubyte1", k_cvtub, 1}, {"v_add3_u32", k_add3, 1}, {"v_lshrrev_b32", k_lshrrev, 1},
        {"v_cndmask_b32_e64 (sgpr mask)", k_cnd64, 1}, {"v_fmaak_f32 (literal)", k_fmak, 1}, {"v_mul_f32 (sgpr src)", k_mulsgpr, 1}, {"v_fma_f32 (sgpr src)", k_fma2s, 1},
        {"v_cmp + v_cndmask (vcc) pair", k_cmpcnd, 2}, {"v_cmp_e64 + v_cndmask_e64 (s[4:5]) pair", k_cmpcnd64, 2}, {"v_cndmask_b32 vcc, no dst dependency", k_cndvccset, 1},
        {"v_mov_b32_dpp wave_shr:1", k_dpp_wshr, 1}, {"v_mov_b32_dpp wave_shl:1 bound_ctrl", k_dpp_wshl, 1}, {"v_fmac_f32_dpp row_shr:1", k_dpp_fmac, 1},
        {"v_mov_b32_dpp row_shr:1", k_dpp_shr, 1}, {"v_add_f32_dpp quad_perm", k_dpp_add, 1}, {"v_mov_b32_dpp row_bcast:15", k_dpp_bcast, 1},
        {"v_lshl_add_u64", k_lshladd64, 1}, {"v_mad_u64_u32", k_mad64, 1}, {"v_cvt_flr_i32_f32", k_cvtflr, 1}, {"v_mul_u32_u24", k_mulu24, 1}, {"v_ashrrev_i32", k_ashr, 1}, {"v_sub_u32", k_subu, 1},
        {"v_cvt_f32_i32", k_cvtfi, 1}, {"v_cvt_f32_ubyte0", k_cvtub0, 1}, {"v_fma_mix_f32 (f16 hi src)", k_fmamixhi, 1}, {"v_fma_mixlo_f16", k_fmamixlo, 1}, {"v_cmp_ge_u32", k_cmpu, 1},
        {"v_mul_f32_e64 |abs|", k_mulabs, 1}, {"v_fma_f32 neg src", k_fmaneg, 1}, {"v_fma_f32 inline 1.0", k_fmainl, 1}, {"v_mul_f32 literal", k_mullit, 1},
        {"v_add_f32 (sgpr src)", k_addsgpr, 1}, {"v_and_b32 (sgpr src)", k_andsgpr, 1}, {"v_fmac_f32 (sgpr src)", k_fmacs, 1}, {"v_mul_f16", k_mulf16, 1}, {"v_fma_f16", k_fmaf16, 1},
        {"v_pk_mul_f16", k_pkmulf16, 1}, {"v_cvt_f32_f16_sdwa WORD_1", k_sdwacvt, 1}, {"v_mul_f32_sdwa", k_sdwamul, 1}, {"v_max3_f32", k_max3, 1}, {"v_subrev_f32", k_subrev, 1},
        {"v_mul_legacy_f32", k_mulleg, 1}, {"v_add_co_u32", k_addco, 1}, {"v_rndne_f32", k_rndne, 1}, {"v_trunc_f32", k_trunc, 1}, {"v_cvt_pk_u8_f32", k_cvtpku8, 1}, {"v_not_b32", k_not, 1},
        {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_mbcnt_lo", k_mbcnt, 1}, {"v_readfirstlane_b32", k_readlane, 1}, {"v_lshlrev_b32 by 16", k_lshl16, 1}, {"v_mul_u32_u24 by 16", k_mulpow2, 1},
        {"v_permlane32_swap", k_permlane, 1}, {"ds_bpermute_b32 (+wait)", k_bperm, 1}, {"ds_swizzle_b32 (+wait)", k_swiz, 1},
    };
    const int wavesPerSimd = 8;
    const int blocks = cus * wavesPerSimd; // 256 threads = 4 waves = one per SIMD; 8 blocks per CU -> 8 waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-30s %10s %12s %14s %12s\n", "instruction", "ms", "cyc@2.4GHz", "Ginstr/s chip", "memtime");
    for (auto& t : tests) {
        t.k<<<blocks, 256>>>(out, clk, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        t.k<<<blocks, 256>>>(out, clk, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instrPerSimd = (double)ITER * 8 * wavesPerSimd * t.instrPerSlot;
        const double secPerInstr = ms * 1e-3 / instrPerSimd;
        unsigned long long ticks = 0;
        CHECK(hipMemcpy(&ticks, clk, sizeof(ticks), hipMemcpyDeviceToHost));
        // ticks: s_memtime delta of one wave over its ITER x 8 x instrPerSlot instructions (constant 100 MHz reference on gfx950, not the shader clock)
        printf("%-30s %10.3f %12.2f %14.1f %12llu\n", t.name, ms, secPerInstr * 2.4e9, instrPerSimd * cus * 4 / (ms * 1e-3) / 1e9, ticks);
    }
    return 0;
}
