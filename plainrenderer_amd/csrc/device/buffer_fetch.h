// Texel fetches through buffer instructions with the ELEMENT INDEX in a VGPR (MUBUF, idxen): the texture addresser scales the index by the stride in the
// resource descriptor and adds the base, so a fetch of base[index] costs no address arithmetic on the VALU. As a global load the same fetch needs a
// v_lshl_add_u64 per site (shift + 64-bit base), 4.65 cycles per wave64 on MI355X (profiles/r04_valu_rates.txt) - in kernels that issue at 3.6 - 3.7 cycles
// per instruction and spend 88 - 94 % of their time issuing (profiles/*_isa_mix.txt). The builtins of this compiler only cover the raw (byte offset) form,
// so the structured form is reached through the LLVM intrinsic's name; the compiler tracks these loads (s_waitcnt vmcnt) like any other.
// The descriptor is four SGPRs built from a kernel-argument pointer: base, stride, no range check (the callers clamp their indices), dword 3 as the
// compiler's own buffer resources on gfx90a / gfx942 / gfx950 (DATA_FORMAT = 32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plr {

typedef int BufferDesc __attribute__((ext_vector_type(4)));
typedef int BufferWords2 __attribute__((ext_vector_type(2)));
typedef int BufferWords4 __attribute__((ext_vector_type(4)));

__device__ short plrStructBufferLoadShort(BufferDesc rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.i16");
__device__ int plrStructBufferLoad1(BufferDesc rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.i32");
__device__ BufferWords2 plrStructBufferLoad2(BufferDesc rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v2i32");
__device__ BufferWords4 plrStructBufferLoad4(BufferDesc rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4i32");

// elementBytes: 2, 4, 8 or 16 (the stride field has 14 bits)
__device__ __forceinline__ BufferDesc texelBuffer(const void* base, uint32_t elementBytes) {
    const uint64_t a = (uint64_t)(uintptr_t)base;
    BufferDesc d;
    d.x = (int)(uint32_t)a;
    d.y = (int)(((uint32_t)(a >> 32) & 0xffffu) | (elementBytes << 16));
    d.z = -1; // NUM_RECORDS: every index is in range
    d.w = 0x00020000;
    return d;
}
__device__ __forceinline__ uint32_t fetch16(BufferDesc b, uint32_t index) { return (uint32_t)(uint16_t)plrStructBufferLoadShort(b, (int)index, 0, 0, 0); }
__device__ __forceinline__ uint32_t fetch32(BufferDesc b, uint32_t index) { return (uint32_t)plrStructBufferLoad1(b, (int)index, 0, 0, 0); }
__device__ __forceinline__ uint2 fetch64(BufferDesc b, uint32_t index) { const BufferWords2 v = plrStructBufferLoad2(b, (int)index, 0, 0, 0); return make_uint2((uint32_t)v.x, (uint32_t)v.y); }
__device__ __forceinline__ uint4 fetch128(BufferDesc b, uint32_t index) {
    const BufferWords4 v = plrStructBufferLoad4(b, (int)index, 0, 0, 0);
    return make_uint4((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w);
}

} // namespace plr
