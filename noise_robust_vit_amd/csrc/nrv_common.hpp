// Common device/host helpers for the gfx950 (MI355X, CDNA4) kernels of libnrv_hip.so.
// wave = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "nrv.h"

#define NRV_WAVE 64
#define NRV_INTERNAL __attribute__((visibility("hidden")))     // cross-file helpers: not part of the C ABI

typedef unsigned short bf16_t;                                    // raw bfloat16 storage
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;       // one MFMA 16x16x32 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;       // one ds_read_b64_tr_b16 result
typedef __attribute__((ext_vector_type(4))) float f32x4_t;        // one MFMA 16x16 accumulator tile / 16 B
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---------------------------------------------------------------------------------------------
// bf16 <-> f32 (round-to-nearest-even through the compiler cast: v_cvt_pk_bf16_f32, NaN stays NaN)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16hi_to_f32(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    f32x2_t f = {lo, hi};
    bf16x2v_t h = __builtin_convertvector(f, bf16x2v_t);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ unsigned short f32_to_bf16(float a) {
    __bf16 h = (__bf16)a;
    return __builtin_bit_cast(unsigned short, h);
}

// ---------------------------------------------------------------------------------------------
// wave reductions (64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// exact-erf GELU pieces.  erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7): one v_rcp, one v_exp.
//   Phi(u) = 0.5 (1 + erf(u / sqrt 2));   phi(u) = exp(-u^2/2) / sqrt(2 pi)
// gelu(u) = u Phi(u)  (nn.GELU() default, simple_vit.py:40);  gelu'(u) = Phi(u) + u phi(u)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void gelu_parts(float u, float& Phi, float& phi) {
    const float x = fabsf(u) * 0.70710678118654752f;               // |u| / sqrt(2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    // e = exp(-x^2) = exp2(-u^2/2 * log2 e)
    const float e = __builtin_amdgcn_exp2f(u * u * -0.72134752044448170f);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfc_abs = p * t * e;                              // 1 - erf(|x|)
    const float half_erfc = 0.5f * erfc_abs;
    Phi = (u >= 0.0f) ? (1.0f - half_erfc) : half_erfc;
    phi = e * 0.39894228040143268f;
}
__device__ __forceinline__ float gelu_fwd(float u) {
    float Phi, phi;
    gelu_parts(u, Phi, phi);
    return u * Phi;
}
__device__ __forceinline__ void gelu_both(float u, float& g, float& dg) {
    float Phi, phi;
    gelu_parts(u, Phi, phi);
    g = u * Phi;
    dg = fmaf(u, phi, Phi);
}
// four elements at once, written on vectors so that hipcc emits packed f32 instructions (v_pk_fma_f32 / v_pk_mul_f32:
// two elements per issue slot; the epilogue that calls this is VALU-issue bound).  The 0.5 of Phi is folded into the
// polynomial, |u| into the fma's source modifier, the sign select into one v_bfi (copysign).
__device__ __forceinline__ void gelu_both4(f32x4_t u, f32x4_t& g, f32x4_t& dg) {
    f32x4_t t, e;
    const f32x4_t w = u * 0.84932180028801904f;                                  // sqrt(log2(e) / 2): w^2 = u^2/2 * log2 e
    const f32x4_t w2 = w * w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[j] = __builtin_amdgcn_rcpf(fmaf(fabsf(u[j]), 0.3275911f * 0.70710678118654752f, 1.0f));
        e[j] = __builtin_amdgcn_exp2f(-w2[j]);                                   // exp(-u^2 / 2): the negation is a source modifier
    }
    // h = erfc(|u| / sqrt 2) / 2 in [0, 0.5].  Phi = 1/2 + sign(u) (1/2 - h) is written 1/2 + copysign(h - 1/2, u): copysign
    // ignores the sign of its first argument, so no operand has to be negated (hipcc spends a v_xor per element on a
    // negated operand of a packed f32 instruction)
    f32x4_t q = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
    q = q * t + (0.5f * 1.421413741f);
    q = q * t + (0.5f * -0.284496736f);
    q = q * t + (0.5f * 0.254829592f);
    q = q * t;
    const f32x4_t z = e * q + (-0.5f);
    f32x4_t Phi;
#pragma unroll
    for (int j = 0; j < 4; ++j) Phi[j] = 0.5f + __builtin_copysignf(z[j], u[j]);
    const f32x4_t phi = e * 0.39894228040143268f;
    g = u * Phi;
    dg = u * phi + Phi;
}
__device__ __forceinline__ float gelu_grad(float u) {
    float Phi, phi;
    gelu_parts(u, Phi, phi);
    return fmaf(u, phi, Phi);
}

// ---------------------------------------------------------------------------------------------
// buffer resources: out-of-range offsets read as zero, which is how every tile tail is handled.
// The descriptor is built from wave-uniform values only (blockIdx / kernel arguments).
// ---------------------------------------------------------------------------------------------
#define NRV_OOB 0x80000000u   // a voffset that is always >= num_records (records are clamped below this)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint64_t bytes) {
    const unsigned rec = bytes > 0x7fffffffull ? 0x7fffffffu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)rec, 0x00020000);
}

// 16-byte LDS-DMA: lane l's 16 bytes land at lds_base + 16*l (lds_base wave-uniform).
// Issued through inline asm so that hipcc's waitcnt pass does not know an LDS-DMA is in flight: with the builtin
// it drains vmcnt(0) in front of every later ds_read_b64_tr_b16 (seen in the TN kernel's .s), which serialises
// load and compute.  The kernels wait for these loads themselves (s_waitcnt vmcnt(0) + barrier before the reads).
// M0 (the LDS base of the DMA) is written inside the statement and declared clobbered; one wait state between the
// scalar write of M0 and the buffer instruction that reads it (what the compiler itself emits for the intrinsic).  Saving and
// restoring M0 around the load with 5 wait states cost the GEMM K loops 2-3 % (profiles/r02_ab_light_m0.txt).
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, unsigned voffset) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_base));
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %0, %1, 0 offen lds"
        :
        : "v"(voffset), "s"(rsrc), "s"(lds_addr)
        : "memory", "m0");
}

// the same, non-temporal
__device__ __forceinline__ void dma16_nt(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, unsigned voffset) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_base));
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %0, %1, 0 offen nt lds"
        :
        : "v"(voffset), "s"(rsrc), "s"(lds_addr)
        : "memory", "m0");
}

// the same with a scalar byte offset added by the memory unit (soffset): a K loop advances the SCALAR and keeps the per-lane
// offset constant, so a DMA costs no vector-ALU instruction (the range check subtracts soffset from the record count)
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, unsigned voffset, unsigned soffset) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_base));
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %0, %1, %3 offen lds"
        :
        : "v"(voffset), "s"(rsrc), "s"(lds_addr), "s"(soffset)
        : "memory", "m0");
}

// the same, LDS destination given as a (wave-uniform) LDS byte address: no generic-pointer cast per instruction
__device__ __forceinline__ void dma16s_at(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr, unsigned voffset, unsigned soffset) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %0, %1, %3 offen lds"
        :
        : "v"(voffset), "s"(rsrc), "s"(lds_addr), "s"(soffset)
        : "memory", "m0");
}

__device__ __forceinline__ bf16x8_t lds_read_b128(const void* p) {
    return *reinterpret_cast<const bf16x8_t*>(p);
}
__device__ __forceinline__ bf16x8_t lds_read_b128_at(unsigned lds_addr) {      // by LDS byte address
    return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8_t*>(static_cast<uintptr_t>(lds_addr));
}
__device__ __forceinline__ bf16x4_t lds_read_tr16_b64(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) bf16x4_t*)(p));
}
__device__ __forceinline__ bf16x4_t lds_read_tr16_b64_at(unsigned lds_addr) {  // by LDS byte address
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        reinterpret_cast<__attribute__((address_space(3))) bf16x4_t*>(static_cast<uintptr_t>(lds_addr)));
}
__device__ __forceinline__ bf16x8_t cat4(bf16x4_t a, bf16x4_t b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// XCD-aware bijective block remap (8 XCDs, blocks are dealt round-robin): consecutive logical
// ids end up on one XCD so that neighbouring tiles share an L2.  Speed only, never correctness.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, idx = bid >> 3;
    const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
#define NRV_CHECK_LAUNCH()                          \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

static inline bool nrv_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t nrv_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
