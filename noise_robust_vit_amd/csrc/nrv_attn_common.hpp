// Shared device helpers of the fused attention kernels (softmax: nrv_attn.hip, Sinkhorn: nrv_sinkhorn.hip).
// LDS images of one head: [rows][64] bf16 = 128-byte rows.
#pragma once
#include "nrv_common.hpp"

namespace nrv_attn {

constexpr int DH = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// row image with 128-byte rows, 16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7)
__device__ __forceinline__ int img_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }
// V image: 32-byte unit u of row r at unit position u ^ ((r >> 1) & 3) (conflict-free transposed reads)
__device__ __forceinline__ int vimg_off(int r, int c) { return r * 128 + ((((c >> 1) ^ ((r >> 1) & 3)) << 5) | ((c & 1) << 4)); }

// cooperative load of one [N x 64] bf16 head slice (row stride ld elements) into an LDS image of NP rows;
// rows >= N are zero filled.
template <int NP, bool VIMG, int THREADS>
__device__ __forceinline__ void load_image(char* img, const bf16_t* src, long long ld, int N, int tid) {
    for (int idx = tid; idx < NP * 8; idx += THREADS) {
        const int r = idx >> 3, c = idx & 7;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (r < N) v = *reinterpret_cast<const u32x4_t*>(src + (long long)r * ld + c * 8);
        *reinterpret_cast<u32x4_t*>(img + (VIMG ? vimg_off(r, c) : img_off(r, c))) = v;
    }
}

__device__ __forceinline__ bf16x8_t load_frag_global(const bf16_t* p) {
    return *reinterpret_cast<const bf16x8_t*>(p);
}

__device__ __forceinline__ bf16x8_t pack_frag(const f32x4_t& lo, const f32x4_t& hi) {
    u32x4_t u = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
    return __builtin_bit_cast(bf16x8_t, u);
}

__device__ __forceinline__ void store_bf16x4(bf16_t* p, const f32x4_t& v) {
    u32x2_t pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *reinterpret_cast<u32x2_t*>(p) = pk;
}

// transposed fragment (A operand, 16 columns x 32 rows) from a GEMM-swizzled row image:
// rows rb + 16 r + 4 g + q (r = 0,1), columns 16 dt .. 16 dt + 15
__device__ __forceinline__ bf16x8_t tr_frag_img(const char* img, int rb, int dt, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int r0 = rb + 4 * g + q;
    const int c = 2 * dt + (pp >> 1);
    const char* a0 = img + img_off(r0, c) + (pp & 1) * 8;
    const char* a1 = img + img_off(r0 + 16, c) + (pp & 1) * 8;
    return cat4(lds_read_tr16_b64(a0), lds_read_tr16_b64(a1));
}
__device__ __forceinline__ bf16x8_t tr_frag_vimg(const char* img, int rb, int dt, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int r0 = rb + 4 * g + q;
    const char* a0 = img + r0 * 128 + ((dt ^ ((r0 >> 1) & 3)) << 5) + pp * 8;
    const char* a1 = img + (r0 + 16) * 128 + ((dt ^ (((r0 + 16) >> 1) & 3)) << 5) + pp * 8;
    return cat4(lds_read_tr16_b64(a0), lds_read_tr16_b64(a1));
}
// row fragment (16 rows x 32 k) from a GEMM-swizzled row image: row rb + (lane & 15), k = 32 ks + 8 (lane >> 4) ..
__device__ __forceinline__ bf16x8_t row_frag_img(const char* img, int rb, int ks, int lane) {
    return lds_read_b128(img + img_off(rb + (lane & 15), ks * 4 + (lane >> 4)));
}


}  // namespace nrv_attn
