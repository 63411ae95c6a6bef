// Fused multi-head self-attention, forward and backward, for gfx950 (MI355X).
//
// ViT heads are tiny (N <= 256 tokens, dh = 64): one workgroup owns one (batch, head), the whole K and V (or Q and dO)
// of the head live in LDS as swizzled row images, and the [N,N] score matrix never leaves the CU (the reference
// materialises it three times: simple_vit.py:70-74).  q/k/v are read in place from the QKV projection's
// [B, N, 3*H*dh] output (128-byte head segments) and the output is written in 'b n (h d)' order, so the einops
// rearrange copies (simple_vit.py:68,75) disappear.
//
// MFMA orientation (16x16x32 bf16): scores are computed TRANSPOSED, S^T = K Q^T, so a lane owns one query column and
// its keys sit in registers: row max / row sum are register reductions plus two shuffles, and the bf16 P^T accumulator
// IS the B operand of O^T = V^T P^T (no lane movement).  V^T (and K^T, Q^T, dO^T in the backward) fragments come from
// ds_read_b64_tr_b16.
//
// "Fat waves": 4 waves of <= 256 VGPRs per workgroup (one per SIMD), 2-3 workgroups per CU; a wave owns TWO 16-row
// tiles so that every fragment read from LDS feeds both; images arrive by 16-byte LDS-DMA.  What the earlier
// thin-wave form (8 waves x 128 VGPRs, one tile per wave, register-staged images) lost, measured with s_memtime stamps:
// every ds_read was followed by a full lgkmcnt(0) wait (no registers to run reads ahead), 6.8 us of arithmetic per head.
//
// Backward = two kernels with the same structure and no atomics:
//   dq kernel : wave owns 32 queries, sweeps keys    -> dQ, and delta = rowsum(dO * O)
//   dkv kernel: wave owns 32 keys,    sweeps queries -> dK, dV
// P is recomputed from q, k and the saved log-sum-exp.
#include <type_traits>

#include "nrv_attn_common.hpp"

// nrv_attn_gen.hip: streaming kernels for the shapes the single-pass kernels below do not hold on chip (N > 256, dh != 64)
NRV_INTERNAL int nrv_attn_gen_supported(int B, int N, int H, int dh);
NRV_INTERNAL int nrv_attn_gen_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int dh, float scale, hipStream_t s);
NRV_INTERNAL int nrv_attn_gen_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta_ws,
                     int B, int N, int H, int dh, float scale, hipStream_t s);
NRV_INTERNAL int nrv_attn_gen_probs(const void* qkv, const float* lse, float* probs, int B, int N, int H, int dh, float scale, hipStream_t s);

namespace {

using namespace nrv_attn;
struct AttnParams {
    const bf16_t* qkv;     // q / k / v of (b, h), token t, feature d at  which * q_ws + h * q_hs + b * q_bs + t * q_rs + d  (elements)
    const bf16_t* out;     // (bwd) o / dO of (b, h), token t at  h * o_hs + b * o_bs + t * o_rs + d
    const bf16_t* dout;    // (bwd)
    bf16_t* o;             // fwd output, the o layout
    bf16_t* dqkv;          // bwd output, the qkv layout
    float* lse;            // [B, H, N]
    float* delta;          // [B, H, N]
    int B, N, H;
    float scale;
    // NRV_LAYOUT_ROWMAJOR: qkv [B*N, 3*H*64] (q_rs = 3 H 64, q_ws = H 64, q_hs = 64, q_bs = N q_rs), o [B*N, H*64]
    // NRV_LAYOUT_BLOCKED : qkv [3*H][B*N][64]  (q_rs = 64, q_ws = H B N 64, q_hs = B N 64, q_bs = N 64), o [H][B*N][64]:
    //                      every head slice [N x 64] is ONE contiguous 128 N bytes instead of N segments of 128 bytes
    long long q_rs, q_ws, q_hs, q_bs, o_rs, o_hs, o_bs;
};

void set_layout(AttnParams& p, int layout, int dh) {
    const long long T = (long long)p.B * p.N;
    if (layout & NRV_ATTN_QKV_BLOCKED) { p.q_rs = dh; p.q_ws = (long long)p.H * T * dh; p.q_hs = T * dh; p.q_bs = (long long)p.N * dh; }
    else { p.q_rs = 3ll * p.H * dh; p.q_ws = (long long)p.H * dh; p.q_hs = dh; p.q_bs = p.N * p.q_rs; }
    if (layout & NRV_ATTN_OUT_BLOCKED) { p.o_rs = dh; p.o_hs = T * dh; p.o_bs = (long long)p.N * dh; }
    else { p.o_rs = (long long)p.H * dh; p.o_hs = dh; p.o_bs = p.N * p.o_rs; }
}

// ---------------------------------------------------------------------------------------------
// forward, 4 fat waves: one workgroup of 4 waves (one per SIMD, <= 256 VGPRs) per (batch, head), TWO workgroups per CU.
//   * K and V images arrive by 16-byte LDS-DMA (swizzles applied to the per-lane SOURCE address, the DMA destination is
//     lane-linear; rows >= N read as zero through the buffer descriptor).  While one workgroup of a CU waits for its
//     images the other one computes: the per-head HBM latency is hidden by the co-resident workgroup, not by a
//     software pipeline.
//   * A wave owns TWO 16-query tiles (32 queries) at a time: every K / V^T fragment read from LDS feeds both tiles (half
//     the LDS traffic) and the in-order wave has two independent MFMA -> softmax -> MFMA chains.
//   * Single pass: all N/16 score tiles of both query tiles stay in registers, K.Q^T is computed once; K fragments run
//     3 key tiles ahead of their MFMAs, V^T fragments one 32-key step ahead (s_memtime stamps: with one tile ahead the
//     S phase is 2x slower; with lgkmcnt(0) after every read, as in the thin-wave kernel above, the whole head is).
// Measured on MI355X (ViT-B/16, B = 256, N = 197): see DESIGN.md.
// ---------------------------------------------------------------------------------------------
// Waves per workgroup: 4, except for heads of at most 64 / 32 tokens (MAE's 50-token encoder): a wave owns 32 rows, so a head of
// N <= 64 keeps only 2 (N <= 32: 1) of 4 waves busy while the idle ones still take registers -- at 3 waves per SIMD that halves the
// heads a CU has in flight, and these kernels are latency-bound per head.  Such heads get workgroups of 2 (1) waves.
template <int NT>
constexpr int atf_waves() { return NT <= 2 ? 1 : NT <= 4 ? 2 : 4; }

// NT = number of 16-key tiles = image rows / 16.  N = 196 / 197 => NT = 13: 2 x 13 x 2 KiB = 53,248 B of LDS, THREE
// workgroups per CU (with rows padded to 32 it would be 57,344 B and two).
template <int NT>
__global__ __launch_bounds__(64 * atf_waves<NT>(), NT <= 13 ? 3 : 2) void attn_fwd_fat_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = NT * 16;                // image rows
    constexpr int IMG = NP * 128;              // one image
    constexpr int NS = (NT + 1) / 2;           // 32-key steps of P.V (the last one is half empty when NT is odd)
    constexpr int NJ = 2 * NP / 8;             // DMA instructions (1 KiB = 8 rows each): K rows then V rows
    constexpr int ATF_WAVES = atf_waves<NT>();
    constexpr int JPW = NJ / ATF_WAVES;        // = NT with four waves
    constexpr int MT = (NT == 13 || NT == 14) ? 1 : 2;   // trailing key tiles that can hold padding rows (launch_fwd_fat rounds other counts up to even)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, H = p.H;
    const long long ldq = p.q_rs;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const bf16_t* hb = p.qkv + (long long)b * p.q_bs + h * p.q_hs;
    const int g = lane >> 4, qc = lane & 15;
    const float sc = p.scale * LOG2E;
    const char* kimg = smem;
    const char* vimg = smem + IMG;

    // first pair's query fragments, then the image DMAs (vmcnt is in order: the fragments complete first)
    const int npairs = (N + 31) >> 5;
    bf16x8_t qf[2][2];
    auto load_q = [&](int pair) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = pair * 32 + t * 16 + qc;
            const int qr = q < N ? q : N - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[t][ks] = load_frag_global(hb + (long long)qr * ldq + ks * 32 + g * 8);
        }
    };
    if (wave < npairs) load_q(wave);
    {
        const __amdgpu_buffer_rsrc_t rk = make_rsrc(hb + p.q_ws, 0x7fffffffull), rv = make_rsrc(hb + 2 * p.q_ws, 0x7fffffffull);
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // scalar: the descriptor select below must stay in SGPRs
#pragma unroll
        for (int i = 0; i < JPW; ++i) {
            const int j = wave_u * JPW + i;
            const bool isv = j >= NP / 8;
            const int r = 8 * (isv ? j - NP / 8 : j) + (lane >> 3);
            const int pos = lane & 7;
            const int c = isv ? ((((pos >> 1) ^ ((r >> 1) & 3)) << 1) | (pos & 1)) : (pos ^ ((r >> 1) & 7));
            const unsigned vo = (r < N) ? (unsigned)(r * ldq * 2 + c * 16) : NRV_OOB;
            if (isv) dma16_nt(rv, smem + j * 1024, vo);
            else dma16_nt(rk, smem + j * 1024, vo);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): through the builtin so that hipcc's own counting stays exact
    asm volatile("" ::: "memory");
    __syncthreads();

    const bf16x4_t zero4 = {0, 0, 0, 0};
    // V^T fragment of 32-key step kk, feature tile dt; the upper 16 keys of the last step do not exist when NT is odd
    auto v_frag = [&](int kk, int dt) {
        const int q = (lane & 15) >> 2, pp = lane & 3;
        const int r0 = kk * 32 + 4 * g + q;
        const bf16x4_t lo = lds_read_tr16_b64(vimg + r0 * 128 + ((dt ^ ((r0 >> 1) & 3)) << 5) + pp * 8);
        if (2 * kk + 1 >= NT) return cat4(lo, zero4);
        return cat4(lo, lds_read_tr16_b64(vimg + (r0 + 16) * 128 + ((dt ^ (((r0 + 16) >> 1) & 3)) << 5) + pp * 8));
    };
    const int d_lane = (g & 1) ? 16 + 4 * (g - 1) : 4 * g;       // first feature of the lane's 8 within a 32-feature pair
    for (int pair = wave; pair < npairs; pair += ATF_WAVES) {
        // ---- S^T = K Q^T for every key tile, both query tiles
        f32x4_t st[2][NT];
        constexpr int KD = NT < 4 ? NT : 4;                        // fragment ring: 3 key tiles (12 MFMAs) ahead of their use
        bf16x8_t kf[KD][2];
#pragma unroll
        for (int j = 0; j < KD - 1; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[j][ks] = row_frag_img(kimg, j * 16, ks, lane);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kt + KD - 1 < NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) kf[(kt + KD - 1) % KD][ks] = row_frag_img(kimg, (kt + KD - 1) * 16, ks, lane);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a = mfma16(kf[kt % KD][ks], qf[t][ks], a);
                st[t][kt] = a;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next pair's query fragments and the first V^T fragments are in flight during the softmax arithmetic
        const int q0 = pair * 32 + qc;
        if (pair + ATF_WAVES < npairs) load_q(pair + ATF_WAVES);
        bf16x8_t vf[2][4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vf[0][dt] = v_frag(0, dt);
        // ---- softmax over the keys (a lane holds 4 keys of each tile for one query; 4 lanes share a query)
        float m[2], l[2];
        bf16x8_t pf[2][NS];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float mm = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (kt >= NT - MT) mm = fmaxf(mm, (kt * 16 + 4 * g + e < N) ? st[t][kt][e] : -INFINITY);
                    else mm = fmaxf(mm, st[t][kt][e]);
                }
            }
            mm = fmaxf(mm, __shfl_xor(mm, 16, 64));
            mm = fmaxf(mm, __shfl_xor(mm, 32, 64));
            mm *= sc;                                               // sc > 0: max commutes with the scaling
            float ll = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pv = __builtin_amdgcn_exp2f(fmaf(st[t][kt][e], sc, -mm));
                    if (kt >= NT - MT) pv = (kt * 16 + 4 * g + e < N) ? pv : 0.f;
                    st[t][kt][e] = pv;
                    ll += pv;
                }
                if (kt & 1) pf[t][kt >> 1] = pack_frag(st[t][kt - 1], st[t][kt]);
                else if (kt == NT - 1) pf[t][kt >> 1] = pack_frag(st[t][kt], f32x4_t{0.f, 0.f, 0.f, 0.f});
            }
            ll += __shfl_xor(ll, 16, 64);
            ll += __shfl_xor(ll, 32, 64);
            m[t] = mm;
            l[t] = ll;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- O^T = V^T P^T, V^T fragments one step ahead
        f32x4_t o[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NS; ++kk) {
            if (kk + 1 < NS) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) vf[(kk + 1) & 1][dt] = v_frag(kk + 1, dt);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int t = 0; t < 2; ++t) o[t][dt] = mfma16(vf[kk & 1][dt], pf[t][kk], o[t][dt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // normalise, pack, and regroup through v_permlane16_swap: lane (q, g) holds features 16 dt + 4 g + {0..3}; after the
        // swap of the (dt = 2 pr, 2 pr + 1) pair an even-g lane holds 16 (2 pr) + 4 g + {0..7}, an odd-g lane
        // 16 (2 pr + 1) + 4 (g - 1) + {0..7}: 16-byte stores (the store tail is issue-bound, guide T21)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = q0 + t * 16;
            const float inv = 1.0f / l[t];
            u32x4_t ov[2];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const f32x4_t oa = o[t][2 * pr] * inv, ob = o[t][2 * pr + 1] * inv;
                const auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(oa[0], oa[1]), pack_bf16x2(ob[0], ob[1]), false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(oa[2], oa[3]), pack_bf16x2(ob[2], ob[3]), false, false);
                ov[pr] = u32x4_t{lo[0], hi[0], lo[1], hi[1]};
            }
            if (q < N) {
                bf16_t* dst = p.o + (long long)b * p.o_bs + h * p.o_hs + (long long)q * p.o_rs + d_lane;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) *reinterpret_cast<u32x4_t*>(dst + 32 * pr) = ov[pr];
                if (g == 0) p.lse[((long long)b * H + h) * N + q] = (m[t] + __builtin_amdgcn_logf(l[t])) * LN2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, fat waves (same recipe as attn_fwd_fat_kernel: 4 waves of <= 256 VGPRs per (batch, head), two workgroups per
// CU, images by LDS-DMA, a wave owns TWO 16-row tiles so that every fragment read from LDS feeds both).
//   dq kernel : images K, V; wave owns 32 queries, sweeps the keys 32 at a time  -> dQ, delta = rowsum(dO * O)
//   dkv kernel: images Q, dO (+ lse, delta); wave owns 32 keys, sweeps the queries -> dK, dV
// P is recomputed from the saved log-sum-exp; dS is scaled once at the end (dQ, dK are linear in it).
// ---------------------------------------------------------------------------------------------
// Backward images are read BOTH as rows (ds_read_b128, lane = row) and transposed (ds_read_b64_tr_b16): they use the 32-byte
// PAIR swizzle of the forward V image (vimg_off), which is conflict-free for both patterns.  With the row-only swizzle
// (img_off) the transposed reads were 2-way conflicted: SQ_LDS_BANK_CONFLICT = 24-29 % of SQ_LDS_IDX_ACTIVE in the PMC pass.
__device__ __forceinline__ bf16x8_t row_frag_bwd(const char* img, int rb, int ks, int lane) {
    return lds_read_b128(img + vimg_off(rb + (lane & 15), ks * 4 + (lane >> 4)));
}

// a 16 x 64 fp32 tile held as acc[dt][e] (lane (r = lane & 15, g = lane >> 4): row r, features 16 dt + 4 g + e) -> bf16,
// regrouped with v_permlane16_swap so that a lane owns 8 consecutive features: two 16-byte stores per row
__device__ __forceinline__ void store_tile_bf16(bf16_t* row_ptr /* row of this lane, feature 0 */, const f32x4_t (&acc)[4],
                                                float mul, int g, bool ok) {
    const int d_lane = (g & 1) ? 16 + 4 * (g - 1) : 4 * g;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const f32x4_t a = acc[2 * pr] * mul, b = acc[2 * pr + 1] * mul;
        const auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a[0], a[1]), pack_bf16x2(b[0], b[1]), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a[2], a[3]), pack_bf16x2(b[2], b[3]), false, false);
        if (ok) *reinterpret_cast<u32x4_t*>(row_ptr + 32 * pr + d_lane) = u32x4_t{lo[0], hi[0], lo[1], hi[1]};
    }
}

// DMA of two [N x 64] head slices into two pair-swizzled row images of NT * 16 rows (rows >= N: zero)
template <int NT>
__device__ __forceinline__ void dma_two_images(char* smem, const bf16_t* src0, long long ld0, const bf16_t* src1, long long ld1,
                                               int N, int wave, int lane) {
    constexpr int NP = NT * 16;
    constexpr int PW = 4 * NT / atf_waves<NT>();    // 2 * NP / 8 = 4 NT instructions over the workgroup's waves
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(src0, 0x7fffffffull), r1 = make_rsrc(src1, 0x7fffffffull);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // scalar: the descriptor select below must stay in SGPRs
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int j = wave_u * PW + i;
        const bool second = j >= NP / 8;
        const int r = 8 * (second ? j - NP / 8 : j) + (lane >> 3);
        const int pos = lane & 7;
        const int c = (((pos >> 1) ^ ((r >> 1) & 3)) << 1) | (pos & 1);      // source chunk that lands at position pos (pair swizzle)
        const unsigned vo = (r < N) ? (unsigned)(r * (second ? ld1 : ld0) * 2 + c * 16) : NRV_OOB;
        if (second) dma16(r1, smem + j * 1024, vo);
        else dma16(r0, smem + j * 1024, vo);
    }
}

template <int NT>
__global__ __launch_bounds__(64 * atf_waves<NT>(), NT <= 13 ? 3 : 2) void attn_bwd_dq_fat_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ATF_WAVES = atf_waves<NT>();
    constexpr int NP = NT * 16;
    constexpr int NS = (NT + 1) / 2;
    constexpr int MT = (NT == 13 || NT == 14) ? 1 : 2;
    const char* kimg = smem;
    const char* vimg = smem + NP * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, H = p.H;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const long long ldq = p.q_rs, ldo = p.o_rs;
    const bf16_t* qbase = p.qkv + (long long)b * p.q_bs + h * p.q_hs;
    const bf16_t* obase = p.out + (long long)b * p.o_bs + h * p.o_hs;
    const bf16_t* dobase = p.dout + (long long)b * p.o_bs + h * p.o_hs;
    const int g = lane >> 4, qc = lane & 15;
    const float sc = p.scale * LOG2E;
    const int npairs = (N + 31) >> 5;

    bf16x8_t qf[2][2], dof[2][2];
    float dl[2], lse2[2];
    auto load_rows = [&](int pair) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = pair * 32 + t * 16 + qc;
            const int qr = q < N ? q : N - 1;
            float d = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[t][ks] = load_frag_global(qbase + (long long)qr * ldq + ks * 32 + g * 8);
                dof[t][ks] = load_frag_global(dobase + (long long)qr * ldo + ks * 32 + g * 8);
                const bf16x8_t of = load_frag_global(obase + (long long)qr * ldo + ks * 32 + g * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) d += bf16_to_f32((unsigned short)dof[t][ks][e]) * bf16_to_f32((unsigned short)of[e]);
            }
            d += __shfl_xor(d, 16, 64);
            d += __shfl_xor(d, 32, 64);
            dl[t] = d;
            const long long sidx = ((long long)b * H + h) * N + qr;
            lse2[t] = p.lse[sidx] * LOG2E;
            if (g == 0 && q < N) p.delta[sidx] = d;
        }
    };
    if (wave < npairs) load_rows(wave);
    dma_two_images<NT>(smem, qbase + p.q_ws, ldq, qbase + 2 * p.q_ws, ldq, N, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __syncthreads();

    const bf16x4_t zero4 = {0, 0, 0, 0};
    for (int pair = wave; pair < npairs; pair += ATF_WAVES) {
        f32x4_t dq[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        auto step = [&](const int kk, auto last_c) {
            constexpr bool LAST = decltype(last_c)::value;      // the last step: padding rows, and no upper half when NT is odd
            f32x4_t ds[2][2];                                   // [tile][half]
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int kt = 2 * kk + hf;
                if (!LAST || hf == 0 || (NT % 2 == 0)) {
                    bf16x8_t kr[2], vr[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        kr[ks] = row_frag_bwd(kimg, kt * 16, ks, lane);
                        vr[ks] = row_frag_bwd(vimg, kt * 16, ks, lane);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        f32x4_t st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            st = mfma16(kr[ks], qf[t][ks], st);
                            dp = mfma16(vr[ks], dof[t][ks], dp);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float pv = __builtin_amdgcn_exp2f(fmaf(st[e], sc, -lse2[t]));
                            if (LAST) pv = (kt * 16 + 4 * g + e < N) ? pv : 0.f;
                            ds[t][hf][e] = pv * (dp[e] - dl[t]);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) ds[t][hf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
            }
            bf16x8_t dsf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) dsf[t] = pack_frag(ds[t][0], ds[t][1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                // K^T fragment: keys kk*32 .. +31 (the upper 16 do not exist in the last step when NT is odd), features 16 dt ..
                const int q4 = (lane & 15) >> 2, pp = lane & 3;
                const int r0 = kk * 32 + 4 * g + q4;
                const int c = 2 * dt + (pp >> 1);
                const bf16x4_t lo = lds_read_tr16_b64(kimg + vimg_off(r0, c) + (pp & 1) * 8);
                const bf16x4_t hi = (!LAST || NT % 2 == 0) ? lds_read_tr16_b64(kimg + vimg_off(r0 + 16, c) + (pp & 1) * 8) : zero4;
                const bf16x8_t ktr = cat4(lo, hi);
#pragma unroll
                for (int t = 0; t < 2; ++t) dq[t][dt] = mfma16(ktr, dsf[t], dq[t][dt]);
            }
        };
        // padding rows (key >= N) live in the last MT key tiles = the last step (MT == 2 only when NT is even)
#pragma unroll 1
        for (int kk = 0; kk < NS - 1; ++kk) step(kk, std::false_type{});
        step(NS - 1, std::true_type{});
        // next pair's rows while this pair's results are packed and stored
        const int q0 = pair * 32 + qc;
        f32x4_t keep[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) keep[t][dt] = dq[t][dt];
        if (pair + ATF_WAVES < npairs) load_rows(pair + ATF_WAVES);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = q0 + t * 16;
            store_tile_bf16(p.dqkv + (long long)b * p.q_bs + h * p.q_hs + (long long)(q < N ? q : 0) * ldq, keep[t], p.scale, g, q < N);
        }
    }
}

template <int NT>
__global__ __launch_bounds__(64 * atf_waves<NT>(), 2) void attn_bwd_dkv_fat_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ATF_WAVES = atf_waves<NT>(), ATF_THREADS = 64 * ATF_WAVES;
    constexpr int NP = NT * 16;
    constexpr int NS = (NT + 1) / 2;
    constexpr int NPS = NS * 32;                       // lse / delta arrays cover whole 32-query steps
    const char* qimg = smem;
    const char* doimg = smem + NP * 128;
    float* lse2s = reinterpret_cast<float*>(smem + 2 * NP * 128);
    float* dels = lse2s + NPS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, H = p.H;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const long long ldq = p.q_rs, ldo = p.o_rs;
    const bf16_t* qbase = p.qkv + (long long)b * p.q_bs + h * p.q_hs;
    const bf16_t* dobase = p.dout + (long long)b * p.o_bs + h * p.o_hs;
    const int g = lane >> 4, kc = lane & 15;
    const float sc = p.scale * LOG2E;
    const int npairs = (N + 31) >> 5;

    bf16x8_t kf[2][2], vf[2][2];
    auto load_rows = [&](int pair) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = pair * 32 + t * 16 + kc;
            const int kr = key < N ? key : N - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[t][ks] = load_frag_global(qbase + p.q_ws + (long long)kr * ldq + ks * 32 + g * 8);
                vf[t][ks] = load_frag_global(qbase + 2 * p.q_ws + (long long)kr * ldq + ks * 32 + g * 8);
            }
        }
    };
    if (wave < npairs) load_rows(wave);
    for (int i = tid; i < NPS; i += ATF_THREADS) {
        const long long sidx = ((long long)b * H + h) * N + i;
        lse2s[i] = i < N ? p.lse[sidx] * LOG2E : INFINITY;     // exp2(s - inf) = 0 for padded queries
        dels[i] = i < N ? p.delta[sidx] : 0.f;
    }
    dma_two_images<NT>(smem, qbase, ldq, dobase, ldo, N, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __syncthreads();

    const bf16x4_t zero4 = {0, 0, 0, 0};
    auto tr_frag = [&](const char* img, int qq, int dt, auto last_c) {     // rows qq*32 .. +31 transposed, features 16 dt ..
        const int q4 = (lane & 15) >> 2, pp = lane & 3;
        const int r0 = qq * 32 + 4 * g + q4;
        const int c = 2 * dt + (pp >> 1);
        const bf16x4_t lo = lds_read_tr16_b64(img + vimg_off(r0, c) + (pp & 1) * 8);
        const bf16x4_t hi = (!decltype(last_c)::value || NT % 2 == 0) ? lds_read_tr16_b64(img + vimg_off(r0 + 16, c) + (pp & 1) * 8) : zero4;
        return cat4(lo, hi);
    };
    for (int pair = wave; pair < npairs; pair += ATF_WAVES) {
        f32x4_t dk[2][4], dv[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dk[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                dv[t][dt] = dk[t][dt];
            }
        auto step = [&](const int qq, auto last_c) {
            constexpr bool LAST = decltype(last_c)::value;      // the last step has no upper half when NT is odd
            f32x4_t pt[2][2], ds[2][2];                         // [key tile][query half]
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int qt = 2 * qq + hf;
                if (!LAST || hf == 0 || (NT % 2 == 0)) {
                    bf16x8_t qr[2], dor[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        qr[ks] = row_frag_bwd(qimg, qt * 16, ks, lane);
                        dor[ks] = row_frag_bwd(doimg, qt * 16, ks, lane);
                    }
                    const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse2s + qt * 16 + 4 * g);
                    const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dels + qt * 16 + 4 * g);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        f32x4_t st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            st = mfma16(qr[ks], kf[t][ks], st);
                            dp = mfma16(dor[ks], vf[t][ks], dp);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float pv = __builtin_amdgcn_exp2f(fmaf(st[e], sc, -l4[e]));
                            pt[t][hf][e] = pv;
                            ds[t][hf][e] = pv * (dp[e] - d4[e]);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        pt[t][hf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                        ds[t][hf] = pt[t][hf];
                    }
                }
            }
            bf16x8_t pf[2], dsf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                pf[t] = pack_frag(pt[t][0], pt[t][1]);
                dsf[t] = pack_frag(ds[t][0], ds[t][1]);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8_t dotr = tr_frag(doimg, qq, dt, last_c);
                const bf16x8_t qtr = tr_frag(qimg, qq, dt, last_c);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    dv[t][dt] = mfma16(dotr, pf[t], dv[t][dt]);
                    dk[t][dt] = mfma16(qtr, dsf[t], dk[t][dt]);
                }
            }
        };
#pragma unroll 1
        for (int qq = 0; qq < NS - 1; ++qq) step(qq, std::false_type{});
        step(NS - 1, std::true_type{});
        const int k0 = pair * 32 + kc;
        if (pair + ATF_WAVES < npairs) load_rows(pair + ATF_WAVES);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = k0 + t * 16;
            bf16_t* row = p.dqkv + (long long)b * p.q_bs + h * p.q_hs + (long long)(key < N ? key : 0) * ldq;
            store_tile_bf16(row + p.q_ws, dk[t], p.scale, g, key < N);
            store_tile_bf16(row + 2 * p.q_ws, dv[t], 1.0f, g, key < N);
        }
    }
}

template <int NT>
int launch_fwd_fat_nt(const AttnParams& p, hipStream_t s) {
    constexpr int lds = 2 * NT * 16 * 128;
    static int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_fat_kernel<NT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != 0) return attr;
    hipLaunchKernelGGL((attn_fwd_fat_kernel<NT>), dim3(p.B * p.H), dim3(64 * atf_waves<NT>()), lds, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}

// key tiles: the instantiated counts cover every N <= 256 with at most one tile of padding, and 13 exactly (N = 196 / 197)
int launch_fwd_fat(const AttnParams& p, hipStream_t s) {
    const int nt = (p.N + 15) / 16;
    if (nt <= 2) return launch_fwd_fat_nt<2>(p, s);
    if (nt <= 4) return launch_fwd_fat_nt<4>(p, s);
    if (nt <= 6) return launch_fwd_fat_nt<6>(p, s);
    if (nt <= 8) return launch_fwd_fat_nt<8>(p, s);
    if (nt <= 10) return launch_fwd_fat_nt<10>(p, s);
    if (nt <= 12) return launch_fwd_fat_nt<12>(p, s);
    if (nt == 13) return launch_fwd_fat_nt<13>(p, s);
    if (nt == 14) return launch_fwd_fat_nt<14>(p, s);
    return launch_fwd_fat_nt<16>(p, s);
}

template <int NT>
int launch_bwd_fat_nt(const AttnParams& p, hipStream_t s) {
    constexpr int lds_dq = 2 * NT * 16 * 128;
    constexpr int lds_dkv = 2 * NT * 16 * 128 + 2 * ((NT + 1) / 2) * 32 * 4;
    static int a1 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_fat_kernel<NT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds_dq);
    static int a2 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_fat_kernel<NT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds_dkv);
    if (a1 != 0) return a1;
    if (a2 != 0) return a2;
    hipLaunchKernelGGL((attn_bwd_dq_fat_kernel<NT>), dim3(p.B * p.H), dim3(64 * atf_waves<NT>()), lds_dq, s, p);
    NRV_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn_bwd_dkv_fat_kernel<NT>), dim3(p.B * p.H), dim3(64 * atf_waves<NT>()), lds_dkv, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}

int launch_bwd_fat(const AttnParams& p, hipStream_t s) {
    const int nt = (p.N + 15) / 16;
    if (nt <= 2) return launch_bwd_fat_nt<2>(p, s);
    if (nt <= 4) return launch_bwd_fat_nt<4>(p, s);
    if (nt <= 6) return launch_bwd_fat_nt<6>(p, s);
    if (nt <= 8) return launch_bwd_fat_nt<8>(p, s);
    if (nt <= 10) return launch_bwd_fat_nt<10>(p, s);
    if (nt <= 12) return launch_bwd_fat_nt<12>(p, s);
    if (nt == 13) return launch_bwd_fat_nt<13>(p, s);
    if (nt == 14) return launch_bwd_fat_nt<14>(p, s);
    return launch_bwd_fat_nt<16>(p, s);
}

// ---------------------------------------------------------------------------------------------
// introspection (Recorder-style attention maps, recorder.py:24-31): P[b,h,q,k] = exp(scale q.k - lse[b,h,q]) written out
// in fp32.  NOT on the training path (which never materialises P): a plain VALU kernel, 16 queries per workgroup, the 16
// query rows in LDS as fp32, one (query, key) dot product of 64 per thread and step.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_probs_kernel(const AttnParams p, float* __restrict__ probs) {
    __shared__ float qs[16][DH + 1];
    const bf16_t* __restrict__ qkv = p.qkv;
    const float* __restrict__ lse = p.lse;
    const int N = p.N, H = p.H;
    const float scale = p.scale;
    const int bh = blockIdx.x, q0 = blockIdx.y * 16;
    const int b = bh / H, h = bh - b * H;
    const long long ldq = p.q_rs;
    const bf16_t* base = qkv + (long long)b * p.q_bs + h * p.q_hs;
    for (int i = threadIdx.x; i < 16 * DH; i += 256) {
        const int r = i / DH, d = i - r * DH;
        qs[r][d] = (q0 + r < N) ? bf16_to_f32(base[(long long)(q0 + r) * ldq + d]) : 0.f;
    }
    __syncthreads();
    const int qi = threadIdx.x & 15;
    const int q = q0 + qi;
    const float l = q < N ? lse[((long long)b * H + h) * N + q] : 0.f;
    for (int key = threadIdx.x >> 4; key < N; key += 16) {
        const bf16_t* kp = base + p.q_ws + (long long)key * ldq;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            const bf16x8_t kv = load_frag_global(kp + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qs[qi][c * 8 + e], bf16_to_f32((unsigned short)kv[e]), acc);
        }
        if (q < N) probs[(((long long)b * H + h) * N + q) * N + key] = __expf(acc * scale - l);
    }
}

// shape class of a call: 1 = the single-pass kernels of this file (dh 64, N <= 256), 2 = the streaming kernels of
// nrv_attn_gen.hip (any N, dh 32 / 64 / 80 / 96 / 128), 0 = not supported
int shape_class(int B, int N, int H, int dh) {
    if (B <= 0 || N <= 0 || H <= 0) return 0;
    if (dh == DH && N <= 256) return (long long)B * H > 0x7fffffffll ? 0 : 1;
    return nrv_attn_gen_supported(B, N, H, dh) ? 2 : 0;
}

}  // namespace

// layouts the single-pass kernels address through strides; the streaming kernels take the row-major form only
static bool layout_ok(int layout, int cls, int B, int N, int H, int dh) {
    if (layout & ~(NRV_ATTN_QKV_BLOCKED | NRV_ATTN_OUT_BLOCKED)) return false;
    if (layout != 0 && cls != 1) return false;
    // byte offsets of the k / v slices from the head's base stay below the 2 GiB window of a buffer descriptor
    return 2ll * 3 * H * (long long)B * N * dh < 0x7fffffffll;
}

extern "C" int nrv_attn_fwd(const void* qkv_bf16, void* out_bf16, float* lse,
                            int B, int N, int H, int dh, float scale, int layout, void* stream) {
    if (!qkv_bf16 || !out_bf16 || !lse) return NRV_ERR_NULL;
    const int cls = shape_class(B, N, H, dh);
    if (cls == 0 || !layout_ok(layout, cls, B, N, H, dh)) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(qkv_bf16) || !nrv_aligned16(out_bf16)) return NRV_ERR_ALIGN;
    if (cls == 2) return nrv_attn_gen_fwd(qkv_bf16, out_bf16, lse, B, N, H, dh, scale, static_cast<hipStream_t>(stream));
    AttnParams p{};
    p.qkv = static_cast<const bf16_t*>(qkv_bf16);
    p.o = static_cast<bf16_t*>(out_bf16);
    p.lse = lse;
    p.B = B; p.N = N; p.H = H; p.scale = scale;
    set_layout(p, layout, dh);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return launch_fwd_fat(p, s);
}

extern "C" int nrv_attn_bwd(const void* qkv_bf16, const void* out_bf16, const void* dout_bf16, const float* lse,
                            void* dqkv_bf16, float* delta_ws,
                            int B, int N, int H, int dh, float scale, int layout, void* stream) {
    if (!qkv_bf16 || !out_bf16 || !dout_bf16 || !lse || !dqkv_bf16 || !delta_ws) return NRV_ERR_NULL;
    const int cls = shape_class(B, N, H, dh);
    if (cls == 0 || !layout_ok(layout, cls, B, N, H, dh)) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(qkv_bf16) || !nrv_aligned16(out_bf16) || !nrv_aligned16(dout_bf16) || !nrv_aligned16(dqkv_bf16))
        return NRV_ERR_ALIGN;
    if (cls == 2)
        return nrv_attn_gen_bwd(qkv_bf16, out_bf16, dout_bf16, lse, dqkv_bf16, delta_ws, B, N, H, dh, scale, static_cast<hipStream_t>(stream));
    AttnParams p{};
    p.qkv = static_cast<const bf16_t*>(qkv_bf16);
    p.out = static_cast<const bf16_t*>(out_bf16);
    p.dout = static_cast<const bf16_t*>(dout_bf16);
    p.dqkv = static_cast<bf16_t*>(dqkv_bf16);
    p.lse = const_cast<float*>(lse);
    p.delta = delta_ws;
    p.B = B; p.N = N; p.H = H; p.scale = scale;
    set_layout(p, layout, dh);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return launch_bwd_fat(p, s);
}

extern "C" int nrv_attn_probs(const void* qkv_bf16, const float* lse, float* probs,
                              int B, int N, int H, int dh, float scale, int layout, void* stream) {
    if (!qkv_bf16 || !lse || !probs) return NRV_ERR_NULL;
    const int cls = shape_class(B, N, H, dh);
    if (cls == 0 || !layout_ok(layout, cls, B, N, H, dh)) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(qkv_bf16)) return NRV_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (cls == 2) return nrv_attn_gen_probs(qkv_bf16, lse, probs, B, N, H, dh, scale, s);
    AttnParams p{};
    p.qkv = static_cast<const bf16_t*>(qkv_bf16);
    p.lse = const_cast<float*>(lse);
    p.B = B; p.N = N; p.H = H; p.scale = scale;
    set_layout(p, layout, dh);
    hipLaunchKernelGGL(attn_probs_kernel, dim3((unsigned)(B * H), (unsigned)((N + 15) / 16)), dim3(256), 0, s, p, probs);
    NRV_CHECK_LAUNCH();
    return 0;
}
