// Fused multi-head self-attention, forward and backward, for gfx950 (MI355X).
//
// ViT heads are tiny (N <= 256 tokens, dh = 64), so one workgroup (4 waves) owns one (batch, head):
// the whole K and V of the head live in LDS (2 x N x 128 B), a wave owns 16 query rows at a time and
// holds the full score row block in registers -- single-pass softmax, no online rescale, and the
// [N,N] score matrix never leaves the CU (the reference materialises it three times:
// simple_vit.py:70-74).  q/k/v are read in place from the QKV projection's [B, N, 3*H*dh] output
// (128-byte head segments), and the output is written in 'b n (h d)' order, so the einops
// rearrange copies (simple_vit.py:68,75) disappear.
//
// MFMA orientation (16x16x32 bf16): scores are computed TRANSPOSED, S^T = K Q^T, so a lane owns one
// query column and its keys sit in registers: row max / row sum are register reductions plus two
// shuffles, and the bf16 P^T accumulator IS the B operand of O^T = V^T P^T (no lane movement).
// V^T (and K^T, Q^T, dO^T in the backward) fragments come from ds_read_b64_tr_b16.
//
// Backward = two kernels with the same structure and no atomics:
//   dq kernel : wave owns 16 queries, sweeps keys  -> dQ, and delta = rowsum(dO * O)
//   dkv kernel: wave owns 16 keys,   sweeps queries -> dK, dV
// P is recomputed from q, k and the saved log-sum-exp.
#include "nrv_attn_common.hpp"

namespace {

using namespace nrv_attn;
constexpr int ATT_THREADS = 512;      // 8 waves per (batch, head); 2 workgroups per CU -> 4 waves per SIMD
constexpr int ATT_WAVES = ATT_THREADS / 64;

struct AttnParams {
    const bf16_t* qkv;     // [B, N, 3*H*64]
    const bf16_t* out;     // [B, N, H*64]      (bwd)
    const bf16_t* dout;    // [B, N, H*64]      (bwd)
    bf16_t* o;             // fwd output
    bf16_t* dqkv;          // bwd output
    float* lse;            // [B, H, N]
    float* delta;          // [B, H, N]
    int B, N, H;
    float scale;
};

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(ATT_THREADS, 4) void attn_fwd_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kimg = smem;
    char* vimg = smem + NP * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int N = p.N;
    const long long ldq = 3ll * p.H * DH;
    const bf16_t* qbase = p.qkv + (long long)b * N * ldq + h * DH;
    load_image<NP, false, ATT_THREADS>(kimg, qbase + p.H * DH, ldq, N, tid);
    load_image<NP, true, ATT_THREADS>(vimg, qbase + 2 * p.H * DH, ldq, N, tid);

    const int g = lane >> 4, qc = lane & 15;
    const float sc = p.scale * LOG2E;
    const int nqt = (N + 15) >> 4;
    // the wave's first query fragments are fetched while the K/V images are still landing
    bf16x8_t qf[2];
    {
        const int q0 = wave * 16 + qc;
        const int qr0 = q0 < N ? q0 : N - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = load_frag_global(qbase + (long long)qr0 * ldq + ks * 32 + g * 8);
    }
    __syncthreads();

    // Two passes over the keys keep the register footprint small (<= 128 VGPRs => two workgroups = 16 waves per CU,
    // which is what hides the per-head load latency): pass 1 finds the row maximum, pass 2 recomputes the score
    // tiles, exponentiates and feeds P.V.  The extra K.Q^T MFMAs are free -- this kernel is latency-, not MFMA-bound.
    for (int qt = wave; qt < nqt; qt += ATT_WAVES) {
        const int q = qt * 16 + qc;
        float m = -INFINITY;
#pragma unroll 2
        for (int kt = 0; kt < NP / 16; ++kt) {
            f32x4_t st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) st = mfma16(row_frag_img(kimg, kt * 16, ks, lane), qf[ks], st);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = kt * 16 + 4 * g + e;
                m = fmaxf(m, key < N ? st[e] * sc : -INFINITY);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));

        float l = 0.f;
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kk = 0; kk < NP / 32; ++kk) {
            f32x4_t pt[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4_t st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) st = mfma16(row_frag_img(kimg, kk * 32 + hf * 16, ks, lane), qf[ks], st);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = kk * 32 + hf * 16 + 4 * g + e;
                    const float pv = key < N ? __builtin_amdgcn_exp2f(st[e] * sc - m) : 0.f;
                    pt[hf][e] = pv;
                    l += pv;
                }
            }
            const bf16x8_t pf = pack_frag(pt[0], pt[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(tr_frag_vimg(vimg, kk * 32, dt, lane), pf, o[dt]);
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        // next tile's query fragments (if any) before the stores of this one
        const int qn = (qt + ATT_WAVES) * 16 + qc;
        if (qt + ATT_WAVES < nqt) {
            const int qrn = qn < N ? qn : N - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[ks] = load_frag_global(qbase + (long long)qrn * ldq + ks * 32 + g * 8);
        }
        if (q < N) {
            const float inv = 1.0f / l;
            bf16_t* dst = p.o + ((long long)b * N + q) * (p.H * DH) + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store_bf16x4(dst + dt * 16, o[dt] * inv);
            if (g == 0) p.lse[((long long)b * p.H + h) * N + q] = (m + __builtin_amdgcn_logf(l)) * LN2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, query-owner pass: dQ and delta
// ---------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(ATT_THREADS, 4) void attn_bwd_dq_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kimg = smem;
    char* vimg = smem + NP * 128;      // GEMM-swizzled row image here (row reads only)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int N = p.N;
    const long long ldq = 3ll * p.H * DH, ldo = (long long)p.H * DH;
    const bf16_t* qbase = p.qkv + (long long)b * N * ldq + h * DH;
    const bf16_t* obase = p.out + (long long)b * N * ldo + h * DH;
    const bf16_t* dobase = p.dout + (long long)b * N * ldo + h * DH;
    load_image<NP, false, ATT_THREADS>(kimg, qbase + p.H * DH, ldq, N, tid);
    load_image<NP, false, ATT_THREADS>(vimg, qbase + 2 * p.H * DH, ldq, N, tid);
    __syncthreads();

    const int g = lane >> 4, qc = lane & 15;
    const float sc = p.scale * LOG2E;
    const int nqt = (N + 15) >> 4;
    for (int qt = wave; qt < nqt; qt += ATT_WAVES) {
        const int q = qt * 16 + qc;
        const int qr = q < N ? q : N - 1;
        bf16x8_t qf[2], dof[2];
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = load_frag_global(qbase + (long long)qr * ldq + ks * 32 + g * 8);
            dof[ks] = load_frag_global(dobase + (long long)qr * ldo + ks * 32 + g * 8);
            const bf16x8_t of = load_frag_global(obase + (long long)qr * ldo + ks * 32 + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                dl += bf16_to_f32((unsigned short)dof[ks][e]) * bf16_to_f32((unsigned short)of[e]);
        }
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        const long long sidx = ((long long)b * p.H + h) * N + qr;
        const float lse2 = p.lse[sidx] * LOG2E;
        if (g == 0 && q < N) p.delta[sidx] = dl;

        f32x4_t dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kk = 0; kk < NP / 32; ++kk) {
            f32x4_t ds[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int kb = kk * 32 + hf * 16;
                f32x4_t st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    st = mfma16(row_frag_img(kimg, kb, ks, lane), qf[ks], st);
                    dp = mfma16(row_frag_img(vimg, kb, ks, lane), dof[ks], dp);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = kb + 4 * g + e;
                    const float pv = key < N ? __builtin_amdgcn_exp2f(st[e] * sc - lse2) : 0.f;
                    ds[hf][e] = pv * (dp[e] - dl) * p.scale;
                }
            }
            const bf16x8_t dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16(tr_frag_img(kimg, kk * 32, dt, lane), dsf, dq[dt]);
        }
        if (q < N) {
            bf16_t* dst = p.dqkv + ((long long)b * N + q) * ldq + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store_bf16x4(dst + dt * 16, dq[dt]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, key-owner pass: dK and dV
// ---------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(ATT_THREADS, 4) void attn_bwd_dkv_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* qimg = smem;
    char* doimg = smem + NP * 128;
    float* lse2s = reinterpret_cast<float*>(smem + 2 * NP * 128);
    float* dels = lse2s + NP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int N = p.N;
    const long long ldq = 3ll * p.H * DH, ldo = (long long)p.H * DH;
    const bf16_t* qbase = p.qkv + (long long)b * N * ldq + h * DH;
    const bf16_t* dobase = p.dout + (long long)b * N * ldo + h * DH;
    load_image<NP, false, ATT_THREADS>(qimg, qbase, ldq, N, tid);
    load_image<NP, false, ATT_THREADS>(doimg, dobase, ldo, N, tid);
    for (int i = tid; i < NP; i += ATT_THREADS) {
        const long long sidx = ((long long)b * p.H + h) * N + i;
        lse2s[i] = i < N ? p.lse[sidx] * LOG2E : INFINITY;     // exp2(s - inf) = 0 for padded queries
        dels[i] = i < N ? p.delta[sidx] : 0.f;
    }
    __syncthreads();

    const int g = lane >> 4, kc = lane & 15;
    const float sc = p.scale * LOG2E;
    const int nkt = (N + 15) >> 4;
    for (int kt = wave; kt < nkt; kt += ATT_WAVES) {
        const int key = kt * 16 + kc;
        const int kr = key < N ? key : N - 1;
        bf16x8_t kf[2], vf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[ks] = load_frag_global(qbase + p.H * DH + (long long)kr * ldq + ks * 32 + g * 8);
            vf[ks] = load_frag_global(qbase + 2 * p.H * DH + (long long)kr * ldq + ks * 32 + g * 8);
        }
        f32x4_t dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dv[dt] = dk[dt];
        }
#pragma unroll 1
        for (int qq = 0; qq < NP / 32; ++qq) {
            f32x4_t pt[2], ds[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int qb = qq * 32 + hf * 16;
                f32x4_t st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    st = mfma16(row_frag_img(qimg, qb, ks, lane), kf[ks], st);
                    dp = mfma16(row_frag_img(doimg, qb, ks, lane), vf[ks], dp);
                }
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse2s + qb + 4 * g);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dels + qb + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(st[e] * sc - l4[e]);
                    pt[hf][e] = pv;
                    ds[hf][e] = pv * (dp[e] - d4[e]) * p.scale;
                }
            }
            const bf16x8_t pf = pack_frag(pt[0], pt[1]);
            const bf16x8_t dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = mfma16(tr_frag_img(doimg, qq * 32, dt, lane), pf, dv[dt]);
                dk[dt] = mfma16(tr_frag_img(qimg, qq * 32, dt, lane), dsf, dk[dt]);
            }
        }
        if (key < N) {
            bf16_t* dst = p.dqkv + ((long long)b * N + key) * ldq + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                store_bf16x4(dst + p.H * DH + dt * 16, dk[dt]);
                store_bf16x4(dst + 2 * p.H * DH + dt * 16, dv[dt]);
            }
        }
    }
}

template <int NP>
int launch_fwd(const AttnParams& p, hipStream_t s) {
    constexpr int lds = 2 * NP * 128;
    static int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<NP>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != 0) return attr;
    hipLaunchKernelGGL((attn_fwd_kernel<NP>), dim3(p.B * p.H), dim3(ATT_THREADS), lds, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}

template <int NP>
int launch_bwd(const AttnParams& p, hipStream_t s) {
    constexpr int lds_dq = 2 * NP * 128;
    constexpr int lds_dkv = 2 * NP * 128 + 2 * NP * 4;
    static int attr1 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<NP>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_dq);
    static int attr2 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<NP>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_dkv);
    if (attr1 != 0) return attr1;
    if (attr2 != 0) return attr2;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<NP>), dim3(p.B * p.H), dim3(ATT_THREADS), lds_dq, s, p);
    NRV_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<NP>), dim3(p.B * p.H), dim3(ATT_THREADS), lds_dkv, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}

int check_shape(int B, int N, int H, int dh) {
    if (B <= 0 || N <= 0 || H <= 0) return NRV_ERR_SHAPE;
    if (dh != DH || N > 256) return NRV_ERR_SHAPE;
    if ((long long)B * H > 0x7fffffffll) return NRV_ERR_SHAPE;
    return 0;
}

#define NRV_DISPATCH_NP(N, CALL)                                   \
    switch (((N) + 31) / 32) {                                     \
        case 1: { constexpr int NPV = 32; return CALL; }           \
        case 2: { constexpr int NPV = 64; return CALL; }           \
        case 3: { constexpr int NPV = 96; return CALL; }           \
        case 4: { constexpr int NPV = 128; return CALL; }          \
        case 5: { constexpr int NPV = 160; return CALL; }          \
        case 6: { constexpr int NPV = 192; return CALL; }          \
        case 7: { constexpr int NPV = 224; return CALL; }          \
        default: { constexpr int NPV = 256; return CALL; }         \
    }

}  // namespace

extern "C" int nrv_attn_fwd(const void* qkv_bf16, void* out_bf16, float* lse,
                            int B, int N, int H, int dh, float scale, void* stream) {
    if (!qkv_bf16 || !out_bf16 || !lse) return NRV_ERR_NULL;
    if (int e = check_shape(B, N, H, dh)) return e;
    if (!nrv_aligned16(qkv_bf16) || !nrv_aligned16(out_bf16)) return NRV_ERR_ALIGN;
    AttnParams p{};
    p.qkv = static_cast<const bf16_t*>(qkv_bf16);
    p.o = static_cast<bf16_t*>(out_bf16);
    p.lse = lse;
    p.B = B; p.N = N; p.H = H; p.scale = scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    NRV_DISPATCH_NP(N, launch_fwd<NPV>(p, s));
}

extern "C" int nrv_attn_bwd(const void* qkv_bf16, const void* out_bf16, const void* dout_bf16, const float* lse,
                            void* dqkv_bf16, float* delta_ws,
                            int B, int N, int H, int dh, float scale, void* stream) {
    if (!qkv_bf16 || !out_bf16 || !dout_bf16 || !lse || !dqkv_bf16 || !delta_ws) return NRV_ERR_NULL;
    if (int e = check_shape(B, N, H, dh)) return e;
    if (!nrv_aligned16(qkv_bf16) || !nrv_aligned16(out_bf16) || !nrv_aligned16(dout_bf16) || !nrv_aligned16(dqkv_bf16))
        return NRV_ERR_ALIGN;
    AttnParams p{};
    p.qkv = static_cast<const bf16_t*>(qkv_bf16);
    p.out = static_cast<const bf16_t*>(out_bf16);
    p.dout = static_cast<const bf16_t*>(dout_bf16);
    p.dqkv = static_cast<bf16_t*>(dqkv_bf16);
    p.lse = const_cast<float*>(lse);
    p.delta = delta_ws;
    p.B = B; p.N = N; p.H = H; p.scale = scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    NRV_DISPATCH_NP(N, launch_bwd<NPV>(p, s));
}
