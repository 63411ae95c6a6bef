// "robust" attention for gfx950: softmax followed by Sinkhorn row/column normalisation, fused on chip.
//
// Reference: SinkhornAttention.forward, /root/reference/vit_pytorch_robust/utils.py:1031-1037
//     P = softmax(S);  3 x { P /= rowsum(P);  P /= colsum(P) };  P /= rowsum(P)          (wired at simple_vit.py:56-57)
//
// Observation that makes the fusion possible: every normalisation only rescales rows or columns, so at all times
//     P = diag(a) . P0 . diag(b),   P0 = softmax(S)
// and one half-iteration is a matrix-vector product with P0:  a_i = 1 / sum_j P0_ij b_j,   b_j = 1 / sum_i a_i P0_ij.
// P0 of one head (N <= 256) stays in registers, distributed over the workgroup exactly as in the softmax kernel
// (16 waves x one 16-query tile; a lane owns one query column, its keys sit in registers):
//     row sums  = register reduction + 2 shuffles (wave-local, no barrier);
//     col sums  = DPP reduction over the 16 query lanes, per-wave partials in LDS, one barrier pair per iteration.
// The 7 scaling vectors are saved ([B,H,7,N] fp32) together with the softmax LSE; nothing [N,N]-sized is stored by
// the forward.
//
// Backward (query-owner kernel): recompute P0, G = dP7 = dO V^T, walk the 7 normalisations backwards
//     row step:  G_ij <- (G_ij - alpha_i sum_j G_ij P0_ij beta_j) * alpha_i / alpha_prev_i
//     col step:  G_ij <- (G_ij - beta_j sum_i G_ij alpha_i P0_ij) * beta_j / beta_prev_j
// then the softmax backward, dQ = dS K on MFMA, and dS / P7 are handed to the key-owner kernel as bf16 [key][q]
// scratch (the only [N,N] traffic of the robust path; the reference materialises ~10 fp32 copies).
// Key-owner kernel: dK = dS^T Q, dV = P7^T dO (pure MFMA, transposed LDS reads of Q / dO).
#include "nrv_attn_common.hpp"

namespace {

using namespace nrv_attn;

constexpr int SK_THREADS = 1024;      // 16 waves: one 16-query tile each (N <= 256)
constexpr int SK_WAVES = 16;
constexpr int SKB_THREADS = 512;      // key-owner backward kernel
#ifndef NRV_SK_TPW
#define NRV_SK_TPW 2
#endif

struct SinkParams {
    const bf16_t* qkv;     // [B, N, 3*H*64]
    const bf16_t* dout;    // [B, N, H*64]
    bf16_t* o;             // [B, N, H*64]
    bf16_t* dqkv;          // [B, N, 3*H*64]
    float* lse;            // [B, H, N]
    float* scal;           // [B, H, 7, N]   a1 b1 a2 b2 a3 b3 a4
    bf16_t* ws_ds;         // [B*H, NP, NP]  dS^T  [key][q]   (backward scratch)
    bf16_t* ws_p;          // [B*H, NP, NP]  P7^T  [key][q]
    int B, N, H;
    float scale;
};

// sum over the 16 lanes of a DPP row (lanes 16g .. 16g+15); every lane ends up with the total
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
}
__device__ __forceinline__ float inv_or_zero(float x) { return x > 0.f ? 1.0f / x : 0.f; }
__device__ __forceinline__ float ratio_or_zero(float num, float den) { return den > 0.f ? num / den : 0.f; }

// S^T tiles of one query tile -> P0 (normalised softmax, fp32); returns via p0[]; padded keys/queries are zero
template <int NP>
__device__ __forceinline__ void softmax_tile(const char* kimg, const bf16x8_t (&qf)[2], f32x4_t (&p0)[NP / 16],
                                             int N, bool q_ok, float sc, int lane, float& m_out, float& l_out) {
    const int g = lane >> 4;
#pragma unroll
    for (int kt = 0; kt < NP / 16; ++kt) {
        p0[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) p0[kt] = mfma16(row_frag_img(kimg, kt * 16, ks, lane), qf[ks], p0[kt]);
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NP / 16; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = kt * 16 + 4 * g + e;
            const float v = key < N ? p0[kt][e] * sc : -INFINITY;
            p0[kt][e] = v;
            m = fmaxf(m, v);
        }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NP / 16; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float pv = __builtin_amdgcn_exp2f(p0[kt][e] - m);
            p0[kt][e] = pv;
            l += pv;
        }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = q_ok ? 1.0f / l : 0.f;        // padded query rows contribute nothing to column sums
#pragma unroll
    for (int kt = 0; kt < NP / 16; ++kt) p0[kt] *= inv;
    m_out = m;
    l_out = l;
}

// r_i = sum_j P0_ij b_j  for the lane's query
template <int NP>
__device__ __forceinline__ float row_dot(const f32x4_t (&p0)[NP / 16], const float* bvec, int lane) {
    const int g = lane >> 4;
    float r = 0.f;
#pragma unroll
    for (int kt = 0; kt < NP / 16; ++kt) {
        const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(bvec + kt * 16 + 4 * g);
        r += (p0[kt][0] * b4[0] + p0[kt][1] * b4[1]) + (p0[kt][2] * b4[2] + p0[kt][3] * b4[3]);
    }
    r += __shfl_xor(r, 16, 64);
    r += __shfl_xor(r, 32, 64);
    return r;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(SK_THREADS) void sinkhorn_fwd_kernel(const SinkParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kimg = smem;
    char* vimg = smem + NP * 128;
    float* bvec = reinterpret_cast<float*>(smem + 2 * NP * 128);
    float* colpart = bvec + NP;                                  // [SK_WAVES][NP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int N = p.N;
    const long long ldq = 3ll * p.H * DH;
    const bf16_t* qbase = p.qkv + (long long)b * N * ldq + h * DH;
    load_image<NP, false, SK_THREADS>(kimg, qbase + p.H * DH, ldq, N, tid);
    load_image<NP, true, SK_THREADS>(vimg, qbase + 2 * p.H * DH, ldq, N, tid);
    for (int j = tid; j < NP; j += SK_THREADS) bvec[j] = 1.0f;
    __syncthreads();

    const int g = lane >> 4, qc = lane & 15;
    const int nqt = (N + 15) >> 4;
    const bool active = wave < nqt;
    const int q = wave * 16 + qc;
    const bool q_ok = active && q < N;
    const int qr = q < N ? q : N - 1;
    float* scal = p.scal + (long long)bh * 7 * N;

    f32x4_t p0[NP / 16];
    float m = 0.f, l = 1.f;
    {
        bf16x8_t qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = load_frag_global(qbase + (long long)qr * ldq + ks * 32 + g * 8);
        softmax_tile<NP>(kimg, qf, p0, N, q_ok, p.scale * LOG2E, lane, m, l);
    }
    if (q_ok && g == 0) p.lse[(long long)bh * N + q] = (m + __builtin_amdgcn_logf(l)) * LN2;

    float a = 1.f;
#pragma unroll 1
    for (int t = 0; t < 3; ++t) {
        a = inv_or_zero(row_dot<NP>(p0, bvec, lane));                      // P /= rowsum(P)
        if (q_ok && g == 0) scal[(2 * t) * N + q] = a;
#pragma unroll
        for (int kt = 0; kt < NP / 16; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = row16_sum(active ? a * p0[kt][e] : 0.f);
                if (qc == 0) colpart[wave * NP + kt * 16 + 4 * g + e] = v;
            }
        __syncthreads();
        for (int j = tid; j < NP; j += SK_THREADS) {                        // P /= colsum(P)
            float c = 0.f;
            for (int w = 0; w < nqt; ++w) c += colpart[w * NP + j];
            const float bn = inv_or_zero(c);
            bvec[j] = bn;
            if (j < N) scal[(2 * t + 1) * N + j] = bn;
        }
        __syncthreads();
    }
    a = inv_or_zero(row_dot<NP>(p0, bvec, lane));                          // final P /= rowsum(P)
    if (q_ok && g == 0) scal[6 * N + q] = a;

    if (active) {
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NP / 32; ++kk) {
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bvec + kk * 32 + 4 * g);
            const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(bvec + kk * 32 + 16 + 4 * g);
            const bf16x8_t pf = pack_frag(p0[2 * kk] * b0 * a, p0[2 * kk + 1] * b1 * a);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(tr_frag_vimg(vimg, kk * 32, dt, lane), pf, o[dt]);
        }
        if (q < N) {
            bf16_t* dst = p.o + ((long long)b * N + q) * (p.H * DH) + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store_bf16x4(dst + dt * 16, o[dt]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, query-owner kernel: dQ, and dS^T / P7^T scratch for the key-owner kernel.
// 8 FAT waves (2 per SIMD, <= 256 VGPRs), each owning TWO 16-query tiles (wave w: tiles w and w + 8).  The only N x N
// state a tile keeps in registers is G (fp32, NP / 4 registers), walked back through the seven normalisations in place;
// **P0 is recomputed on the MFMA every time it is needed** (S^T = K Q^T, 2 MFMAs + 4 exp2 per key tile and query tile, ten
// passes per head): the MFMA pipe is otherwise idle in this kernel, and with P0 also resident (round 1: 16 waves x one tile
// at 128 VGPRs, 488 spilled registers; first round-2 form: 82) the column steps reloaded G from scratch and took 1.6 of the
// kernel's 2.3 ms per layer (ablation builds, tools/sinkhorn_bench.py).
// ---------------------------------------------------------------------------------------------
// hipcc unrolls the key-tile loops fully (the register arrays need static indices) and then hoists every LDS vector load of
// every iteration to the top: a compiler-level memory fence per iteration keeps the loads where they are written.
#define SK_KEEP_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int NP, int TPW>          // TPW = query tiles per wave: 16 / TPW waves
__global__ __launch_bounds__(1024 / TPW, TPW == 1 ? 4 : 2) void sinkhorn_bwd_q_kernel(const SinkParams p) {
    constexpr int SKQ_WAVES = 16 / TPW, SKQ_THREADS = 64 * SKQ_WAVES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kimg = smem;                                            // row + transposed reads
    char* vimg = smem + NP * 128;                                 // row reads
    float* bv = reinterpret_cast<float*>(smem + 2 * NP * 128);    // [4][NP]: b0 = 1, b1, b2, b3
    float* kap = bv + 4 * NP;                                     // [NP] column correction of the current step
    float* colpart = kap + NP;                                    // [SKQ_WAVES][NP]
    constexpr int NT = NP / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int N = p.N;
    const long long ldq = 3ll * p.H * DH, ldo = (long long)p.H * DH;
    const bf16_t* qbase = p.qkv + (long long)b * N * ldq + h * DH;
    const bf16_t* dobase = p.dout + (long long)b * N * ldo + h * DH;
    const float* scal = p.scal + (long long)bh * 7 * N;
    load_image<NP, false, SKQ_THREADS>(kimg, qbase + p.H * DH, ldq, N, tid);
    load_image<NP, false, SKQ_THREADS>(vimg, qbase + 2 * p.H * DH, ldq, N, tid);
    for (int j = tid; j < NP; j += SKQ_THREADS) {
        bv[j] = j < N ? 1.0f : 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) bv[(t + 1) * NP + j] = j < N ? scal[(2 * t + 1) * N + j] : 0.f;
    }
    __syncthreads();

    const int g = lane >> 4, qc = lane & 15;
    const int nqt = (N + 15) >> 4;
    const float sc = p.scale * LOG2E;
    bool active[TPW], q_ok[TPW];
    int q[TPW];
    float av[TPW][5];               // a0 = 1, a1 .. a4 of the lane's query, per tile
    float lse2[TPW];
    bf16x8_t qf[TPW][2];            // Q fragments: kept for the P0 recomputation
    f32x4_t G[TPW][NT];             // G = dP7^T (fp32), walked back through the normalisations in place
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int tile = wave + SKQ_WAVES * u;
        active[u] = tile < nqt;
        q[u] = tile * 16 + qc;
        q_ok[u] = active[u] && q[u] < N;
        const int qr = q[u] < N ? q[u] : N - 1;
        av[u][0] = q_ok[u] ? 1.f : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) av[u][t + 1] = q_ok[u] ? scal[(2 * t) * N + qr] : 0.f;
        bf16x8_t dof[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[u][ks] = load_frag_global(qbase + (long long)qr * ldq + ks * 32 + g * 8);
            dof[ks] = load_frag_global(dobase + (long long)qr * ldo + ks * 32 + g * 8);
        }
        lse2[u] = q_ok[u] ? p.lse[(long long)bh * N + qr] * LOG2E : INFINITY;      // exp2(s - inf) = 0: padded queries need no mask
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            SK_KEEP_ORDER();
            G[u][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) G[u][kt] = mfma16(row_frag_img(vimg, kt * 16, ks, lane), dof[ks], G[u][kt]);
        }
    }
    // P0 of key tile kt for the wave's query tiles (zero for padded keys / queries)
    auto p0_tiles = [&](int kt, f32x4_t (&p0)[TPW]) {
        SK_KEEP_ORDER();
        bf16x8_t kr[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kr[ks] = row_frag_img(kimg, kt * 16, ks, lane);
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            f32x4_t st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) st = mfma16(kr[ks], qf[u][ks], st);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pv = __builtin_amdgcn_exp2f(fmaf(st[e], sc, -lse2[u]));
                // padded keys (>= N) can only sit in the last two key tiles (NP - N < 32): a per-element mask on every tile
                // costs a hoisted SGPR-pair each (112 scalar registers at N = 197: spilled to VGPR lanes and read back
                // with v_readlane + s_nop in every pass)
                if (kt >= NT - 2) pv = (kt * 16 + 4 * g + e < N) ? pv : 0.f;
                p0[u][e] = pv;
            }
        }
    };

    // hand P7^T to the key-owner kernel:  P7 = a4 P0 b3.  The scratch is written through buffer descriptors: the per-lane
    // offset (key row 4 g + e, query column) is 4 registers per tile, the key tile is the scalar soffset (with plain
    // pointers hipcc precomputes a 64-bit address per store).
    const __amdgpu_buffer_rsrc_t rwp = make_rsrc(p.ws_p + (long long)bh * NP * NP, (unsigned long long)NP * NP * 2);
    const __amdgpu_buffer_rsrc_t rwd = make_rsrc(p.ws_ds + (long long)bh * NP * NP, (unsigned long long)NP * NP * 2);
    unsigned wo[TPW][4];
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) wo[u][e] = (unsigned)(((4 * g + e) * NP + q[u]) * 2);
    auto ws_store = [&](const __amdgpu_buffer_rsrc_t& r, int u, int kt, int e, unsigned short v) {
        __builtin_amdgcn_raw_buffer_store_b16((short)v, r, wo[u][e], kt * 16 * NP * 2, 0);
    };
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        f32x4_t p0[TPW];
        p0_tiles(kt, p0);
        const f32x4_t b3 = *reinterpret_cast<const f32x4_t*>(bv + 3 * NP + kt * 16 + 4 * g);
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            if (active[u]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ws_store(rwp, u, kt, e, f32_to_bf16(av[u][4] * p0[u][e] * b3[e]));
            }
        }
        SK_KEEP_ORDER();
    }

    // walk the normalisations backwards: steps 7 (row) 6 (col) 5 (row) 4 (col) 3 (row) 2 (col) 1 (row)
    // (fully unrolled: av[][] must be indexed statically or it lands in scratch)
#pragma unroll
    for (int t = 3; t >= 0; --t) {
        {   // row step with alpha = a_{t+1}, alpha_prev = a_t, beta = b_t
            const float* bt = bv + t * NP;
            float rho[TPW];
#pragma unroll
            for (int u = 0; u < TPW; ++u) rho[u] = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x4_t p0[TPW];
                p0_tiles(kt, p0);
                const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(bt + kt * 16 + 4 * g);
#pragma unroll
                for (int u = 0; u < TPW; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) rho[u] += G[u][kt][e] * p0[u][e] * b4[e];
            }
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                float r = rho[u];
                r += __shfl_xor(r, 16, 64);
                r += __shfl_xor(r, 32, 64);
                r *= av[u][t + 1];
                const float f = ratio_or_zero(av[u][t + 1], av[u][t]);
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) G[u][kt] = (G[u][kt] - r) * f;
            }
        }
        if (t == 0) break;
        {   // column step with beta = b_t, beta_prev = b_{t-1}, alpha = a_t: the tiles' contributions are added before the
            // 16-lane reduction
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x4_t p0[TPW];
                p0_tiles(kt, p0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = 0.f;                                       // av = 0 for inactive tiles
#pragma unroll
                    for (int u = 0; u < TPW; ++u) x += G[u][kt][e] * av[u][t] * p0[u][e];
                    const float v = row16_sum(x);
                    if (qc == 0) colpart[wave * NP + kt * 16 + 4 * g + e] = v;
                }
            }
            __syncthreads();
            for (int j = tid; j < NP; j += SKQ_THREADS) {
                float c = 0.f;
#pragma unroll
                for (int w = 0; w < SKQ_WAVES; ++w) c += colpart[w * NP + j];
                // kap = beta_j * colsum, and the ratio beta_t / beta_{t-1} of the rescale that follows (once per column
                // here instead of once per element in every wave)
                kap[j] = c * bv[t * NP + j];
                colpart[j] = ratio_or_zero(bv[t * NP + j], bv[(t - 1) * NP + j]);      // row 0 of colpart is free again: all partials were read
            }
            __syncthreads();
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                SK_KEEP_ORDER();
                const f32x4_t k4 = *reinterpret_cast<const f32x4_t*>(kap + kt * 16 + 4 * g);
                const f32x4_t r4 = *reinterpret_cast<const f32x4_t*>(colpart + kt * 16 + 4 * g);
#pragma unroll
                for (int u = 0; u < TPW; ++u) G[u][kt] = (G[u][kt] - k4) * r4;
            }
            __syncthreads();      // kap / colpart are rewritten by the next column step
        }
    }
    // softmax backward: dS = P0 (G - sum_j G P0) * scale
    {
        float sd[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) sd[u] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4_t p0[TPW];
            p0_tiles(kt, p0);
#pragma unroll
            for (int u = 0; u < TPW; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) sd[u] += G[u][kt][e] * p0[u][e];
        }
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            sd[u] += __shfl_xor(sd[u], 16, 64);
            sd[u] += __shfl_xor(sd[u], 32, 64);
        }
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4_t p0[TPW];
            p0_tiles(kt, p0);
#pragma unroll
            for (int u = 0; u < TPW; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) G[u][kt][e] = p0[u][e] * (G[u][kt][e] - sd[u]) * p.scale;
        }
    }

#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int tile = wave + SKQ_WAVES * u;
        if (!active[u] && tile < NT) {
            // query columns nqt*16 .. NP-1 of the scratch are read (against zero Q / dO rows) by the key-owner kernel:
            // make them finite
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ws_store(rwd, u, kt, e, 0);
                    ws_store(rwp, u, kt, e, 0);
                }
        }
        if (active[u]) {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                SK_KEEP_ORDER();
#pragma unroll
                for (int e = 0; e < 4; ++e) ws_store(rwd, u, kt, e, f32_to_bf16(G[u][kt][e]));
            }
            f32x4_t dq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NP / 32; ++kk) {
                SK_KEEP_ORDER();
                const bf16x8_t dsf = pack_frag(G[u][2 * kk], G[u][2 * kk + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16(tr_frag_img(kimg, kk * 32, dt, lane), dsf, dq[dt]);
            }
            if (q[u] < N) {
                bf16_t* dst = p.dqkv + ((long long)b * N + q[u]) * ldq + h * DH + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) store_bf16x4(dst + dt * 16, dq[dt]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, key-owner kernel: dK = dS^T Q, dV = P7^T dO  (dS^T, P7^T come from the scratch as [key][q] rows)
// ---------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(SKB_THREADS, 4) void sinkhorn_bwd_kv_kernel(const SinkParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* qimg = smem;
    char* doimg = smem + NP * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int N = p.N;
    const long long ldq = 3ll * p.H * DH, ldo = (long long)p.H * DH;
    const bf16_t* qbase = p.qkv + (long long)b * N * ldq + h * DH;
    const bf16_t* dobase = p.dout + (long long)b * N * ldo + h * DH;
    load_image<NP, false, SKB_THREADS>(qimg, qbase, ldq, N, tid);
    load_image<NP, false, SKB_THREADS>(doimg, dobase, ldo, N, tid);
    __syncthreads();
    const bf16_t* wsp = p.ws_p + (long long)bh * NP * NP;
    const bf16_t* wsd = p.ws_ds + (long long)bh * NP * NP;
    const int g = lane >> 4, kc = lane & 15;
    const int nkt = (N + 15) >> 4;
    for (int kt = wave; kt < nkt; kt += SKB_THREADS / 64) {
        const int key = kt * 16 + kc;          // < NP always: scratch rows exist (zero / garbage-free for key >= N? guarded below)
        f32x4_t dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dv[dt] = dk[dt];
        }
        const int kr = key < N ? key : N - 1;
#pragma unroll 1
        for (int qq = 0; qq < NP / 32; ++qq) {
            // B operand: lane holds rows-of-scratch key, 8 consecutive queries 32 qq + 8 g ..  (natural k order;
            // the matching A fragments below use the same natural order: rows 32 qq + 8 g + {0..7})
            const bf16x8_t dsf = load_frag_global(wsd + (long long)kr * NP + qq * 32 + g * 8);
            const bf16x8_t pf = load_frag_global(wsp + (long long)kr * NP + qq * 32 + g * 8);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                // transposed A fragments with k = q in natural order: lane (g, i) needs rows 32 qq + 8 g + {0..7}
                const int q4 = (lane & 15) >> 2, pp = lane & 3;
                const int r0 = qq * 32 + 8 * g + q4;
                const int c = 2 * dt + (pp >> 1);
                const bf16x8_t qa = cat4(lds_read_tr16_b64(qimg + img_off(r0, c) + (pp & 1) * 8),
                                         lds_read_tr16_b64(qimg + img_off(r0 + 4, c) + (pp & 1) * 8));
                const bf16x8_t da = cat4(lds_read_tr16_b64(doimg + img_off(r0, c) + (pp & 1) * 8),
                                         lds_read_tr16_b64(doimg + img_off(r0 + 4, c) + (pp & 1) * 8));
                dk[dt] = mfma16(qa, dsf, dk[dt]);
                dv[dt] = mfma16(da, pf, dv[dt]);
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

int np_of(int N) { return (N + 31) / 32 * 32; }

template <int NP>
int launch_sk_fwd(const SinkParams& p, hipStream_t s) {
    constexpr int lds = 2 * NP * 128 + (1 + SK_WAVES) * NP * 4;
    static int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_fwd_kernel<NP>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != 0) return attr;
    hipLaunchKernelGGL((sinkhorn_fwd_kernel<NP>), dim3(p.B * p.H), dim3(SK_THREADS), lds, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}

template <int NP>
int launch_sk_bwd(const SinkParams& p, hipStream_t s) {
    constexpr int TPW = NRV_SK_TPW;
    constexpr int lds_q = 2 * NP * 128 + (5 + 16 / TPW) * NP * 4;
    constexpr int lds_kv = 2 * NP * 128;
    static int attr1 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_bwd_q_kernel<NP, TPW>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
    static int attr2 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_bwd_kv_kernel<NP>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
    if (attr1 != 0) return attr1;
    if (attr2 != 0) return attr2;
    hipLaunchKernelGGL((sinkhorn_bwd_q_kernel<NP, TPW>), dim3(p.B * p.H), dim3(1024 / TPW), lds_q, s, p);
    NRV_CHECK_LAUNCH();
    hipLaunchKernelGGL((sinkhorn_bwd_kv_kernel<NP>), dim3(p.B * p.H), dim3(SKB_THREADS), lds_kv, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}

int sk_check(int B, int N, int H, int dh) {
    if (B <= 0 || N <= 0 || H <= 0) return NRV_ERR_SHAPE;
    if (dh != DH || N > 256) return NRV_ERR_SHAPE;
    return 0;
}

#define NRV_SK_DISPATCH(N, CALL)                                   \
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

extern "C" int nrv_attn_sinkhorn_fwd(const void* qkv_bf16, void* out_bf16, float* lse, float* scalings,
                                     int B, int N, int H, int dh, float scale, void* stream) {
    if (!qkv_bf16 || !out_bf16 || !lse || !scalings) return NRV_ERR_NULL;
    if (int e = sk_check(B, N, H, dh)) return e;
    if (!nrv_aligned16(qkv_bf16) || !nrv_aligned16(out_bf16)) return NRV_ERR_ALIGN;
    SinkParams p{};
    p.qkv = static_cast<const bf16_t*>(qkv_bf16);
    p.o = static_cast<bf16_t*>(out_bf16);
    p.lse = lse; p.scal = scalings;
    p.B = B; p.N = N; p.H = H; p.scale = scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    NRV_SK_DISPATCH(N, launch_sk_fwd<NPV>(p, s));
}

extern "C" size_t nrv_attn_sinkhorn_bwd_workspace(int B, int N, int H) {
    if (B <= 0 || N <= 0 || H <= 0) return 0;
    const size_t np = (size_t)np_of(N);
    return 2 * (size_t)B * H * np * np * 2;
}

extern "C" int nrv_attn_sinkhorn_bwd(const void* qkv_bf16, const void* dout_bf16, const float* lse, const float* scalings,
                                     void* dqkv_bf16, void* workspace, size_t workspace_bytes,
                                     int B, int N, int H, int dh, float scale, void* stream) {
    if (!qkv_bf16 || !dout_bf16 || !lse || !scalings || !dqkv_bf16 || !workspace) return NRV_ERR_NULL;
    if (int e = sk_check(B, N, H, dh)) return e;
    if (!nrv_aligned16(qkv_bf16) || !nrv_aligned16(dout_bf16) || !nrv_aligned16(dqkv_bf16) || !nrv_aligned16(workspace)) return NRV_ERR_ALIGN;
    const size_t np = (size_t)np_of(N);
    const size_t half = (size_t)B * H * np * np * 2;
    if (workspace_bytes < 2 * half) return NRV_ERR_WORKSPACE;
    SinkParams p{};
    p.qkv = static_cast<const bf16_t*>(qkv_bf16);
    p.dout = static_cast<const bf16_t*>(dout_bf16);
    p.dqkv = static_cast<bf16_t*>(dqkv_bf16);
    p.lse = const_cast<float*>(lse); p.scal = const_cast<float*>(scalings);
    p.ws_ds = static_cast<bf16_t*>(workspace);
    p.ws_p = reinterpret_cast<bf16_t*>(static_cast<char*>(workspace) + half);
    p.B = B; p.N = N; p.H = H; p.scale = scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    NRV_SK_DISPATCH(N, launch_sk_bwd<NPV>(p, s));
}
