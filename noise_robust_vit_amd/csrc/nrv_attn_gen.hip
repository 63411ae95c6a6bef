// Fused multi-head self-attention for the shapes the single-pass kernels of nrv_attn.hip do not hold on chip:
// any token count N (N > 256: ViT-B/16 at 384 px = 577 tokens after interpolate_embeddings, vit.py:522-603; vit_h_14 = 257
// tokens, vit.py:512-519) and head dims 32 / 64 / 80 / 96 / 128 (vit_h_14: 1280 / 16 = 80; SimpleViT(dim_head=...),
// simple_vit.py:101-114).  Same interface, layouts and numerics contract as nrv_attn.hip (bf16 operands, fp32 MFMA
// accumulation, fp32 softmax in the exp2 domain, P fed to P.V in bf16 and normalised by the fp32 row sum, natural-log LSE).
//
// Streaming ("flash") form: a workgroup of 4 waves owns 64 queries (forward, dQ) or 64 keys (dK / dV) of one (batch, head)
// and sweeps the other side in tiles of 64 rows staged in LDS; scores never leave the CU.
//   forward : online softmax -- running row max m and row sum l per query, O^T rescaled by exp2(m_old - m_new) per key tile
//   backward: P recomputed from q, k and the saved LSE; query-owner pass (dQ, delta = rowsum(dO * O)) + key-owner pass
//             (dK, dV), no atomics, deterministic
// MFMA orientation as in nrv_attn.hip (16x16x32 bf16): S^T = K Q^T, a lane owns one query column with its keys in
// registers; the bf16 P^T / dS^T accumulators ARE the B operands of O^T = V^T P^T / dQ^T = K^T dS^T (the k slots of a
// 32-key step are keys {4g .. 4g+3, 16+4g .. 16+4g+3} for lane group g in both operands); transposed A operands come from
// ds_read_b64_tr_b16.  The head dim is padded to DHP = 32 KS in LDS and registers (zero columns; dh = 80 -> 96).
//
// Tile image: [64 rows][DHP] bf16, rows of 2 DHP bytes, the 32-byte unit u of row r stored at unit u ^ ((r >> 1) & UM): one
// image serves the row reads (ds_read_b128) and the transposed reads of the same tile.
#include "nrv_attn_common.hpp"

namespace {

using nrv_attn::LN2;
using nrv_attn::LOG2E;
using nrv_attn::pack_frag;

struct GenParams {
    const bf16_t* qkv;     // [B, N, 3*H*dh]
    const bf16_t* out;     // [B, N, H*dh]      (bwd)
    const bf16_t* dout;    // [B, N, H*dh]      (bwd)
    bf16_t* o;             // fwd output
    bf16_t* dqkv;          // bwd output
    float* lse;            // [B, H, N]
    float* delta;          // [B, H, N]
    int B, N, H, dh;
    float scale;
};

constexpr int GT = 64;          // rows of a streamed tile = rows owned by a workgroup (4 waves x 16)
constexpr int GEN_THREADS = 256;

template <int KS>
struct GenCfg {
    static constexpr int DHP = 32 * KS, RB = 2 * DHP, TILE = GT * RB, DT = DHP / 16;
    static constexpr int UNITS = RB / 32;                                  // 2, 4, 6, 8
    static constexpr int UM = UNITS == 2 ? 1 : UNITS == 4 ? 3 : UNITS == 6 ? 1 : 7;   // XOR mask that stays inside a row
};

template <int KS>
__device__ __forceinline__ int tile_off(int r, int c /* 16-byte chunk */) {
    using C = GenCfg<KS>;
    return r * C::RB + ((((c >> 1) ^ ((r >> 1) & C::UM)) << 5) | ((c & 1) << 4));
}

// cooperative load of rows r0 .. r0 + 63 of a [N x dh] head slice (row stride ld elements) into a tile image;
// rows >= N and columns >= dh are zero
template <int KS>
__device__ __forceinline__ void load_tile(char* img, const bf16_t* src, long long ld, int r0, int N, int dh, int tid) {
    using C = GenCfg<KS>;
    constexpr int CPR = C::DHP / 8;                      // chunks per row
#pragma unroll
    for (int i = 0; i < GT * CPR / GEN_THREADS; ++i) {
        const int idx = i * GEN_THREADS + tid;
        const int r = idx / CPR, c = idx - r * CPR;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (r0 + r < N && c * 8 < dh) v = *reinterpret_cast<const u32x4_t*>(src + (long long)(r0 + r) * ld + c * 8);
        *reinterpret_cast<u32x4_t*>(img + tile_off<KS>(r, c)) = v;
    }
}

// the same tile in two halves: global -> registers (issued a tile ahead, in flight during the current tile's arithmetic), registers -> LDS.
// A thread holds KS 16-byte chunks of a tile (64 rows x 4 KS chunks over 256 threads).
template <int KS>
__device__ __forceinline__ void fetch_tile(u32x4_t (&v)[KS], const bf16_t* src, long long ld, int r0, int N, int dh, int tid) {
    using C = GenCfg<KS>;
    constexpr int CPR = C::DHP / 8;
    static_assert(GT * CPR / GEN_THREADS == KS, "chunks per thread");
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        const int idx = i * GEN_THREADS + tid;
        const int r = idx / CPR, c = idx - r * CPR;
        v[i] = u32x4_t{0u, 0u, 0u, 0u};
        if (r0 + r < N && c * 8 < dh) v[i] = *reinterpret_cast<const u32x4_t*>(src + (long long)(r0 + r) * ld + c * 8);
    }
}
template <int KS>
__device__ __forceinline__ void put_tile(char* img, const u32x4_t (&v)[KS], int tid) {
    using C = GenCfg<KS>;
    constexpr int CPR = C::DHP / 8;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        const int idx = i * GEN_THREADS + tid;
        const int r = idx / CPR, c = idx - r * CPR;
        *reinterpret_cast<u32x4_t*>(img + tile_off<KS>(r, c)) = v[i];
    }
}

// row fragment (A or B operand whose 16 rows are tile rows rb .. rb + 15): lane -> row rb + (lane & 15), k = 32 ks + 8 (lane >> 4) ..
template <int KS>
__device__ __forceinline__ bf16x8_t row_frag(const char* img, int rb, int ks, int lane) {
    return lds_read_b128(img + tile_off<KS>(rb + (lane & 15), 4 * ks + (lane >> 4)));
}
// transposed fragment (A operand): columns 16 dt .. 16 dt + 15 of tile rows rb + {4g .. 4g+3, 16+4g .. 16+4g+3}
template <int KS>
__device__ __forceinline__ bf16x8_t tr_frag(const char* img, int rb, int dt, int lane) {
    using C = GenCfg<KS>;
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int r0 = rb + 4 * g + q, r1 = r0 + 16;
    const char* a0 = img + r0 * C::RB + ((dt ^ ((r0 >> 1) & C::UM)) << 5) + pp * 8;
    const char* a1 = img + r1 * C::RB + ((dt ^ ((r1 >> 1) & C::UM)) << 5) + pp * 8;
    return cat4(lds_read_tr16_b64(a0), lds_read_tr16_b64(a1));
}
// the lane's 8 features 32 ks + 8 g .. of row `row` of a [N x dh] slice straight from global memory (zero beyond N / dh)
__device__ __forceinline__ bf16x8_t glob_frag(const bf16_t* src, long long ld, int row, int N, int dh, int ks, int g) {
    const int d0 = 32 * ks + 8 * g;
    if (row < N && d0 < dh) return *reinterpret_cast<const bf16x8_t*>(src + (long long)row * ld + d0);
    return bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
}
__device__ __forceinline__ float quad_max(float v) {        // over the 4 lanes that share a query / key column (lane ^ 16, ^ 32)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(GEN_THREADS) void attn_gen_fwd_kernel(const GenParams p) {
    using C = GenCfg<KS>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::TILE];
    char* kimg = smem;
    char* vimg = smem + C::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, qc = lane & 15;
    const int N = p.N, H = p.H, dh = p.dh;
    const int nqb = (N + GT - 1) / GT;
    const int bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
    const int b = bh / H, h = bh - b * H;
    const long long ldq = 3ll * H * dh;
    const bf16_t* qbase = p.qkv + (long long)b * N * ldq + h * dh;
    const bf16_t* kbase = qbase + (long long)H * dh;
    const bf16_t* vbase = kbase + (long long)H * dh;
    const int q = qb * GT + wave * 16 + qc;               // this lane's query
    const float sc = p.scale * LOG2E;

    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = glob_frag(qbase, ldq, q, N, dh, ks, g);
    float m = -INFINITY, l = 0.f;
    f32x4_t ot[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) ot[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    u32x4_t kreg[KS], vreg[KS];                           // the next K / V tile, requested one tile ahead
    fetch_tile<KS>(kreg, kbase, ldq, 0, N, dh, tid);
    fetch_tile<KS>(vreg, vbase, ldq, 0, N, dh, tid);
    for (int k0 = 0; k0 < N; k0 += GT) {
        __syncthreads();                                  // every wave is done with the previous tile
        put_tile<KS>(kimg, kreg, tid);
        put_tile<KS>(vimg, vreg, tid);
        __syncthreads();
        if (k0 + GT < N) {
            fetch_tile<KS>(kreg, kbase, ldq, k0 + GT, N, dh, tid);
            fetch_tile<KS>(vreg, vbase, ldq, k0 + GT, N, dh, tid);
        }
        // S^T = K Q^T for the four 16-key sub-tiles; lane: keys k0 + 16 sub + 4 g + e of query q
        f32x4_t st[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a = mfma16(row_frag<KS>(kimg, sub * 16, ks, lane), qf[ks], a);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = (k0 + sub * 16 + 4 * g + e < N) ? a[e] * sc : -INFINITY;
                tmax = fmaxf(tmax, a[e]);
            }
            st[sub] = a;
        }
        tmax = quad_max(tmax);
        const float mn = fmaxf(m, tmax);                  // finite: every tile holds at least one key < N
        const float alpha = __builtin_amdgcn_exp2f(m - mn);   // m = -inf on the first tile: 0
        float ps = 0.f;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(st[sub][e] - mn);
                st[sub][e] = pv;
                ps += pv;
            }
        l = l * alpha + quad_sum(ps);
        m = mn;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) ot[dt] *= alpha;
        // O^T += V^T P^T, two 32-key steps
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8_t pf = pack_frag(st[2 * kk], st[2 * kk + 1]);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) ot[dt] = mfma16(tr_frag<KS>(vimg, kk * 32, dt, lane), pf, ot[dt]);
        }
    }
    if (q < N) {
        const float inv = 1.0f / l;
        bf16_t* dst = p.o + ((long long)b * N + q) * ((long long)H * dh) + h * dh;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            const int d0 = dt * 16 + 4 * g;
            if (d0 < dh) nrv_attn::store_bf16x4(dst + d0, ot[dt] * inv);
        }
        if (g == 0) p.lse[((long long)b * H + h) * N + q] = (m + __builtin_amdgcn_logf(l)) * LN2;
    }
}

// ---------------------------------------------------------------------------------------------
// backward, query-owner pass: dQ = scale * dS K with dS = P o (dP - delta), dP = dO V^T; also writes delta = rowsum(dO o O)
// ---------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(GEN_THREADS) void attn_gen_dq_kernel(const GenParams p) {
    using C = GenCfg<KS>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::TILE];
    char* kimg = smem;
    char* vimg = smem + C::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, qc = lane & 15;
    const int N = p.N, H = p.H, dh = p.dh;
    const int nqb = (N + GT - 1) / GT;
    const int bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
    const int b = bh / H, h = bh - b * H;
    const long long ldq = 3ll * H * dh, ldo = (long long)H * dh;
    const bf16_t* qbase = p.qkv + (long long)b * N * ldq + h * dh;
    const bf16_t* kbase = qbase + (long long)H * dh;
    const bf16_t* vbase = kbase + (long long)H * dh;
    const bf16_t* obase = p.out + (long long)b * N * ldo + h * dh;
    const bf16_t* dobase = p.dout + (long long)b * N * ldo + h * dh;
    const int q = qb * GT + wave * 16 + qc;
    const float sc = p.scale * LOG2E;

    bf16x8_t qf[KS], dof[KS];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = glob_frag(qbase, ldq, q, N, dh, ks, g);
        dof[ks] = glob_frag(dobase, ldo, q, N, dh, ks, g);
        const bf16x8_t of = glob_frag(obase, ldo, q, N, dh, ks, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl = fmaf(bf16_to_f32((unsigned short)dof[ks][e]), bf16_to_f32((unsigned short)of[e]), dl);
    }
    dl = quad_sum(dl);
    const long long sidx = ((long long)b * H + h) * N + (q < N ? q : 0);
    const float lse2 = q < N ? p.lse[sidx] * LOG2E : INFINITY;      // exp2(s - inf) = 0 for padded queries
    if (q < N && g == 0) p.delta[sidx] = dl;

    f32x4_t dqt[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) dqt[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    u32x4_t kreg[KS], vreg[KS];                           // the next K / V tile, requested one tile ahead
    fetch_tile<KS>(kreg, kbase, ldq, 0, N, dh, tid);
    fetch_tile<KS>(vreg, vbase, ldq, 0, N, dh, tid);
    for (int k0 = 0; k0 < N; k0 += GT) {
        __syncthreads();
        put_tile<KS>(kimg, kreg, tid);
        put_tile<KS>(vimg, vreg, tid);
        __syncthreads();
        if (k0 + GT < N) {
            fetch_tile<KS>(kreg, kbase, ldq, k0 + GT, N, dh, tid);
            fetch_tile<KS>(vreg, vbase, ldq, k0 + GT, N, dh, tid);
        }
        f32x4_t ds[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = mfma16(row_frag<KS>(kimg, sub * 16, ks, lane), qf[ks], s);
                dp = mfma16(row_frag<KS>(vimg, sub * 16, ks, lane), dof[ks], dp);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = (k0 + sub * 16 + 4 * g + e < N) ? __builtin_amdgcn_exp2f(fmaf(s[e], sc, -lse2)) : 0.f;
                ds[sub][e] = pv * (dp[e] - dl);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8_t df = pack_frag(ds[2 * kk], ds[2 * kk + 1]);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) dqt[dt] = mfma16(tr_frag<KS>(kimg, kk * 32, dt, lane), df, dqt[dt]);
        }
    }
    if (q < N) {
        bf16_t* dst = p.dqkv + ((long long)b * N + q) * ldq + h * dh;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            const int d0 = dt * 16 + 4 * g;
            if (d0 < dh) nrv_attn::store_bf16x4(dst + d0, dqt[dt] * p.scale);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, key-owner pass: dV = P^T dO, dK = scale * dS^T Q.  Scores in [query][key] orientation: a lane owns one key
// column, its queries sit in registers and are the k slots of the dV^T / dK^T products.
// ---------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(GEN_THREADS) void attn_gen_dkv_kernel(const GenParams p) {
    using C = GenCfg<KS>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::TILE + 2 * GT * 4];
    char* qimg = smem;
    char* doimg = smem + C::TILE;
    float* lse2s = reinterpret_cast<float*>(smem + 2 * C::TILE);
    float* dels = lse2s + GT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, kc = lane & 15;
    const int N = p.N, H = p.H, dh = p.dh;
    const int nkb = (N + GT - 1) / GT;
    const int bh = blockIdx.x / nkb, kb = blockIdx.x - bh * nkb;
    const int b = bh / H, h = bh - b * H;
    const long long ldq = 3ll * H * dh, ldo = (long long)H * dh;
    const bf16_t* qbase = p.qkv + (long long)b * N * ldq + h * dh;
    const bf16_t* kbase = qbase + (long long)H * dh;
    const bf16_t* vbase = kbase + (long long)H * dh;
    const bf16_t* dobase = p.dout + (long long)b * N * ldo + h * dh;
    const float* lse = p.lse + ((long long)b * H + h) * N;
    const float* delta = p.delta + ((long long)b * H + h) * N;
    const int key = kb * GT + wave * 16 + kc;             // this lane's key
    const float sc = p.scale * LOG2E;

    bf16x8_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = glob_frag(kbase, ldq, key, N, dh, ks, g);
        vf[ks] = glob_frag(vbase, ldq, key, N, dh, ks, g);
    }
    f32x4_t dkt[C::DT], dvt[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) dkt[dt] = dvt[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    u32x4_t qreg[KS], doreg[KS];                          // the next Q / dO tile and its statistics, requested one tile ahead
    float lreg = INFINITY, dreg = 0.f;
    auto fetch_stats = [&](int q0) {
        if (tid < GT) {
            const int qq = q0 + tid;
            lreg = qq < N ? lse[qq] * LOG2E : INFINITY;              // exp2(s - inf) = 0 for padded queries
            dreg = qq < N ? delta[qq] : 0.f;
        }
    };
    // one tile ahead only where the registers are there: at head dims <= 64 the prefetch registers cost this kernel half its
    // occupancy (82 -> 128 VGPRs) and 17 % of its time (profiles/r04_streaming_attention_prefetch.txt)
    constexpr bool AHEAD = KS >= 3;
    if (AHEAD) {
        fetch_tile<KS>(qreg, qbase, ldq, 0, N, dh, tid);
        fetch_tile<KS>(doreg, dobase, ldo, 0, N, dh, tid);
        fetch_stats(0);
    }
    for (int q0 = 0; q0 < N; q0 += GT) {
        __syncthreads();
        if (!AHEAD) {
            fetch_tile<KS>(qreg, qbase, ldq, q0, N, dh, tid);
            fetch_tile<KS>(doreg, dobase, ldo, q0, N, dh, tid);
            fetch_stats(q0);
        }
        put_tile<KS>(qimg, qreg, tid);
        put_tile<KS>(doimg, doreg, tid);
        if (tid < GT) {
            lse2s[tid] = lreg;
            dels[tid] = dreg;
        }
        __syncthreads();
        if (AHEAD && q0 + GT < N) {
            fetch_tile<KS>(qreg, qbase, ldq, q0 + GT, N, dh, tid);
            fetch_tile<KS>(doreg, dobase, ldo, q0 + GT, N, dh, tid);
            fetch_stats(q0 + GT);
        }
        f32x4_t pt[4], ds[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            // S[query 16 sub + 4 g + e][key] and dP = dO V^T in the same orientation
            f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = mfma16(row_frag<KS>(qimg, sub * 16, ks, lane), kf[ks], s);
                dp = mfma16(row_frag<KS>(doimg, sub * 16, ks, lane), vf[ks], dp);
            }
            const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse2s + sub * 16 + 4 * g);
            const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dels + sub * 16 + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[e], sc, -l4[e]));
                pt[sub][e] = pv;
                ds[sub][e] = pv * (dp[e] - d4[e]);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8_t pf = pack_frag(pt[2 * kk], pt[2 * kk + 1]);
            const bf16x8_t df = pack_frag(ds[2 * kk], ds[2 * kk + 1]);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
                dvt[dt] = mfma16(tr_frag<KS>(doimg, kk * 32, dt, lane), pf, dvt[dt]);
                dkt[dt] = mfma16(tr_frag<KS>(qimg, kk * 32, dt, lane), df, dkt[dt]);
            }
        }
    }
    if (key < N) {
        bf16_t* dk = p.dqkv + ((long long)b * N + key) * ldq + (long long)H * dh + h * dh;
        bf16_t* dv = dk + (long long)H * dh;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            const int d0 = dt * 16 + 4 * g;
            if (d0 < dh) {
                nrv_attn::store_bf16x4(dk + d0, dkt[dt] * p.scale);
                nrv_attn::store_bf16x4(dv + d0, dvt[dt]);
            }
        }
    }
}

// introspection (recorder.py:24-31): P[b,h,q,k] = exp(scale q.k - lse[b,h,q]) in fp32, any N / dh; a plain VALU kernel
__global__ __launch_bounds__(256) void attn_gen_probs_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ lse,
                                                             float* __restrict__ probs, int B, int N, int H, int dh, float scale) {
    __shared__ float qs[16][129];
    const int bh = blockIdx.x, q0 = blockIdx.y * 16;
    const int b = bh / H, h = bh - b * H;
    const long long ldq = 3ll * H * dh;
    const bf16_t* base = qkv + (long long)b * N * ldq + h * dh;
    for (int i = threadIdx.x; i < 16 * dh; i += 256) {
        const int r = i / dh, d = i - r * dh;
        qs[r][d] = (q0 + r < N) ? bf16_to_f32(base[(long long)(q0 + r) * ldq + d]) : 0.f;
    }
    __syncthreads();
    const int qi = threadIdx.x & 15;
    const int q = q0 + qi;
    const float l = q < N ? lse[((long long)b * H + h) * N + q] : 0.f;
    for (int key = threadIdx.x >> 4; key < N; key += 16) {
        const bf16_t* kp = base + (long long)H * dh + (long long)key * ldq;
        float acc = 0.f;
        for (int c = 0; c < dh / 8; ++c) {
            const bf16x8_t kv = *reinterpret_cast<const bf16x8_t*>(kp + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qs[qi][c * 8 + e], bf16_to_f32((unsigned short)kv[e]), acc);
        }
        if (q < N) probs[(((long long)b * H + h) * N + q) * N + key] = __expf(acc * scale - l);
    }
}

int ks_of(int dh) {
    switch (dh) {
        case 32: return 1;
        case 64: return 2;
        case 80: case 96: return 3;
        case 128: return 4;
        default: return 0;
    }
}

template <int KS>
int launch_fwd(const GenParams& p, hipStream_t s) {
    const long long grid = (long long)p.B * p.H * ((p.N + GT - 1) / GT);
    hipLaunchKernelGGL(attn_gen_fwd_kernel<KS>, dim3((unsigned)grid), dim3(GEN_THREADS), 0, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}
template <int KS>
int launch_bwd(const GenParams& p, hipStream_t s) {
    const long long grid = (long long)p.B * p.H * ((p.N + GT - 1) / GT);
    hipLaunchKernelGGL(attn_gen_dq_kernel<KS>, dim3((unsigned)grid), dim3(GEN_THREADS), 0, s, p);
    NRV_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_gen_dkv_kernel<KS>, dim3((unsigned)grid), dim3(GEN_THREADS), 0, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Called by the C ABI entries of nrv_attn.hip for the shapes its single-pass kernels do not take (host-side dispatch on
// N and dh: one code path per shape class).  Arguments are already null- and alignment-checked there.
NRV_INTERNAL int nrv_attn_gen_supported(int B, int N, int H, int dh) {
    if (B <= 0 || N <= 0 || H <= 0 || ks_of(dh) == 0) return 0;
    if ((long long)B * H * ((N + GT - 1) / GT) > 0x7fffffffll) return 0;
    if ((long long)N * 3 * H * dh > 0x7fffffffll) return 0;
    return 1;
}

NRV_INTERNAL int nrv_attn_gen_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int dh, float scale, hipStream_t s) {
    GenParams p{};
    p.qkv = static_cast<const bf16_t*>(qkv);
    p.o = static_cast<bf16_t*>(out);
    p.lse = lse;
    p.B = B; p.N = N; p.H = H; p.dh = dh; p.scale = scale;
    switch (ks_of(dh)) {
        case 1: return launch_fwd<1>(p, s);
        case 2: return launch_fwd<2>(p, s);
        case 3: return launch_fwd<3>(p, s);
        case 4: return launch_fwd<4>(p, s);
        default: return NRV_ERR_SHAPE;
    }
}

NRV_INTERNAL int nrv_attn_gen_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta_ws,
                     int B, int N, int H, int dh, float scale, hipStream_t s) {
    GenParams p{};
    p.qkv = static_cast<const bf16_t*>(qkv);
    p.out = static_cast<const bf16_t*>(out);
    p.dout = static_cast<const bf16_t*>(dout);
    p.dqkv = static_cast<bf16_t*>(dqkv);
    p.lse = const_cast<float*>(lse);
    p.delta = delta_ws;
    p.B = B; p.N = N; p.H = H; p.dh = dh; p.scale = scale;
    switch (ks_of(dh)) {
        case 1: return launch_bwd<1>(p, s);
        case 2: return launch_bwd<2>(p, s);
        case 3: return launch_bwd<3>(p, s);
        case 4: return launch_bwd<4>(p, s);
        default: return NRV_ERR_SHAPE;
    }
}

NRV_INTERNAL int nrv_attn_gen_probs(const void* qkv, const float* lse, float* probs, int B, int N, int H, int dh, float scale, hipStream_t s) {
    hipLaunchKernelGGL(attn_gen_probs_kernel, dim3((unsigned)(B * H), (unsigned)((N + 15) / 16)), dim3(256), 0, s,
                       static_cast<const bf16_t*>(qkv), lse, probs, B, N, H, dh, scale);
    NRV_CHECK_LAUNCH();
    return 0;
}
