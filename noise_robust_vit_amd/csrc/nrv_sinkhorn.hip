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
// then the softmax backward and dQ = dS K on MFMA.  dK = dS^T Q and dV = P7^T dO contract over queries: P7 / dS are
// transposed through an LDS chunk inside the same kernel (nothing [N,N]-sized ever reaches HBM; the reference materialises
// ~10 fp32 copies).
#include "nrv_attn_common.hpp"
#define NRV_DEV_TU sinkhorn
#include <nrv_dev.hpp>       // instrumentation hooks: empty in the product (csrc/nrv_dev.hpp)

namespace {

using namespace nrv_attn;

typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;   // the backward kernel's resident copy of one P0 tile

constexpr int SK_THREADS = 1024;      // 16 waves: one 16-query tile each (N <= 256)
constexpr int SK_WAVES = 16;
constexpr int SK_TPW = 2;             // query tiles per wave of the backward kernel

struct SinkParams {
    const bf16_t* qkv;     // [B, N, 3*H*64]
    const bf16_t* dout;    // [B, N, H*64]
    bf16_t* o;             // [B, N, H*64]
    bf16_t* dqkv;          // [B, N, 3*H*64]
    float* lse;            // [B, H, N]
    float* scal;           // [B, H, 7, N]   a1 b1 a2 b2 a3 b3 a4
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

// Column sums of a key-tile pair (kt, kt + NT / 2) over the 16 query lanes of a DPP row, as a butterfly: after the exchange
// over lane bit 3 a lane keeps the tile of its half, after the one over bit 2 the element pair of its quad, and a quad
// reduction finishes -- 10 cross-lane adds for 8 values instead of 32, and one store per value instead of 16 lanes holding the
// same sum.  Lane (g, qc) ends with the sums of the keys 16 (kt + NT/2 b3) + 4 g + 2 b2 + {0, 1}, b3 = hi8, b2 = hi4:
// dst = partial-sum row of the wave + 16 kt + that lane offset; the lanes with (lane & 3) == 0 write.
// The selects are in the instructions: `v_add_f32_dpp dst, src, src <pattern> bank_mask:M` writes only the lanes of the banks in M
// (a bank = 4 consecutive lanes of a row), so "lanes 0-7 take the sum of the pair's first tile, lanes 8-15 that of its second" is two
// adds, not two selects and an add.  Written as one asm statement (18 cross-lane adds for 8 values): hipcc does not know these are
// DPP reads, so the statement carries its own wait states (a VALU result needs 2 before a DPP read of it).
__device__ __forceinline__ void col_sums_pair(const f32x4_t xa, const f32x4_t xb, float* dst, bool hi8, bool hi4, bool writer) {
    (void)hi8; (void)hi4;               // where a lane's sums belong is in `dst` (the caller's lane offset)
    float t0, t1, t2, t3, u0, u1;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"       // lanes 8-15: the pair's second tile
        "v_add_f32_dpp %1, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"   // lanes 4-7, 12-15: elements 2, 3
        "v_add_f32_dpp %5, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(u0), "=&v"(u1)
        : "v"(xa[0]), "v"(xa[1]), "v"(xa[2]), "v"(xa[3]), "v"(xb[0]), "v"(xb[1]), "v"(xb[2]), "v"(xb[3]));
    if (writer) {
        dst[0] = u0;
        dst[1] = u1;
    }
}

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
    // the row maximum is taken on the raw scores (sc > 0) and the scale folded into the exponent's fma; the row sum is gathered in
    // four packed accumulators: one instruction per element less in each loop
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NP / 16; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // padded keys (>= N) can only sit in the last two key tiles (NP - N < 32): a mask on every tile costs a hoisted SGPR
            // pair each, parked in VGPR lanes and read back (v_readlane) for every head
            if (kt >= NP / 16 - 2) p0[kt][e] = (kt * 16 + 4 * g + e < N) ? p0[kt][e] : -INFINITY;
            m = fmaxf(m, p0[kt][e]);
        }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    m *= sc;
    f32x4_t l4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NP / 16; ++kt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) p0[kt][e] = __builtin_amdgcn_exp2f(fmaf(p0[kt][e], sc, -m));
        l4 += p0[kt];
    }
    float l = (l4[0] + l4[1]) + (l4[2] + l4[3]);
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

// LDS-DMA of one [N x 64] bf16 head slice (row stride ld elements) into a row image with the img_off swizzle, or (VIMG) the
// vimg_off swizzle of the forward's V image; the swizzle is applied to the SOURCE chunk: the LDS side of a DMA instruction is
// lane-linear.  Rows >= N read as zero.  Asynchronous: vmcnt.
template <int NP, int NWAVES, bool VIMG = false>
__device__ __forceinline__ void dma_image(char* img, const bf16_t* src, long long ld, int N, int wave_u, int lane) {
    // the head's base address is uniform but comes out of an integer division (VALU): pin it to scalar registers, the
    // descriptor of the DMA must live in SGPRs
    const unsigned long long a = reinterpret_cast<unsigned long long>(src);
    const unsigned a_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));      // the builtin returns int: go through
    const unsigned a_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);              // unsigned before widening
    const unsigned long long au = ((unsigned long long)a_hi << 32) | (unsigned long long)a_lo;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(reinterpret_cast<const bf16_t*>(au), 0x7fffffffull);
    constexpr int NI = NP / 8;
#pragma unroll
    for (int i = 0; i < (NI + NWAVES - 1) / NWAVES; ++i) {
        const int j = wave_u + NWAVES * i;
        if (j < NI) {                                          // wave-uniform
            const int r = 8 * j + (lane >> 3), pos = lane & 7;
            const int c = VIMG ? ((((pos >> 1) ^ ((r >> 1) & 3)) << 1) | (pos & 1)) : (pos ^ ((r >> 1) & 7));
            dma16(rs, img + j * 1024, r < N ? (unsigned)(r * ld * 2 + c * 16) : NRV_OOB);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// One workgroup per CU walks the heads blockIdx.x, + gridDim.x, ...: the K / V images of the next head are requested (LDS-DMA
// into the other image pair) and its Q fragments loaded while the current head is computed.
template <int NP>
struct SkFwdLds {
    static constexpr int IMG = 2 * NP * 128;                        // K image + V image
    static constexpr int VEC = (2 + SK_WAVES) * NP * 4;             // b vector (two: heads alternate) + the waves' partial column sums
    static constexpr bool TWO = 2 * IMG + VEC <= 160 * 1024;        // room for the next head's images
    static constexpr int BYTES = (TWO ? 2 : 1) * IMG + VEC;
};

template <int NP>
__global__ __launch_bounds__(SK_THREADS) void sinkhorn_fwd_kernel(const SinkParams p) {
    using L = SkFwdLds<NP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bvecs = reinterpret_cast<float*>(smem + (L::TWO ? 2 : 1) * L::IMG);      // [2][NP]
    float* colpart = bvecs + 2 * NP;                             // [SK_WAVES][NP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N;
    const int nbh = p.B * p.H;
    const long long ldq = 3ll * p.H * DH;
    const int g = lane >> 4, qc = lane & 15;
    const int nqt = (N + 15) >> 4;
    const bool active = wave < nqt;                              // a wave beyond the head's query tiles only keeps the barriers
    const int q = wave * 16 + qc;
    const bool q_ok = active && q < N;
    const int qr = q < N ? q : N - 1;
    const bool hi8 = (qc & 8) != 0, hi4 = (qc & 4) != 0;
    float* colw = colpart + wave * NP + (hi8 ? NP / 2 : 0) + 4 * g + (hi4 ? 2 : 0);
    auto qbase_of = [&](int bh_) {
        const int b_ = bh_ / p.H, h_ = bh_ - b_ * p.H;
        return p.qkv + (long long)b_ * N * ldq + h_ * DH;
    };
    bf16x8_t qf_n[2];
    auto request = [&](int bh_, int buf) {
        const bf16_t* qb = qbase_of(bh_);
        dma_image<NP, SK_WAVES>(smem + buf * L::IMG, qb + p.H * DH, ldq, N, wave, lane);
        dma_image<NP, SK_WAVES, true>(smem + buf * L::IMG + NP * 128, qb + 2 * p.H * DH, ldq, N, wave, lane);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf_n[ks] = load_frag_global(qb + (long long)qr * ldq + ks * 32 + g * 8);
    };
    int bh = blockIdx.x;            // < nbh (host)
    request(bh, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0) through the builtin: hipcc's own counting stays exact
    asm volatile("" ::: "memory");
#pragma unroll 1
    for (int it = 0; bh < nbh; ++it, bh += gridDim.x) {
        const int cur = L::TWO ? (it & 1) : 0;
        const char* kimg = smem + cur * L::IMG;
        const char* vimg = kimg + NP * 128;
        float* bvec = bvecs + (it & 1) * NP;                     // the slower waves may still read the previous head's vector
        const int b = bh / p.H, h = bh - b * p.H;
        float* scal = p.scal + (long long)bh * 7 * N;
        for (int j = tid; j < NP; j += SK_THREADS) bvec[j] = 1.0f;
        bf16x8_t qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = qf_n[ks];
        __syncthreads();                           // this head's images (every wave waited for its parts below); every wave has left the previous head
        const bool more = bh + (int)gridDim.x < nbh;
        if (L::TWO) request(more ? bh + (int)gridDim.x : bh, cur ^ 1);

        f32x4_t p0[NP / 16];
        float a = 1.f;
        if (active) {
            float m = 0.f, l = 1.f;
            softmax_tile<NP>(kimg, qf, p0, N, q_ok, p.scale * LOG2E, lane, m, l);
            if (q_ok && g == 0) p.lse[(long long)bh * N + q] = (m + __builtin_amdgcn_logf(l)) * LN2;
        }
#pragma unroll 1
        for (int t = 0; t < 3; ++t) {
            if (active) {
                a = inv_or_zero(row_dot<NP>(p0, bvec, lane));                  // P /= rowsum(P)
                if (q_ok && g == 0) scal[(2 * t) * N + q] = a;
#pragma unroll
                for (int kt = 0; kt < NP / 32; ++kt)
                    col_sums_pair(p0[kt] * a, p0[kt + NP / 32] * a, colw + kt * 16, hi8, hi4, (lane & 3) == 0);
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
        // the next head's images and Q fragments were requested a head ago: waiting for them HERE, before the output stores,
        // keeps the store latency off the next head's start
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        if (active) {
            a = inv_or_zero(row_dot<NP>(p0, bvec, lane));                      // final P /= rowsum(P)
            if (q_ok && g == 0) scal[6 * N + q] = a;
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
        if (!L::TWO) {                             // one image pair: the next head's images replace this head's (grid = heads: no next head)
            __syncthreads();
        }
    }
}

// hipcc unrolls the key-tile loops fully (the register arrays need static indices) and then hoists every LDS vector load of
// every iteration to the top: a compiler-level memory fence per iteration keeps the loads where they are written.
#define SK_KEEP_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// ---------------------------------------------------------------------------------------------
// backward, ONE kernel per head: the query-owner walk above plus the key-owner products dV = P7^T dO and dK = dS^T Q, which
// contract over QUERIES.  The walk holds P7 / dS with a lane per query and its keys in registers (the operand layout of a
// contraction over keys: dQ = dS K); the transposition goes through an LDS chunk instead of 2 x [NP x NP] bf16 of HBM scratch
// per head and a second kernel: the waves write their 16-query tile slot u as bf16 rows [query][key] (one 8-byte store per
// key tile), and every wave then takes its key tiles (wave, wave + 8) and reads the chunk back with ds_read_b64_tr_b16 as the
// MFMA operand whose k index is the query (rows = queries: the same transposed pattern as the Q / dO / K image reads).
// LDS: K image | one image slot that holds dO (dV phase), then V (G = dO V^T), then Q (dK phase) | vectors | chunk of 128
// queries x (NP keys + 16) bf16 (row stride = 8 dwords mod 64: the 8 rows of a transposed read hit 8 different bank groups).
// ---------------------------------------------------------------------------------------------
// LDS size of the one-kernel backward; with room for a third image slot the V image is loaded with K and dO at the start and
// the Q image during the walk (both hidden), otherwise one slot holds dO, V and Q in turn
template <int NP, int TPW>
struct SkBwdLds {
    static constexpr int WAVES = 16 / TPW;
    static constexpr int CH = 16 * WAVES;                      // queries per chunk: tile slot u of every wave
    static constexpr int RS = NP * 2 + 32;                     // chunk row stride (bytes)
    static constexpr int VEC = (5 + WAVES) * NP * 4;
    static constexpr bool THREE = 3 * NP * 128 + VEC + CH * RS <= 160 * 1024;
    static constexpr int IMAGES = THREE ? 3 : 2;
    static constexpr int BYTES = IMAGES * NP * 128 + VEC + CH * RS;
};

template <int NP, int TPW>
__global__ __launch_bounds__(1024 / TPW, TPW == 1 ? 4 : 2) void sinkhorn_bwd_kernel(const SinkParams p) {
    using L = SkBwdLds<NP, TPW>;
    constexpr int SKQ_WAVES = L::WAVES, SKQ_THREADS = 64 * SKQ_WAVES;
    constexpr int CH = L::CH, RS = L::RS;
    constexpr bool THREE = L::THREE;
    static_assert((RS / 4) % 16 == 8, "chunk rows must be 8 dwords mod 16 apart");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const slot0 = smem;
    char* const img1 = smem + NP * 128;                           // dO, then (two slots: V, then) Q
    char* const slot2 = smem + (THREE ? 2 : 1) * NP * 128;        // three slots: K and V alternate between slot 0 and slot 2 from head to head
    float* bv = reinterpret_cast<float*>(smem + L::IMAGES * NP * 128);    // [4][NP]: b0 = 1, b1, b2, b3
    float* kap = bv + 4 * NP;                                     // [NP] column correction of the current step
    float* colpart = kap + NP;                                    // [SKQ_WAVES][NP]
    char* chunk = reinterpret_cast<char*>(colpart + SKQ_WAVES * NP);
    constexpr int NT = NP / 16;
    const int tid = threadIdx.x;
    int lane = tid & 63;            // re-derived (opaquely) at the phase boundaries: see relane()
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    NRV_STAMP_SEQ_VARS(tid);      // phase stamps of thread 0: empty hooks in the product (csrc/nrv_dev.hpp)
#define SK_STAMP() NRV_STAMP_SEQ()
    const int N = p.N;
    const int nbh = p.B * p.H;
    const long long ldq = 3ll * p.H * DH, ldo = (long long)p.H * DH;
    const int nqt = (N + 15) >> 4;
    const float sc = p.scale * LOG2E;
    int g = lane >> 4, qc = lane & 15;
    bool active[TPW];
    int q[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        active[u] = wave + SKQ_WAVES * u < nqt;
        q[u] = (wave + SKQ_WAVES * u) * 16 + qc;
    }
    auto qbase_of = [&](int bh_) {
        const int b_ = bh_ / p.H, h_ = bh_ - b_ * p.H;
        return p.qkv + (long long)b_ * N * ldq + h_ * DH;
    };
    // One workgroup per CU walks the heads blockIdx.x, + gridDim.x, ... (three image slots: otherwise the grid is the head count).
    // What a head needs before it can start -- its K image, Q fragments and LSE -- is requested during the previous head:
    // K into the slot its V image has left (after G is built), the per-lane values into registers during the dK phase; the
    // other operands are requested at the top and land under the P0 computation.
    bf16x8_t qf_n[TPW][2];
    float lse2_n[TPW];
    auto request_q_lse = [&](int bh_) {
        const bf16_t* qb = qbase_of(bh_);
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const bool ok = active[u] && q[u] < N;
            const int qr = q[u] < N ? q[u] : N - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf_n[u][ks] = load_frag_global(qb + (long long)qr * ldq + ks * 32 + g * 8);
            lse2_n[u] = ok ? p.lse[(long long)bh_ * N + qr] * LOG2E : INFINITY;      // exp2(s - inf) = 0: padded queries need no mask
        }
    };
    int bh = blockIdx.x;            // < nbh (host)
    dma_image<NP, SKQ_WAVES>(slot0, qbase_of(bh) + p.H * DH, ldq, N, wave, lane);
    request_q_lse(bh);
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) through the builtin: hipcc's own counting stays exact
    asm volatile("" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int it = 0; bh < nbh; ++it, bh += gridDim.x) {
    NRV_STAMP_SEQ_RESET();
    SK_STAMP();
    asm volatile("" : "+v"(lane));
    g = lane >> 4;
    qc = lane & 15;
#pragma unroll
    for (int u = 0; u < TPW; ++u) q[u] = (wave + SKQ_WAVES * u) * 16 + qc;
    const bool odd = THREE && (it & 1);
    char* const kimg = odd ? slot2 : slot0;                       // row + transposed reads
    char* const vimg = odd ? slot0 : slot2;                       // two slots: img1 (V follows dO there)
    const bool more = bh + (int)gridDim.x < nbh;
    const int b = bh / p.H, h = bh - b * p.H;
    const bf16_t* qbase = p.qkv + (long long)b * N * ldq + h * DH;
    const bf16_t* dobase = p.dout + (long long)b * N * ldo + h * DH;
    const float* scal = p.scal + (long long)bh * 7 * N;
    bool q_ok[TPW];
    float av[TPW][5];               // a0 = 1, a1 .. a4 of the lane's query, per tile
    float lse2[TPW];
    bf16x8_t qf[TPW][2];            // Q fragments (P0 = softmax(Q K^T))
    bf16x8_t dof[TPW][2];           // dO fragments (G = dO V^T): requested here, used after the dV phase
    float sb[3] = {0.f, 0.f, 0.f};  // b1 .. b3 of column tid (SKQ_THREADS >= NP)
    static_assert(SKQ_THREADS >= NP, "one column per thread");
    if (tid < N) {
#pragma unroll
        for (int t = 0; t < 3; ++t) sb[t] = scal[(2 * t + 1) * N + tid];
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        q_ok[u] = active[u] && q[u] < N;
        const int qr = q[u] < N ? q[u] : N - 1;
        av[u][0] = q_ok[u] ? 1.f : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) av[u][t + 1] = q_ok[u] ? scal[(2 * t) * N + qr] : 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[u][ks] = qf_n[u][ks];
            dof[u][ks] = load_frag_global(dobase + (long long)qr * ldo + ks * 32 + g * 8);
        }
        lse2[u] = lse2_n[u];
    }
    dma_image<NP, SKQ_WAVES, true>(img1, dobase, ldo, N, wave, lane);      // dO / Q are only read transposed: the V-image swizzle (conflict-free ds_read_b64_tr_b16)
    if (THREE) dma_image<NP, SKQ_WAVES>(vimg, qbase + 2 * p.H * DH, ldq, N, wave, lane);
    // P0 of key tile kt (zero for padded keys / queries): for all tile slots of the wave, or for one
    auto p0_from = [&](const f32x4_t& st, int kt, int u, f32x4_t& out) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pv = __builtin_amdgcn_exp2f(fmaf(st[e], sc, -lse2[u]));
            // padded keys (>= N) can only sit in the last two key tiles (NP - N < 32): a per-element mask on every tile
            // costs a hoisted SGPR-pair each (spilled to VGPR lanes and read back with v_readlane + s_nop in every pass)
            if (kt >= NT - 2) pv = (kt * 16 + 4 * g + e < N) ? pv : 0.f;
            out[e] = pv;
        }
    };
    // P0 is computed ONCE (in the dV phase) and kept in registers as fp16 (values in [0, 1]: 11 significant bits, everything
    // below 6e-8 -- exp2 of less than -24 -- reads back as zero); the walk reads it through v_fma_mix_f32 (fp16 operand, fp32
    // arithmetic).  Recomputing it instead (MFMA + v_exp_f32, which issues at a quarter of the VALU rate) in each of the 9
    // later passes was 40 % of the kernel's VALU cycles.
    u32x2_t P0h[TPW][NT];           // f16x4_t bit patterns
    auto p0h = [&](int u, int kt) {
        // opaque per use: hipcc would otherwise convert every tile back ONCE and keep the fp32 copy live (spilled)
        asm volatile("" : "+v"(P0h[u][kt]));
        return __builtin_bit_cast(f16x4_t, P0h[u][kt]);
    };

    // key-owner side: this wave's key tiles are wave + SKQ_WAVES s; the chunk rows are queries u CH .. of slot u
    int chunk_rd = (4 * g + ((lane & 15) >> 2)) * RS + (lane & 3) * 8;       // + (32 ks + {0, 16}) RS + 32 kt
    int chunk_wr = (wave * 16 + qc) * RS + 8 * g;                            // + 32 kt: 4 consecutive keys of the lane's query
    // The per-lane indices and LDS offsets are cheap to derive and expensive to keep: between the phases the lane id goes
    // through an empty asm, so everything derived from it is recomputed where it is used instead of living (or being spilled)
    // through the walk, which needs the registers for G and P0.
    auto relane = [&]() {
        asm volatile("" : "+v"(lane));
        g = lane >> 4;
        qc = lane & 15;
        chunk_rd = (4 * g + ((lane & 15) >> 2)) * RS + (lane & 3) * 8;
        chunk_wr = (wave * 16 + qc) * RS + 8 * g;
#pragma unroll
        for (int u = 0; u < TPW; ++u) q[u] = (wave + SKQ_WAVES * u) * 16 + qc;
    };
    auto key_owner_products = [&](f32x4_t (&acc)[TPW][4], const char* img, int u) {      // acc[s][dt] += img^T[d, q] . chunk[q, key]
        const int rows = NP - u * CH < CH ? NP - u * CH : CH;                // multiple of 32 (NP is)
        // per 32-query step: the four image fragments (shared by the wave's key tiles) and one chunk fragment per key tile,
        // read one step ahead of the MFMAs that consume them
        bf16x8_t imf[2][4], cf[2][TPW];
        auto fetch = [&](int ks, int buf) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) imf[buf][dt] = tr_frag_vimg(img, u * CH + ks * 32, dt, lane);
#pragma unroll
            for (int s = 0; s < TPW; ++s) {
                const int ktw = wave + SKQ_WAVES * s;
                if (ktw < NT) {
                    const char* a0 = chunk + chunk_rd + ks * 32 * RS + ktw * 32;
                    cf[buf][s] = cat4(lds_read_tr16_b64(a0), lds_read_tr16_b64(a0 + 16 * RS));
                }
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int ks = 0; ks < CH / 32; ++ks) {
            if (ks * 32 < rows) {
                SK_KEEP_ORDER();
                if ((ks + 1) * 32 < rows) fetch(ks + 1, (ks + 1) & 1);
#pragma unroll
                for (int s = 0; s < TPW; ++s) {
                    if (wave + SKQ_WAVES * s < NT) {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) acc[s][dt] = mfma16(imf[ks & 1][dt], cf[ks & 1][s], acc[s][dt]);
                    }
                }
            }
        }
    };
    auto store_key_rows = [&](const f32x4_t (&acc)[TPW][4], int which /* 1: dK, 2: dV */) {
#pragma unroll
        for (int s = 0; s < TPW; ++s) {
            const int key = (wave + SKQ_WAVES * s) * 16 + qc;
            if (key < N) {
                bf16_t* dst = p.dqkv + ((long long)b * N + key) * ldq + which * p.H * DH + h * DH + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) store_bf16x4(dst + dt * 16, acc[s][dt]);
            }
        }
    };

    // ---- P0 of both tile slots, once: the K fragments of a key tile serve both slots and are read one tile ahead
    {
        bf16x8_t kr[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kr[0][ks] = row_frag_img(kimg, 0, ks, lane);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            SK_KEEP_ORDER();
            if (kt + 1 < NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) kr[(kt + 1) & 1][ks] = row_frag_img(kimg, (kt + 1) * 16, ks, lane);
            }
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                f32x4_t st = {0.f, 0.f, 0.f, 0.f}, p0u;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) st = mfma16(kr[kt & 1][ks], qf[u][ks], st);
                p0_from(st, kt, u, p0u);              // zero for the queries of a tile slot beyond the head (lse2 = inf)
                P0h[u][kt] = __builtin_bit_cast(u32x2_t, __builtin_convertvector(p0u, f16x4_t));
                asm volatile("" : "+v"(P0h[u][kt]));                  // convert here, not at the first use (the fp32 tile would stay live)
            }
        }
    }
    SK_STAMP();           // 1: requests + P0
    __builtin_amdgcn_s_waitcnt(0x0F70);        // dO / V images and the per-lane values of this head (requested before P0) have landed
    asm volatile("" ::: "memory");
    if (tid < NP) {
        bv[tid] = tid < N ? 1.0f : 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) bv[(t + 1) * NP + tid] = sb[t];
    }
    __syncthreads();
    // ---- dV = P7^T dO,  P7 = a4 P0 b3
    {
    f32x4_t dv[TPW][4];
#pragma unroll
    for (int s = 0; s < TPW; ++s)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dv[s][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        if (u * CH < NP) {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                SK_KEEP_ORDER();
                const f16x4_t ph = p0h(u, kt);
                const f32x4_t b3 = *reinterpret_cast<const f32x4_t*>(bv + 3 * NP + kt * 16 + 4 * g);
                const float a4 = av[u][4];                                    // 0 for inactive tiles / padded queries
                const f32x4_t ab = b3 * f32x4_t{a4, a4, a4, a4};
                const u32x2_t pk = {pack_bf16x2((float)ph[0] * ab[0], (float)ph[1] * ab[1]), pack_bf16x2((float)ph[2] * ab[2], (float)ph[3] * ab[3])};
                *reinterpret_cast<u32x2_t*>(chunk + chunk_wr + kt * 32) = pk;
            }
            __syncthreads();
            key_owner_products(dv, img1, u);
            __syncthreads();
        }
    }
    store_key_rows(dv, 2);
    }
    SK_STAMP();           // 2: dV phase

    // ---- G = dP7^T = V dO^T.  Every wave is past its dO reads: the dO slot takes Q now (three slots: lands during the walk)
    // or V (two slots: Q follows at the end)
    if (THREE) {
        dma_image<NP, SKQ_WAVES, true>(img1, qbase, ldq, N, wave, lane);
    } else {
        dma_image<NP, SKQ_WAVES>(img1, qbase + 2 * p.H * DH, ldq, N, wave, lane);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        __syncthreads();
    }
    relane();
    f32x4_t G[TPW][NT];             // fp32, walked back through the normalisations in place
    {
        bf16x8_t vr[2][2];              // V fragments of a key tile: both tile slots, one tile ahead
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) vr[0][ks] = row_frag_img(vimg, 0, ks, lane);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            SK_KEEP_ORDER();
            if (kt + 1 < NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) vr[(kt + 1) & 1][ks] = row_frag_img(vimg, (kt + 1) * 16, ks, lane);
            }
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                G[u][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) G[u][kt] = mfma16(vr[kt & 1][ks], dof[u][ks], G[u][kt]);
            }
        }
    }

    SK_STAMP();           // 3: V image + G
    relane();
    const bool hi8 = (qc & 8) != 0, hi4 = (qc & 4) != 0;
    float* colw = colpart + wave * NP + (hi8 ? 8 * NT : 0) + 4 * g + (hi4 ? 2 : 0);
    const bool col_writer = (lane & 3) == 0;

    // ---- walk the normalisations backwards: steps 7 (row) 6 (col) 5 (row) 4 (col) 3 (row) 2 (col) 1 (row)
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
                SK_KEEP_ORDER();
                const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(bt + kt * 16 + 4 * g);
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const f16x4_t ph = p0h(u, kt);
                    const f32x4_t gb = G[u][kt] * b4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) rho[u] = __builtin_fmaf((float)ph[e], gb[e], rho[u]);
                    asm volatile("" : "+v"(rho[u]));
                }
            }
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                float r = rho[u];
                r += __shfl_xor(r, 16, 64);
                r += __shfl_xor(r, 32, 64);
                r *= av[u][t + 1];
                const float f = ratio_or_zero(av[u][t + 1], av[u][t]);
                const float nrf = -r * f;
                const f32x4_t f4 = {f, f, f, f}, c4 = {nrf, nrf, nrf, nrf};
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) G[u][kt] = G[u][kt] * f4 + c4;              // (G - r) f
            }
        }
        if (t == 0) break;
        {   // column step with beta = b_t, beta_prev = b_{t-1}, alpha = a_t: the tiles' contributions are added before the
            // reduction over the query lanes
            auto col_terms = [&](int kt) {
                f32x4_t x = {0.f, 0.f, 0.f, 0.f};                            // av = 0 for inactive tiles
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const f16x4_t ph = p0h(u, kt);
                    const float au = av[u][t];
                    const f32x4_t a4 = {au, au, au, au};
                    const f32x4_t ga = G[u][kt] * a4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = __builtin_fmaf((float)ph[e], ga[e], x[e]);
                }
                return x;
            };
#pragma unroll
            for (int kt = 0; kt < NT / 2; ++kt) {
                SK_KEEP_ORDER();
                const f32x4_t xa = col_terms(kt), xb = col_terms(kt + NT / 2);
                col_sums_pair(xa, xb, colw + kt * 16, hi8, hi4, col_writer);
            }
            __syncthreads();
            if (THREE && t == 3 && more)      // every wave has built its G: the V slot takes the next head's K image
                dma_image<NP, SKQ_WAVES>(vimg, qbase_of(bh + gridDim.x) + p.H * DH, ldq, N, wave, lane);
            for (int j = tid; j < NP; j += SKQ_THREADS) {
                float c = 0.f;
#pragma unroll
                for (int w = 0; w < SKQ_WAVES; ++w) c += colpart[w * NP + j];
                // the rescale that follows is G <- (G - beta_j colsum) * (beta_t / beta_{t-1}) = G ratio + kap: both factors once
                // per column here instead of once per element in every wave
                const float ratio = ratio_or_zero(bv[t * NP + j], bv[(t - 1) * NP + j]);
                kap[j] = -c * bv[t * NP + j] * ratio;
                colpart[j] = ratio;                                          // row 0 of colpart is free again: all partials were read
            }
            __syncthreads();
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                SK_KEEP_ORDER();
                const f32x4_t k4 = *reinterpret_cast<const f32x4_t*>(kap + kt * 16 + 4 * g);
                const f32x4_t r4 = *reinterpret_cast<const f32x4_t*>(colpart + kt * 16 + 4 * g);
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    G[u][kt] = G[u][kt] * r4 + k4;
                    asm volatile("" : "+v"(G[u][kt]));       // finished HERE: hipcc otherwise issues all 2 NT vector loads first (112 registers)
                }
            }
            __syncthreads();      // kap / colpart are rewritten by the next column step
        }
        SK_STAMP();       // 4, 5, 6: row + column step
    }
    SK_STAMP();           // 7: last row step
    // ---- softmax backward: dS = P0 (G - sum_j G P0) * scale
    {
        float sd[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) sd[u] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const f16x4_t ph = p0h(u, kt);
#pragma unroll
                for (int e = 0; e < 4; ++e) sd[u] = __builtin_fmaf((float)ph[e], G[u][kt][e], sd[u]);
            }
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            sd[u] += __shfl_xor(sd[u], 16, 64);
            sd[u] += __shfl_xor(sd[u], 32, 64);
            const float ns = -sd[u] * p.scale;
            const f32x4_t s4 = {p.scale, p.scale, p.scale, p.scale}, n4 = {ns, ns, ns, ns};
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const f16x4_t ph = p0h(u, kt);
                const f32x4_t d = G[u][kt] * s4 + n4;
#pragma unroll
                for (int e = 0; e < 4; ++e) G[u][kt][e] = (float)ph[e] * d[e];
            }
        }
    }

    SK_STAMP();           // 8: softmax backward
    relane();
    // ---- dQ = dS K: the K fragments of a 32-key step serve both tile slots and are read one step ahead
    {
        f32x4_t dq[TPW][4];
#pragma unroll
        for (int u = 0; u < TPW; ++u)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[u][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        bf16x8_t kf[2][4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) kf[0][dt] = tr_frag_img(kimg, 0, dt, lane);
#pragma unroll
        for (int kk = 0; kk < NP / 32; ++kk) {
            SK_KEEP_ORDER();
            if (kk + 1 < NP / 32) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) kf[(kk + 1) & 1][dt] = tr_frag_img(kimg, (kk + 1) * 32, dt, lane);
            }
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const bf16x8_t dsf = pack_frag(G[u][2 * kk], G[u][2 * kk + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[u][dt] = mfma16(kf[kk & 1][dt], dsf, dq[u][dt]);
            }
        }
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            if (active[u] && q[u] < N) {
                bf16_t* dst = p.dqkv + ((long long)b * N + q[u]) * ldq + h * DH + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) store_bf16x4(dst + dt * 16, dq[u][dt]);
            }
        }
    }

    SK_STAMP();           // 9: dQ
    // ---- dK = dS^T Q  (two slots: Q replaces V now; the last V read was before the walk's barriers)
    if (!THREE) dma_image<NP, SKQ_WAVES, true>(img1, qbase, ldq, N, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);        // this wave's part of the Q image; the barrier below publishes all parts
    asm volatile("" ::: "memory");
    f32x4_t dk[TPW][4];
#pragma unroll
    for (int s = 0; s < TPW; ++s)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[s][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        if (u * CH < NP) {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const f32x4_t d = active[u] ? G[u][kt] : f32x4_t{0.f, 0.f, 0.f, 0.f};      // rows of absent tiles are read against zero Q rows: keep them finite
                const u32x2_t pk = {pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3])};
                *reinterpret_cast<u32x2_t*>(chunk + chunk_wr + kt * 32) = pk;
            }
            if (u == 0) request_q_lse(more ? bh + (int)gridDim.x : bh);       // unconditional: qf_n / lse2_n are dead through the rest of the loop body
            __syncthreads();          // also orders the Q image stores (u == 0) before the transposed reads
            key_owner_products(dk, img1, u);
            __syncthreads();
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);        // the next head's Q fragments / LSE (its K image: waited for above, published by the barriers) --
    asm volatile("" ::: "memory");             // before the dK stores, whose latency then overlaps the next head's start
    store_key_rows(dk, 1);
    SK_STAMP();           // 10: dK phase
    }
#undef SK_STAMP
}

int np_of(int N) { return (N + 31) / 32 * 32; }
int sk_cus() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
        return v > 0 ? v : 256;
    }();
    return n;
}

template <int NP>
int launch_sk_fwd(const SinkParams& p, hipStream_t s) {
    constexpr int lds = SkFwdLds<NP>::BYTES;
    static int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_fwd_kernel<NP>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != 0) return attr;
    const int heads = p.B * p.H;
    const int grid = SkFwdLds<NP>::TWO && heads > sk_cus() ? sk_cus() : heads;
    hipLaunchKernelGGL((sinkhorn_fwd_kernel<NP>), dim3(grid), dim3(SK_THREADS), lds, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}

template <int NP>
int launch_sk_bwd(const SinkParams& p, hipStream_t s) {
    constexpr int TPW = SK_TPW;
    constexpr int lds = SkBwdLds<NP, TPW>::BYTES;
    static int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_bwd_kernel<NP, TPW>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != 0) return attr;
    // three image slots: one workgroup per CU walks the heads (the next head's operands arrive during the current one)
    const int heads = p.B * p.H;
    const int grid = SkBwdLds<NP, TPW>::THREE && heads > sk_cus() ? sk_cus() : heads;
    hipLaunchKernelGGL((sinkhorn_bwd_kernel<NP, TPW>), dim3(grid), dim3(1024 / TPW), lds, s, p);
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

extern "C" int nrv_attn_sinkhorn_bwd(const void* qkv_bf16, const void* dout_bf16, const float* lse, const float* scalings,
                                     void* dqkv_bf16, int B, int N, int H, int dh, float scale, void* stream) {
    if (!qkv_bf16 || !dout_bf16 || !lse || !scalings || !dqkv_bf16) return NRV_ERR_NULL;
    if (int e = sk_check(B, N, H, dh)) return e;
    if (!nrv_aligned16(qkv_bf16) || !nrv_aligned16(dout_bf16) || !nrv_aligned16(dqkv_bf16)) return NRV_ERR_ALIGN;
    SinkParams p{};
    p.qkv = static_cast<const bf16_t*>(qkv_bf16);
    p.dout = static_cast<const bf16_t*>(dout_bf16);
    p.dqkv = static_cast<bf16_t*>(dqkv_bf16);
    p.lse = const_cast<float*>(lse); p.scal = const_cast<float*>(scalings);
    p.B = B; p.N = N; p.H = H; p.scale = scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    NRV_SK_DISPATCH(N, launch_sk_bwd<NPV>(p, s));
}
