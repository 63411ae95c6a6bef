// Batched small GEMM with arbitrary strides for gfx950: C[g1,g2] = alpha * A[g1,g2] . B[g1,g2], bf16 MFMA, fp32 accumulate.
//
// Used by the COMPOSED robust=True attention for the shapes the fused Sinkhorn kernels do not hold on chip (N > 256 or head
// dim != 64: vit_h_14(robust=True), vit.py:512-519; robust models at 384 px, vit.py:522-603; SimpleViT(dim_head != 64,
// robust=True), simple_vit.py:56-57,101-114).  There the reference's own structure is followed -- scores, SinkhornAttention on
// the materialised [B,H,N,N] tensor (utils.py:1025-1037 = nrv_sinkhorn_fwd/bwd), P.V -- and the matrix products around it are
// this kernel:    S = scale q k^T,  O = P7 v,  dP7 = dO v^T,  dV = P7^T dO,  dQ = scale dS k,  dK = scale dS^T q.
// Every operand is addressed as  base + g1 * b1 + g2 * b2 + row * rs + col * cs  (elements), so head slices of the packed
// [B*N, 3*H*dh] projection, transposes and fp32 [B,H,N,N] matrices need no copies.  Operands are rounded to bf16 when they are
// staged into LDS (P7 and dS enter their products in bf16 exactly as in the fused kernels, DESIGN.md "Numerics").
//
// One workgroup (4 waves) per 64 x 64 tile of C, K-steps of 32: both operands are staged element-wise into [64][32] bf16 row
// images (k contiguous, rows padded to 80 bytes), wave w owns rows 16 w .. 16 w + 15 of the tile and all 64 columns
// (4 MFMA 16x16x32 accumulators).  Not tuned: the HBM-bound Sinkhorn sweeps over the materialised matrices dominate this path.
#include "nrv_common.hpp"

namespace {

constexpr int BG_THREADS = 256;
constexpr int BG_T = 64;                 // tile rows / columns
constexpr int BG_K = 32;                 // K-step
constexpr int BG_RS = 80;                // bytes per image row (64 data + 16 pad: the 16 rows of a fragment read spread over the banks)

struct BgOperand {
    const void* p;
    long long rs, cs, b1, b2;            // element strides: row, column, batch level 1, batch level 2
    int f32;                             // 1: fp32, 0: bf16
};

struct BgParams {
    BgOperand A, B;                      // A [M, K], B [K, N]
    void* C;
    long long c_rs, c_cs, c_b1, c_b2;
    int c_f32;
    int G2, M, N, K;
    int tiles_m, tiles_n;
    float alpha;
};

__device__ __forceinline__ float bg_load(const BgOperand& o, long long off) {
    return o.f32 ? static_cast<const float*>(o.p)[off] : bf16_to_f32(static_cast<const bf16_t*>(o.p)[off]);
}

__global__ __launch_bounds__(BG_THREADS) void bgemm_kernel(const BgParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BG_T * BG_RS];
    char* aimg = smem;                   // [64 m][32 k]
    char* bimg = smem + BG_T * BG_RS;    // [64 n][32 k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned bid = blockIdx.x;
    const int tn = bid % p.tiles_n; bid /= p.tiles_n;
    const int tm = bid % p.tiles_m; bid /= p.tiles_m;
    const int g2 = bid % p.G2, g1 = bid / p.G2;
    const int m0 = tm * BG_T, n0 = tn * BG_T;
    const long long abase = (long long)g1 * p.A.b1 + (long long)g2 * p.A.b2;
    const long long bbase = (long long)g1 * p.B.b1 + (long long)g2 * p.B.b2;
    // thread -> (row, k) of a [64][32] image: the fast index follows the operand's unit stride so that a wave's loads coalesce
    const bool a_kfast = p.A.cs == 1 || p.A.rs != 1;          // A[m][k]: k is its column
    const bool b_kfast = p.B.rs == 1 && p.B.cs != 1;          // B[k][n]: k is its row

    f32x4_t acc[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < p.K; k0 += BG_K) {
#pragma unroll
        for (int i = 0; i < BG_T * BG_K / BG_THREADS; ++i) {
            const int idx = i * BG_THREADS + tid;
            {
                const int r = a_kfast ? idx / BG_K : idx % BG_T, k = a_kfast ? idx % BG_K : idx / BG_T;
                float v = 0.f;
                if (m0 + r < p.M && k0 + k < p.K) v = bg_load(p.A, abase + (long long)(m0 + r) * p.A.rs + (long long)(k0 + k) * p.A.cs);
                *reinterpret_cast<bf16_t*>(aimg + r * BG_RS + k * 2) = f32_to_bf16(v);
            }
            {
                const int r = b_kfast ? idx / BG_K : idx % BG_T, k = b_kfast ? idx % BG_K : idx / BG_T;
                float v = 0.f;
                if (n0 + r < p.N && k0 + k < p.K) v = bg_load(p.B, bbase + (long long)(k0 + k) * p.B.rs + (long long)(n0 + r) * p.B.cs);
                *reinterpret_cast<bf16_t*>(bimg + r * BG_RS + k * 2) = f32_to_bf16(v);
            }
        }
        __syncthreads();
        // fragments: lane -> row (lane & 15) of its 16-row block, k = 8 (lane >> 4) .. + 7
        const bf16x8_t af = lds_read_b128(aimg + (wave * 16 + (lane & 15)) * BG_RS + (lane >> 4) * 16);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const bf16x8_t bf = lds_read_b128(bimg + (ni * 16 + (lane & 15)) * BG_RS + (lane >> 4) * 16);
            acc[ni] = mfma16(bf, af, acc[ni]);       // lane holds row m = 16 wave + (lane & 15), columns 16 ni + 4 (lane >> 4) + {0..3}
        }
        __syncthreads();
    }
    const int m = m0 + wave * 16 + (lane & 15);
    if (m >= p.M) return;
    const long long cbase = (long long)g1 * p.c_b1 + (long long)g2 * p.c_b2 + (long long)m * p.c_rs;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + ni * 16 + 4 * (lane >> 4) + e;
            if (n < p.N) {
                const float v = acc[ni][e] * p.alpha;
                const long long off = cbase + (long long)n * p.c_cs;
                if (p.c_f32) static_cast<float*>(p.C)[off] = v;
                else static_cast<bf16_t*>(p.C)[off] = f32_to_bf16(v);
            }
        }
    }
}

}  // namespace

extern "C" int nrv_bgemm(const void* A, int a_dtype, int64_t a_rs, int64_t a_cs, int64_t a_b1, int64_t a_b2,
                         const void* B, int b_dtype, int64_t b_rs, int64_t b_cs, int64_t b_b1, int64_t b_b2,
                         void* C, int c_dtype, int64_t c_rs, int64_t c_cs, int64_t c_b1, int64_t c_b2,
                         int G1, int G2, int M, int N, int K, float alpha, void* stream) {
    if (!A || !B || !C) return NRV_ERR_NULL;
    if (G1 <= 0 || G2 <= 0 || M <= 0 || N <= 0 || K <= 0) return NRV_ERR_SHAPE;
    for (int d : {a_dtype, b_dtype, c_dtype})
        if (d != NRV_F32 && d != NRV_BF16) return NRV_ERR_DTYPE;
    const long long tiles_m = nrv_cdiv(M, BG_T), tiles_n = nrv_cdiv(N, BG_T);
    const long long blocks = tiles_m * tiles_n * (long long)G1 * G2;
    if (blocks > 0x7fffffffll) return NRV_ERR_SHAPE;
    BgParams p;
    p.A = BgOperand{A, a_rs, a_cs, a_b1, a_b2, a_dtype == NRV_F32};
    p.B = BgOperand{B, b_rs, b_cs, b_b1, b_b2, b_dtype == NRV_F32};
    p.C = C; p.c_rs = c_rs; p.c_cs = c_cs; p.c_b1 = c_b1; p.c_b2 = c_b2; p.c_f32 = c_dtype == NRV_F32;
    p.G2 = G2; p.M = M; p.N = N; p.K = K;
    p.tiles_m = (int)tiles_m; p.tiles_n = (int)tiles_n;
    p.alpha = alpha;
    hipLaunchKernelGGL(bgemm_kernel, dim3((unsigned)blocks), dim3(BG_THREADS), 0, static_cast<hipStream_t>(stream), p);
    NRV_CHECK_LAUNCH();
    return 0;
}
