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
// One workgroup (4 waves) per 64 x 64 tile of C, K-steps of 32: both operands are staged into [64][32] bf16 row images (k contiguous,
// rows padded to 80 bytes; 16-byte loads along the operand's unit stride where its addresses allow, else element by element), wave w owns rows 16 w .. 16 w + 15 of the tile and all 64 columns
// (4 MFMA 16x16x32 accumulators).  The vector path keeps its per-item addresses across the K loop and issues the next K-step's loads before
// the current MFMAs; it streams the fp32 [N, N] operands at ~2.5 TB/s (profiles/r04_robust_composed_path_timing.txt).
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
    int a_vec, b_vec;                    // 16-byte vector loads along the operand's unit stride (bg_vec_ok)
    int c_vec;                           // C rows can take dword-aligned vector stores (unit column stride, base / strides aligned)
    float alpha;
};

__device__ __forceinline__ float bg_load(const BgOperand& o, long long off) {
    return o.f32 ? static_cast<const float*>(o.p)[off] : bf16_to_f32(static_cast<const bf16_t*>(o.p)[off]);
}

// 16-byte vectors that are only dword-aligned in memory (rows of an fp32 [N, N] matrix with odd N; head slices at any even element)
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x2_u __attribute__((ext_vector_type(2), aligned(4)));

// Stage one [64 rows][32 k] operand tile into its bf16 row image (element (r, k) at img + r * BG_RS + 2 k).  `rs_row` / `rs_k` are the
// element strides along the tile's rows / along k.  `vec` (host): the operand's unit-stride direction can be read 16 bytes at a
// time (8 bf16 / 4 fp32 per load: one or two loads per thread and K-step instead of eight); otherwise element by element.
__device__ __forceinline__ void bg_stage(const BgOperand& o, char* img, long long base, long long rs_row, long long rs_k,
                                         int row0, int rows, int k0, int K, int vec, int tid) {
    if (vec == 0) {
        const bool kfast = rs_k == 1 || rs_row != 1;
#pragma unroll
        for (int i = 0; i < BG_T * BG_K / BG_THREADS; ++i) {
            const int idx = i * BG_THREADS + tid;
            const int r = kfast ? idx / BG_K : idx % BG_T, k = kfast ? idx % BG_K : idx / BG_T;
            float v = 0.f;
            if (row0 + r < rows && k0 + k < K) v = bg_load(o, base + (long long)(row0 + r) * rs_row + (long long)(k0 + k) * rs_k);
            *reinterpret_cast<bf16_t*>(img + r * BG_RS + k * 2) = f32_to_bf16(v);
        }
        return;
    }
    const int VEC = o.f32 ? 4 : 8;
    const bool kfast = rs_k == 1;                                   // else rs_row == 1 (host)
    const int items = BG_T * BG_K / VEC;                            // 256 (bf16) or 512 (fp32)
    for (int it = tid; it < items; it += BG_THREADS) {
        int r, k;
        if (kfast) { const int per = BG_K / VEC; r = it / per; k = (it - r * per) * VEC; }
        else { const int per = BG_T / VEC; k = it / per; r = (it - k * per) * VEC; }
        const long long off = base + (long long)(row0 + r) * rs_row + (long long)(k0 + k) * rs_k;
        float v[8];
        const bool full = kfast ? (row0 + r < rows && k0 + k + VEC <= K) : (row0 + r + VEC <= rows && k0 + k < K);
        if (full) {
            if (o.f32) {
                const f32x4_u t = *reinterpret_cast<const f32x4_u*>(static_cast<const float*>(o.p) + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = t[e];
            } else {
                const u32x4_u t = *reinterpret_cast<const u32x4_u*>(static_cast<const bf16_t*>(o.p) + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] = bf16lo_to_f32(t[e]); v[2 * e + 1] = bf16hi_to_f32(t[e]); }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = 0.f;
                if (e < VEC) {
                    const int rr = kfast ? r : r + e, kk = kfast ? k + e : k;
                    if (row0 + rr < rows && k0 + kk < K) v[e] = bg_load(o, off + (long long)e);       // unit stride along the vector
                }
            }
        }
        if (kfast) {                                                // VEC consecutive k of one row: one 8- or 16-byte LDS store
            if (o.f32) {
                *reinterpret_cast<u32x2_t*>(img + r * BG_RS + k * 2) = u32x2_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            } else {
                *reinterpret_cast<u32x4_t*>(img + r * BG_RS + k * 2) =
                    u32x4_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            }
        } else {                                                    // VEC consecutive rows of one k
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < VEC) *reinterpret_cast<bf16_t*>(img + (r + e) * BG_RS + k * 2) = f32_to_bf16(v[e]);
        }
    }
}

// The vector-load path with everything that does not change from K-step to K-step taken out of the loop: a thread's items of an
// operand tile (one 16-byte load each: 8 bf16 or 4 fp32 along the operand's unit stride) are fixed by (tid, item), so their
// tile coordinates, LDS offsets and global offsets at k = 0 are computed once; a K-step adds a constant to the offset and tests
// k against K.  The loads of step s + 1 are issued before the MFMAs of step s (register double buffer).
struct BgItem {
    long long off;                  // element offset at k0 = 0
    int r, k;                       // tile coordinates of the item's first element
    int rows_left;                  // rows of the matrix from the item's first row on (<= 0: none)
};
template <bool F32>
__device__ __forceinline__ BgItem bg_item(long long base, long long rs_row, long long rs_k, int row0, int rows, int it) {
    constexpr int VEC = F32 ? 4 : 8;
    BgItem q;
    if (rs_k == 1) { constexpr int per = BG_K / VEC; q.r = it / per; q.k = (it - q.r * per) * VEC; }
    else { constexpr int per = BG_T / VEC; q.k = it / per; q.r = (it - q.k * per) * VEC; }
    q.off = base + (long long)(row0 + q.r) * rs_row + (long long)q.k * rs_k;
    q.rows_left = rows - (row0 + q.r);
    return q;
}
// the item's VEC values of K-step k0 (zero outside the matrix)
template <bool F32>
__device__ __forceinline__ void bg_item_load(const BgOperand& o, const BgItem& q, long long rs_k, int k0, int K, float (&v)[8]) {
    constexpr int VEC = F32 ? 4 : 8;
    const bool kfast = rs_k == 1;
    const long long at = q.off + (long long)k0 * rs_k;
    const int kk = k0 + q.k;
    const bool full = kfast ? (q.rows_left > 0 && kk + VEC <= K) : (q.rows_left >= VEC && kk < K);
    if (full) {
        if (F32) {
            const f32x4_u t = *reinterpret_cast<const f32x4_u*>(static_cast<const float*>(o.p) + at);
            v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        } else {
            const u32x4_u t = *reinterpret_cast<const u32x4_u*>(static_cast<const bf16_t*>(o.p) + at);
            v[0] = bf16lo_to_f32(t[0]); v[1] = bf16hi_to_f32(t[0]); v[2] = bf16lo_to_f32(t[1]); v[3] = bf16hi_to_f32(t[1]);
            v[4] = bf16lo_to_f32(t[2]); v[5] = bf16hi_to_f32(t[2]); v[6] = bf16lo_to_f32(t[3]); v[7] = bf16hi_to_f32(t[3]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const bool ok = kfast ? (q.rows_left > 0 && kk + e < K) : (e < q.rows_left && kk < K);
            v[e] = 0.f;
            if (ok) v[e] = F32 ? static_cast<const float*>(o.p)[at + e] : bf16_to_f32(static_cast<const bf16_t*>(o.p)[at + e]);
        }
    }
}
template <bool F32>
__device__ __forceinline__ void bg_item_store(char* img, const BgItem& q, bool kfast, const float (&v)[8]) {
    constexpr int VEC = F32 ? 4 : 8;
    if (kfast) {                                                    // VEC consecutive k of one row: one 8- or 16-byte LDS store
        if (F32) {
            *reinterpret_cast<u32x2_t*>(img + q.r * BG_RS + q.k * 2) = u32x2_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        } else {
            *reinterpret_cast<u32x4_t*>(img + q.r * BG_RS + q.k * 2) =
                u32x4_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        }
    } else {                                                        // VEC consecutive rows of one k
#pragma unroll
        for (int e = 0; e < VEC; ++e) *reinterpret_cast<bf16_t*>(img + (q.r + e) * BG_RS + q.k * 2) = f32_to_bf16(v[e]);
    }
}

__device__ __forceinline__ void bg_mma_step(const char* aimg, const char* bimg, f32x4_t (&acc)[4], int wave, int lane) {
    // fragments: lane -> row (lane & 15) of its 16-row block, k = 8 (lane >> 4) .. + 7
    const bf16x8_t af = lds_read_b128(aimg + (wave * 16 + (lane & 15)) * BG_RS + (lane >> 4) * 16);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const bf16x8_t bf = lds_read_b128(bimg + (ni * 16 + (lane & 15)) * BG_RS + (lane >> 4) * 16);
        acc[ni] = mfma16(bf, af, acc[ni]);       // lane holds row m = 16 wave + (lane & 15), columns 16 ni + 4 (lane >> 4) + {0..3}
    }
}

// both operands through the vector path
template <bool AF32, bool BF32>
__device__ __forceinline__ void bg_loop_vec(const BgParams& p, char* aimg, char* bimg, long long abase, long long bbase, int m0, int n0,
                                            f32x4_t (&acc)[4], int tid, int wave, int lane) {
    constexpr int NA = AF32 ? 2 : 1, NB = BF32 ? 2 : 1;            // items per thread: 64 x 32 elements / (4 | 8 per load) / 256 threads
    const long long a_row = p.A.rs, a_k = p.A.cs, b_row = p.B.cs, b_k = p.B.rs;     // A[m][k]; B[k][n] staged as rows n
    const bool a_kfast = a_k == 1, b_kfast = b_k == 1;
    const BgItem qa0 = bg_item<AF32>(abase, a_row, a_k, m0, p.M, tid), qa1 = bg_item<AF32>(abase, a_row, a_k, m0, p.M, tid + BG_THREADS);
    const BgItem qb0 = bg_item<BF32>(bbase, b_row, b_k, n0, p.N, tid), qb1 = bg_item<BF32>(bbase, b_row, b_k, n0, p.N, tid + BG_THREADS);
    float va0[8], va1[8], vb0[8], vb1[8];
    auto load_all = [&](int k0) {
        bg_item_load<AF32>(p.A, qa0, a_k, k0, p.K, va0);
        if (NA > 1) bg_item_load<AF32>(p.A, qa1, a_k, k0, p.K, va1);
        bg_item_load<BF32>(p.B, qb0, b_k, k0, p.K, vb0);
        if (NB > 1) bg_item_load<BF32>(p.B, qb1, b_k, k0, p.K, vb1);
    };
    load_all(0);
    for (int k0 = 0; k0 < p.K; k0 += BG_K) {
        bg_item_store<AF32>(aimg, qa0, a_kfast, va0);
        if (NA > 1) bg_item_store<AF32>(aimg, qa1, a_kfast, va1);
        bg_item_store<BF32>(bimg, qb0, b_kfast, vb0);
        if (NB > 1) bg_item_store<BF32>(bimg, qb1, b_kfast, vb1);
        __syncthreads();
        if (k0 + BG_K < p.K) load_all(k0 + BG_K);                    // the next step's operands travel while this step multiplies
        bg_mma_step(aimg, bimg, acc, wave, lane);
        __syncthreads();
    }
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

    f32x4_t acc[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (p.a_vec && p.b_vec) {            // uniform
        if (p.A.f32) {
            if (p.B.f32) bg_loop_vec<true, true>(p, aimg, bimg, abase, bbase, m0, n0, acc, tid, wave, lane);
            else bg_loop_vec<true, false>(p, aimg, bimg, abase, bbase, m0, n0, acc, tid, wave, lane);
        } else {
            if (p.B.f32) bg_loop_vec<false, true>(p, aimg, bimg, abase, bbase, m0, n0, acc, tid, wave, lane);
            else bg_loop_vec<false, false>(p, aimg, bimg, abase, bbase, m0, n0, acc, tid, wave, lane);
        }
    } else {
        for (int k0 = 0; k0 < p.K; k0 += BG_K) {
            bg_stage(p.A, aimg, abase, p.A.rs, p.A.cs, m0, p.M, k0, p.K, p.a_vec, tid);       // A[m][k]: rows m, k along its columns
            bg_stage(p.B, bimg, bbase, p.B.cs, p.B.rs, n0, p.N, k0, p.K, p.b_vec, tid);       // B[k][n]: rows n of the image, k along B's rows
            __syncthreads();
            bg_mma_step(aimg, bimg, acc, wave, lane);
            __syncthreads();
        }
    }
    const int m = m0 + wave * 16 + (lane & 15);
    if (m >= p.M) return;
    const long long cbase = (long long)g1 * p.c_b1 + (long long)g2 * p.c_b2 + (long long)m * p.c_rs;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int nb = n0 + ni * 16 + 4 * (lane >> 4);
        const f32x4_t v4 = acc[ni] * p.alpha;
        if (p.c_cs == 1 && p.c_vec && nb + 4 <= p.N) {              // 4 consecutive columns: one 16-byte (fp32) or 8-byte (bf16) store
            const long long off = cbase + nb;
            if (p.c_f32) *reinterpret_cast<f32x4_u*>(static_cast<float*>(p.C) + off) = f32x4_u{v4[0], v4[1], v4[2], v4[3]};
            else *reinterpret_cast<u32x2_u*>(static_cast<bf16_t*>(p.C) + off) = u32x2_u{pack_bf16x2(v4[0], v4[1]), pack_bf16x2(v4[2], v4[3])};
            continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = nb + e;
            if (n < p.N) {
                const long long off = cbase + (long long)n * p.c_cs;
                if (p.c_f32) static_cast<float*>(p.C)[off] = v4[e];
                else static_cast<bf16_t*>(p.C)[off] = f32_to_bf16(v4[e]);
            }
        }
    }
}

// can the operand's unit-stride direction be read as 16-byte vectors?  One stride is 1; a bf16 vector needs a dword-aligned address:
// base pointer and every other stride even (K-steps and tile origins are multiples of 8)
int bg_vec_ok(const void* ptr, int f32, int64_t s_row, int64_t s_k, int64_t b1, int64_t b2) {
    if (s_row != 1 && s_k != 1) return 0;
    if (f32) return (reinterpret_cast<uintptr_t>(ptr) & 3u) == 0;
    const int64_t other = s_row == 1 ? s_k : s_row;
    if (s_row == 1 && s_k == 1) return 0;
    return (reinterpret_cast<uintptr_t>(ptr) & 3u) == 0 && !(other & 1) && !(b1 & 1) && !(b2 & 1);
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
    p.a_vec = bg_vec_ok(A, p.A.f32, a_rs, a_cs, a_b1, a_b2);
    p.b_vec = bg_vec_ok(B, p.B.f32, b_cs, b_rs, b_b1, b_b2);
    // 4 consecutive columns of a C row as one store: dword-aligned addresses (bf16: even row / batch strides; tile origins are multiples of 4)
    p.c_vec = c_cs == 1 && (reinterpret_cast<uintptr_t>(C) & 3u) == 0 && (p.c_f32 || (!(c_rs & 1) && !(c_b1 & 1) && !(c_b2 & 1)));
    hipLaunchKernelGGL(bgemm_kernel, dim3((unsigned)blocks), dim3(BG_THREADS), 0, static_cast<hipStream_t>(stream), p);
    NRV_CHECK_LAUNCH();
    return 0;
}
