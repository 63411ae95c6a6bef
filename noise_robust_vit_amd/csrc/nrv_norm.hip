// LayerNorm forward/backward and column reductions for gfx950 -- HBM-bound kernels.
//
// LPR lanes own one token row at a time -- a whole wave (LPR = 64) for rows of more than 512 elements, HALF a wave (LPR = 32,
// two rows per wave) below: at D = 384 (ViT-S) a 64-lane row leaves half of the second chunk's lanes idle and moved 4.3 / 4.6
// TB/s where D = 768 moves 5.2 / 6.1.  A lane holds the 4-element chunks l, l + LPR, ... of its row in registers (16-byte
// loads for fp32 rows, 8-byte for bf16), so a row is read exactly once; mean / variance / the two backward dot products are
// shuffles inside the row's lane group.  gamma/beta gradients
// and bias gradients are deterministic two-pass column reductions (per-workgroup partial rows, then
// a small finalize kernel) -- no float atomics, bitwise reproducible.
//
// Reference call sites replaced: nn.LayerNorm at simple_vit.py:38,54,65 and vit.py:104,115,167
// (eps 1e-5 / 1e-6), plus the bias-gradient sums of nn.Linear's backward.
#include "nrv_common.hpp"

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_WAVES = LN_THREADS / 64;
constexpr int LN_FWD_BLOCKS = 4096;      // measured on [50432 x 768] fp32: 512 -> 55 us, 1024 -> 43, 2048 -> 45, 4096 -> 39, 8192 -> 39, 12608 -> 42
constexpr int LN_BWD_BLOCKS = 1024;      // measured on [50432 x 768]: 512 -> 0.154 ms, 1024 -> 0.110 ms, 2048 -> 0.137 ms
constexpr int COLSUM_ROWCHUNKS = 128;

template <bool F32>
__device__ __forceinline__ f32x4_t load4(const void* base, long long idx) {
    if (F32) {
        return *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(base) + idx);
    } else {
        const u32x2_t a = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const bf16_t*>(base) + idx);
        return f32x4_t{bf16lo_to_f32(a[0]), bf16hi_to_f32(a[0]), bf16lo_to_f32(a[1]), bf16hi_to_f32(a[1])};
    }
}
// the same, non-temporal: streams read for the LAST time (the backward's x, dy and residual-gradient reads) do not displace
// the weights and operand panels the neighbouring GEMMs keep in L2 / the Infinity Cache
template <bool F32>
__device__ __forceinline__ f32x4_t load4_nt(const void* base, long long idx) {
    if (F32) {
        return __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(base) + idx));
    } else {
        const u32x2_t a = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(reinterpret_cast<const bf16_t*>(base) + idx));
        return f32x4_t{bf16lo_to_f32(a[0]), bf16hi_to_f32(a[0]), bf16lo_to_f32(a[1]), bf16hi_to_f32(a[1])};
    }
}
__device__ __forceinline__ void store4_bf16(bf16_t* base, long long idx, f32x4_t v) {
    u32x2_t pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *reinterpret_cast<u32x2_t*>(base + idx) = pk;
}
__device__ __forceinline__ float sum4(f32x4_t v) { return (v[0] + v[1]) + (v[2] + v[3]); }
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {           // over the LPR lanes that share a row
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool X_F32, int MAXJ, int LPR>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            long long rows, int dim, float eps) {
    constexpr int RPW = 64 / LPR;                                   // rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    const float inv_dim = 1.0f / (float)dim;
    f32x4_t g4[MAXJ], b4[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int c = (l + LPR * j) * 4;
        if (c < dim) {
            g4[j] = *reinterpret_cast<const f32x4_t*>(gamma + c);
            b4[j] = *reinterpret_cast<const f32x4_t*>(beta + c);
        } else {
            g4[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            b4[j] = g4[j];
        }
    }
    for (long long row0 = ((long long)blockIdx.x * LN_WAVES + wave) * RPW; row0 < rows; row0 += (long long)gridDim.x * LN_WAVES * RPW) {
        const long long row = row0 + sub;
        const bool ok = row < rows;                                 // the second row of the last pair may not exist
        f32x4_t v[MAXJ];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int c = (l + LPR * j) * 4;
            if (ok && c < dim) {
                v[j] = load4_nt<X_F32>(x, row * dim + c);      // the stream's next reader is the residual epilogue four kernels on: by then it is evicted anyway
                s += sum4(v[j]);
            } else {
                v[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float mu = group_sum<LPR>(s) * inv_dim;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int c = (l + LPR * j) * 4;
            if (c < dim) {
                const f32x4_t d = v[j] - mu;
                q += sum4(d * d);
            }
        }
        const float var = group_sum<LPR>(q) * inv_dim;
        const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int c = (l + LPR * j) * 4;
            if (ok && c < dim) store4_bf16(y, row * dim + c, (v[j] - mu) * rs * g4[j] + b4[j]);
        }
        if (ok && l == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// DRES: 0 = none, 1 = fp32, 2 = bf16
template <bool X_F32, int DRES, int MAXJ, int LPR>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const void* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const void* __restrict__ dres,
                                                            float* __restrict__ dx_f32, bf16_t* __restrict__ dx_bf16,
                                                            float* __restrict__ partial, long long rows, int dim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    const float inv_dim = 1.0f / (float)dim;
    f32x4_t g4[MAXJ], adg[MAXJ], adb[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int c = (l + LPR * j) * 4;
        g4[j] = (c < dim) ? *reinterpret_cast<const f32x4_t*>(gamma + c) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        adg[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        adb[j] = adg[j];
    }
    for (long long row0 = ((long long)blockIdx.x * LN_WAVES + wave) * RPW; row0 < rows; row0 += (long long)gridDim.x * LN_WAVES * RPW) {
        const long long row = row0 + sub;
        const bool ok = row < rows;
        const float mu = ok ? mean[row] : 0.f, rs = ok ? rstd[row] : 0.f;
        f32x4_t xh[MAXJ], g[MAXJ];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int c = (l + LPR * j) * 4;
            if (ok && c < dim) {
                const f32x4_t d = load4_nt<false>(dy, row * dim + c);
                xh[j] = (load4_nt<X_F32>(x, row * dim + c) - mu) * rs;
                g[j] = d * g4[j];
                c1 += sum4(g[j] * xh[j]);
                c2 += sum4(g[j]);
                adg[j] += d * xh[j];
                adb[j] += d;
            } else {
                xh[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                g[j] = xh[j];
            }
        }
        c1 = group_sum<LPR>(c1) * inv_dim;
        c2 = group_sum<LPR>(c2) * inv_dim;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int c = (l + LPR * j) * 4;
            if (ok && c < dim) {
                f32x4_t d = (g[j] - c2 - xh[j] * c1) * rs;
                if (DRES == 1) d += load4_nt<true>(dres, row * dim + c);
                if (DRES == 2) d += load4_nt<false>(dres, row * dim + c);
                // the fp32 stream gradient is next read by the LayerNorm backward of the preceding half, five GEMM-sized kernels on
                if (dx_f32 != nullptr) __builtin_nontemporal_store(d, reinterpret_cast<f32x4_t*>(dx_f32 + row * dim + c));
                if (dx_bf16 != nullptr) store4_bf16(dx_bf16, row * dim + c, d);
            }
        }
    }
    // reduction of the column partials over the row groups of the workgroup through LDS: [LN_WAVES * RPW][dim] floats,
    // dgamma then dbeta (fixed order)
    float* red = reinterpret_cast<float*>(smem);
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int c = (l + LPR * j) * 4;
            if (c < dim) *reinterpret_cast<f32x4_t*>(red + (wave * RPW + sub) * dim + c) = pass == 0 ? adg[j] : adb[j];
        }
        __syncthreads();
        for (int c = threadIdx.x * 4; c < dim; c += LN_THREADS * 4) {
            f32x4_t s = *reinterpret_cast<const f32x4_t*>(red + c);
#pragma unroll
            for (int w = 1; w < LN_WAVES * RPW; ++w) s += *reinterpret_cast<const f32x4_t*>(red + w * dim + c);
            *reinterpret_cast<f32x4_t*>(partial + ((long long)blockIdx.x * 2 + pass) * dim + c) = s;
        }
    }
}

// partial column sums of a bf16 matrix: partial[blockIdx.y][n] = sum over the block's row chunk
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ X, long long ld, float* __restrict__ partial,
                                                             long long T, int N, int rows_per_chunk) {
    __shared__ __attribute__((aligned(16))) float red[4][512];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + cg * 8;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > T) r1 = T;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    if (col < N) {
        for (long long r = r0 + rl; r < r1; r += 4) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(X + r * ld + col);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[2 * j] += bf16lo_to_f32(v[j]);
                a[2 * j + 1] += bf16hi_to_f32(v[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = a[j];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int gc = blockIdx.x * 512 + c;
        if (gc < N) partial[(long long)blockIdx.y * N + gc] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// out[c] = beta * out[c] + sum_p partial[p][c]  (fixed summation order)
// 16 columns x 64 row-lanes per workgroup: with ~1000 partial rows and ~1500 columns this gives ~100 workgroups and 16
// loads per thread (a 64-column x 16-lane layout left the kernel on 24 CUs with 64 dependent loads per thread: 20 us).
constexpr int RR_COLS = 16, RR_LANES = 64;
__global__ __launch_bounds__(RR_COLS * RR_LANES) void reduce_rows_kernel(const float* __restrict__ partial, int P, int ncols,
                                                                         float* __restrict__ out0, float* __restrict__ out1,
                                                                         int split_col, float beta) {
    __shared__ float red[RR_LANES][RR_COLS + 1];
    const int cl = threadIdx.x % RR_COLS, rl = threadIdx.x / RR_COLS;
    const int c = blockIdx.x * RR_COLS + cl;
    float s = 0.f;
    if (c < ncols)
        for (int p = rl; p < P; p += RR_LANES) s += partial[(long long)p * ncols + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < ncols) {
        float t = 0.f;
#pragma unroll 8
        for (int k = 0; k < RR_LANES; ++k) t += red[k][cl];
        float* dst = c < split_col ? out0 + c : out1 + (c - split_col);
        *dst = beta != 0.f ? beta * (*dst) + t : t;
    }
}

int ln_lpr(int dim) { return dim <= 512 ? 32 : 64; }             // lanes per row: half a wave for short rows
int ln_maxj(int dim) {                                             // 4-element chunks per lane, rounded up to a dispatched count
    const int lpr = ln_lpr(dim);
    const int chunks = (dim / 4 + lpr - 1) / lpr;
    static const int steps[8] = {1, 2, 3, 4, 5, 6, 8, 16};
    for (int k = 0; k < 8; ++k)
        if (chunks <= steps[k]) return steps[k];
    return 16;
}

template <bool X_F32, int MAXJ>
void launch_ln_fwd(const void* x, const float* g, const float* b, bf16_t* y, float* mean, float* rstd,
                   long long rows, int dim, float eps, int grid, hipStream_t s) {
    if (ln_lpr(dim) == 32)
        hipLaunchKernelGGL((ln_fwd_kernel<X_F32, (MAXJ > 4 ? 4 : MAXJ), 32>), dim3(grid), dim3(LN_THREADS), 0, s, x, g, b, y, mean, rstd, rows, dim, eps);
    else
        hipLaunchKernelGGL((ln_fwd_kernel<X_F32, MAXJ, 64>), dim3(grid), dim3(LN_THREADS), 0, s, x, g, b, y, mean, rstd, rows, dim, eps);
}

template <bool X_F32, int DRES, int MAXJ>
void launch_ln_bwd(const bf16_t* dy, const void* x, const float* g, const float* mean, const float* rstd, const void* dres,
                   float* dxf, bf16_t* dxb, float* partial, long long rows, int dim, int grid, hipStream_t s) {
    if (ln_lpr(dim) == 32)
        hipLaunchKernelGGL((ln_bwd_kernel<X_F32, DRES, (MAXJ > 4 ? 4 : MAXJ), 32>), dim3(grid), dim3(LN_THREADS), (size_t)LN_WAVES * 2 * dim * 4, s,
                           dy, x, g, mean, rstd, dres, dxf, dxb, partial, rows, dim);
    else
        hipLaunchKernelGGL((ln_bwd_kernel<X_F32, DRES, MAXJ, 64>), dim3(grid), dim3(LN_THREADS), (size_t)LN_WAVES * dim * 4, s,
                           dy, x, g, mean, rstd, dres, dxf, dxb, partial, rows, dim);
}

#define NRV_DISPATCH_MAXJ(MJ, CALL)            \
    switch (MJ) {                              \
        case 1: { constexpr int J = 1; CALL; } break;   \
        case 2: { constexpr int J = 2; CALL; } break;   \
        case 3: { constexpr int J = 3; CALL; } break;   \
        case 4: { constexpr int J = 4; CALL; } break;   \
        case 5: { constexpr int J = 5; CALL; } break;   \
        case 6: { constexpr int J = 6; CALL; } break;   \
        case 8: { constexpr int J = 8; CALL; } break;   \
        default: { constexpr int J = 16; CALL; } break; \
    }

int ln_bwd_grid(int64_t rows, int dim) {
    int64_t g = nrv_cdiv(rows, LN_WAVES * (64 / ln_lpr(dim)));
    if (g > LN_BWD_BLOCKS) g = LN_BWD_BLOCKS;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int nrv_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta,
                                 void* y_bf16, float* mean, float* rstd,
                                 int64_t rows, int dim, float eps, void* stream) {
    if (!x || !gamma || !beta || !y_bf16 || !mean || !rstd) return NRV_ERR_NULL;
    if (rows <= 0 || dim <= 0 || (dim & 7) || dim > 4096) return NRV_ERR_SHAPE;
    if (x_dtype != NRV_F32 && x_dtype != NRV_BF16) return NRV_ERR_DTYPE;
    if (!nrv_aligned16(x) || !nrv_aligned16(gamma) || !nrv_aligned16(beta) || !nrv_aligned16(y_bf16)) return NRV_ERR_ALIGN;
    int64_t g = nrv_cdiv(rows, LN_WAVES * (64 / ln_lpr(dim)));
    if (g > LN_FWD_BLOCKS) g = LN_FWD_BLOCKS;
    const int grid = (int)g;
    hipStream_t s = static_cast<hipStream_t>(stream);
    bf16_t* y = static_cast<bf16_t*>(y_bf16);
    const int mj = ln_maxj(dim);
    if (x_dtype == NRV_F32) {
        NRV_DISPATCH_MAXJ(mj, (launch_ln_fwd<true, J>(x, gamma, beta, y, mean, rstd, rows, dim, eps, grid, s)));
    } else {
        NRV_DISPATCH_MAXJ(mj, (launch_ln_fwd<false, J>(x, gamma, beta, y, mean, rstd, rows, dim, eps, grid, s)));
    }
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t nrv_layernorm_bwd_workspace(int64_t rows, int dim) {
    if (rows <= 0 || dim <= 0) return 0;
    return (size_t)ln_bwd_grid(rows, dim) * 2 * (size_t)dim * 4;
}

extern "C" int nrv_layernorm_bwd(const void* dy_bf16, const void* x, int x_dtype, const float* gamma,
                                 const float* mean, const float* rstd,
                                 const void* dres, int dres_dtype,
                                 float* dx_f32, void* dx_bf16,
                                 float* dgamma, float* dbeta, int accumulate,
                                 void* workspace, size_t workspace_bytes,
                                 int64_t rows, int dim, void* stream) {
    if (!dy_bf16 || !x || !gamma || !mean || !rstd || !dgamma || !dbeta || !workspace) return NRV_ERR_NULL;
    if (!dx_f32 && !dx_bf16) return NRV_ERR_NULL;
    if (rows <= 0 || dim <= 0 || (dim & 7) || dim > 4096) return NRV_ERR_SHAPE;
    if (x_dtype != NRV_F32 && x_dtype != NRV_BF16) return NRV_ERR_DTYPE;
    if (dres && dres_dtype != NRV_F32 && dres_dtype != NRV_BF16) return NRV_ERR_DTYPE;
    if (!nrv_aligned16(dy_bf16) || !nrv_aligned16(x) || !nrv_aligned16(gamma) || !nrv_aligned16(workspace) ||
        (dres && !nrv_aligned16(dres)) || (dx_f32 && !nrv_aligned16(dx_f32)) || (dx_bf16 && !nrv_aligned16(dx_bf16)))
        return NRV_ERR_ALIGN;
    const int grid = ln_bwd_grid(rows, dim);
    if (workspace_bytes < (size_t)grid * 2 * (size_t)dim * 4) return NRV_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16_t* dy = static_cast<const bf16_t*>(dy_bf16);
    bf16_t* dxb = static_cast<bf16_t*>(dx_bf16);
    float* partial = static_cast<float*>(workspace);
    const int mj = ln_maxj(dim);
    const int dr = !dres ? 0 : (dres_dtype == NRV_F32 ? 1 : 2);
#define NRV_LN_BWD(XF, DR) NRV_DISPATCH_MAXJ(mj, (launch_ln_bwd<XF, DR, J>(dy, x, gamma, mean, rstd, dres, dx_f32, dxb, partial, rows, dim, grid, s)))
    if (x_dtype == NRV_F32) {
        if (dr == 0) { NRV_LN_BWD(true, 0); } else if (dr == 1) { NRV_LN_BWD(true, 1); } else { NRV_LN_BWD(true, 2); }
    } else {
        if (dr == 0) { NRV_LN_BWD(false, 0); } else if (dr == 1) { NRV_LN_BWD(false, 1); } else { NRV_LN_BWD(false, 2); }
    }
#undef NRV_LN_BWD
    NRV_CHECK_LAUNCH();
    // partial rows are [grid*2][dim] with (block, pass) interleaved: view as [grid][2*dim]
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((2 * dim + RR_COLS - 1) / RR_COLS), dim3(RR_COLS * RR_LANES), 0, s,
                       partial, grid, 2 * dim, dgamma, dbeta, dim, accumulate ? 1.0f : 0.0f);
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t nrv_colsum_workspace(int64_t T, int64_t N) {
    if (T <= 0 || N <= 0) return 0;
    return (size_t)COLSUM_ROWCHUNKS * (size_t)N * 4;
}

extern "C" int nrv_colsum_bf16(const void* X, int64_t ld, float* out, int64_t T, int64_t N, float beta,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!X || !out || !workspace) return NRV_ERR_NULL;
    if (T <= 0 || N <= 0 || (N & 7) || (ld & 7) || ld < N || N > 0x7fffff00ll) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(X) || !nrv_aligned16(workspace)) return NRV_ERR_ALIGN;
    if (workspace_bytes < (size_t)COLSUM_ROWCHUNKS * (size_t)N * 4) return NRV_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int chunks = COLSUM_ROWCHUNKS;
    int64_t rpc = nrv_cdiv(T, chunks);
    if (rpc < 4) { rpc = 4; }
    chunks = (int)nrv_cdiv(T, rpc);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nrv_cdiv(N, 512), (unsigned)chunks), dim3(256), 0, s,
                       static_cast<const bf16_t*>(X), (long long)ld, partial, (long long)T, (int)N, (int)rpc);
    NRV_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)nrv_cdiv(N, RR_COLS)), dim3(RR_COLS * RR_LANES), 0, s,
                       partial, chunks, (int)N, out, out, (int)N, beta);
    NRV_CHECK_LAUNCH();
    return 0;
}
