// Stand-alone SinkhornAttention(scores) for gfx950: the reference's exported module (utils.py:1025-1037) applied to a
// MATERIALISED score tensor [..., R, C] -- softmax over the last dimension, `iters` x {rows / row sums; columns / column
// sums}, a final row normalisation.  The training path never comes here (robust=True attention is fused into
// nrv_attn_sinkhorn_*, which never materialises the scores): this is the op for users who call the module directly.
//
// P = diag(a) softmax(S) diag(b): only the scaling vectors iterate.  One workgroup per matrix keeps a (R floats) and b (C
// floats) in LDS and sweeps P0 = softmax(S) -- written once into the output buffer -- once per normalisation step: row steps
// with one wave per row (lanes along the row), column steps with one thread per column (consecutive threads read
// consecutive columns of a row): every access is coalesced, nothing is transposed.  The 2 iters + 1 row scalings and iters
// column scalings of every step are saved; the backward walks the steps in reverse on a gradient matrix kept in the dS
// buffer:  row step   Y = X / r :  dX = (dY - rowsum(dY o Y)) / r      (1 / r_i = a_k[i] / a_{k-1}[i])
//          column step likewise;   softmax:  dS = P0 o (dY0 - rowsum(dY0 o P0)).
// Deterministic (fixed reduction orders).  HBM / L2 bound: (2 iters + 3) sweeps of an R x C fp32 matrix forward.
#include "nrv_common.hpp"

namespace {

constexpr int SN_THREADS = 1024;
constexpr int SN_WAVES = SN_THREADS / 64;
constexpr int SN_CHUNK = 512;               // columns of one column-step pass: 8 per lane
constexpr int SN_PART_FLOATS = SN_WAVES * SN_CHUNK;

// column sums  t[j] = sum_i w_i(j),  w given per (row, column) by `term(i, j)`: wave w adds the rows i = w, w + 16, ... for 8 columns per
// lane (independent accumulators: 8 loads in flight per lane instead of one dependent chain down a column -- the first form, one
// thread per column walking all R rows, was latency-bound: 12 ms per call at 384 heads of 577 x 577), the 16 partial sums meet in LDS
// and are added in wave order (deterministic).  `emit(j, sum)` runs once per column.
template <typename Term, typename Emit>
__device__ __forceinline__ void column_sums(float* part /* [SN_WAVES][SN_CHUNK] */, int R, int C, int tid, Term term, Emit emit) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int c0 = 0; c0 < C; c0 += SN_CHUNK) {
        float acc[SN_CHUNK / 64];
#pragma unroll
        for (int k = 0; k < SN_CHUNK / 64; ++k) acc[k] = 0.f;
        for (int i = wave; i < R; i += SN_WAVES) {
#pragma unroll
            for (int k = 0; k < SN_CHUNK / 64; ++k) {
                const int j = c0 + lane + 64 * k;
                if (j < C) acc[k] += term(i, j);
            }
        }
#pragma unroll
        for (int k = 0; k < SN_CHUNK / 64; ++k) part[wave * SN_CHUNK + lane + 64 * k] = acc[k];
        __syncthreads();
        for (int jj = tid; jj < SN_CHUNK && c0 + jj < C; jj += SN_THREADS) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < SN_WAVES; ++w) sum += part[w * SN_CHUNK + jj];
            emit(c0 + jj, sum);
        }
        __syncthreads();
    }
}

// forward: scores [G,R,C] -> out [G,R,C]; lse [G,R]; avec [G, iters + 1, R] (cumulative row scalings a_1 .. a_{iters+1});
// bvec [G, iters, C] (cumulative column scalings b_1 .. b_iters)
__global__ __launch_bounds__(SN_THREADS) void sinknorm_fwd_kernel(const float* __restrict__ S, float* __restrict__ out,
                                                                 float* __restrict__ lse, float* __restrict__ avec,
                                                                 float* __restrict__ bvec, int R, int C, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* a = sm;            // [R]
    float* b = sm + R;        // [C]
    float* part = sm + R + C; // [SN_WAVES][SN_CHUNK]
    const long long g = blockIdx.x;
    const float* Sg = S + g * (long long)R * C;
    float* Pg = out + g * (long long)R * C;
    float* lg = lse + g * R;
    float* ag = avec + g * (long long)(iters + 1) * R;
    float* bg = bvec + g * (long long)iters * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // P0 = softmax over the row
    for (int i = wave; i < R; i += SN_WAVES) {
        const float* row = Sg + (long long)i * C;
        float m = -INFINITY;
        for (int j = lane; j < C; j += 64) m = fmaxf(m, row[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float s = 0.f;
        for (int j = lane; j < C; j += 64) s += __expf(row[j] - m);
        s = wave_sum(s);
        const float l = m + __logf(s);
        if (lane == 0) lg[i] = l;
        for (int j = lane; j < C; j += 64) Pg[(long long)i * C + j] = __expf(row[j] - l);
    }
    for (int j = tid; j < C; j += SN_THREADS) b[j] = 1.0f;
    __syncthreads();
    for (int it = 0; it <= iters; ++it) {
        // row step: a_i = 1 / sum_j P0_ij b_j
        for (int i = wave; i < R; i += SN_WAVES) {
            const float* row = Pg + (long long)i * C;
            float s = 0.f;
            for (int j = lane; j < C; j += 64) s = fmaf(row[j], b[j], s);
            s = wave_sum(s);
            if (lane == 0) {
                const float v = 1.0f / s;
                a[i] = v;
                ag[(long long)it * R + i] = v;
            }
        }
        __syncthreads();
        if (it == iters) break;
        // column step: b_j = 1 / sum_i a_i P0_ij
        column_sums(part, R, C, tid, [&](int i, int j) { return a[i] * Pg[(long long)i * C + j]; },
                    [&](int j, float sum) {
                        const float v = 1.0f / sum;
                        b[j] = v;
                        bg[(long long)it * C + j] = v;
                    });
    }
    // P = diag(a) P0 diag(b), in place
    for (int i = wave; i < R; i += SN_WAVES) {
        float* row = Pg + (long long)i * C;
        const float ai = a[i];
        for (int j = lane; j < C; j += 64) row[j] = ai * row[j] * b[j];
    }
}

// backward: dS [G,R,C] from dP (dout), the scores and the saved statistics; dS doubles as the running gradient matrix
__global__ __launch_bounds__(SN_THREADS) void sinknorm_bwd_kernel(const float* __restrict__ S, const float* __restrict__ dP,
                                                                 const float* __restrict__ lse, const float* __restrict__ avec,
                                                                 const float* __restrict__ bvec, float* __restrict__ dS,
                                                                 int R, int C, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* a = sm;            // [R]  a_k of the step being undone
    float* b = sm + R;        // [C]
    float* t = sm + R + C;    // [max(R, C)]  per-row / per-column correction
    float* part = t + (R > C ? R : C);     // [SN_WAVES][SN_CHUNK]
    const long long g = blockIdx.x;
    const float* Sg = S + g * (long long)R * C;
    const float* dPg = dP + g * (long long)R * C;
    float* Gg = dS + g * (long long)R * C;
    const float* lg = lse + g * R;
    const float* ag = avec + g * (long long)(iters + 1) * R;
    const float* bg = bvec + g * (long long)iters * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    auto p0 = [&](int i, int j) { return __expf(Sg[(long long)i * C + j] - lg[i]); };
    // start: G = dP
    for (long long e = tid; e < (long long)R * C; e += SN_THREADS) Gg[e] = dPg[e];
    // undo the steps k = 2 iters + 1 .. 1; row steps are the odd ones: row step number it (0-based) has a = avec[it],
    // b = bvec[it - 1] (ones for it = 0); column step it has a = avec[it], b = bvec[it]
    for (int it = iters; it >= 0; --it) {
        // ---- row step it:  Y = diag(avec[it]) P0 diag(b_prev),  1 / r_i = avec[it][i] / avec[it - 1][i]  (avec[-1] = 1)
        for (int i = tid; i < R; i += SN_THREADS) a[i] = ag[(long long)it * R + i];
        for (int j = tid; j < C; j += SN_THREADS) b[j] = it > 0 ? bg[(long long)(it - 1) * C + j] : 1.0f;
        __syncthreads();
        for (int i = wave; i < R; i += SN_WAVES) {
            float* grow = Gg + (long long)i * C;
            const float ai = a[i];
            float s = 0.f;
            for (int j = lane; j < C; j += 64) s = fmaf(grow[j], ai * p0(i, j) * b[j], s);
            s = wave_sum(s);
            const float inv_r = it > 0 ? ai / ag[(long long)(it - 1) * R + i] : ai;
            for (int j = lane; j < C; j += 64) grow[j] = (grow[j] - s) * inv_r;
        }
        __syncthreads();
        if (it == 0) break;
        // ---- column step it - 1:  Y = diag(avec[it - 1]) P0 diag(bvec[it - 1]),  1 / c_j = bvec[it - 1][j] / bvec[it - 2][j]
        for (int i = tid; i < R; i += SN_THREADS) a[i] = ag[(long long)(it - 1) * R + i];
        for (int j = tid; j < C; j += SN_THREADS) b[j] = bg[(long long)(it - 1) * C + j];
        __syncthreads();
        column_sums(part, R, C, tid, [&](int i, int j) { return Gg[(long long)i * C + j] * (a[i] * p0(i, j) * b[j]); },
                    [&](int j, float sum) { t[j] = sum; });
        for (int i = wave; i < R; i += SN_WAVES) {
            float* grow = Gg + (long long)i * C;
            for (int j = lane; j < C; j += 64) {
                const float inv_c = it > 1 ? b[j] / bg[(long long)(it - 2) * C + j] : b[j];
                grow[j] = (grow[j] - t[j]) * inv_c;
            }
        }
        __syncthreads();
    }
    // softmax backward
    for (int i = wave; i < R; i += SN_WAVES) {
        float* grow = Gg + (long long)i * C;
        float s = 0.f;
        for (int j = lane; j < C; j += 64) s = fmaf(grow[j], p0(i, j), s);
        s = wave_sum(s);
        for (int j = lane; j < C; j += 64) grow[j] = p0(i, j) * (grow[j] - s);
    }
}

int sn_check(int64_t G, int R, int C, int iters) {
    if (G <= 0 || R <= 0 || C <= 0 || iters < 0 || iters > 64) return NRV_ERR_SHAPE;
    if (G > 0x7fffffffll || R > 4096 || C > 4096) return NRV_ERR_SHAPE;
    return 0;
}

}  // namespace

extern "C" int nrv_sinkhorn_fwd(const float* scores, float* out, float* lse, float* avec, float* bvec,
                                int64_t G, int R, int C, int iters, void* stream) {
    if (!scores || !out || !lse || !avec || (iters > 0 && !bvec)) return NRV_ERR_NULL;
    if (int e = sn_check(G, R, C, iters)) return e;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(sinknorm_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (4096 * 2 + SN_PART_FLOATS) * 4);
    if (attr != 0) return attr;
    hipLaunchKernelGGL(sinknorm_fwd_kernel, dim3((unsigned)G), dim3(SN_THREADS), (size_t)(R + C + SN_PART_FLOATS) * 4, s,
                       scores, out, lse, avec, bvec, R, C, iters);
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" int nrv_sinkhorn_bwd(const float* scores, const float* dout, const float* lse, const float* avec, const float* bvec,
                                float* dscores, int64_t G, int R, int C, int iters, void* stream) {
    if (!scores || !dout || !lse || !avec || (iters > 0 && !bvec) || !dscores) return NRV_ERR_NULL;
    if (int e = sn_check(G, R, C, iters)) return e;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(sinknorm_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (4096 * 3 + SN_PART_FLOATS) * 4);
    if (attr != 0) return attr;
    hipLaunchKernelGGL(sinknorm_bwd_kernel, dim3((unsigned)G), dim3(SN_THREADS), (size_t)(R + C + (R > C ? R : C) + SN_PART_FLOATS) * 4, s,
                       scores, dout, lse, avec, bvec, dscores, R, C, iters);
    NRV_CHECK_LAUNCH();
    return 0;
}
