// Stand-alone SinkhornAttention(scores) for gfx950: the reference's exported module (utils.py:1025-1037) applied to a
// MATERIALISED score tensor [..., R, C] -- softmax over the last dimension, `iters` x {rows / row sums; columns / column
// sums}, a final row normalisation.  The fused training kernels (nrv_attn_sinkhorn_*, N <= 256, head dim 64) never
// materialise the scores; this is the op for users who call the module directly and the middle of the composed robust
// attention at every other shape (kernels._attn_sinkhorn_*_composed).
//
// P = diag(a) softmax(S) diag(b): only the scaling vectors iterate.  One workgroup per matrix; a wave owns the rows wave,
// wave + 8, ... and holds ONE ROW AT A TIME IN REGISTERS (lane l: columns l, l + 64, ...), so that everything that is local
// to a row costs no extra pass over the matrix:
//   forward   sweep it = 0 .. iters reads S once: P0 = exp(S - lse) (the softmax statistics in sweep 0), the row step
//             a = 1 / (P0 . b), and either the column sums of a P0 for the next b (per-lane accumulators over the wave's rows,
//             the 8 waves meet in LDS) or, in the last sweep, the output a P0 b:  iters + 1 reads + 1 write of the matrix
//             (the first version wrote P0, swept it once per step and rescaled it in place: 2 iters + 6 passes);
//   backward  sweep it = iters .. 0 reads G and S and writes G once: the column rescale whose sums the previous sweep
//             gathered, the row step, the terms of the next column sums -- and in the last sweep the softmax backward:
//             3 (iters + 1) passes instead of 6 iters + 7.
//     row step   Y = X / r :  dX = (dY - rowsum(dY o Y)) / r      (1 / r_i = a_k[i] / a_{k-1}[i]);  column step likewise;
//     softmax:   dS = P0 o (dY0 - rowsum(dY0 o P0)).
// Deterministic (fixed reduction orders).  The 2 iters + 1 cumulative scaling vectors are saved by the forward.
#include "nrv_common.hpp"

namespace {

constexpr int SN_THREADS = 512;
constexpr int SN_WAVES = SN_THREADS / 64;
constexpr int SN_CHUNK = 512;               // columns of one cross-wave reduction pass: 8 per lane

// the 8 waves' per-lane column accumulators (column lane + 64 k in acc[k]) -> emit(j, sum over the waves in wave order), 512 columns
// per pass through LDS.  Every wave calls it (barriers).
template <int MAXJ, typename Emit>
__device__ __forceinline__ void reduce_columns(float* part /* [SN_WAVES][SN_CHUNK] */, const float (&acc)[MAXJ], int C, int tid, Emit emit) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int c = 0; c < (MAXJ + 7) / 8; ++c) {
        if (c * SN_CHUNK < C) {                                  // uniform
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (c * 8 + kk < MAXJ) part[wave * SN_CHUNK + lane + 64 * kk] = acc[c * 8 + kk];
            __syncthreads();
            const int j = c * SN_CHUNK + tid;                    // SN_THREADS == SN_CHUNK: one column per thread
            if (j < C) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < SN_WAVES; ++w) sum += part[w * SN_CHUNK + tid];
                emit(j, sum);
            }
            __syncthreads();
        }
    }
}
static_assert(SN_THREADS == SN_CHUNK, "one column per thread in the cross-wave pass");

// forward: scores [G,R,C] -> out [G,R,C]; lse [G,R]; avec [G, iters + 1, R] (cumulative row scalings a_1 .. a_{iters+1});
// bvec [G, iters, C] (cumulative column scalings b_1 .. b_iters).  LDS: b [64 MAXJ] | lse of the rows [R] | part.
template <int MAXJ>
__global__ __launch_bounds__(SN_THREADS) void sinknorm_fwd_kernel(const float* __restrict__ S, float* __restrict__ out,
                                                                 float* __restrict__ lse, float* __restrict__ avec,
                                                                 float* __restrict__ bvec, int R, int C, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int CP = 64 * MAXJ;
    float* b = sm;                 // [CP]
    float* lrow = sm + CP;         // [R]
    float* part = lrow + R;        // [SN_WAVES][SN_CHUNK]
    const long long g = blockIdx.x;
    const float* Sg = S + g * (long long)R * C;
    float* Pg = out + g * (long long)R * C;
    float* lg = lse + g * R;
    float* ag = avec + g * (long long)(iters + 1) * R;
    float* bg = bvec + g * (long long)iters * C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int j = tid; j < CP; j += SN_THREADS) b[j] = 1.0f;
    __syncthreads();
    for (int it = 0; it <= iters; ++it) {
        const bool last = it == iters;
        float acc[MAXJ];
#pragma unroll
        for (int k = 0; k < MAXJ; ++k) acc[k] = 0.f;
        for (int i = wave; i < R; i += SN_WAVES) {
            const float* row = Sg + (long long)i * C;
            float x[MAXJ];
#pragma unroll
            for (int k = 0; k < MAXJ; ++k) {
                const int j = lane + 64 * k;
                x[k] = j < C ? row[j] : -INFINITY;
            }
            float l;
            if (it == 0) {                                       // softmax statistics of the row
                float m = x[0];
#pragma unroll
                for (int k = 1; k < MAXJ; ++k) m = fmaxf(m, x[k]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < MAXJ; ++k) s += __expf(x[k] - m);
                s = wave_sum(s);
                l = m + __logf(s);
                if (lane == 0) { lg[i] = l; lrow[i] = l; }
            } else {
                l = lrow[i];                                     // written by this wave in sweep 0 (the same wave owns row i in every sweep)
            }
            // row step: a_i = 1 / sum_j P0_ij b_j
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < MAXJ; ++k) {
                x[k] = __expf(x[k] - l);                         // P0; exp(-inf) = 0 beyond the row
                r = fmaf(x[k], b[lane + 64 * k], r);
            }
            r = wave_sum(r);
            const float a = 1.0f / r;
            if (lane == 0) ag[(long long)it * R + i] = a;
            if (!last) {                                         // terms of the column step: b_j = 1 / sum_i a_i P0_ij
#pragma unroll
                for (int k = 0; k < MAXJ; ++k) acc[k] = fmaf(a, x[k], acc[k]);
            } else {                                             // P = diag(a) P0 diag(b)
                float* orow = Pg + (long long)i * C;
#pragma unroll
                for (int k = 0; k < MAXJ; ++k) {
                    const int j = lane + 64 * k;
                    if (j < C) orow[j] = a * x[k] * b[j];
                }
            }
        }
        if (last) break;
        // (the first barrier inside reduce_columns comes before any emit: every wave is done reading b by then)
        reduce_columns<MAXJ>(part, acc, C, tid, [&](int j, float sum) {
            const float v = 1.0f / sum;
            b[j] = v;
            bg[(long long)it * C + j] = v;
        });
    }
}

// backward: dS [G,R,C] from dP (dout), the scores and the saved statistics; dS doubles as the running gradient matrix (every lane
// re-reads only what it wrote itself).  LDS: bp | cinv | t [64 MAXJ each] | part.
template <int MAXJ>
__global__ __launch_bounds__(SN_THREADS) void sinknorm_bwd_kernel(const float* __restrict__ S, const float* __restrict__ dP,
                                                                 const float* __restrict__ lse, const float* __restrict__ avec,
                                                                 const float* __restrict__ bvec, float* __restrict__ dS,
                                                                 int R, int C, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int CP = 64 * MAXJ;
    float* bp = sm;                // [CP]  b of row step `it` = b of column step it - 1: bvec[it - 1] (ones for it = 0)
    float* cinv = sm + CP;         // [CP]  rescale of column step `it`: bvec[it] / bvec[it - 1]
    float* t = sm + 2 * CP;        // [CP]  column sums of column step `it` (gathered by the previous sweep)
    float* part = sm + 3 * CP;     // [SN_WAVES][SN_CHUNK]
    const long long g = blockIdx.x;
    const float* Sg = S + g * (long long)R * C;
    const float* dPg = dP + g * (long long)R * C;
    float* Gg = dS + g * (long long)R * C;
    const float* lg = lse + g * R;
    const float* ag = avec + g * (long long)(iters + 1) * R;
    const float* bg = bvec + g * (long long)iters * C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int j = tid; j < CP; j += SN_THREADS) t[j] = 0.f;
    // sweep `it` undoes column step it (its sums are in t; none for it = iters), then row step it, and gathers the sums of column
    // step it - 1; the last sweep ends with the softmax backward
    for (int it = iters; it >= 0; --it) {
        for (int j = tid; j < CP; j += SN_THREADS) {
            const float bprev = (it > 0 && j < C) ? bg[(long long)(it - 1) * C + j] : 1.0f;
            bp[j] = bprev;
            cinv[j] = (it < iters && j < C) ? bg[(long long)it * C + j] / bprev : 1.0f;
        }
        __syncthreads();
        float acc[MAXJ];
#pragma unroll
        for (int k = 0; k < MAXJ; ++k) acc[k] = 0.f;
        const float* gin = it == iters ? dPg : Gg;
        for (int i = wave; i < R; i += SN_WAVES) {
            const float* srow = Sg + (long long)i * C;
            const float* grow = gin + (long long)i * C;
            float x[MAXJ], G[MAXJ];
#pragma unroll
            for (int k = 0; k < MAXJ; ++k) {
                const int j = lane + 64 * k;
                x[k] = j < C ? srow[j] : -INFINITY;
                G[k] = j < C ? grow[j] : 0.f;
            }
            const float l = lg[i];
            const float a = ag[(long long)it * R + i];
            const float aprev = it > 0 ? ag[(long long)(it - 1) * R + i] : 1.0f;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < MAXJ; ++k) {
                const int j = lane + 64 * k;
                x[k] = __expf(x[k] - l) * bp[j];                 // P0 b_prev (zero beyond the row)
                G[k] = (G[k] - t[j]) * cinv[j];                  // column step `it` (t = 0, cinv = 1 in the first sweep)
                s = fmaf(G[k], x[k], s);
            }
            s = wave_sum(s) * a;                                 // rowsum(G o Y),  Y = a P0 b_prev
            const float inv_r = a / aprev;
            float* orow = Gg + (long long)i * C;
            if (it > 0) {
#pragma unroll
                for (int k = 0; k < MAXJ; ++k) {
                    const int j = lane + 64 * k;
                    G[k] = (G[k] - s) * inv_r;                   // row step `it`
                    acc[k] = fmaf(G[k], aprev * x[k], acc[k]);   // column step it - 1: sum_i G o (a_{it-1} P0 b_{it-1})
                    if (j < C) orow[j] = G[k];
                }
            } else {                                             // row step 0 (b_prev = 1: x is P0), then the softmax backward
                float sd = 0.f;
#pragma unroll
                for (int k = 0; k < MAXJ; ++k) {
                    G[k] = (G[k] - s) * inv_r;
                    sd = fmaf(G[k], x[k], sd);
                }
                sd = wave_sum(sd);
#pragma unroll
                for (int k = 0; k < MAXJ; ++k) {
                    const int j = lane + 64 * k;
                    if (j < C) orow[j] = x[k] * (G[k] - sd);
                }
            }
        }
        if (it == 0) break;
        reduce_columns<MAXJ>(part, acc, C, tid, [&](int j, float sum) { t[j] = sum; });
    }
}

int sn_check(int64_t G, int R, int C, int iters) {
    if (G <= 0 || R <= 0 || C <= 0 || iters < 0 || iters > 64) return NRV_ERR_SHAPE;
    if (G > 0x7fffffffll || R > 4096 || C > 4096) return NRV_ERR_SHAPE;
    return 0;
}

template <int MAXJ>
int launch_sn_fwd(const float* scores, float* out, float* lse, float* avec, float* bvec, int64_t G, int R, int C, int iters, hipStream_t s) {
    const size_t lds = (size_t)(64 * MAXJ + R + SN_WAVES * SN_CHUNK) * 4;
    static int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(sinknorm_fwd_kernel<MAXJ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (64 * MAXJ + 4096 + SN_WAVES * SN_CHUNK) * 4);
    if (attr != 0) return attr;
    hipLaunchKernelGGL((sinknorm_fwd_kernel<MAXJ>), dim3((unsigned)G), dim3(SN_THREADS), lds, s, scores, out, lse, avec, bvec, R, C, iters);
    NRV_CHECK_LAUNCH();
    return 0;
}

template <int MAXJ>
int launch_sn_bwd(const float* scores, const float* dout, const float* lse, const float* avec, const float* bvec, float* dscores,
                  int64_t G, int R, int C, int iters, hipStream_t s) {
    const size_t lds = (size_t)(3 * 64 * MAXJ + SN_WAVES * SN_CHUNK) * 4;
    static int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(sinknorm_bwd_kernel<MAXJ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != 0) return attr;
    hipLaunchKernelGGL((sinknorm_bwd_kernel<MAXJ>), dim3((unsigned)G), dim3(SN_THREADS), lds, s, scores, dout, lse, avec, bvec, dscores, R, C, iters);
    NRV_CHECK_LAUNCH();
    return 0;
}

// columns per lane of a row held in registers: the smallest instantiated count that covers C
#define NRV_SN_DISPATCH(C, CALL)                                   \
    do {                                                           \
        const int need = ((C) + 63) / 64;                          \
        if (need <= 4) { constexpr int MJ = 4; return CALL; }      \
        if (need <= 10) { constexpr int MJ = 10; return CALL; }    \
        if (need <= 16) { constexpr int MJ = 16; return CALL; }    \
        if (need <= 32) { constexpr int MJ = 32; return CALL; }    \
        { constexpr int MJ = 64; return CALL; }                    \
    } while (0)

}  // namespace

extern "C" int nrv_sinkhorn_fwd(const float* scores, float* out, float* lse, float* avec, float* bvec,
                                int64_t G, int R, int C, int iters, void* stream) {
    if (!scores || !out || !lse || !avec || (iters > 0 && !bvec)) return NRV_ERR_NULL;
    if (int e = sn_check(G, R, C, iters)) return e;
    hipStream_t s = static_cast<hipStream_t>(stream);
    NRV_SN_DISPATCH(C, launch_sn_fwd<MJ>(scores, out, lse, avec, bvec, G, R, C, iters, s));
}

extern "C" int nrv_sinkhorn_bwd(const float* scores, const float* dout, const float* lse, const float* avec, const float* bvec,
                                float* dscores, int64_t G, int R, int C, int iters, void* stream) {
    if (!scores || !dout || !lse || !avec || (iters > 0 && !bvec) || !dscores) return NRV_ERR_NULL;
    if (int e = sn_check(G, R, C, iters)) return e;
    hipStream_t s = static_cast<hipStream_t>(stream);
    NRV_SN_DISPATCH(C, launch_sn_bwd<MJ>(scores, dout, lse, avec, bvec, dscores, G, R, C, iters, s));
}
