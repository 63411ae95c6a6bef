// Optimizer step on flat fp32 buffers for gfx950 (MI355X): global gradient-norm clipping + AdamW in two HBM-bound passes.
//
// The reference harness does, per step (examples/CIFAR100.py:90-97,191-192; baseline.py:127):
//     torch.nn.utils.clip_grad_norm_(params, 5.0);  torch.optim.AdamW(lr, betas (0.9, 0.999), eps 1e-8, weight_decay 0.05).step()
// which PyTorch-ROCm runs as ~6 multi-tensor launches over 150 parameter tensors.  Here all parameters (and their
// gradients, which the DP reducer already keeps in one flat buffer) are one contiguous range each:
//   pass 1  sum of squares of the gradient buffer       -> one float in HBM (two-stage, deterministic)
//   pass 2  p, m, v update with the clip coefficient computed on device from that float (no host round trip)
// Algorithmic traffic: pass 1 reads 4 B/param, pass 2 reads 16 and writes 12 B/param.
#include "nrv_common.hpp"

namespace {

constexpr int OPT_THREADS = 256;
constexpr int SUMSQ_BLOCKS = 1024;

// four gradients at index 4 i: fp32, or bf16 (the reduced bucket slabs of a bf16 gradient exchange: parallel.GradReducer)
template <bool BF16>
__device__ __forceinline__ f32x4_t load4(const void* __restrict__ x, long long i) {
    if (BF16) {
        const u32x2_t r = *reinterpret_cast<const u32x2_t*>(static_cast<const bf16_t*>(x) + 4 * i);
        return f32x4_t{bf16lo_to_f32(r[0]), bf16hi_to_f32(r[0]), bf16lo_to_f32(r[1]), bf16hi_to_f32(r[1])};
    }
    return *reinterpret_cast<const f32x4_t*>(static_cast<const float*>(x) + 4 * i);
}
template <bool BF16>
__device__ __forceinline__ float load1(const void* __restrict__ x, long long i) {
    return BF16 ? bf16_to_f32(static_cast<const bf16_t*>(x)[i]) : static_cast<const float*>(x)[i];
}

template <bool BF16>
__global__ __launch_bounds__(OPT_THREADS) void sumsq_partial_kernel(const void* __restrict__ x, long long n4, long long n,
                                                                   float* __restrict__ partial) {
    float s = 0.f;
    const long long stride = (long long)gridDim.x * OPT_THREADS;
    for (long long i = (long long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += stride) {
        const f32x4_t v = load4<BF16>(x, i);
        s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
    }
    if (blockIdx.x == 0)
        for (long long i = 4 * n4 + threadIdx.x; i < n; i += OPT_THREADS) { const float v = load1<BF16>(x, i); s = fmaf(v, v, s); }
    __shared__ float red[OPT_THREADS / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < OPT_THREADS / 64; ++w) t += red[w];
        partial[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(OPT_THREADS) void sumsq_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out) {
    // fixed summation order: thread t adds partial[t], partial[t + 256], ...; then the waves; then 4 wave sums in order
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += OPT_THREADS) s += partial[i];
    __shared__ float red[OPT_THREADS / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < OPT_THREADS / 64; ++w) t += red[w];
        out[0] = t;
    }
}

struct AdamWArgs {
    float lr, beta1, beta2, eps, decay;       // decay = 1 - lr * weight_decay
    float omb1, omb2;                         // 1 - beta, rounded from double as torch does (1 - 0.999f in float is off by 1.3e-5)
    float step_size, inv_bc2_sqrt;            // lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)
    float max_norm;                           // <= 0: no clipping
};

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamWArgs& a, float coef) {
    g *= coef;
    p *= a.decay;
    m = fmaf(a.beta1, m, a.omb1 * g);                  // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(a.beta2, v, a.omb2 * g * g);              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = fmaf(sqrtf(v), a.inv_bc2_sqrt, a.eps);
    p -= a.step_size * (m / denom);
}

template <bool GBF16>
__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(float* __restrict__ p, const void* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           long long n4, long long n, const float* __restrict__ gnorm_sq,
                                                           const float* __restrict__ step_scalars, AdamWArgs a) {
    if (step_scalars != nullptr) {          // the step-dependent scalars from device memory (a captured HIP graph replays this
        a.decay = step_scalars[0];          // launch with other learning rates and bias corrections: optim.FusedAdamW)
        a.step_size = step_scalars[1];
        a.inv_bc2_sqrt = step_scalars[2];
    }
    float coef = 1.0f;
    if (gnorm_sq != nullptr && a.max_norm > 0.f) {
        // clip_grad_norm_: coef = clamp(max_norm / (total_norm + 1e-6), max = 1)
        const float c = a.max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f);
        coef = c < 1.0f ? c : 1.0f;
    }
    const long long stride = (long long)gridDim.x * OPT_THREADS;
    for (long long i = (long long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += stride) {
        f32x4_t pv = *reinterpret_cast<const f32x4_t*>(p + 4 * i);
        const f32x4_t gv = load4<GBF16>(g, i);
        f32x4_t mv = *reinterpret_cast<const f32x4_t*>(m + 4 * i);
        f32x4_t vv = *reinterpret_cast<const f32x4_t*>(v + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pv[e], me = mv[e], ve = vv[e];
            adamw_one(pe, gv[e], me, ve, a, coef);
            pv[e] = pe; mv[e] = me; vv[e] = ve;
        }
        *reinterpret_cast<f32x4_t*>(p + 4 * i) = pv;
        *reinterpret_cast<f32x4_t*>(m + 4 * i) = mv;
        *reinterpret_cast<f32x4_t*>(v + 4 * i) = vv;
    }
    if (blockIdx.x == 0)
        for (long long i = 4 * n4 + threadIdx.x; i < n; i += OPT_THREADS) adamw_one(p[i], load1<GBF16>(g, i), m[i], v[i], a, coef);
}

}  // namespace

extern "C" size_t nrv_sumsq_workspace(int64_t n) {
    return n > 0 ? (size_t)SUMSQ_BLOCKS * 4 : 0;
}

extern "C" int nrv_sumsq_f32(const void* x, int x_dtype, int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || !workspace) return NRV_ERR_NULL;
    if (n <= 0) return NRV_ERR_SHAPE;
    if (x_dtype != NRV_F32 && x_dtype != NRV_BF16) return NRV_ERR_DTYPE;
    if (!nrv_aligned16(x)) return NRV_ERR_ALIGN;
    if (workspace_bytes < (size_t)SUMSQ_BLOCKS * 4) return NRV_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n4 = n / 4;
    long long blocks = (n4 + OPT_THREADS - 1) / OPT_THREADS;
    if (blocks > SUMSQ_BLOCKS) blocks = SUMSQ_BLOCKS;
    if (blocks < 1) blocks = 1;
    float* partial = static_cast<float*>(workspace);
    if (x_dtype == NRV_BF16) hipLaunchKernelGGL(sumsq_partial_kernel<true>, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, s, x, n4, (long long)n, partial);
    else hipLaunchKernelGGL(sumsq_partial_kernel<false>, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, s, x, n4, (long long)n, partial);
    NRV_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(OPT_THREADS), 0, s, partial, (int)blocks, out);
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" int nrv_adamw_f32(float* p, const void* g, int g_dtype, float* m, float* v, int64_t n,
                             double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                             const float* gnorm_sq, float max_norm, const float* step_scalars, void* stream) {
    if (!p || !g || !m || !v) return NRV_ERR_NULL;
    if (g_dtype != NRV_F32 && g_dtype != NRV_BF16) return NRV_ERR_DTYPE;
    if (n <= 0 || step < 1) return NRV_ERR_SHAPE;
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(p) || !nrv_aligned16(g) || !nrv_aligned16(m) || !nrv_aligned16(v)) return NRV_ERR_ALIGN;
    AdamWArgs a;
    // hyper-parameters arrive as doubles (Python floats) and are rounded once, as torch rounds its scalar arguments
    a.lr = (float)lr; a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps;
    a.decay = (float)(1.0 - lr * weight_decay);
    a.omb1 = (float)(1.0 - beta1);
    a.omb2 = (float)(1.0 - beta2);
    // bias corrections in double on the host, as torch computes them from a Python float step
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    a.step_size = (float)(lr / bc1);
    a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    a.max_norm = max_norm;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n4 = n / 4;
    long long blocks = (n4 + OPT_THREADS - 1) / OPT_THREADS;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (g_dtype == NRV_BF16) hipLaunchKernelGGL(adamw_kernel<true>, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, s, p, g, m, v, n4, (long long)n, gnorm_sq, step_scalars, a);
    else hipLaunchKernelGGL(adamw_kernel<false>, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, s, p, g, m, v, n4, (long long)n, gnorm_sq, step_scalars, a);
    NRV_CHECK_LAUNCH();
    return 0;
}
