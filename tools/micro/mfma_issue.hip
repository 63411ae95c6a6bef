// How long does a wave take to ISSUE n independent v_mfma_f32_16x16x32_bf16 (s_memtime in front, s_memtime behind, no use of
// the results in between), and how long until the results are there (a dependent VALU read behind them)?  1 or 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/mfma_issue tools/micro/mfma_issue.hip && tools/_build/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int N>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 acc[N];
    for (int i = 0; i < N; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    unsigned long long t[3][8];
    for (int rep = 0; rep < 8; ++rep) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) s += acc[i][0];            // waits for every result
        asm volatile("" :: "v"(s));
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        t[0][rep] = t0; t[1][rep] = t1; t[2][rep] = t2;
        __syncthreads();
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        for (int rep = 0; rep < 8; ++rep) {
            out[((size_t)blockIdx.x * 8 + w) * 16 + rep * 2] = t[1][rep] - t[0][rep];
            out[((size_t)blockIdx.x * 8 + w) * 16 + rep * 2 + 1] = t[2][rep] - t[0][rep];
        }
    }
    float s = 0.f;
    for (int i = 0; i < N; ++i) s += acc[i][1];
    if (s == 123.456f) sink[0] = s;
}

template <int N>
void run(int threads) {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * 16 * 8); hipMalloc(&sink, 4);
    hipMemset(d, 0, 256 * 8 * 16 * 8);
    hipLaunchKernelGGL(k<N>, dim3(256), dim3(threads), 0, 0, d, sink);
    hipDeviceSynchronize();
    static unsigned long long h[256 * 8 * 16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double issue = 0, done = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) for (int rep = 2; rep < 8; ++rep) {
        issue += (double)h[(b * 8 + w) * 16 + rep * 2]; done += (double)h[(b * 8 + w) * 16 + rep * 2 + 1]; ++n;
    }
    printf("%2d MFMAs 16x16x32 bf16, %d waves per SIMD: issued after %6.1f cycles, results read after %6.1f cycles (pipe time %d)\n",
           N, threads / 256, issue / n, done / n, N * 16);
    hipFree(d); hipFree(sink);
}

int main() {
    run<4>(256); run<8>(256); run<16>(256); run<20>(256); run<32>(256);
    run<4>(512); run<8>(512); run<16>(512); run<20>(512); run<32>(512);
    return 0;
}
