// Microbenchmark (dev tool, standalone HIP program): what does one CU sustain for 16-byte LDS-DMA (buffer_load ... lds), alone
// and next to MFMA work?  One 512-thread workgroup per CU; per iteration every wave issues NDMA 1-KiB LDS-DMA instructions
// from an L2-resident 64 KiB region of its own and, optionally, NMFMA bare MFMAs on registers; vmcnt(0) + barrier per
// iteration (the structure of the GEMM K loop).  Prints shader cycles per iteration (median over workgroups).
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/dma_rate tools/micro/dma_rate.hip && tools/_build/dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, unsigned voffset, unsigned soffset) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_base));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voffset), "s"(rsrc), "s"(lds_addr), "s"(soffset) : "memory");
}

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int NDMA, int NMFMA, int MODE /* 0 DMA, 1 global_load + ds_write */>
__global__ __launch_bounds__(512, 2) void k(const char* src, unsigned long long* out, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * 65536;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 65536, 0x00020000);
    f32x4_t acc[16];
    bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    a[0] = (short)lane; b[1] = (short)tid;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        constexpr int STEP = NMFMA > 0 ? (NMFMA + (NDMA > 0 ? NDMA : 1) - 1) / (NDMA > 0 ? NDMA : 1) : 0;
#pragma unroll
        for (int d = 0; d < (NDMA > NMFMA / 8 ? NDMA : (NMFMA + 7) / 8); ++d) {
            if (d < NDMA) {
                const int blk = (d * 8 + wave) & 63;
                if (MODE == 0) {
                    dma16s(rs, smem + buf * 65536 + blk * 1024, (unsigned)lane * 16, (unsigned)blk * 1024);
                } else {
                    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, blk * 1024, 0);
                    *reinterpret_cast<u32x4_t*>(smem + buf * 65536 + blk * 1024 + lane * 16) = v;
                }
            }
            if (NMFMA > 0) {
#pragma unroll
                for (int m = 0; m < (NDMA > 0 ? (NMFMA + NDMA - 1) / NDMA : 8); ++m) {
                    const int idx = (d * 8 + m) & 15;
                    if (d * ((NDMA > 0 ? (NMFMA + NDMA - 1) / NDMA : 8)) + m < NMFMA)
                        acc[idx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[idx], 0, 0, 0);
                }
            }
        }
        (void)STEP;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) sink[0] = s + smem[tid];
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int NDMA, int NMFMA, int MODE>
void run(const char* name, const char* src, unsigned long long* dout, float* sink, int grid) {
    const int iters = 400;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NDMA, NMFMA, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<NDMA, NMFMA, MODE>), dim3(grid), dim3(512), 131072, 0, src, dout, iters, sink);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), dout, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[grid / 2] / iters;
    const double kib = NDMA * 8.0;
    printf("%-44s %2d x 1 KiB per wave + %2d MFMA per wave: %7.0f cycles / iteration", name, NDMA, NMFMA, cyc);
    if (NDMA) printf("  = %5.1f cycles per KiB per CU", cyc / kib);
    if (NMFMA) printf("  (MFMA alone would be %d)", NMFMA * 16 * 2);
    printf("\n");
}


// K-tile shaped iteration: 8 groups of 8 MFMAs per wave; each wave issues one DMA per group (NDMA = 8).
// PLACE 0: every wave issues its DMA at the start of the group (both waves of a SIMD reach it together);
// PLACE 1: waves 4-7 issue it in the MIDDLE of the group (after 4 of the 8 MFMAs): out of phase with waves 0-3;
// PLACE 2: waves 0-3 issue TWO DMAs in groups 0-3, waves 4-7 two in groups 4-7 (disjoint halves of the K-tile).
// LIGHT 1: M0 is not saved / restored and only 2 wait states follow its write.
template <int PLACE, int LIGHT>
__global__ __launch_bounds__(512, 2) void k2(const char* src, unsigned long long* out, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const char* base = src + (size_t)blockIdx.x * 65536;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 65536, 0x00020000);
    f32x4_t acc[16];
    bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    a[0] = (short)lane; b[1] = (short)tid;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto dma = [&](int buf, int blk) {
        char* dst = smem + buf * 65536 + blk * 1024;
        if (LIGHT) {
            const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(dst));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 1\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                         :: "v"((unsigned)lane * 16), "s"(rs), "s"(lds_addr), "s"((unsigned)blk * 1024) : "memory");
        } else {
            dma16s(rs, dst, (unsigned)lane * 16, (unsigned)blk * 1024);
        }
    };
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (PLACE == 0 || (PLACE == 1 && grp == 0)) dma(buf, (g * 8 + wave) & 63);
            if (PLACE == 2 && grp == 0 && g < 4) { dma(buf, (2 * g * 8 + wave) & 63); dma(buf, ((2 * g + 1) * 8 + wave) & 63); }
            if (PLACE == 2 && grp == 1 && g >= 4) { dma(buf, (2 * (g - 4) * 8 + wave) & 63); dma(buf, ((2 * (g - 4) + 1) * 8 + wave) & 63); }
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[(g * 8 + m) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[(g * 8 + m) & 15], 0, 0, 0);
            if (PLACE == 1 && grp == 1) dma(buf, (g * 8 + wave) & 63);
#pragma unroll
            for (int m = 4; m < 8; ++m) acc[(g * 8 + m) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[(g * 8 + m) & 15], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) sink[0] = s + smem[tid];
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int PLACE, int LIGHT>
void run2(const char* name, const char* src, unsigned long long* dout, float* sink, int grid) {
    const int iters = 400;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k2<PLACE, LIGHT>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k2<PLACE, LIGHT>), dim3(grid), dim3(512), 131072, 0, src, dout, iters, sink);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), dout, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-78s %7.0f cycles / K-tile (MFMA alone 2048)\n", name, (double)h[grid / 2] / iters);
}

// same K-tile mix on the 32x32x16 MFMA: 32 MFMAs per wave (32 cycles each: the same 2048 cycles per SIMD), which hold the
// SIMD's vector issue for 8 of their 32 cycles instead of 8 of 16
template <int NDMA, int LIGHT>
__global__ __launch_bounds__(512, 2) void k3(const char* src, unsigned long long* out, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * 65536;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 65536, 0x00020000);
    f32x16_t acc[8];
    bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    a[0] = (short)lane; b[1] = (short)tid;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g < NDMA) {
                char* dst = smem + buf * 65536 + ((g * 8 + wave) & 63) * 1024;
                if (LIGHT) {
                    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(dst));
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 1\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                                 :: "v"((unsigned)lane * 16), "s"(rs), "s"(lds_addr), "s"((unsigned)(((g * 8 + wave) & 63) * 1024)) : "memory");
                } else dma16s(rs, dst, (unsigned)lane * 16, (unsigned)(((g * 8 + wave) & 63) * 1024));
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[(g * 4 + m) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(g * 4 + m) & 7], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
    if (s == 12345.678f) sink[0] = s + smem[tid];
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}
template <int NDMA, int LIGHT>
void run3(const char* name, const char* src, unsigned long long* dout, float* sink, int grid) {
    const int iters = 400;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k3<NDMA, LIGHT>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k3<NDMA, LIGHT>), dim3(grid), dim3(512), 131072, 0, src, dout, iters, sink);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), dout, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-78s %7.0f cycles / K-tile (MFMA alone 2048)\n", name, (double)h[grid / 2] / iters);
}

// K-tile mix WITH the LDS fragment reads of the two GEMM kernels: per group of 8 MFMAs either 3 ds_read_b128 (NT: 24 per
// K-tile and wave) or 6 ds_read_b64_tr_b16 (TN: 48), results consumed by the next group's MFMAs (one group ahead).
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
template <int READS /* 0 none, 1 b128 x3, 2 tr_b64 x6 */, int NDMA, int SHAPE = 0>
__global__ __launch_bounds__(512, 2) void k4(const char* src, unsigned long long* out, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * 65536;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 65536, 0x00020000);
    f32x4_t acc[16];
    f32x16_t acc2[8];
    bf16x8_t fa[2][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) { fa[0][j] = bf16x8_t{1, 2, 3, 4, 5, 6, 7, 8}; fa[0][j][0] = (short)(lane + j); fa[1][j] = fa[0][j]; }
    // conflict-free per-lane read offsets (row = lane & 15 style with an XOR swizzle), different per wave
    const int rd = ((wave * 16 + (lane & 15)) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4)) & 0x7fff;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        const char* sa = smem + (buf ^ 1) * 65536;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (READS == 1) {
#pragma unroll
                for (int j = 0; j < 3; ++j) fa[(g + 1) & 1][j] = *reinterpret_cast<const bf16x8_t*>(sa + ((rd + (g * 3 + j) * 2048) & 0xffff));
            } else if (READS == 2) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(sa + ((rd + (g * 6 + 2 * j) * 1024) & 0xfff8)));
                    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(sa + ((rd + (g * 6 + 2 * j + 1) * 1024) & 0xfff8)));
                    fa[(g + 1) & 1][j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            if (g < NDMA) dma16s(rs, smem + buf * 65536 + ((g * 8 + wave) & 63) * 1024, (unsigned)lane * 16, (unsigned)(((g * 8 + wave) & 63) * 1024));
            if (SHAPE == 0) {
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    acc[(g * 8 + m) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[g & 1][m % 3], fa[g & 1][(m + 1) % 3], acc[(g * 8 + m) & 15], 0, 0, 0);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc2[(g * 4 + m) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g & 1][m % 3], fa[g & 1][(m + 1) % 3], acc2[(g * 4 + m) & 7], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc2[i][0] + acc2[i][15];
    if (s == 12345.678f) sink[0] = s + smem[tid];
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}
template <int READS, int NDMA, int SHAPE = 0>
void run4(const char* name, const char* src, unsigned long long* dout, float* sink, int grid) {
    const int iters = 400;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k4<READS, NDMA, SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k4<READS, NDMA, SHAPE>), dim3(grid), dim3(512), 131072, 0, src, dout, iters, sink);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), dout, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-78s %7.0f cycles / K-tile (MFMA alone 2048), %6.1f ns wall\n", name, (double)h[grid / 2] / iters, ms * 1e6 / iters);
}

// The same CU tile (256 x 256 x 64 per K-tile) with FOUR waves, one per SIMD, each owning 128 x 128 of the output
// (256 accumulator registers): LDS fragment traffic drops from 8 x (128 + 64) to 4 x (128 + 128) rows per k (-33 %), the
// wave count per SIMD from 2 to 1.  SHAPE 0: v_mfma_f32_16x16x32_bf16 (128 per wave and K-tile), 1: 32x32x16 (64).
template <int SHAPE, int NDMA /* per wave and K-tile: 16 = the real tile */, int READS>
__global__ __launch_bounds__(256, 1) void k5(const char* src, unsigned long long* out, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * 65536;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 65536, 0x00020000);
    const int rd = ((wave * 16 + (lane & 15)) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4)) & 0x7fff;
    bf16x8_t fr[2][16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { fr[0][j] = bf16x8_t{1, 2, 3, 4, 5, 6, 7, 8}; fr[0][j][0] = (short)(lane + j); fr[1][j] = fr[0][j]; }
    float s = 0.f;
    unsigned long long t0, t1;
    if (SHAPE == 0) {
        f32x4_t acc[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            const int buf = it & 1;
            const char* sa = smem + (buf ^ 1) * 65536;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (READS) {
                        fr[(ks + 1) & 1][2 * i] = *reinterpret_cast<const bf16x8_t*>(sa + ((rd + (ks * 16 + 2 * i) * 2048) & 0xffff));
                        fr[(ks + 1) & 1][2 * i + 1] = *reinterpret_cast<const bf16x8_t*>(sa + ((rd + (ks * 16 + 2 * i + 1) * 2048) & 0xffff));
                    }
                    if (ks * 8 + i < NDMA) dma16s(rs, smem + buf * 65536 + (((ks * 8 + i) * 4 + wave) & 63) * 1024, (unsigned)lane * 16, (unsigned)((((ks * 8 + i) * 4 + wave) & 63) * 1024));
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc[i * 8 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[ks & 1][i], fr[ks & 1][8 + j], acc[i * 8 + j], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 64; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16_t acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            const int buf = it & 1;
            const char* sa = smem + (buf ^ 1) * 65536;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (READS) {
                        fr[(ks + 1) & 1][2 * i] = *reinterpret_cast<const bf16x8_t*>(sa + ((rd + (ks * 8 + 2 * i) * 2048) & 0xffff));
                        fr[(ks + 1) & 1][2 * i + 1] = *reinterpret_cast<const bf16x8_t*>(sa + ((rd + (ks * 8 + 2 * i + 1) * 2048) & 0xffff));
                    }
                    if (ks * 4 + i < NDMA) dma16s(rs, smem + buf * 65536 + (((ks * 4 + i) * 4 + wave) & 63) * 1024, (unsigned)lane * 16, (unsigned)((((ks * 4 + i) * 4 + wave) & 63) * 1024));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ks & 1][i], fr[ks & 1][4 + j], acc[i * 4 + j], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][15];
    }
    if (s == 12345.678f) sink[0] = s + smem[tid];
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}
template <int SHAPE, int NDMA, int READS>
void run5(const char* name, const char* src, unsigned long long* dout, float* sink, int grid) {
    const int iters = 400;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k5<SHAPE, NDMA, READS>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k5<SHAPE, NDMA, READS>), dim3(grid), dim3(256), 131072, 0, src, dout, iters, sink);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), dout, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-78s %7.0f cycles / K-tile (MFMA alone 2048)\n", name, (double)h[grid / 2] / iters);
}

int main() {
    const int grid = 256;
    char* src; unsigned long long* dout; float* sink;
    hipMalloc(&src, (size_t)grid * 65536); hipMemset(src, 1, (size_t)grid * 65536);
    hipMalloc(&dout, grid * 8); hipMalloc(&sink, 64);
    run<8, 0, 0>("LDS-DMA only", src, dout, sink, grid);
    run<4, 0, 0>("LDS-DMA only", src, dout, sink, grid);
    run<16, 0, 0>("LDS-DMA only (2 x 64 KiB in flight)", src, dout, sink, grid);
    run<0, 64, 0>("MFMA only", src, dout, sink, grid);
    run<8, 64, 0>("LDS-DMA + MFMA (the GEMM K-tile's mix)", src, dout, sink, grid);
    run<4, 64, 0>("LDS-DMA + MFMA (half the bytes per FLOP)", src, dout, sink, grid);
    run<8, 0, 1>("buffer_load -> VGPR -> ds_write_b128 only", src, dout, sink, grid);
    run<8, 64, 1>("buffer_load -> VGPR -> ds_write_b128 + MFMA", src, dout, sink, grid);
    run2<0, 0>("8 DMA + 64 MFMA per wave, DMA at group start in every wave", src, dout, sink, grid);
    run2<1, 0>("same, waves 4-7 issue their DMA in the middle of the group (out of phase)", src, dout, sink, grid);
    run2<2, 0>("same, waves 0-3 issue in groups 0-3 (2 each), waves 4-7 in groups 4-7", src, dout, sink, grid);
    run2<0, 1>("DMA at group start, light M0 handling (no save/restore, s_nop 1)", src, dout, sink, grid);
    run2<1, 1>("out of phase + light M0 handling", src, dout, sink, grid);
    run3<0, 0>("32x32x16 MFMA: 32 MFMA per wave, no DMA", src, dout, sink, grid);
    run3<8, 0>("32x32x16 MFMA: 8 DMA + 32 MFMA per wave", src, dout, sink, grid);
    run3<8, 1>("32x32x16 MFMA: 8 DMA + 32 MFMA per wave, light M0 handling", src, dout, sink, grid);
    run4<0, 0>("64 MFMA per wave, operands in registers, no DMA", src, dout, sink, grid);
    run4<1, 0>("64 MFMA + 24 ds_read_b128 per wave (NT fragment reads), no DMA", src, dout, sink, grid);
    run4<2, 0>("64 MFMA + 48 ds_read_b64_tr_b16 per wave (TN fragment reads), no DMA", src, dout, sink, grid);
    run4<1, 8>("64 MFMA + 24 ds_read_b128 + 8 DMA per wave  (the NT K-tile)", src, dout, sink, grid);
    run4<2, 8>("64 MFMA + 48 ds_read_b64_tr_b16 + 8 DMA per wave  (the TN K-tile)", src, dout, sink, grid);
    run4<0, 0, 1>("32 MFMA 32x32x16 per wave, registers only", src, dout, sink, grid);
    run4<1, 0, 1>("32 MFMA 32x32x16 + 24 ds_read_b128 per wave", src, dout, sink, grid);
    run4<1, 8, 1>("32 MFMA 32x32x16 + 24 ds_read_b128 + 8 DMA per wave (NT K-tile, 32x32 shape)", src, dout, sink, grid);
    run4<2, 8, 1>("32 MFMA 32x32x16 + 48 ds_read_b64_tr_b16 + 8 DMA per wave (TN K-tile, 32x32 shape)", src, dout, sink, grid);
    run5<0, 0, 0>("4 waves x 128x128: 128 MFMA 16x16x32 per wave, registers only", src, dout, sink, grid);
    run5<0, 0, 1>("4 waves x 128x128: 128 MFMA 16x16x32 + 32 ds_read_b128", src, dout, sink, grid);
    run5<0, 16, 1>("4 waves x 128x128: 128 MFMA 16x16x32 + 32 ds_read_b128 + 16 DMA (full K-tile)", src, dout, sink, grid);
    run5<1, 0, 0>("4 waves x 128x128: 64 MFMA 32x32x16 per wave, registers only", src, dout, sink, grid);
    run5<1, 0, 1>("4 waves x 128x128: 64 MFMA 32x32x16 + 32 ds_read_b128", src, dout, sink, grid);
    run5<1, 16, 1>("4 waves x 128x128: 64 MFMA 32x32x16 + 32 ds_read_b128 + 16 DMA (full K-tile)", src, dout, sink, grid);
    return 0;
}
