// Data-movement kernels around the encoder hot path (gfx950): patch unfold, weight staging
// (fp32 -> bf16 + transposed bf16), casts, token gather/scatter, and the hardware-assumption probes.
#include "nrv_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// patch unfold: img [B,C,H,W] -> patches bf16 [B*hh*ww, C*p*p]
//   layout 0 (p1 p2 c): einops 'b c (h p1) (w p2) -> b h w (p1 p2 c)'   simple_vit.py:126-129
//   layout 1 (c p1 p2): Conv2d(k=s=p) weight.reshape(D,-1) order         vit.py:237-242,323
// one thread = 8 consecutive output features (one 16-byte store)
// ---------------------------------------------------------------------------------------------
template <bool IMG_F32>
__device__ __forceinline__ float ld_img(const void* img, long long i) {
    return IMG_F32 ? reinterpret_cast<const float*>(img)[i] : bf16_to_f32(reinterpret_cast<const bf16_t*>(img)[i]);
}

template <bool IMG_F32, int LAYOUT>
__global__ __launch_bounds__(256) void patch_unfold_kernel(const void* __restrict__ img, bf16_t* __restrict__ out,
                                                           int B, int C, int H, int W, int p) {
    const int hh = H / p, ww = W / p;
    const int F = C * p * p, FP = (F + 7) & ~7, F8 = FP >> 3;      // rows are FP wide: features >= F (p = 14: 588 -> 592) are zero
    const long long total = (long long)B * hh * ww * F8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long t = i / F8;
        const int f0 = (int)(i - t * F8) * 8;
        const int b = (int)(t / (hh * ww));
        const int rem = (int)(t - (long long)b * hh * ww);
        const int py = rem / ww, px = rem - py * ww;
        if (LAYOUT == NRV_PATCH_CP1P2 && !IMG_F32 && (p & 7) == 0 && (W & 7) == 0) {
            // (c, p1, p2) order with p % 8 == 0: the 8 features are 8 consecutive pixels of one image row: one 16-byte copy
            const int c = f0 / (p * p);
            const int pp = f0 - c * p * p;
            const int p1 = pp / p, p2 = pp - p1 * p;
            const long long src = (((long long)b * C + c) * H + (py * p + p1)) * W + px * p + p2;
            *reinterpret_cast<u32x4_t*>(out + t * FP + f0) = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(img) + src);
            continue;
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = f0 + j;
            int c, p1, p2;
            if (LAYOUT == NRV_PATCH_P1P2C) {
                c = f % C;
                const int pp = f / C;
                p1 = pp / p;
                p2 = pp - p1 * p;
            } else {
                c = f / (p * p);
                const int pp = f - c * p * p;
                p1 = pp / p;
                p2 = pp - p1 * p;
            }
            v[j] = f < F ? ld_img<IMG_F32>(img, (((long long)b * C + c) * H + (py * p + p1)) * W + px * p + p2) : 0.f;
        }
        u32x4_t pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        *reinterpret_cast<u32x4_t*>(out + t * FP + f0) = pk;
    }
}

// ---------------------------------------------------------------------------------------------
// weight staging: w fp32 [R,C] -> wb bf16 [R,C], wt bf16 [C,R]
// ---------------------------------------------------------------------------------------------
// 64 x 64 tile per workgroup: 16-byte fp32 loads (256 contiguous bytes per 16 lanes), 8-byte bf16 stores for both the
// straight and the transposed copy (the 32 x 32 / 2-byte-store version ran at 2.7 TB/s and cost 0.34 ms per optimizer step).
__device__ __forceinline__ void cast_transpose_tile(float (&tile)[64][65], const float* __restrict__ w, bf16_t* __restrict__ wb,
                                                    bf16_t* __restrict__ wt, long long R, long long C, long long r0, long long c0) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // 16 x 16
    const bool vec_ok = (C & 3) == 0;                            // rows stay 16-byte aligned
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long r = r0 + ty + 16 * k, c = c0 + 4 * tx;
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
            if (vec_ok && c + 3 < C) {
                v = *reinterpret_cast<const f32x4_t*>(w + r * C + c);
                u32x2_t pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *reinterpret_cast<u32x2_t*>(wb + r * C + c) = pk;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < C) {
                        v[j] = w[r * C + c + j];
                        wb[r * C + c + j] = f32_to_bf16(v[j]);
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[ty + 16 * k][4 * tx + j] = v[j];
    }
    if (wt == nullptr) return;
    __syncthreads();
    const bool vec_t = (R & 3) == 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long c = c0 + ty + 16 * k, r = r0 + 4 * tx;      // output row c of wt, columns r .. r + 3
        if (c < C) {
            const float a0 = tile[4 * tx + 0][ty + 16 * k], a1 = tile[4 * tx + 1][ty + 16 * k];
            const float a2 = tile[4 * tx + 2][ty + 16 * k], a3 = tile[4 * tx + 3][ty + 16 * k];
            if (vec_t && r + 3 < R) {
                u32x2_t pk = {pack_bf16x2(a0, a1), pack_bf16x2(a2, a3)};
                *reinterpret_cast<u32x2_t*>(wt + c * R + r) = pk;
            } else {
                const float a[4] = {a0, a1, a2, a3};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r + j < R) wt[c * R + r + j] = f32_to_bf16(a[j]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ w, bf16_t* __restrict__ wb,
                                                             bf16_t* __restrict__ wt, long long R, long long C) {
    __shared__ float tile[64][65];
    cast_transpose_tile(tile, w, wb, wt, R, C, (long long)blockIdx.y * 64, (long long)blockIdx.x * 64);
}

// all weight matrices of a model in ONE launch: blockIdx.x = global tile number, the job table (device memory) maps it to
// a matrix.  49 separate launches of ~6 us each are launch-latency bound (0.30 ms per step for 344 MB + 344 MB).
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(const nrv_cast_job* __restrict__ jobs, int njobs) {
    __shared__ float tile[64][65];
    const long long id = blockIdx.x;
    int lo = 0, hi = njobs - 1;                    // last job with tile_start <= id
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile_start <= id) lo = mid; else hi = mid - 1;
    }
    const nrv_cast_job j = jobs[lo];
    const long long t = id - j.tile_start;
    const long long tr = t / j.tiles_c, tc = t - tr * j.tiles_c;
    cast_transpose_tile(tile, j.w, static_cast<bf16_t*>(j.w_bf16), static_cast<bf16_t*>(j.wT_bf16), j.R, j.C, tr * 64, tc * 64);
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long n) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(x + i * 4);
        u32x2_t pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2_t*>(y + i * 4) = pk;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = n4 * 4 + threadIdx.x;
        y[i] = f32_to_bf16(x[i]);
    }
}

// Dropout (p > 0 in training: vit.py:100-101,112,125,154,175).  The keep mask is DATA (one byte per element, drawn by the caller's
// generator and saved for the backward); these two kernels apply it.  scale = 1 / (1 - p) is a float argument: a bf16 mask of
// 1 / (1 - p) would round the scale itself to 8 bits.
//   dropout_add:  out = x + y * (keep ? scale : 0)      the residual stream takes a dropped branch output (fp32)
//   mask_mul:     out = a * (keep ? scale : 0)           bf16 tensors: the GELU output, the gelu' stream, the branch's incoming gradient
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const unsigned char* __restrict__ keep, float* __restrict__ out, float scale, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const u32x2_t k = *reinterpret_cast<const u32x2_t*>(keep + i * 8);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4_t xv = *reinterpret_cast<const f32x4_t*>(x + i * 8 + 4 * h);
            const f32x4_t yv = *reinterpret_cast<const f32x4_t*>(y + i * 8 + 4 * h);
            f32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = xv[e] + (((k[h] >> (8 * e)) & 0xffu) ? yv[e] * scale : 0.f);
            *reinterpret_cast<f32x4_t*>(out + i * 8 + 4 * h) = o;
        }
    }
}
__global__ __launch_bounds__(256) void mask_mul_kernel(const bf16_t* __restrict__ a, const unsigned char* __restrict__ keep,
                                                       bf16_t* __restrict__ out, float scale, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const u32x2_t k = *reinterpret_cast<const u32x2_t*>(keep + i * 8);
        const u32x4_t av = *reinterpret_cast<const u32x4_t*>(a + i * 8);
        u32x4_t o;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned kb = k[w >> 1] >> (16 * (w & 1));
            const float lo = (kb & 0xffu) ? bf16lo_to_f32(av[w]) * scale : 0.f;
            const float hi = (kb & 0xff00u) ? bf16hi_to_f32(av[w]) * scale : 0.f;
            o[w] = pack_bf16x2(lo, hi);
        }
        *reinterpret_cast<u32x4_t*>(out + i * 8) = o;
    }
}

// the fp32 form (attention probabilities [B,H,N,N] of the composed attention-dropout path: any element count, not a hot path)
__global__ __launch_bounds__(256) void mask_mul_f32_kernel(const float* __restrict__ a, const unsigned char* __restrict__ keep,
                                                           float* __restrict__ out, float scale, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = keep[i] ? a[i] * scale : 0.f;
}

// rows of the fp32 residual stream: gather out[r] = src[index[r]]; scatter dsrc[index[r]] = dout[r].  The indices are device data
// (a permutation computed by the caller's kernels): an index outside [0, rows_src) can never become an address -- the gather
// writes a zero row for it, the scatter drops the row (ABI 11; the host cannot validate device data without a sync).
template <bool SCATTER>
__global__ __launch_bounds__(256) void move_rows_kernel(const float* __restrict__ a, const long long* __restrict__ index,
                                                        float* __restrict__ o, long long rows, long long rows_src, int dim) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
        const long long s = index[r];
        const bool ok = s >= 0 && s < rows_src;                  // wave-uniform
        if (SCATTER && !ok) continue;
        const float* src = SCATTER ? a + r * dim : a + (ok ? s : 0) * dim;
        float* dst = SCATTER ? o + s * dim : o + r * dim;
        for (int c = lane * 4; c < dim; c += 256)
            *reinterpret_cast<f32x4_t*>(dst + c) = ok ? *reinterpret_cast<const f32x4_t*>(src + c) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
}

// ---------------------------------------------------------------------------------------------
// probes (single wave): make the hardware facts the kernels rely on visible to the test-suite
//   0: MFMA 16x16x32 bf16 lane maps.  in = A[16][32] then Bt[16][32] (B^T) bf16; out = D[16][16] fp32, D = A.B
//   1: ds_read_b64_tr_b16.  in = image [16 rows][64 cols] bf16, linear; every 16-lane group g reads the
//      4 x 16 block (rows 4g..4g+3, cols 0..15) with lane 4q+pp supplying row q / cols 4pp..4pp+3;
//      out[lane][e] = received element e
//   2: LDS-DMA.  in = n bytes (>= 768); lanes 0..47 fetch 16 B at offset 16*lane, lanes 48..63 use an
//      out-of-range offset; out[0..511] = the 1 KiB LDS block viewed as 512 bf16 -> fp32 (expect linear
//      placement and zeros in the tail)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void probe_kernel(int which, const bf16_t* __restrict__ in, float* __restrict__ out, int n) {
    __shared__ __attribute__((aligned(16))) char lds[2048];
    const int lane = threadIdx.x;
    if (which == 0) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(in + (lane & 15) * 32 + (lane >> 4) * 8);
        const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(in + 512 + (lane & 15) * 32 + (lane >> 4) * 8);
        f32x4_t d = {0.f, 0.f, 0.f, 0.f};
        d = mfma16(a, b, d);
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = d[r];
    } else if (which == 1) {
        for (int i = lane; i < 128; i += 64) *reinterpret_cast<u32x4_t*>(lds + i * 16) = *reinterpret_cast<const u32x4_t*>(in + i * 8);
        __syncthreads();
        const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
        const bf16x4_t t = lds_read_tr16_b64(lds + (4 * g + q) * 128 + pp * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[lane * 4 + e] = bf16_to_f32((unsigned short)t[e]);
    } else {
        for (int i = lane; i < 128; i += 64) *reinterpret_cast<u32x4_t*>(lds + i * 16) = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        __syncthreads();
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(in, (unsigned long long)n);
        dma16(rs, lds, lane < 48 ? (unsigned)lane * 16u : NRV_OOB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = lane; i < 512; i += 64) out[i] = bf16_to_f32(reinterpret_cast<const bf16_t*>(lds)[i]);
    }
}

int grid_for(long long work_items, int block) {
    long long g = (work_items + block - 1) / block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int nrv_abi_version(void) { return NRV_ABI_VERSION; }

extern "C" const char* nrv_error_string(int code) {
    switch (code) {
        case NRV_OK: return "ok";
        case NRV_ERR_NULL: return "required pointer is NULL";
        case NRV_ERR_SHAPE: return "unsupported shape or stride";
        case NRV_ERR_DTYPE: return "unknown dtype code";
        case NRV_ERR_WORKSPACE: return "workspace too small";
        case NRV_ERR_ALIGN: return "pointer or leading dimension not 16-byte aligned";
        case NRV_ERR_EPILOGUE: return "unknown epilogue or missing epilogue operand";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown nrv error";
    }
}

extern "C" int nrv_patch_unfold(const void* img, int img_dtype, void* patches_bf16,
                                int B, int C, int H, int W, int p, int layout, void* stream) {
    if (!img || !patches_bf16) return NRV_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0 || H % p || W % p) return NRV_ERR_SHAPE;
    if (img_dtype != NRV_F32 && img_dtype != NRV_BF16) return NRV_ERR_DTYPE;
    if (layout != NRV_PATCH_P1P2C && layout != NRV_PATCH_CP1P2) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(patches_bf16)) return NRV_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long total = (long long)B * (H / p) * (W / p) * ((C * p * p + 7) >> 3);
    const int grid = grid_for(total, 256);
    bf16_t* out = static_cast<bf16_t*>(patches_bf16);
    if (img_dtype == NRV_F32) {
        if (layout == NRV_PATCH_P1P2C) hipLaunchKernelGGL((patch_unfold_kernel<true, NRV_PATCH_P1P2C>), dim3(grid), dim3(256), 0, s, img, out, B, C, H, W, p);
        else hipLaunchKernelGGL((patch_unfold_kernel<true, NRV_PATCH_CP1P2>), dim3(grid), dim3(256), 0, s, img, out, B, C, H, W, p);
    } else {
        if (layout == NRV_PATCH_P1P2C) hipLaunchKernelGGL((patch_unfold_kernel<false, NRV_PATCH_P1P2C>), dim3(grid), dim3(256), 0, s, img, out, B, C, H, W, p);
        else hipLaunchKernelGGL((patch_unfold_kernel<false, NRV_PATCH_CP1P2>), dim3(grid), dim3(256), 0, s, img, out, B, C, H, W, p);
    }
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" int nrv_cast_transpose(const float* w, void* w_bf16, void* wT_bf16, int64_t R, int64_t C, void* stream) {
    if (!w || !w_bf16) return NRV_ERR_NULL;
    if (R <= 0 || C <= 0 || R > 0xffffll * 64 || C > 0x7fffffll * 64) return NRV_ERR_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((unsigned)nrv_cdiv(C, 64), (unsigned)nrv_cdiv(R, 64)), dim3(256), 0, s,
                       w, static_cast<bf16_t*>(w_bf16), static_cast<bf16_t*>(wT_bf16), (long long)R, (long long)C);
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" int nrv_cast_transpose_batched(const nrv_cast_job* jobs_dev, int njobs, int64_t total_tiles, void* stream) {
    if (!jobs_dev) return NRV_ERR_NULL;
    if (njobs <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffll) return NRV_ERR_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(cast_transpose_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, s, jobs_dev, njobs);
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" int nrv_cast_f32_bf16(const float* x, void* y_bf16, int64_t n, void* stream) {
    if (!x || !y_bf16) return NRV_ERR_NULL;
    if (n <= 0) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(x) || !nrv_aligned16(y_bf16)) return NRV_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n >> 2, 256)), dim3(256), 0, s, x, static_cast<bf16_t*>(y_bf16), (long long)n);
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" int nrv_dropout_add_f32(const float* x, const float* y, const unsigned char* keep, float* out, float scale, int64_t n, void* stream) {
    if (!x || !y || !keep || !out) return NRV_ERR_NULL;
    if (n <= 0 || (n & 7)) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(x) || !nrv_aligned16(y) || !nrv_aligned16(out) || (reinterpret_cast<uintptr_t>(keep) & 7u)) return NRV_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(dropout_add_kernel, dim3(grid_for(n >> 3, 256)), dim3(256), 0, s, x, y, keep, out, scale, (long long)(n >> 3));
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" int nrv_mask_mul_bf16(const void* a_bf16, const unsigned char* keep, void* out_bf16, float scale, int64_t n, void* stream) {
    if (!a_bf16 || !keep || !out_bf16) return NRV_ERR_NULL;
    if (n <= 0 || (n & 7)) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(a_bf16) || !nrv_aligned16(out_bf16) || (reinterpret_cast<uintptr_t>(keep) & 7u)) return NRV_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mask_mul_kernel, dim3(grid_for(n >> 3, 256)), dim3(256), 0, s, static_cast<const bf16_t*>(a_bf16), keep,
                       static_cast<bf16_t*>(out_bf16), scale, (long long)(n >> 3));
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" int nrv_mask_mul_f32(const float* a, const unsigned char* keep, float* out, float scale, int64_t n, void* stream) {
    if (!a || !keep || !out) return NRV_ERR_NULL;
    if (n <= 0) return NRV_ERR_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mask_mul_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, a, keep, out, scale, (long long)n);
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" int nrv_gather_rows_f32(const float* src, const int64_t* index, float* out,
                                   int64_t rows_out, int64_t rows_src, int dim, void* stream) {
    if (!src || !index || !out) return NRV_ERR_NULL;
    if (rows_out <= 0 || rows_src <= 0 || dim <= 0 || (dim & 3)) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(src) || !nrv_aligned16(out)) return NRV_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL((move_rows_kernel<false>), dim3(grid_for(rows_out, 4)), dim3(256), 0, s,
                       src, reinterpret_cast<const long long*>(index), out, (long long)rows_out, (long long)rows_src, dim);
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" int nrv_scatter_rows_f32(const float* dout, const int64_t* index, float* dsrc,
                                    int64_t rows_out, int64_t rows_src, int dim, void* stream) {
    if (!dout || !index || !dsrc) return NRV_ERR_NULL;
    if (rows_out <= 0 || rows_src <= 0 || dim <= 0 || (dim & 3)) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(dout) || !nrv_aligned16(dsrc)) return NRV_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL((move_rows_kernel<true>), dim3(grid_for(rows_out, 4)), dim3(256), 0, s,
                       dout, reinterpret_cast<const long long*>(index), dsrc, (long long)rows_out, (long long)rows_src, dim);
    NRV_CHECK_LAUNCH();
    return 0;
}

extern "C" int nrv_probe(int which, const void* in, void* out, int n, void* stream) {
    if (!in || !out) return NRV_ERR_NULL;
    if (which < 0 || which > 2) return NRV_ERR_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, s, which, static_cast<const bf16_t*>(in), static_cast<float*>(out), n);
    NRV_CHECK_LAUNCH();
    return 0;
}
