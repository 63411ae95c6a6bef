// MFMA bf16 GEMM kernels for gfx950 (MI355X).
//
//   gemm_nt_kernel : C[M,N] = A[M,K] . B[N,K]^T (+ fused epilogue)      -- nn.Linear forward / dX
//   gemm_tn_kernel : C[M,N] = sum_t A[t,M] B[t,N], split over t          -- nn.Linear dW
//
// Shared structure (one workgroup = one 256x256 output tile, 512 threads = 8 waves as 2(M) x 4(N),
// each wave owns 128x64 = 8x4 MFMA 16x16x32 accumulator tiles = 128 VGPRs):
//   * operands are staged HBM -> LDS by 16-byte LDS-DMA (buffer_load ... lds): no VGPR round trip,
//     out-of-range rows / k-chunks read as zero through the buffer descriptor, which is the only
//     tail handling the main loop needs;
//   * 2 LDS stages of (32 KiB A + 32 KiB B), BK = 64: tile k+1 is in flight while tile k is
//     multiplied; one vmcnt(0) + barrier per K-step (cdna_hip_programming.md §5, "glds, 2 LDS
//     buffers, BK=64" row);
//   * the LDS image is lane-linear per DMA instruction, so bank-conflict swizzles are applied to
//     the per-lane SOURCE address and undone on the fragment read (rule 21);
//   * accumulators are held "transposed" (MFMA(Bfrag, Afrag)): a lane owns 4 consecutive output
//     columns of one row, the epilogue transposes 16x64 slabs through a wave-private LDS patch and
//     every global access of the epilogue is a full 16-byte, row-contiguous access.
//
// Reference call sites replaced: simple_vit.py:39,41,61,62,130 ; vit.py:40-47 ; utils.py:115,579.

#include "nrv_common.hpp"
#include <atomic>
#include <type_traits>
#define NRV_DEV_TU gemm
#include <nrv_dev.hpp>       // instrumentation hooks: empty in the product (csrc/nrv_dev.hpp)

namespace {

constexpr int BK = 64;                      // K-tile depth of both kernels
constexpr int GEMM_THREADS = 512;
constexpr int DMA_GROUPS = 4;               // 2-stage K loops: MFMA groups of a K-step that carry the next K-step's DMA issue (swept 2 .. 8)

// NT tile configurations: WM x WN waves, each wave MI x NI MFMA tiles of 16x16 (NI is 4 everywhere: the
// epilogue transposes 16 x 64 slabs).  One workgroup per CU.
//   Cfg256: 256x256 tile, 8 waves, 128 KiB LDS -- least L2 traffic per FLOP among the square tiles
//   Cfg320: 320x256 tile: fewer rounds of workgroups on the N = 768 shapes (474 tiles = 2 rounds instead of 591 = 3)
// (192x128 / 128x256 tiles with two workgroups per CU, a 4-slot ring variant and a persistent deferred-epilogue kernel
// were measured slower in round 1 and are not part of the library any more: DESIGN.md "tried and rejected".)
template <int WM_, int WN_, int MI_, int NI_>
struct TileCfg {
    static constexpr int WM = WM_, WN = WN_, MI = MI_, NI = NI_;
    static constexpr int TBM = WM * MI * 16, TBN = WN * NI * 16;
    static constexpr int NWAVES = WM * WN, THREADS = 64 * NWAVES;
    static constexpr int A_BYTES = TBM * 128, B_BYTES = TBN * 128, STAGE = A_BYTES + B_BYTES;
    static constexpr int CA = TBM * 8 / THREADS, CB = TBN * 8 / THREADS;     // DMA instructions per thread per tile
    static constexpr int LDS = 2 * STAGE;
    static_assert(NI == 4, "epilogue slab is 64 columns");
    static_assert(TBM * 8 % THREADS == 0 && TBN * 8 % THREADS == 0, "whole DMA instructions");
};
using Cfg128 = TileCfg<2, 4, 4, 4>;      // 128x256 and 192x256: row counts that leave a 256-row grid a fraction of a round short
using Cfg192 = TileCfg<2, 4, 6, 4>;      // of the 256 CUs (MAE: 12544 rows x N 768 = 147 tiles of 256 rows, 198 of 192)
using Cfg256 = TileCfg<2, 4, 8, 4>;
using Cfg320 = TileCfg<2, 4, 10, 4>;
// 128-column tiles (4 x 2 waves) for N that is a multiple of 128 but wastes a quarter or more of a 256-column grid
// (N = 384: ViT-S): the same 64 KiB of operands per K-tile as Cfg256 for 3/4 of its MFMA work, but no padding columns
using Cfg384n = TileCfg<4, 2, 6, 4>;     // 384 x 128  (256 x 128 measured slower on every shape); phased persistent kernel since round 4
constexpr int EPI_ROW_F32 = 68;             // 64 floats + 4 pad  (272 B rows: conflict-free b128 writes)
constexpr int EPI_PATCH_BYTES = 16 * EPI_ROW_F32 * 4;   // 4352 B per wave

struct EpiParams {
    void* C;
    const float* bias;
    const void* aux;
    void* aux_out;
    long long ldc, ld_aux, ld_aux_out;
    int M, N;
    int aux_row_mod;
    int out_group, out_group_stride, out_row_offset;
};

struct GemmNTParams {
    const bf16_t* A;
    const bf16_t* B;
    long long lda, ldb;
    int K;
    int tiles_n;
    int gn;                       // column-group width of the tile order (0 = plain row-major sweep)
    int ntiles;                   // gemm_nt8_kernel (persistent): tiles of the whole output, >= gridDim.x
    EpiParams e;
};

struct GemmTNParams {
    const bf16_t* A;
    const bf16_t* B;
    long long lda, ldb;
    int T;
    int tiles_n, tiles_mn;
    int splits, kt_q, kt_r;       // K-tiles (of 64 token rows) per split: kt_q, the first kt_r splits one more
    int a_group, a_group_stride, a_row_offset;
    long long slab_stride;        // elements between split slabs (0 when splits == 1)
    float* bias_ws;               // optional [splits][M] column sums of A (bias gradient), nullptr = off
    EpiParams e;
};

// rows are < 2^31 (host-checked), so the remap stays in 32-bit integer arithmetic
__device__ __forceinline__ int remap_row(int m, int group, int group_stride, int offset) {
    if (group <= 0) return m;
    const int g = m / group;
    return g * group_stride + (m - g * group) + offset;
}

// ---------------------------------------------------------------------------------------------
// Epilogues.  acc[mi][ni] holds, for lane l: row  m = 16 mi + (l & 15), columns n = 16 ni + 4 (l >> 4) + {0,1,2,3} of
// the wave's (16 MI) x 64 block.  A 16 x 64 slab goes through a wave-private LDS patch so that every global access is a
// 16-byte, row-contiguous access.
//
// The epilogue is VALU-ISSUE bound (one wave64 VALU instruction per ~4.7 cycles per SIMD, two waves per SIMD: the round-1
// epilogue times of 3.4 / 12.4 / 17.5 us per 256 x 256 tile are 2 x 735 / 2648 / 4014 instructions x 4.7 cycles), so
// `epilogue_lin` spends no vector instruction on addresses: C, the epilogue operand and the gelu' stream are addressed
// through buffer descriptors whose base is the wave's block; the per-lane offset (row within a pass, column chunk) is ONE
// VGPR computed once, the row of each pass is a scalar soffset, and rows >= M / columns >= N are dropped by the
// descriptor's range check (a pass whose first row is >= M is skipped by a scalar branch, so soffset never exceeds the
// record count).  `epilogue_remap` keeps per-row pointer arithmetic for the one launch per step that scatters rows
// (class-token slot of the patch embedding) or broadcasts the operand rows (positional table).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned rawx4_t;
typedef __attribute__((ext_vector_type(2))) unsigned rawx2_t;

// Buffer stores of the epilogue (row offset = scalar soffset).  The hardware reads the store data AFTER issue.  For a store of
// more than 8 bytes hipcc keeps the wait states the ISA asks for before a VALU overwrite of the data registers only when
// soffset is not a register; with a register soffset it emits none.  On gfx950 with the vector-memory path busy (eight waves
// storing, LDS-DMA of the next tile in flight) an immediately following VALU write of a data register did reach the store:
// element 1 of an fp32 pass came out as the next pass's row index, intermittently (tools/nt_diag.py; first seen with the
// persistent kernel, latent in every build before it).  The data registers stay live across two wait states behind the store.
// AUX = 2: non-temporal (the gelu' stream: written once per layer, next read in the backward pass a whole model later)
template <int AUX = 0>
__device__ __forceinline__ void store_b128_row(rawx4_t v, __amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, vo, so, AUX);
    asm volatile("s_nop 1" :: "v"(v) : "memory");
}
template <int AUX = 0>
__device__ __forceinline__ void store_b64_row(rawx2_t v, __amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, vo, so, AUX);
    asm volatile("s_nop 1" :: "v"(v) : "memory");
}

// the 8-bit gelu' stream (NRV_EPI_BIAS_GELU_Q8 writes it, NRV_EPI_DGELU_Q8 reads it): gelu_erf' lies in [-0.1290, 1.1290]; a byte q stands for
// (q - 26) / 202 (step 0.00495: the rounding error <= 0.0025 is what bf16 leaves on values in [0.5, 1)); half the bytes of the bf16 stream
// in the two HBM-bound epilogues of a layer (fc1: 620 -> 465 MB written, dU: 392 -> 237 MB read at ViT-B/16 batch 256)
constexpr bool epi_gelu(int e) { return e == NRV_EPI_BIAS_GELU || e == NRV_EPI_BIAS_GELU_Q8; }
constexpr bool epi_dgelu(int e) { return e == NRV_EPI_DGELU || e == NRV_EPI_DGELU_Q8; }
constexpr bool epi_q8(int e) { return e == NRV_EPI_BIAS_GELU_Q8 || e == NRV_EPI_DGELU_Q8; }
#define NRV_Q8_SCALE 202.0f
#define NRV_Q8_ZERO 26.0f

// four gelu' values -> four bytes (v_cvt_pk_u8_f32 rounds to nearest, measured: tests/test_kernels_gpu.py)
__device__ __forceinline__ unsigned q8_pack4(f32x4_t g) {
    unsigned r = 0;
    const float y0 = fmaf(g[0], NRV_Q8_SCALE, NRV_Q8_ZERO), y1 = fmaf(g[1], NRV_Q8_SCALE, NRV_Q8_ZERO);
    const float y2 = fmaf(g[2], NRV_Q8_SCALE, NRV_Q8_ZERO), y3 = fmaf(g[3], NRV_Q8_SCALE, NRV_Q8_ZERO);
    asm("v_cvt_pk_u8_f32 %0, %1, 0, %0\n\tv_cvt_pk_u8_f32 %0, %2, 1, %0\n\tv_cvt_pk_u8_f32 %0, %3, 2, %0\n\tv_cvt_pk_u8_f32 %0, %4, 3, %0"
        : "+v"(r) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));
    return r;
}
// four bytes -> four gelu' values
__device__ __forceinline__ f32x4_t q8_unpack4(unsigned d) {
    float q0, q1, q2, q3;
    asm("v_cvt_f32_ubyte0 %0, %4\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_cvt_f32_ubyte2 %2, %4\n\tv_cvt_f32_ubyte3 %3, %4"
        : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(d));
    constexpr float c = 1.0f / NRV_Q8_SCALE, z = -NRV_Q8_ZERO / NRV_Q8_SCALE;
    return f32x4_t{fmaf(q0, c, z), fmaf(q1, c, z), fmaf(q2, c, z), fmaf(q3, c, z)};
}

template <int EPI, bool OUT_F32, bool AUX_F32, int MI>
__device__ __forceinline__ void epilogue_lin(f32x4_t (&acc)[MI][4], float* stg /* the wave's 4-KiB LDS patch */, const EpiParams& e,
                                             int row_base /* global row of the wave's block (wave-uniform) */,
                                             int col_base /* global col of the wave's block (wave-uniform) */,
                                             int lane) {
    const int rows_left = e.M - row_base, cols_left = e.N - col_base;
    if (rows_left <= 0 || cols_left <= 0) return;                       // uniform: the whole block is outside
    // patch: 16 rows of 64 floats, the 16-byte unit u of row r stored at unit u ^ r: conflict-free b128 writes (16 lanes =
    // 16 rows of one column group) and reads (16 lanes = 16 / 2 x 8 units of one / two rows) without padding
    const int wc_row = lane & 15, wg = lane >> 4;       // write side: row within slab, column group
    constexpr int CW = OUT_F32 ? 4 : 8;                 // columns per lane on the read side: one 16-byte store
    constexpr int V = CW / 4;                           // float4 pieces per lane
    constexpr int LPR = 64 / CW;                        // lanes per 64-column row
    constexpr int RPI = 64 / LPR;                       // rows covered per pass
    constexpr int NIT = 16 / RPI;                       // passes per 16-row slab
    const int rcol = lane % LPR, rrow = lane / LPR;
    const bool col_ok = rcol * CW < cols_left;          // N % 8 == 0: a chunk is entirely inside or outside
    const int rows_here = rows_left < MI * 16 ? rows_left : MI * 16;
    const int cols_here = cols_left < 64 ? cols_left : 64;
    constexpr bool HAS_AUX = EPI == NRV_EPI_BIAS_RESIDUAL || epi_dgelu(EPI);
    constexpr bool Q8 = epi_q8(EPI);                    // the gelu' stream as bytes (bf16 C only)
    static_assert(!Q8 || !OUT_F32, "the 8-bit gelu' stream goes with bf16 outputs: 8 columns = 8 bytes per lane");
    constexpr bool AUX32 = EPI == NRV_EPI_BIAS_RESIDUAL && AUX_F32;
    constexpr int ES = OUT_F32 ? 4 : 2, AS = AUX32 ? 4 : (Q8 ? 1 : 2), US = Q8 ? 1 : 2;
    constexpr int AW = CW * AS / 4;                     // dwords of the epilogue operand per lane and pass: 2, 4 or 8
    constexpr int HALF = MI > 8 ? (MI + 3) / 4 : (MI + 1) / 2;      // operand-prefetch depth, bounded by the register file

    // descriptors: base = the wave's block, records end with the last valid column of the last valid row
    // row strides: opaque per call, so that inside a persistent tile loop the per-pass offsets r0 * stride are computed here (one
    // s_mul each) instead of being hoisted out of the loop as 20 - 80 scalar registers that live, spilled, through the K loops
    int c_rs = (int)e.ldc * ES;
    asm volatile("" : "+s"(c_rs));
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(static_cast<char*>(e.C) + ((long long)row_base * e.ldc + col_base) * ES,
                                                ((unsigned long long)(rows_here - 1) * e.ldc + cols_here) * ES);
    const unsigned c_vo = col_ok ? (unsigned)(rrow * c_rs + rcol * CW * ES) : NRV_OOB;
    int a_rs = HAS_AUX ? (int)e.ld_aux * AS : 0;
    asm volatile("" : "+s"(a_rs));
    // The 8-bit stream is stored in ROW PAIRS: byte (m, n) at (m >> 1) * 2 ld + (n >> 6) * 128 + (m & 1) * 64 + (n & 63) (include/nrv.h), so that the 64 columns
    // x 2 rows a wave touches are ONE 128-byte line.  Row-major, a wave's 64 bytes per row are half a line, and what a CU can take in is a number of LINES
    // (~1 per 6.4 cycles: the K loop's own bound): the half lines cost the dU launch as much as the bf16 stream's full ones
    // (profiles/r04_gelu_stream_8bit_and_epilogue_bounds.txt).  row_base and every pass's first row are even, N % 64 == 0 (host).
    constexpr bool QA = Q8 && epi_dgelu(EPI), QU = Q8 && epi_gelu(EPI);
    const int last_row = rows_here - 1;
    const unsigned long long q8_records = (unsigned long long)(last_row >> 1) * 2ull * (unsigned long long)(QA ? e.ld_aux : e.ld_aux_out) + (last_row & 1) * 64 + 64;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(HAS_AUX ? static_cast<const char*>(e.aux) + (QA ? (long long)row_base * e.ld_aux + (col_base >> 6) * 128
                                                                                                   : ((long long)row_base * e.ld_aux + col_base) * AS) : nullptr,
                                                HAS_AUX ? (QA ? q8_records : ((unsigned long long)(rows_here - 1) * e.ld_aux + cols_here) * AS) : 0ull);
    const unsigned a_vo = (HAS_AUX && col_ok) ? (QA ? (unsigned)((rrow >> 1) * 2 * a_rs + (rrow & 1) * 64 + rcol * 8) : (unsigned)(rrow * a_rs + rcol * CW * AS)) : NRV_OOB;
    const bool want_u = epi_gelu(EPI) && e.aux_out != nullptr;
    int u_rs = want_u ? (int)e.ld_aux_out * US : 0;
    asm volatile("" : "+s"(u_rs));
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(want_u ? static_cast<char*>(e.aux_out) + (QU ? (long long)row_base * e.ld_aux_out + (col_base >> 6) * 128
                                                                                                  : ((long long)row_base * e.ld_aux_out + col_base) * US) : nullptr,
                                                want_u ? (QU ? q8_records : ((unsigned long long)(rows_here - 1) * e.ld_aux_out + cols_here) * US) : 0ull);
    const unsigned u_vo = (want_u && col_ok) ? (QU ? (unsigned)((rrow >> 1) * 2 * u_rs + (rrow & 1) * 64 + rcol * 8) : (unsigned)(rrow * u_rs + rcol * CW * US)) : NRV_OOB;

    f32x4_t bias4[V];
#pragma unroll
    for (int v = 0; v < V; ++v) bias4[v] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr bool HAS_BIAS = EPI == NRV_EPI_BIAS || epi_gelu(EPI) || EPI == NRV_EPI_BIAS_RESIDUAL;
    const bool add_bias = HAS_BIAS && e.bias != nullptr;                 // uniform
    if (add_bias && col_ok) {
#pragma unroll
        for (int v = 0; v < V; ++v) bias4[v] = *reinterpret_cast<const f32x4_t*>(e.bias + col_base + rcol * CW + 4 * v);
    }

#pragma unroll
    for (int h0 = 0; h0 < MI; h0 += HALF) {
        // the epilogue operand does not depend on the LDS transposition: issue the loads of a group of slabs up front
        unsigned auxr[HAS_AUX ? HALF : 1][NIT][AW];
        if (HAS_AUX) {
#pragma unroll
            for (int mh = 0; mh < HALF; ++mh) {
                const int mi = h0 + mh;
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    if (mi < MI) {
                        const int r0 = mi * 16 + RPI * i;                // first row of the pass (uniform)
                        const int so = (r0 < rows_here ? r0 : 0) * a_rs;  // keep soffset inside the records; the lanes are dropped below
                        const unsigned vo = r0 < rows_here ? a_vo : NRV_OOB;
                        constexpr int LAUX = epi_dgelu(EPI) ? 2 : 0;      // the gelu' stream is read once: non-temporal
                        if (AW == 2) {
                            const rawx2_t t = __builtin_amdgcn_raw_buffer_load_b64(ra, vo, so, LAUX);
                            auxr[mh][i][0] = t[0]; auxr[mh][i][1] = t[1];
                        } else {
#pragma unroll
                            for (int q = 0; q < AW / 4; ++q) {
                                const rawx4_t t = __builtin_amdgcn_raw_buffer_load_b128(ra, vo + 16 * q, so, LAUX);
#pragma unroll
                                for (int j = 0; j < 4; ++j) auxr[mh][i][4 * q + j] = t[j];
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int mh = 0; mh < HALF; ++mh) {
            const int mi = h0 + mh;
            if (mi < MI) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    *reinterpret_cast<f32x4_t*>(stg + wc_row * 64 + (((ni * 4 + wg) ^ wc_row) << 2)) = acc[mi][ni];
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    const int r = rrow + RPI * i;
                    f32x4_t val[V];
#pragma unroll
                    for (int v = 0; v < V; ++v) val[v] = *reinterpret_cast<const f32x4_t*>(stg + r * 64 + (((rcol * V + v) ^ r) << 2));
                    const int r0 = mi * 16 + RPI * i;                    // first row of the pass within the block (uniform)
                    if (r0 < rows_here) {                                // scalar branch; later rows of the pass: range check
                        unsigned pk[2 * V], pku[2 * V];
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            f32x4_t x = val[v];
                            if (HAS_BIAS) x += bias4[v];
                            if (epi_gelu(EPI)) {
                                // one erf/exp evaluation gives both gelu(u) (the output) and gelu'(u) (saved for the backward)
                                f32x4_t dg;
                                gelu_both4(x, x, dg);
                                if (Q8) {
                                    pku[v] = q8_pack4(dg);
                                } else {
                                    pku[2 * v] = pack_bf16x2(dg[0], dg[1]);
                                    pku[2 * v + 1] = pack_bf16x2(dg[2], dg[3]);
                                }
                            }
                            if (HAS_AUX) {
                                f32x4_t a;
                                if (AUX32) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) a[j] = __uint_as_float(auxr[mh][i][4 * v + j]);
                                } else if (Q8) {
                                    a = q8_unpack4(auxr[mh][i][v]);
                                } else {
                                    const unsigned a0 = auxr[mh][i][2 * v], a1 = auxr[mh][i][2 * v + 1];
                                    a = f32x4_t{bf16lo_to_f32(a0), bf16hi_to_f32(a0), bf16lo_to_f32(a1), bf16hi_to_f32(a1)};
                                }
                                if (EPI == NRV_EPI_BIAS_RESIDUAL) x += a;
                                else x *= a;
                            }
                            if (OUT_F32) {
                                store_b128_row(__builtin_bit_cast(rawx4_t, x), rc, c_vo + 16 * v, r0 * c_rs);
                            } else {
                                pk[2 * v] = pack_bf16x2(x[0], x[1]);
                                pk[2 * v + 1] = pack_bf16x2(x[2], x[3]);
                            }
                        }
                        if (!OUT_F32) {      // V == 2: one 16-byte store of 8 bf16
                            const rawx4_t o = {pk[0], pk[1], pk[2 * V - 2], pk[2 * V - 1]};
                            store_b128_row(o, rc, c_vo, r0 * c_rs);
                        }
                        if (epi_gelu(EPI) && want_u) {
                            if (Q8) {
                                const rawx2_t o = {pku[0], pku[1]};
                                store_b64_row<2>(o, ru, u_vo, r0 * u_rs);
                            } else if (V == 2) {
                                const rawx4_t o = {pku[0], pku[1], pku[2 * V - 2], pku[2 * V - 1]};
                                store_b128_row<2>(o, ru, u_vo, r0 * u_rs);
                            } else {
                                const rawx2_t o = {pku[0], pku[1]};
                                store_b64_row<2>(o, ru, u_vo, r0 * u_rs);
                            }
                        }
                    }
                }
            }
        }
    }
}

// per-row pointer arithmetic (row remap / operand row broadcast): the patch-embedding launch only
template <int EPI, bool OUT_F32, bool AUX_F32, int MI>
__device__ __forceinline__ void epilogue_remap(f32x4_t (&acc)[MI][4], char* smem, const EpiParams& e,
                                         int row_base /* global row of the wave's block */,
                                         int col_base /* global col of the wave's block */,
                                         int lane, int wave) {
    float* stg = reinterpret_cast<float*>(smem + wave * EPI_PATCH_BYTES);
    const int wc_row = lane & 15, wg = lane >> 4;       // write side: row within slab, column group
    // read side: a lane owns CW consecutive columns of one row, so that every global access is 16 bytes wide
    // (bf16 results: 8 columns = one dwordx4 store -- the store tail is issue-bound, guide T21; fp32: 4 columns)
    constexpr int CW = OUT_F32 ? 4 : 8;
    constexpr int V = CW / 4;                           // float4 pieces per lane
    constexpr int LPR = 64 / CW;                        // lanes per 64-column row
    constexpr int RPI = 64 / LPR;                       // rows covered per pass
    constexpr int NIT = 16 / RPI;                       // passes per 16-row slab
    const int rcol = lane % LPR, rrow = lane / LPR;
    const int ncol = col_base + rcol * CW;
    const bool col_ok = ncol < e.N;                     // N % 8 == 0: a chunk is entirely inside or outside
    constexpr bool HAS_AUX = EPI == NRV_EPI_BIAS_RESIDUAL || EPI == NRV_EPI_DGELU;
    constexpr bool AUX32 = EPI == NRV_EPI_BIAS_RESIDUAL && AUX_F32;
    constexpr int HALF = MI > 8 ? (MI + 3) / 4 : (MI + 1) / 2;      // operand-prefetch depth, bounded by the register file

    f32x4_t bias4[V];
#pragma unroll
    for (int v = 0; v < V; ++v) bias4[v] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (EPI == NRV_EPI_BIAS || EPI == NRV_EPI_BIAS_GELU || EPI == NRV_EPI_BIAS_RESIDUAL) {
        if (e.bias != nullptr && col_ok) {
#pragma unroll
            for (int v = 0; v < V; ++v) bias4[v] = *reinterpret_cast<const f32x4_t*>(e.bias + ncol + 4 * v);
        }
    }

#pragma unroll
    for (int h0 = 0; h0 < MI; h0 += HALF) {
        // Epilogue operands (residual stream / saved pre-activation) do not depend on the LDS transposition:
        // issue the loads of half the wave's block up front so that many requests per lane are in flight
        // (one dependent load per slab made this phase latency-bound: 27 us per 256x256 fp32-residual tile).
        f32x4_t aux32[AUX32 ? HALF : 1][NIT][V];
        u32x2_t aux16[(HAS_AUX && !AUX32) ? HALF : 1][NIT][V];
        if (HAS_AUX) {
#pragma unroll
            for (int mh = 0; mh < HALF; ++mh) {
                const int mi = h0 + mh;
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    const int m = row_base + mi * 16 + rrow + RPI * i;
                    const bool ok = mi < MI && m < e.M && col_ok;
                    const long long orow = remap_row(m, e.out_group, e.out_group_stride, e.out_row_offset);
                    const long long arow = (EPI == NRV_EPI_BIAS_RESIDUAL && e.aux_row_mod > 0) ? (long long)(m % e.aux_row_mod) : orow;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        if (AUX32) {
                            aux32[mh][i][v] = ok ? *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(e.aux) + arow * e.ld_aux + ncol + 4 * v)
                                                 : f32x4_t{0.f, 0.f, 0.f, 0.f};
                        } else {
                            aux16[mh][i][v] = ok ? *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const bf16_t*>(e.aux) + arow * e.ld_aux + ncol + 4 * v)
                                                 : u32x2_t{0u, 0u};
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int mh = 0; mh < HALF; ++mh) {
            const int mi = h0 + mh;
            if (mi < MI) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    *reinterpret_cast<f32x4_t*>(stg + wc_row * EPI_ROW_F32 + ni * 16 + wg * 4) = acc[mi][ni];
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    const int r = rrow + RPI * i;
                    f32x4_t val[V];
#pragma unroll
                    for (int v = 0; v < V; ++v) val[v] = *reinterpret_cast<const f32x4_t*>(stg + r * EPI_ROW_F32 + rcol * CW + 4 * v);
                    const int m = row_base + mi * 16 + r;
                    if (m < e.M && col_ok) {
                        const long long orow = remap_row(m, e.out_group, e.out_group_stride, e.out_row_offset);
                        unsigned pk[2 * V], pku[2 * V];
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            f32x4_t x = val[v] + bias4[v];
                            if (EPI == NRV_EPI_BIAS_GELU) {
                                // one erf/exp evaluation gives both gelu(u) (the output) and gelu'(u) (saved for the backward)
                                f32x4_t dg;
#pragma unroll
                                for (int j = 0; j < 4; ++j) { float gv, dv; gelu_both(x[j], gv, dv); x[j] = gv; dg[j] = dv; }
                                pku[2 * v] = pack_bf16x2(dg[0], dg[1]);
                                pku[2 * v + 1] = pack_bf16x2(dg[2], dg[3]);
                            }
                            if (EPI == NRV_EPI_BIAS_RESIDUAL) {
                                if (AUX32) {
                                    x += aux32[mh][i][v];
                                } else {
                                    const u32x2_t a = aux16[mh][i][v];
                                    x[0] += bf16lo_to_f32(a[0]); x[1] += bf16hi_to_f32(a[0]);
                                    x[2] += bf16lo_to_f32(a[1]); x[3] += bf16hi_to_f32(a[1]);
                                }
                            }
                            if (EPI == NRV_EPI_DGELU) {
                                const u32x2_t a = aux16[mh][i][v];
                                x[0] *= bf16lo_to_f32(a[0]); x[1] *= bf16hi_to_f32(a[0]);
                                x[2] *= bf16lo_to_f32(a[1]); x[3] *= bf16hi_to_f32(a[1]);
                            }
                            if (OUT_F32) {
                                *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(e.C) + orow * e.ldc + ncol + 4 * v) = x;
                            } else {
                                pk[2 * v] = pack_bf16x2(x[0], x[1]);
                                pk[2 * v + 1] = pack_bf16x2(x[2], x[3]);
                            }
                        }
                        if (!OUT_F32) {      // V == 2: one 16-byte store of 8 bf16
                            const u32x4_t o = {pk[0], pk[1], pk[2 * V - 2], pk[2 * V - 1]};
                            *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(e.C) + orow * e.ldc + ncol) = o;
                        }
                        if (EPI == NRV_EPI_BIAS_GELU && e.aux_out != nullptr) {
                            if (V == 2) {
                                const u32x4_t o = {pku[0], pku[1], pku[2 * V - 2], pku[2 * V - 1]};
                                *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(e.aux_out) + orow * e.ld_aux_out + ncol) = o;
                            } else {
                                const u32x2_t o = {pku[0], pku[1]};
                                *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(e.aux_out) + orow * e.ld_aux_out + ncol) = o;
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// NT kernel.  LDS tile image: [256 rows][64 k] bf16 = 128-byte rows; 16-byte chunk c of row r is
// stored at chunk position c ^ ((r >> 1) & 7): conflict-free ds_read_b128 for the MFMA fragment
// pattern (lane & 15 = row, lane >> 4 = chunk).
// ---------------------------------------------------------------------------------------------
template <typename C, int EPI, bool OUT_F32, bool AUX_F32, bool REMAP>
__global__ __launch_bounds__(C::THREADS, 2) void gemm_nt_kernel(const GemmNTParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // provably wave-uniform: descriptors, LDS bases and
    const int wr = wave / C::WN, wc = wave - wr * C::WN;                 // the epilogue's row offsets stay in scalar registers

    const unsigned id = xcd_remap(blockIdx.x, gridDim.x);
    // tile order: consecutive ids run concurrently on one XCD (xcd_remap).  With column groups of `gn` tiles the 32
    // tiles an XCD holds at a time form an (32/gn) x gn block: fewer distinct A/B panels per round than a 1-D sweep,
    // so more of the panel re-reads hit the XCD's 4 MiB L2 instead of the fabric.
    int tm, tn;
    if (p.gn > 0 && p.tiles_n > p.gn) {
        const int tiles_m = gridDim.x / p.tiles_n;
        const int gsize = tiles_m * p.gn;
        const int grp = id / gsize, within = id - grp * gsize;
        const int gw = (grp + 1) * p.gn <= p.tiles_n ? p.gn : p.tiles_n - grp * p.gn;     // width of this column group
        tm = within / gw;
        tn = grp * p.gn + (within - tm * gw);
    } else {
        tm = id / p.tiles_n;
        tn = id - tm * p.tiles_n;
    }
    const int m0 = tm * C::TBM, n0 = tn * C::TBN;
    const int M = p.e.M, N = p.e.N, K = p.K;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A + (long long)m0 * p.lda, (unsigned long long)(M - m0) * p.lda * 2ull);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B + (long long)n0 * p.ldb, (unsigned long long)(N - n0) * p.ldb * 2ull);

    // staging: DMA instruction i of this wave fills rows 8*(NWAVES i + wave) .. +7 of a tile
    unsigned st_a[C::CA], st_b[C::CB];
#pragma unroll
    for (int i = 0; i < C::CA; ++i) {
        const int r = (i * C::NWAVES + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        st_a[i] = (m0 + r < M) ? (unsigned)((long long)r * p.lda * 2) + c * 16 : NRV_OOB;
    }
#pragma unroll
    for (int i = 0; i < C::CB; ++i) {
        const int r = (i * C::NWAVES + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        st_b[i] = (n0 + r < N) ? (unsigned)((long long)r * p.ldb * 2) + c * 16 : NRV_OOB;
    }
    // one DMA instruction (index d of the CA + CB that make up a K-tile) of tile kt into stage buffer buf.  The per-lane
    // offset is a constant of the tile (row, swizzled chunk; NRV_OOB for rows beyond the matrix), the K position is the
    // scalar soffset: no vector instruction per DMA.  Only a K that is not a multiple of 64 needs a per-lane test, in
    // its last tile (chunks at k >= K must read zero, and they are not at the end of the buffer).
    const int nk = (K + BK - 1) / BK;
    const bool ktail = (K & (BK - 1)) != 0;
    auto dma_one = [&](int buf, int kt, int d) {
        char* base = smem + buf * C::STAGE;
        const int k0 = kt * BK;
        const bool isA = d < C::CA;
        const int i = isA ? d : d - C::CA;
        unsigned vo = isA ? st_a[isA ? i : 0] : st_b[isA ? 0 : i];
        if (ktail && kt == nk - 1) {
            const int r = (i * C::NWAVES + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            vo = (k0 + c * 8 < K) ? vo : NRV_OOB;
        }
        dma16s(isA ? ra : rb, base + (isA ? 0 : C::A_BYTES) + (i * C::NWAVES + wave) * 1024, vo, (unsigned)(k0 * 2));
    };
    constexpr int ND = C::CA + C::CB;            // DMA instructions per thread per K-tile
    constexpr int NG = C::MI;                    // MFMA groups per K-tile: 2 k-steps x MI/2 row pairs
    constexpr int NGD = DMA_GROUPS < NG ? DMA_GROUPS : NG;   // groups that carry the next tile's DMA issue

    // fragment read offsets
    const int fr = lane & 15, fg = lane >> 4;
    const int swz = (fg ^ ((fr >> 1) & 7)) << 4;
    const int a_rd = (wr * (C::MI * 16) + fr) * 128 + swz;
    const int b_rd = C::A_BYTES + (wc * (C::NI * 16) + fr) * 128 + swz;

    f32x4_t acc[C::MI][C::NI];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    NRV_STAMP_VARS(4);
    NRV_WACC_VARS;
    NRV_STAMP(0);
#pragma unroll
    for (int d = 0; d < ND; ++d) dma_one(0, 0, d);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        NRV_WACC(2);                                  // section 2: MFMA groups + fragment reads + DMA issue
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NRV_WACC(0);                                  // section 0: parked on vmcnt(0)
        __syncthreads();
        NRV_WACC(1);                                  // section 1: parked on the barrier
        if (kt == 0) { NRV_STAMP(1); NRV_WACC_MARK(); }
        const bool more = kt + 1 < nk;
        const char* sa = smem + cur * C::STAGE;
        // The next tile's DMA instructions are spread over the MFMA groups of this tile (each costs the issuing
        // wave tens to >100 cycles): issued in one burst after the barrier they stall both waves of every SIMD.
        // fragment reads run one MFMA group ahead of their use (register double buffering)
        constexpr int GH = C::MI / 2;                 // groups per k-step
        bf16x8_t bfr[2][C::NI], af[2][2];
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) bfr[0][ni] = lds_read_b128(sa + (b_rd + ni * 2048));
#pragma unroll
        for (int j = 0; j < 2; ++j) af[0][j] = lds_read_b128(sa + (a_rd + j * 2048));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ks = g / GH, mp = g % GH;
            if (g + 1 < NG) {
                const int ks1 = (g + 1) / GH, mp1 = (g + 1) % GH;
                if (mp1 == 0) {
#pragma unroll
                    for (int ni = 0; ni < C::NI; ++ni) bfr[ks1 & 1][ni] = lds_read_b128(sa + ((b_rd ^ (ks1 << 6)) + ni * 2048));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) af[(g + 1) & 1][j] = lds_read_b128(sa + ((a_rd ^ (ks1 << 6)) + (2 * mp1 + j) * 2048));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ni = 0; ni < C::NI; ++ni)
                    acc[2 * mp + j][ni] = mfma16(bfr[ks & 1][ni], af[g & 1][j], acc[2 * mp + j][ni]);
            if (more && g < NGD) {
#pragma unroll
                for (int d = g * ND / NGD; d < (g + 1) * ND / NGD; ++d) dma_one(cur ^ 1, kt + 1, d);
            }
        }
    }
    __syncthreads();     // every wave is done with the tile buffers: reuse them as epilogue patches
    NRV_STAMP(2);
    if (REMAP) epilogue_remap<EPI, OUT_F32, AUX_F32, C::MI>(acc, smem, p.e, m0 + wr * (C::MI * 16), n0 + wc * (C::NI * 16), lane, wave);
    else epilogue_lin<EPI, OUT_F32, AUX_F32, C::MI>(acc, reinterpret_cast<float*>(smem + wave * EPI_PATCH_BYTES), p.e, m0 + wr * (C::MI * 16), n0 + wc * (C::NI * 16), lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no-ops in the product: the hooks below are empty
    NRV_STAMP(3);
    NRV_STAMP_FLUSH_WG(4, tid);
    NRV_WACC_FLUSH(C::NWAVES, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// NT kernel, phased main loop (cdna_hip_programming.md "The 256^2 8-phase template", T3 + T4 + T5): the same output
// tile, wave layout, swizzle and epilogues as gemm_nt_kernel, another K loop.
//
//   * Each operand tile of a K-step is staged as two HALF-tiles: A half h = the rows (16 MI) wr + 8 MI h + [0, 8 MI) of both
//     wave rows, B half h = the columns 64 wc + 32 h + [0, 32) of the four wave columns, so that every wave owns one
//     quadrant (A half i) x (B half j) of its block per PHASE.  A K-step is four phases:
//         phase 0: read B0, A0 | quadrant (A0, B0)      phase 2: read A1 | quadrant (A1, B1)
//         phase 1: read B1     | quadrant (A0, B1)      phase 3: --      | quadrant (A1, B0)   (B0 stays in registers)
//     each phase = [fragment reads + ONE half-tile of LDS-DMA + counted vmcnt] s_barrier [MFMAs] s_barrier.
//   * Waves 4-7 (wave row 1 of a 2 x 4 layout, wave rows 2-3 of the 4 x 2 layout of the 384 x 128 tile; the SIMD partners of waves 0-3)
//     run ONE barrier behind waves 0-3: on every SIMD one wave is in
//     its MFMA section while the other reads fragments and issues DMA.
//   * Half-tile op n = 4 t + {A0, B0, B1, A1} of K-step t is issued in global phase n - 6 (a half is re-staged >= 2 phases
//     after its last fragment read in either wave group) and first read in phase >= n - 1: five phases (2.5 K-step
//     quarters) of flight.  The wait in phase g - 1 for what phase g reads leaves the four youngest ops (one of each
//     kind) in flight: ONE counted vmcnt value for the whole loop, never 0 before the last K-step.  RAW: every wave's
//     counted wait precedes a barrier that every reader passes before its read (also across the stagger).
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// vector-memory instructions of one full-block epilogue_lin per wave (stores + epilogue-operand loads; the bias loads are not
// counted: they are the oldest and a LOWER bound is what the counted waits behind an epilogue need)
template <int EPI, bool OUT_F32, bool AUX_F32, int MI>
constexpr int epilogue_vm_ops() {
    constexpr int CW = OUT_F32 ? 4 : 8, NIT = 16 / (64 / (64 / CW));
    constexpr bool HAS_AUX = EPI == NRV_EPI_BIAS_RESIDUAL || epi_dgelu(EPI);
    constexpr int AS = (EPI == NRV_EPI_BIAS_RESIDUAL && AUX_F32) ? 4 : (epi_q8(EPI) ? 1 : 2), AW = CW * AS / 4;
    return MI * NIT * (1 + (epi_gelu(EPI) ? 1 : 0) + (HAS_AUX ? (AW <= 4 ? 1 : AW / 4) : 0));
}

template <typename C>
constexpr int nt8_stage_bytes() { return (C::TBM + C::TBN) * 128; }
template <typename C>
constexpr bool nt8_patches_behind() { return 2 * nt8_stage_bytes<C>() + 8 * 4096 <= 160 * 1024; }
template <typename C>
constexpr int nt8_lds_bytes() { return 2 * nt8_stage_bytes<C>() + (nt8_patches_behind<C>() ? 8 : 4) * 4096; }

struct Nt8Tile {                     // what the staging ops and the epilogue of one output tile need (scalar registers)
    __amdgpu_buffer_rsrc_t ra0, ra1, rb0, rb1;
    unsigned ka1, kb1;
    int m0, n0;
};

// PERSISTENT over tiles: workgroup b computes the tiles b, b + gridDim.x, ... (one workgroup per CU).  The half-tile op
// pipeline runs THROUGH the tile boundary: the six phases of a tile's last 1.5 K-steps, which have no op of their own tile
// left to issue, issue ops 0 .. 5 of the NEXT tile (its K-step 0 whole, A0 and B0 of its K-step 1), so a tile's epilogue runs
// with the next tile's first operands landing behind it and the next K loop starts without a prologue, a workgroup
// turnaround or a cold start (measured on the round-3 one-tile-per-workgroup kernel: 1.7 - 2.7 us prologue + ~3 us between a
// workgroup's last store and its successor's first stamp, per 29 - 36 us tile at K = 768).
//   * LDS during an epilogue: both stage buffers are taken except the A1 / B1 halves of the stage the tile's last K-step
//     used.  The wave-private epilogue patches of waves 4-7 live in that A1 half (its next writer, op A1 of the next tile's
//     K-step 1, is issued behind a barrier every wave reaches after its epilogue), those of waves 0-3 in 16 KiB behind the
//     stage buffers.  The wave groups are re-aligned in front of the epilogue and staggered again behind it.
//   * vmcnt is in issue order for loads and stores alike, so the waits of the next tile's K-step 0 count the epilogue's own
//     stores as "younger ops that may stay in flight" (E of them for a full block; an edge block, whose epilogue skips
//     passes, falls back to the plain count and waits for its stores): the K loop does not wait for the store drain.
template <typename C, int EPI, bool OUT_F32, bool AUX_F32>
__global__ __launch_bounds__(C::THREADS, 2) void gemm_nt8_kernel(const GemmNTParams p) {
    static_assert(((C::WM == 2 && C::WN == 4) || (C::WM == 4 && C::WN == 2)) && C::NI == 4 && C::MI % 2 == 0, "2 x 4 or 4 x 2 waves, 64-column wave blocks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MH = C::MI / 2;                                   // row tiles per wave and half
    constexpr int AH = C::TBM / 2 * 128, BH = C::TBN / 2 * 128;     // bytes of a half-tile image (128-byte rows)
    constexpr int STG = 2 * AH + 2 * BH;
    constexpr int PA = AH / 1024, PB = BH / 1024;                   // 1-KiB DMA pieces (8 rows) per half
    static_assert(PB % 8 == 0 && PA % 4 == 0, "pieces divide over the waves");
    constexpr bool PATCH_BEHIND = nt8_patches_behind<C>();          // all eight patches behind the stage buffers, or waves 4-7 in an A half
    static_assert(PATCH_BEHIND || AH >= 4 * 4096, "four epilogue patches fit an A half");
    // pieces of an A half per wave: PA / 8, the remainder (MI = 10: 4 pieces) goes to waves 0-3 in half 0 and to waves 4-7
    // in half 1, so that every wave issues the same number per K-step and per window of four consecutive ops
    constexpr int NA = (PA + 7) / 8, NB = PB / 8;
    constexpr bool A_UNEVEN = (PA % 8) != 0;
    constexpr int W4 = (A_UNEVEN ? 2 * NA - 1 : 2 * NA) + 2 * NB;   // DMA instructions of four consecutive ops (one per kind)
    constexpr int E_OPS = epilogue_vm_ops<EPI, OUT_F32, AUX_F32, C::MI>();
    constexpr int W4E = W4 + E_OPS > 63 ? 63 : W4 + E_OPS;          // vmcnt is a 6-bit field: 63 outstanding = at most the 63 youngest

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / C::WN, wc = wave % C::WN;                 // the wave's block of the tile
    const int grp = wave >> 2, wq = wave & 3;                       // wave group (waves w and w + 4 share a SIMD) and index within it
    NRV_TILE_STAMP_VARS(wave);       // hooks: empty in the product (csrc/nrv_dev.hpp)
    NRV_WACC_VARS;
    NRV_TILE_STAMP();                // [0] start

    const int M = p.e.M, N = p.e.N;
    const int nk = p.K / BK;                                        // host: K % 64 == 0, nk >= 3
    const unsigned ntiles = (unsigned)p.ntiles;

    // tile t of this workgroup's sequence.  One descriptor per half: based at the half's first row / column, records = the
    // valid rows behind it, so that rows >= M - m0 / >= N - n0 read as zero and the per-lane offsets are the same for both
    // halves.  The K position is the scalar soffset, which the range check subtracts from the records: a half without a
    // valid row (records 0) keeps soffset 0.
    auto make_tile = [&](unsigned t, bool valid) {                  // !valid: descriptors without records (their DMA writes zeros)
        const unsigned id = xcd_remap(valid ? t : 0u, ntiles);
        int tm, tn;
        if (p.gn > 0 && p.tiles_n > p.gn) {
            const int tiles_m = (int)ntiles / p.tiles_n;
            const int gsize = tiles_m * p.gn;
            const int grp = id / gsize, within = id - grp * gsize;
            const int gw = (grp + 1) * p.gn <= p.tiles_n ? p.gn : p.tiles_n - grp * p.gn;
            tm = within / gw;
            tn = grp * p.gn + (within - tm * gw);
        } else {
            tm = id / p.tiles_n;
            tn = id - tm * p.tiles_n;
        }
        // the divisions above run on the vector ALU (reciprocal); bring the results back explicitly so that everything
        // derived from them (descriptors, masks) stays in scalar registers
        tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
        Nt8Tile d;
        d.m0 = tm * C::TBM; d.n0 = tn * C::TBN;
        const int ar0 = valid ? M - d.m0 : 0, br0 = valid ? N - d.n0 : 0;
        const int ar1 = ar0 - 8 * C::MI, br1 = br0 - 32;
        d.ra0 = make_rsrc(p.A + (long long)d.m0 * p.lda, (unsigned long long)ar0 * p.lda * 2ull);
        d.ra1 = make_rsrc(p.A + (long long)(d.m0 + (ar1 > 0 ? 8 * C::MI : 0)) * p.lda, (unsigned long long)(ar1 > 0 ? ar1 : 0) * p.lda * 2ull);
        d.rb0 = make_rsrc(p.B + (long long)d.n0 * p.ldb, (unsigned long long)br0 * p.ldb * 2ull);
        d.rb1 = make_rsrc(p.B + (long long)(d.n0 + (br1 > 0 ? 32 : 0)) * p.ldb, (unsigned long long)(br1 > 0 ? br1 : 0) * p.ldb * 2ull);
        // all-ones when the half has a valid row, by arithmetic: a select here is sunk through the cur / nxt hand-over as a
        // loop-carried i1, which lives in a lane mask and drags the masks into vector registers
        d.ka1 = (unsigned)__builtin_amdgcn_readfirstlane((-ar1) >> 31); d.kb1 = (unsigned)__builtin_amdgcn_readfirstlane((-br1) >> 31);
        asm volatile("" : "+s"(d.ka1), "+s"(d.kb1));                // ... and the readfirstlane is not sunk behind the hand-over either
        return d;
    };

    // staging: piece q = 8 i + wave of a half covers its local rows 8 q .. 8 q + 7; local row r' of A half h is tile row
    // (16 MI) (r' / (8 MI)) + 8 MI h + r' % (8 MI), local row r' of B half h is tile column 64 (r' / 32) + 32 h + r' % 32
    unsigned st_a[NA], st_b[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int q = i * 8 + wave;
        if (A_UNEVEN && i == NA - 1) q = (NA - 1) * 8 + wq;                  // issued by waves 0-3 (h = 0) / 4-7 (h = 1) only
        const int rl = q * 8 + (lane >> 3);
        const int r = (C::MI * 16) * (rl / (8 * C::MI)) + rl % (8 * C::MI);
        const int c = (lane & 7) ^ ((rl >> 1) & 7);
        st_a[i] = (unsigned)((long long)r * p.lda * 2) + c * 16;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int rl = (i * 8 + wave) * 8 + (lane >> 3);
        const int r = 64 * (rl >> 5) + (rl & 31);
        const int c = (lane & 7) ^ ((rl >> 1) & 7);
        st_b[i] = (unsigned)((long long)r * p.ldb * 2) + c * 16;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(smem));    // LDS byte address of the tile buffers
    // half-tile op `kind` (0: A0, 1: B0, 2: B1, 3: A1) of K-step t of tile d into stage buffer `par` (the parity of the
    // workgroup's running K-step count: the stage buffers alternate through the tile boundaries)
    auto stage_half = [&](auto kind_c, __amdgpu_buffer_rsrc_t r, unsigned mask, int t, int par) {
        constexpr int kind = decltype(kind_c)::value;
        constexpr bool isA = kind == 0 || kind == 3;
        constexpr int h = (kind == 2 || kind == 3) ? 1 : 0;
        const unsigned base = lds0 + par * STG + (isA ? h * AH : 2 * AH + h * BH);
        // uniform by construction; the readfirstlane costs nothing where hipcc sees that (the product) and keeps the operand a
        // scalar register where it does not (instrumented builds: a thread-0 branch in the tile loop)
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(t * (BK * 2)) & mask));
        if constexpr (isA) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (A_UNEVEN && i == NA - 1) {
                    if (grp == h) dma16s_at(r, base + ((NA - 1) * 8 + wq) * 1024, st_a[i], so);
                } else {
                    dma16s_at(r, base + (i * 8 + wave) * 1024, st_a[i], so);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) dma16s_at(r, base + (i * 8 + wave) * 1024, st_b[i], so);
        }
    };
    auto stage = [&](auto kind_c, const Nt8Tile& d, int t, int par) {
        constexpr int kind = decltype(kind_c)::value;
        if constexpr (kind == 0) stage_half(kind_c, d.ra0, ~0u, t, par);
        else if constexpr (kind == 1) stage_half(kind_c, d.rb0, ~0u, t, par);
        else if constexpr (kind == 2) stage_half(kind_c, d.rb1, d.kb1, t, par);
        else stage_half(kind_c, d.ra1, d.ka1, t, par);
    };

    // fragment read addresses (LDS byte addresses of k-step 0 / 1 in the CURRENT stage buffer, half 0; half 1 = + AH / + BH
    // as an immediate); they move to the other stage buffer in place after every K-step (no per-K-step copies)
    const int fr = lane & 15, fg = lane >> 4;
    const int swz = (fg ^ ((fr >> 1) & 7)) << 4;
    unsigned ard[2], brd[2];
    ard[0] = lds0 + (wr * (8 * C::MI) + fr) * 128 + swz;      ard[1] = ard[0] ^ 64u;
    brd[0] = lds0 + 2 * AH + (wc * 32 + fr) * 128 + swz;      brd[1] = brd[0] ^ 64u;
    int dstg = STG;                                                 // > 0: the current stage buffer is buffer 0
    auto next_stage = [&]() {
        ard[0] += dstg; ard[1] += dstg; brd[0] += dstg; brd[1] += dstg;
        dstg = -dstg;
    };

    f32x4_t acc[C::MI][4];                                          // started by the MFMAs of every tile's K-step 0 (C = 0)
    bf16x8_t a[MH][2], b0[2][2], b1[2][2];

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // Late starts against lockstep.  Equal tiles keep the 256 CUs in step, so their epilogues coincide: 256 x 327 KB (gelu) ask
    // for twice the HBM write rate while no CU computes, and nobody stores while all compute.  Workgroups that have one tile
    // fewer than the busiest ones start late, spread over 3/4 of a tile time: free (their last tile still ends before the
    // busiest workgroups' last), + 2.6 % on fc1 and + 4 % on dU of ViT-B/16.  Spreading the busiest workgroups too (30 / 60 % of
    // a tile time) only added the delay on every shape, fp32-residual epilogues included (profiles/r03_nt8_late_start_spread.txt).
    {
        const unsigned rem = ntiles % gridDim.x;
        if (ntiles > gridDim.x && rem != 0 && blockIdx.x >= rem) {
            const unsigned tile_cycles = (unsigned)nk * (C::MI * 330u) + 6000u;
            const unsigned span = tile_cycles - tile_cycles / 4;
            const unsigned naps = (blockIdx.x - rem) * (span / 8128u) / (gridDim.x - rem);  // s_sleep 127 = 8128 cycles
            for (unsigned i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
    Nt8Tile cur = make_tile(blockIdx.x, true), nxt = cur;
    int kb = 0;                                                     // parity of the K-steps this workgroup ran before the current tile

    // one phase of K-step kt of the current tile.  OP: 1 this phase's half-tile op is of the current tile, 2 of the next tile
    // (K-step index counted on behind the current tile's last).  Counted wait before the barrier: vmcnt WA if `alt` else WB
    // (-1: none); `alt` is wave-uniform, and only the wait sits behind the scalar branch
    auto phase = [&](auto P_c, auto OP_c, auto WA_c, auto WB_c, int kt, bool alt, auto FIRST_c) {
        constexpr int P = decltype(P_c)::value;
        constexpr bool FIRST = decltype(FIRST_c)::value;          // K-step 0 of a tile: the accumulators start from the constant 0
        constexpr int OP = decltype(OP_c)::value;
        constexpr int WA = decltype(WA_c)::value, WB = decltype(WB_c)::value;
        if constexpr (P == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nl = 0; nl < 2; ++nl) b0[nl][ks] = lds_read_b128_at(brd[ks] + nl * 2048);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ml = 0; ml < MH; ++ml) a[ml][ks] = lds_read_b128_at(ard[ks] + ml * 2048);
        } else if constexpr (P == 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nl = 0; nl < 2; ++nl) b1[nl][ks] = lds_read_b128_at(brd[ks] + BH + nl * 2048);
        } else if constexpr (P == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ml = 0; ml < MH; ++ml) a[ml][ks] = lds_read_b128_at(ard[ks] + AH + ml * 2048);
        }
        if constexpr (OP != 0) {
            // global phase g = 4 kt + P issues op g + 6: B1 / A1 of K-step kt + 1 in phases 0 / 1, A0 / B0 of kt + 2 in 2 / 3
            const int kk = kt + (P < 2 ? 1 : 2);
            const int par = (kb + kk) & 1;
            using KIND = std::integral_constant<int, P == 0 ? 2 : P == 1 ? 3 : P == 2 ? 0 : 1>;
            if constexpr (OP == 1) {
                stage(KIND{}, cur, kk, par);
            } else {                                              // the op pipeline runs on into the next tile: K-step kk - nk of `nxt`
                // in the hand-over loop (kt >= nk - 2) phases 2 and 3 always issue for the next tile (kk = kt + 2 >= nk); phases
                // 0 and 1 do in its second K-step only
                const bool over = P >= 2 ? true : kk >= nk;
                constexpr int kind = KIND::value;
                const __amdgpu_buffer_rsrc_t r = kind == 0 ? (over ? nxt.ra0 : cur.ra0) : kind == 1 ? (over ? nxt.rb0 : cur.rb0)
                                               : kind == 2 ? (over ? nxt.rb1 : cur.rb1) : (over ? nxt.ra1 : cur.ra1);
                const unsigned mask = kind == 2 ? (over ? nxt.kb1 : cur.kb1) : kind == 3 ? (over ? nxt.ka1 : cur.ka1) : ~0u;
                stage_half(KIND{}, r, mask, over ? kk - nk : kk, par);
            }
        }
        NRV_WACC(4 * P + 0);                                      // per phase P: section 0: fragment-read and DMA issue
        if constexpr (WA == WB) {
            if constexpr (WA >= 0) wait_vm<WA>();
        } else {
            if (alt) { if constexpr (WA >= 0) wait_vm<WA>(); }
            else { if constexpr (WB >= 0) wait_vm<WB>(); }
        }
        NRV_WACC(4 * P + 1);                                      // section 1: counted vmcnt wait
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0), through the builtin: hipcc's own counter restarts at 0
        NRV_WACC(4 * P + 2);                                      // section 2: barrier + fragment-read latency
        __builtin_amdgcn_sched_barrier(0);
        constexpr int mh = (P >= 2) ? MH : 0;                    // first row tile of the quadrant
        constexpr int nh = (P == 1 || P == 2) ? 2 : 0;           // first column tile
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ml = 0; ml < MH; ++ml)
#pragma unroll
                for (int nl = 0; nl < 2; ++nl)
                    acc[mh + ml][nh + nl] = mfma16((P == 1 || P == 2) ? b1[nl][ks] : b0[nl][ks], a[ml][ks],
                                                   (FIRST && ks == 0) ? f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[mh + ml][nh + nl]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        NRV_WACC(4 * P + 3);                                      // section 3: MFMA section + closing barrier
    };
    using OPC = std::integral_constant<int, 1>; using OPX = std::integral_constant<int, 2>;
    using T_ = std::true_type; using F_ = std::false_type;
    using WN_ = std::integral_constant<int, -1>;                    // no wait
    using WF = std::integral_constant<int, W4>;                     // steady state: four ops in flight
    using WE = std::integral_constant<int, W4E>;                    // ... and the previous tile's epilogue stores behind them
    // prologue (first tile only): ops 0 .. 5 (K-step 0 whole, A0 and B0 of K-step 1); A0, B0 of K-step 0 landed before the first barrier
    stage(I0{}, cur, 0, 0); stage(I1{}, cur, 0, 0); stage(I2{}, cur, 0, 0); stage(I3{}, cur, 0, 0); stage(I0{}, cur, 1, 1); stage(I1{}, cur, 1, 1);
    wait_vm<W4>();
    __builtin_amdgcn_s_barrier();
    if (NRV_TUNE_STAGGER(grp == 1)) __builtin_amdgcn_s_barrier();                     // stagger: waves 4-7 one barrier behind
    NRV_WACC_MARK();

    bool count_stores = false;       // the previous epilogue of this wave issued exactly E_OPS counted instructions
    for (unsigned t = blockIdx.x;;) {
        NRV_TILE_STAMP();            // [1 + 4 i] tile i: first K-step ready
        const unsigned tnext = t + gridDim.x;
        const bool has_next = tnext < ntiles;                       // uniform
        nxt = make_tile(tnext, has_next);
        // K-step 0: behind an epilogue its three waits also leave that epilogue's stores in flight
        phase(I0{}, OPC{}, WE{}, WF{}, 0, count_stores, T_{});
        phase(I1{}, OPC{}, WE{}, WF{}, 0, count_stores, T_{});
        phase(I2{}, OPC{}, WN_{}, WN_{}, 0, false, T_{});
        phase(I3{}, OPC{}, WE{}, WF{}, 0, count_stores, T_{});
        next_stage();
        // K-steps 1 .. nk - 3: every op belongs to the current tile
        int kt = 1;
        for (; kt < nk - 2; ++kt) {
            phase(I0{}, OPC{}, WF{}, WF{}, kt, false, F_{});
            phase(I1{}, OPC{}, WF{}, WF{}, kt, false, F_{});
            phase(I2{}, OPC{}, WN_{}, WN_{}, kt, false, F_{});
            phase(I3{}, OPC{}, WF{}, WF{}, kt, false, F_{});
            next_stage();
        }
        // K-steps nk - 2, nk - 1.  In the last six phases the op to issue belongs to the next tile (its ops 0 .. 5: K-step 0
        // whole, A0 and B0 of K-step 1): same schedule, one op per phase and one wait count, only the descriptor is selected.
        // Behind the workgroup's last tile those descriptors have no records: the six ops write zeros into free halves.
        // (A loop of its own rather than straight-line code: peeled in front of the epilogue it spilled accumulators.)
#pragma clang loop unroll(disable)
        for (; kt < nk; ++kt) {
            phase(I0{}, OPX{}, WF{}, WF{}, kt, false, F_{});
            phase(I1{}, OPX{}, WF{}, WF{}, kt, false, F_{});
            phase(I2{}, OPX{}, WN_{}, WN_{}, kt, false, F_{});
            phase(I3{}, OPX{}, WF{}, WF{}, kt, false, F_{});
            next_stage();
        }
        // re-align the wave groups for the epilogue (staggered, waves 4-7 would sit in their last barrier through the epilogue
        // of waves 0-3 and run theirs afterwards: the two epilogues one after the other, measured +2 .. 7 us per tile)
        NRV_TILE_STAMP();            // [2 + 4 i] K loop done
        if (NRV_TUNE_STAGGER(grp == 0)) __builtin_amdgcn_s_barrier();
        NRV_TILE_STAMP();            // [3 + 4 i] wave groups re-aligned
        // epilogue patches: waves 0-3 behind the stage buffers, waves 4-7 in the A1 half of the stage the last K-step used
        // (after next_stage() that is the buffer the read addresses do NOT point at)
        {
            const unsigned ybase = dstg > 0 ? (unsigned)STG : 0u;
            char* patch = smem + ((PATCH_BEHIND || grp == 0) ? 2 * STG + wave * 4096 : (int)ybase + AH + wq * 4096);
            const int row_base = cur.m0 + wr * (C::MI * 16), col_base = cur.n0 + wc * 64;
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));        // opaque per tile: the epilogue's per-lane offsets are recomputed here, not kept in registers through the K loops
            epilogue_lin<EPI, OUT_F32, AUX_F32, C::MI>(acc, reinterpret_cast<float*>(patch), p.e, row_base, col_base, lane_e);
            count_stores = M - row_base >= C::MI * 16 && N - col_base >= 64 && (!epi_gelu(EPI) || p.e.aux_out != nullptr);
        }
        NRV_TILE_STAMP();            // [4 + 4 i] epilogue issued
        if (!has_next) break;
        if (NRV_TUNE_STAGGER(grp == 1)) __builtin_amdgcn_s_barrier();                 // stagger again: waves 4-7 one barrier behind
        kb = (kb + nk) & 1;
        cur = nxt;
        t = tnext;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // stamp hooks below are empty in the product
    NRV_TILE_STAMP();                // last: stores drained
    NRV_WACC_FLUSH(C::NWAVES, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// TN kernel.  LDS tile image: [64 token rows][TBM (A) / TBN (B) cols] bf16, rows of 256 / 512 / 768 bytes; the 32-byte
// unit u of row R is stored at unit position u ^ (R & 7) (rows are multiples of 8 units, so the XOR stays inside the
// row): conflict-free ds_read_b64_tr_b16 (a 32-lane half reads 8 rows x 32 bytes).  Both operands are read with the same
// transposed pattern, so the k-order permutation inside a 32-deep MFMA step is the same for A and B.
//
// Tile configurations (8 waves as WM x WN, wave block (16 MI) x 64, 64 KiB of operands per K-tile in both):
//   TnCfg256: 256 x 256 -- the only TN tile (a 384 x 128 variant removed the 25-44 % padding of a 256 x 256 grid on the ViT-S
//             widths and won in isolated launches, but lost inside the step: profiles/r02_step_ab_in_process.txt)
// ---------------------------------------------------------------------------------------------
template <int WM_, int WN_, int MI_>
struct TnCfg {
    static constexpr int WM = WM_, WN = WN_, MI = MI_;
    static constexpr int TBM = WM * MI * 16, TBN = WN * 64;
    static constexpr int RA = TBM * 2, RB = TBN * 2;                      // bytes per token row of the LDS images
    static constexpr int A_BYTES = BK * RA, B_BYTES = BK * RB, STAGE = A_BYTES + B_BYTES;
    static constexpr int CA = A_BYTES / 1024 / 8, CB = B_BYTES / 1024 / 8;   // DMA instructions per wave and K-tile
    static constexpr int LDS = 2 * STAGE;
    static_assert(WM * WN == 8, "8 waves");
    static_assert(TBM % 128 == 0 && TBN % 128 == 0, "rows are multiples of 8 swizzle units");
    static_assert(A_BYTES % 8192 == 0 && B_BYTES % 8192 == 0, "whole DMA instructions per wave");
    static_assert(MI % 2 == 0, "row blocks are processed in pairs");
};
using TnCfg256 = TnCfg<2, 4, 8>;

template <typename C>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_kernel(const GemmTNParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / C::WN, wc = wave - wr * C::WN;

    const unsigned id = xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / p.tiles_mn;
    const int tile = id - split * p.tiles_mn;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * C::TBM, n0 = tn * C::TBN;
    const int M = p.e.M, N = p.e.N;

    const int t_begin = (split * p.kt_q + (split < p.kt_r ? split : p.kt_r)) * BK;
    int t_end = t_begin + (p.kt_q + (split < p.kt_r ? 1 : 0)) * BK;
    if (t_end > p.T) t_end = p.T;
    const int nk = t_end > t_begin ? (t_end - t_begin + BK - 1) / BK : 0;

    const long long arow0 = remap_row(t_begin, p.a_group, p.a_group_stride, p.a_row_offset);
    const bf16_t* abase = p.A + arow0 * p.lda + m0;
    const bf16_t* bbase = p.B + (long long)t_begin * p.ldb + n0;
    const int trem = t_end - t_begin;                       // token rows of this split (<= 0: nothing to do)
    const int acols = M - m0 < C::TBM ? M - m0 : C::TBM, bcols = N - n0 < C::TBN ? N - n0 : C::TBN;
    // plain case: the records end with the last valid column of the last token row of the split, so rows >= t_end read as
    // zero through the range check and a DMA costs no vector instruction (per-lane offset constant, K position = soffset).
    // row-remapped A (class-token slot of the patch embedding): per-DMA address arithmetic, window of 2 GiB.
    const bool a_remap = p.a_group > 0;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(abase, (a_remap || trem <= 0) ? 0x7fffffffull : ((unsigned long long)(trem - 1) * p.lda + acols) * 2ull);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(bbase, trem <= 0 ? 0ull : ((unsigned long long)(trem - 1) * p.ldb + bcols) * 2ull);

    // staging: DMA instruction i of this wave fills LDS bytes (8 i + wave) KiB .. +1 KiB of an operand image, lane-linear:
    // lane -> (token row R, 16-byte position within the row); the swizzle is applied to the SOURCE column
    int st_r[C::CA], st_col[C::CA];
    unsigned va0[C::CA], vb0[C::CB];
#pragma unroll
    for (int i = 0; i < C::CA; ++i) {
        const int L = (i * 8 + wave) * 1024 + lane * 16;
        const int R = L / C::RA, pos = (L - R * C::RA) >> 4;
        const int ul = (pos >> 1) ^ (R & 7);
        st_r[i] = R;
        st_col[i] = (ul * 2 + (pos & 1)) * 8;
        va0[i] = st_col[i] < acols ? (unsigned)(((long long)R * p.lda + st_col[i]) * 2) : NRV_OOB;
    }
#pragma unroll
    for (int i = 0; i < C::CB; ++i) {
        const int L = (i * 8 + wave) * 1024 + lane * 16;
        const int R = L / C::RB, pos = (L - R * C::RB) >> 4;
        const int ul = (pos >> 1) ^ (R & 7);
        const int col = (ul * 2 + (pos & 1)) * 8;
        vb0[i] = col < bcols ? (unsigned)(((long long)R * p.ldb + col) * 2) : NRV_OOB;
    }
    const unsigned a_step = (unsigned)(BK * p.lda * 2), b_step = (unsigned)(BK * p.ldb * 2);

    // one DMA instruction d (0 .. CA-1: A image, CA .. CA+CB-1: B image) of K-tile kt into stage buffer buf
    auto dma_one = [&](int buf, int kt, int d) {
        char* base = smem + buf * C::STAGE;
        const bool isA = d < C::CA;
        const int i = isA ? d : d - C::CA;
        if (isA) {
            if (a_remap) {
                const int t = t_begin + kt * BK + st_r[i];
                const long long ar = remap_row(t, p.a_group, p.a_group_stride, p.a_row_offset) - arow0;
                const unsigned va = (t < t_end && st_col[i] < acols) ? (unsigned)((ar * p.lda + st_col[i]) * 2) : NRV_OOB;
                dma16(ra, base + (i * 8 + wave) * 1024, va);
            } else {
                dma16s(ra, base + (i * 8 + wave) * 1024, va0[i], (unsigned)kt * a_step);
            }
        } else {
            dma16s(rb, base + C::A_BYTES + (i * 8 + wave) * 1024, vb0[isA ? 0 : i], (unsigned)kt * b_step);
        }
    };
    constexpr int ND = C::CA + C::CB;
    constexpr int NG = C::MI, GH = C::MI / 2;            // MFMA groups per K-tile: 2 k-steps x MI/2 row-block pairs
    constexpr int NGD = DMA_GROUPS < NG ? DMA_GROUPS : NG;      // groups that carry the next tile's DMA issue

    // transposed fragment read offsets: lane (g = l>>4, q = (l&15)>>2, pp = l&3) supplies row 4g+q (+16 r + 32 ks);
    // one offset register per 16-column block of the wave (its unit index is not a multiple of 8 in every configuration,
    // so the swizzle XOR is folded in here instead of at the read)
    const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
    const int Rl = 4 * fg + fq;
    const int x = Rl & 7;
    int a_off[C::MI], b_off[4];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi) a_off[mi] = Rl * C::RA + (((wr * C::MI + mi) ^ x) << 5) + fp * 8;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b_off[ni] = C::A_BYTES + Rl * C::RB + (((wc * 4 + ni) ^ x) << 5) + fp * 8;

    f32x4_t acc[C::MI][4];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fused bias gradient: db[m] = sum_t A[t,m] = (ones . A) on the MFMA; only the first column tile's workgroups do
    // it, and the WN column waves share the MI/2 row-block pairs (pair mp goes to wave column mp % WN)
    constexpr int NBP = (GH + C::WN - 1) / C::WN;
    const bool do_bias = p.bias_ws != nullptr && tn == 0;
    const u32x4_t ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_u);
    f32x4_t accb[NBP][2];
#pragma unroll
    for (int b = 0; b < NBP; ++b) accb[b][0] = accb[b][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto ld_a = [&](const char* sa, int ks, int mi) {
        const char* q = sa + a_off[mi] + ks * (32 * C::RA);
        return cat4(lds_read_tr16_b64(q), lds_read_tr16_b64(q + 16 * C::RA));
    };
    auto ld_b = [&](const char* sa, int ks, int ni) {
        const char* q = sa + b_off[ni] + ks * (32 * C::RB);
        return cat4(lds_read_tr16_b64(q), lds_read_tr16_b64(q + 16 * C::RB));
    };

    if (nk > 0) {
#pragma unroll
        for (int d = 0; d < ND; ++d) dma_one(0, 0, d);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool more = kt + 1 < nk;
        const char* sa = smem + cur * C::STAGE;
        // same schedule as the NT kernel: fragment reads one MFMA group ahead, next tile's DMA issue spread over
        // the first groups
        bf16x8_t bfr[2][4], af[2][2];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bfr[0][ni] = ld_b(sa, 0, ni);
#pragma unroll
        for (int j = 0; j < 2; ++j) af[0][j] = ld_a(sa, 0, j);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ks = g / GH, mp = g % GH;
            if (g + 1 < NG) {
                const int ks1 = (g + 1) / GH, mp1 = (g + 1) % GH;
                if (mp1 == 0) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) bfr[ks1 & 1][ni] = ld_b(sa, ks1, ni);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) af[(g + 1) & 1][j] = ld_a(sa, ks1, 2 * mp1 + j);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[2 * mp + j][ni] = mfma16(bfr[ks & 1][ni], af[g & 1][j], acc[2 * mp + j][ni]);
            if (do_bias && (mp % C::WN) == wc) {
#pragma unroll
                for (int j = 0; j < 2; ++j) accb[mp / C::WN][j] = mfma16(ones, af[g & 1][j], accb[mp / C::WN][j]);
            }
            if (more && g < NGD) {
#pragma unroll
                for (int d = g * ND / NGD; d < (g + 1) * ND / NGD; ++d) dma_one(cur ^ 1, kt + 1, d);
            }
        }
    }
    __syncthreads();
    if (do_bias && lane < 16) {
#pragma unroll
        for (int b = 0; b < NBP; ++b) {
            const int mp = b * C::WN + wc;
            if (mp < GH) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int m = m0 + wr * (C::MI * 16) + (2 * mp + j) * 16 + lane;
                    if (m < M) p.bias_ws[(long long)split * M + m] = accb[b][j][0];
                }
            }
        }
    }
    EpiParams e = p.e;
    e.C = reinterpret_cast<float*>(p.e.C) + (long long)split * p.slab_stride;
    epilogue_lin<NRV_EPI_NONE, true, true, C::MI>(acc, reinterpret_cast<float*>(smem + wave * EPI_PATCH_BYTES), e, m0 + wr * (C::MI * 16), n0 + wc * 64, lane);
}

// ---------------------------------------------------------------------------------------------
// TN kernel, phased K loop: the schedule of gemm_nt8_kernel (four phases per K-step of 64 token rows, waves 4-7 one barrier
// behind, half-tile LDS-DMA ops issued six phases ahead of their first read, ONE counted vmcnt value) on the TN operand
// images.  A half image = [64 token rows][the 8 MI columns of half h of both wave rows] (256-byte rows, 32-byte units
// swizzled by the row), B half image = [64 token rows][the 32 columns of half h of the four wave columns]; both are read
// with ds_read_b64_tr_b16.  Host: no row remap, every split has at least three K-steps.
// ---------------------------------------------------------------------------------------------
// one unit of TN work: the 256 x 256 output tile (m0, n0) of C = A^T B over the token rows [t_begin, t_begin + trem), nk K-steps
// (the last one may be partial), written as an fp32 tile to `out` (leading dimension out_ld: a slab of the whole matrix, or a
// dense tile slot) and, with bias_out, the column sums of A of those rows to bias_out[0 .. 255] (first column tile only)
struct TnSeg {
    const bf16_t* A;
    const bf16_t* B;
    long long lda, ldb;
    int M, N, m0, n0;
    int t_begin, trem, nk;
    float* out;                      // element (m0, n0) of the destination
    long long out_ld;
    float* bias_out;                 // element m0 of the destination, or nullptr
};

template <typename C>
__device__ __forceinline__ void tn8_segment(const TnSeg& sg, char* smem) {
    static_assert(C::WM == 2 && C::WN == 4 && C::MI % 2 == 0, "2 x 4 waves");
    constexpr int MH = C::MI / 2;
    constexpr int RAH = C::TBM, RBH = C::TBN;                 // bytes per token row of a half image (TBM / 2 columns x 2 bytes)
    constexpr int AH = BK * RAH, BH = BK * RBH, STG = 2 * AH + 2 * BH;
    constexpr int NA = AH / 8192, NB = BH / 8192;             // DMA instructions per wave and half-tile op
    static_assert(AH % 8192 == 0 && BH % 8192 == 0 && RAH % 256 == 0 && RBH % 256 == 0, "whole pieces, rows of 8 units");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    NRV_WACC_VARS;                   // hooks: empty in the product (csrc/nrv_dev.hpp)

    const int m0 = sg.m0, n0 = sg.n0;
    const int M = sg.M, N = sg.N;
    const int nk = sg.nk;                                     // >= 3 (host)
    const int t_begin = sg.t_begin;
    const int trem = sg.trem;                                 // > 64 (nk - 1) >= 128
    struct { const bf16_t* A; const bf16_t* B; long long lda, ldb; } p = {sg.A, sg.B, sg.lda, sg.ldb};

    const int acols = M - m0 < C::TBM ? M - m0 : C::TBM, bcols = N - n0 < C::TBN ? N - n0 : C::TBN;
    // records end with the last valid column of the segment's last token row: rows beyond it read as zero, the K position is
    // the scalar soffset (< the records for every K-step of the segment)
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A + (long long)t_begin * p.lda + m0, ((unsigned long long)(trem - 1) * p.lda + acols) * 2ull);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B + (long long)t_begin * p.ldb + n0, ((unsigned long long)(trem - 1) * p.ldb + bcols) * 2ull);

    // staging: piece 8 i + wave of a half image is lane-linear: lane -> (token row R, 16-byte position); the unit swizzle is
    // applied to the SOURCE column; column c' of A half h is tile column (16 MI)(c' / (8 MI)) + 8 MI h + c' % (8 MI), of B
    // half h tile column 64 (c' / 32) + 32 h + c' % 32
    unsigned va[2][NA], vb[2][NB];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int L = (i * 8 + wave) * 1024 + lane * 16;
            const int R = L / RAH, pos = (L - R * RAH) >> 4;
            const int ch = ((((pos >> 1) ^ (R & 7)) << 1) + (pos & 1)) * 8;
            const int col = (16 * C::MI) * (ch / (8 * C::MI)) + 8 * C::MI * h + ch % (8 * C::MI);
            va[h][i] = col < acols ? (unsigned)(((long long)R * p.lda + col) * 2) : NRV_OOB;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int L = (i * 8 + wave) * 1024 + lane * 16;
            const int R = L / RBH, pos = (L - R * RBH) >> 4;
            const int ch = ((((pos >> 1) ^ (R & 7)) << 1) + (pos & 1)) * 8;
            const int col = 64 * (ch >> 5) + 32 * h + (ch & 31);
            vb[h][i] = col < bcols ? (unsigned)(((long long)R * p.ldb + col) * 2) : NRV_OOB;
        }
    }
    const unsigned a_step = (unsigned)(BK * p.lda * 2), b_step = (unsigned)(BK * p.ldb * 2);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(smem));
    auto stage = [&](auto kind_c, int t) {                    // op kind 0: A0, 1: B0, 2: B1, 3: A1 of K-step t
        constexpr int kind = decltype(kind_c)::value;
        constexpr bool isA = kind == 0 || kind == 3;
        constexpr int h = (kind == 2 || kind == 3) ? 1 : 0;
        const unsigned base = lds0 + (t & 1) * STG + (isA ? h * AH : 2 * AH + h * BH);
        if constexpr (isA) {
#pragma unroll
            for (int i = 0; i < NA; ++i) dma16s_at(ra, base + (i * 8 + wave) * 1024, va[h][i], (unsigned)t * a_step);
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) dma16s_at(rb, base + (i * 8 + wave) * 1024, vb[h][i], (unsigned)t * b_step);
        }
    };

    // transposed fragment reads: lane (g = l >> 4, q = (l & 15) >> 2, pp = l & 3) supplies row 4 g + q (+ 16 for the second
    // read, + 32 per k-step), unit (16-column block) u of the wave ^ (row & 7), 8 pp bytes into it
    const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
    const int Rl = 4 * fg + fq;
    const int x = Rl & 7;
    unsigned a_off[MH], b_off[2];
#pragma unroll
    for (int ml = 0; ml < MH; ++ml) a_off[ml] = lds0 + Rl * RAH + (((wr * MH + ml) ^ x) << 5) + fp * 8;
#pragma unroll
    for (int nl = 0; nl < 2; ++nl) b_off[nl] = lds0 + 2 * AH + Rl * RBH + (((wc * 2 + nl) ^ x) << 5) + fp * 8;
    int dstg = STG;
    auto next_stage = [&]() {
#pragma unroll
        for (int ml = 0; ml < MH; ++ml) a_off[ml] += dstg;
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) b_off[nl] += dstg;
        dstg = -dstg;
    };
    auto ld_a = [&](int h, int ml, int ks) {
        const unsigned q = a_off[ml] + h * AH + ks * (32 * RAH);
        return cat4(lds_read_tr16_b64_at(q), lds_read_tr16_b64_at(q + 16 * RAH));
    };
    auto ld_b = [&](int h, int nl, int ks) {
        const unsigned q = b_off[nl] + h * BH + ks * (32 * RBH);
        return cat4(lds_read_tr16_b64_at(q), lds_read_tr16_b64_at(q + 16 * RBH));
    };

    f32x4_t acc[C::MI][4];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t a[MH][2], bs[2][2][2];            // as in gemm_nt8_kernel: bs[q] / bs[q ^ 1] = B0 / B1 of a K-step of parity q

    // fused bias gradient db[m] = sum_t A[t, m] = (ones . A) on the MFMA, first column tile only; wave column wc takes the row
    // blocks 2 wc, 2 wc + 1 of its wave row: half wc >> 1, fragments (wc & 1) * MH / 2 + {0, 1}
    static_assert(MH == 4, "bias-gradient block assignment");
    const bool do_bias = sg.bias_out != nullptr;
    const u32x4_t ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_u);
    f32x4_t accb[2];
    accb[0] = accb[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // the phase of gemm_nt8_kernel (reads: A0 | B1 | A1 | next B0; ops: A1 (kt + 1) | B0 | A0 | B1 (kt + 2); five phases of flight)
    auto phase = [&](auto P_c, auto PAR_c, auto ST_c, auto WAIT_c, auto RD_c, auto MF_c, int kt) {
        constexpr int P = decltype(P_c)::value, PAR = decltype(PAR_c)::value;
        constexpr bool ST = decltype(ST_c)::value, RD = decltype(RD_c)::value, MF = decltype(MF_c)::value;
        constexpr int WAIT = decltype(WAIT_c)::value;
        if constexpr (P == 3 && MF) next_stage();
        if constexpr (RD) {
            if constexpr (P == 0) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int ml = 0; ml < MH; ++ml) a[ml][ks] = ld_a(0, ml, ks);
            } else if constexpr (P == 1) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int nl = 0; nl < 2; ++nl) bs[PAR ^ 1][nl][ks] = ld_b(1, nl, ks);
            } else if constexpr (P == 2) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int ml = 0; ml < MH; ++ml) a[ml][ks] = ld_a(1, ml, ks);
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int nl = 0; nl < 2; ++nl) bs[PAR ^ 1][nl][ks] = ld_b(0, nl, ks);
            }
        }
        if constexpr (ST) {
            if constexpr (P == 0) stage(I3{}, kt + 1);
            else if constexpr (P == 1) stage(I1{}, kt + 2);
            else if constexpr (P == 2) stage(I0{}, kt + 2);
            else stage(I2{}, kt + 2);
        }
        NRV_WACC(4 * P + 0);                                      // section 0: fragment reads + DMA issue
        if constexpr (WAIT >= 0) wait_vm<WAIT>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        NRV_WACC(4 * P + 2);                                      // section 2: counted vmcnt wait + barrier + fragment-read latency
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MF) {
            constexpr int mh = (P >= 2) ? MH : 0;
            constexpr int nh = (P == 1 || P == 2) ? 2 : 0;
            constexpr int bq = (P == 1 || P == 2) ? (PAR ^ 1) : PAR;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ml = 0; ml < MH; ++ml)
#pragma unroll
                    for (int nl = 0; nl < 2; ++nl)
                        acc[mh + ml][nh + nl] = mfma16(bs[bq][nl][ks], a[ml][ks], acc[mh + ml][nh + nl]);
            NRV_WACC(4 * P + 1);                                  // section 1: the quadrant's 16 MFMAs issued
            if constexpr (P == 0 || P == 2) {
                if (do_bias && (wc >> 1) == P / 2) {              // uniform
                    if (wc & 1) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int j = 0; j < 2; ++j) accb[j] = mfma16(ones, a[2 + j][ks], accb[j]);
                    } else {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int j = 0; j < 2; ++j) accb[j] = mfma16(ones, a[j][ks], accb[j]);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        NRV_WACC(4 * P + 3);                                      // section 3: bias-gradient MFMAs + closing barrier
    };
    using T = std::true_type; using F = std::false_type;
    using WN_ = std::integral_constant<int, -1>;
    using WE = std::integral_constant<int, 3 * NA + 2 * NB>;        // windows of the five youngest ops (gemm_nt8_kernel)
    using WO = std::integral_constant<int, 2 * NA + 3 * NB>;
    using WT1 = std::integral_constant<int, 2 * NA + 2 * NB>;
    using WT2 = std::integral_constant<int, 2 * NA + NB>;
    using WT3 = std::integral_constant<int, NB + NA>;
    using WL0 = std::integral_constant<int, NA>;
    using W0 = std::integral_constant<int, 0>;
    auto kstep = [&](auto PAR_c, int kt) {
        phase(I0{}, PAR_c, T{}, WE{}, T{}, T{}, kt);
        phase(I1{}, PAR_c, T{}, WO{}, T{}, T{}, kt);
        phase(I2{}, PAR_c, T{}, WE{}, T{}, T{}, kt);
        phase(I3{}, PAR_c, T{}, WO{}, T{}, T{}, kt);
    };

    stage(I1{}, 0); stage(I0{}, 0); stage(I2{}, 0); stage(I3{}, 0); stage(I1{}, 1); stage(I0{}, 1); stage(I2{}, 1);
    wait_vm<3 * NA + 3 * NB>();
    __builtin_amdgcn_s_barrier();
    if (NRV_TUNE_STAGGER(wr == 1)) __builtin_amdgcn_s_barrier();
    NRV_WACC_MARK();

    const int nmain = nk - 2;
    int kt = 0;
    if (nmain & 1) {
        phase(I3{}, I0{}, F{}, WO{}, T{}, F{}, -1);
        kstep(I1{}, 0);
        kt = 1;
    } else {
        phase(I3{}, I1{}, F{}, WO{}, T{}, F{}, -1);
    }
    for (; kt < nmain; kt += 2) {
        kstep(I0{}, kt);
        kstep(I1{}, kt + 1);
    }
    phase(I0{}, I0{}, T{}, WE{}, T{}, T{}, kt);
    phase(I1{}, I0{}, F{}, WT1{}, T{}, T{}, kt);
    phase(I2{}, I0{}, F{}, WT2{}, T{}, T{}, kt);
    phase(I3{}, I0{}, F{}, WT3{}, T{}, T{}, kt);
    ++kt;
    phase(I0{}, I1{}, F{}, WL0{}, T{}, T{}, kt);
    phase(I1{}, I1{}, F{}, W0{}, T{}, T{}, kt);
    phase(I2{}, I1{}, F{}, WN_{}, T{}, T{}, kt);
    phase(I3{}, I1{}, F{}, WN_{}, F{}, T{}, kt);
    if (NRV_TUNE_STAGGER(wr == 0)) __builtin_amdgcn_s_barrier();
    NRV_WACC_FLUSH(8, wave, lane);

    if (do_bias && lane < 16) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ml = wr * (C::MI * 16) + (2 * wc + j) * 16 + lane;
            if (m0 + ml < M) sg.bias_out[ml] = accb[j][0];
        }
    }
    // the destination is addressed relative to the tile: rows / columns beyond the matrix are dropped by the range check
    EpiParams e;
    e.C = sg.out; e.bias = nullptr; e.aux = nullptr; e.aux_out = nullptr;
    e.ldc = sg.out_ld; e.ld_aux = 0; e.ld_aux_out = 0;
    e.M = acols; e.N = bcols; e.aux_row_mod = 0; e.out_group = 0; e.out_group_stride = 0; e.out_row_offset = 0;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));            // keeps the epilogue's per-lane address arithmetic below the K loop
    epilogue_lin<NRV_EPI_NONE, true, true, C::MI>(acc, reinterpret_cast<float*>(smem + wave * EPI_PATCH_BYTES), e, wr * (C::MI * 16), wc * 64, lane_e);
}

template <typename C>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_tn8_kernel(const GemmTNParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned id = xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / p.tiles_mn;
    const int tile = id - split * p.tiles_mn;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    TnSeg sg;
    sg.A = p.A; sg.B = p.B; sg.lda = p.lda; sg.ldb = p.ldb;
    sg.M = p.e.M; sg.N = p.e.N; sg.m0 = tm * C::TBM; sg.n0 = tn * C::TBN;
    sg.nk = p.kt_q + (split < p.kt_r ? 1 : 0);               // >= 3 (host)
    sg.t_begin = (split * p.kt_q + (split < p.kt_r ? split : p.kt_r)) * BK;
    int t_end = sg.t_begin + sg.nk * BK;
    if (t_end > p.T) t_end = p.T;
    sg.trem = t_end - sg.t_begin;
    sg.out = reinterpret_cast<float*>(p.e.C) + (long long)split * p.slab_stride + (long long)sg.m0 * p.e.ldc + sg.n0;
    sg.out_ld = p.e.ldc;
    sg.bias_out = (p.bias_ws != nullptr && tn == 0) ? p.bias_ws + (long long)split * p.e.M + sg.m0 : nullptr;
    tn8_segment<C>(sg, smem);
}

// ---------------------------------------------------------------------------------------------
// Grouped TN GEMM: ALL weight gradients of a layer in one launch.  The 256 x 256 output tiles of up to four problems (same token
// count T) are numbered consecutively; one workgroup per CU, every one with the same number of K-steps (Lc = ceil(ntiles nk / W)):
//   * COHORTS: workgroup v < F ntiles computes tile v % ntiles over the K-steps [c Lc, (c + 1) Lc), c = v / ntiles.  All
//     workgroups of a cohort start at the same token row and move in step, so at any time the chip reads F (+ 1) token
//     positions and every operand row is fetched from HBM once per cohort and shared through L2 / Infinity Cache by the tiles
//     that need it -- as in one split-K launch per gradient.  (A first version gave every workgroup ONE contiguous range of
//     the flattened (tile, K-step) space: 256 workgroups at 256 token positions, no operand row shared, 5.6 GB of reads per
//     ViT-B/16 layer instead of 1.2 -- HBM-bound at 6 TB/s, the TN time went from 8.4 to 12.0 ms per step.)
//   * REMAINDER: the K-steps [F Lc, nk) of all tiles (R per tile) are split evenly, tile after tile, over the Wr = W - F ntiles
//     workgroups that are left (stream-K: a workgroup takes a contiguous range of that space as up to segs_r segments, range
//     ends snapped onto tile boundaries within 3 K-steps so that every segment has the >= 3 K-steps the phased loop needs).
// Every segment writes its partial tile to a dense slot; tng_fixup_kernel adds a tile's slots in K order into C: deterministic,
// no atomics, no waiting on other workgroups.  Against one split-K launch per gradient: every CU is busy whatever the tile
// counts (a 768 x 768 gradient has 9 tiles), F + 1..2 partials per tile instead of 7 - 64, one reduction launch per layer.
// ---------------------------------------------------------------------------------------------
constexpr int TNG_MAX = 4;                  // problems per launch

struct TngProblem {
    const bf16_t* A;
    const bf16_t* B;
    float* C;
    float* dbias;                           // or nullptr
    long long lda, ldb, ldc;
    int M, N;
    int tiles_n, tile0;                     // column tiles; global index of the problem's first tile
    float beta, dbias_beta;
};

struct GemmTNGParams {
    TngProblem pr[TNG_MAX];
    int nprob, ntiles, nk, T, W;
    int Lc, F, Wr, R, segs_r;               // cohort length, cohorts, remainder workgroups, remainder K-steps per tile, slots per remainder workgroup
    float* slots;                           // [F ntiles + Wr segs_r][256 * 256] partial tiles
    float* bias_slots;                      // [F ntiles + Wr segs_r][256] partial column sums
};

// remainder space: units = ntiles * R, workgroup j of Wr takes [bound(j), bound(j + 1))
__host__ __device__ __forceinline__ long long tng_bound(long long j, long long units, int Wr, int R) {
    long long b = j * units / Wr;
    const int r = (int)(b % R);
    if (r < 3) b -= r;
    else if (R - r < 3) b += R - r;
    return b;
}

__device__ __forceinline__ int tng_problem_of(const GemmTNGParams& p, int tile) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < TNG_MAX; ++i)
        if (i < p.nprob && tile >= p.pr[i].tile0) pi = i;
    return pi;
}

template <typename C>
__device__ __forceinline__ void tng_run(const GemmTNGParams& p, int tile, int k0, int k1, int slot, char* smem) {
    const TngProblem& q = p.pr[tng_problem_of(p, tile)];
    const int lt = tile - q.tile0;
    const int tm = lt / q.tiles_n, tn = lt - tm * q.tiles_n;
    TnSeg sg;
    sg.A = q.A; sg.B = q.B; sg.lda = q.lda; sg.ldb = q.ldb;
    sg.M = q.M; sg.N = q.N; sg.m0 = tm * C::TBM; sg.n0 = tn * C::TBN;
    sg.nk = k1 - k0;
    sg.t_begin = k0 * BK;
    const int t_end = k1 * BK < p.T ? k1 * BK : p.T;
    sg.trem = t_end - sg.t_begin;
    sg.out = p.slots + (long long)slot * (C::TBM * C::TBN);
    sg.out_ld = C::TBN;
    sg.bias_out = (q.dbias != nullptr && tn == 0) ? p.bias_slots + (long long)slot * C::TBM : nullptr;
    tn8_segment<C>(sg, smem);
}

template <typename C>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_tng_kernel(const GemmTNGParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // consecutive logical workgroups (one cohort's consecutive tiles: shared operand panels) sit on one XCD
    const int v = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int ncoh = p.F * p.ntiles;
    if (v < ncoh) {
        const int c = v / p.ntiles, tile = v - c * p.ntiles;
        const int k0 = c * p.Lc;
        const int k1 = (c == p.F - 1 && p.Wr == 0) ? p.nk : k0 + p.Lc;       // no remainder workgroups: the last cohort runs to the end
        tng_run<C>(p, tile, k0, k1, v, smem);
        return;
    }
    const int j = v - ncoh;
    if (j >= p.Wr) return;
    const long long units = (long long)p.ntiles * p.R;
    const long long b0 = tng_bound(j, units, p.Wr, p.R), b1 = tng_bound(j + 1, units, p.Wr, p.R);
    const int first_tile = (int)(b0 / p.R);
    const int kr = p.F * p.Lc;                                               // first K-step of the remainder
    for (long long u = b0; u < b1;) {                                        // workgroup-uniform
        const int tile = (int)(u / p.R);
        const int r0 = (int)(u - (long long)tile * p.R);
        const long long tile_end = (long long)(tile + 1) * p.R;
        const int r1 = (int)((b1 < tile_end ? b1 : tile_end) - (long long)tile * p.R);
        tng_run<C>(p, tile, kr + r0, kr + r1, ncoh + j * p.segs_r + (tile - first_tile), smem);
        u = (long long)tile * p.R + r1;
        __syncthreads();                                                     // the epilogue's LDS patches are the next segment's stage buffers
    }
}

// C tile = beta * C + the tile's partial slots in K order (cohort 0 .. F - 1, then the remainder workgroups in index order).
// Block = 4 waves on 64 consecutive 16-byte chunks of one tile; wave g sums the contributors g, g + 4, ..., the four sums meet in
// LDS and are added in the fixed order ((p0 + p1) + (p2 + p3)).  The first block of a first-column tile also reduces the bias slots.
__global__ __launch_bounds__(256) void tng_fixup_kernel(const GemmTNGParams p) {
    __shared__ f32x4_t part[4][64];
    constexpr int TILE = 256, CHUNKS = TILE * TILE / 4, BLOCKS_PER_TILE = CHUNKS / 64;
    const int tile = blockIdx.x / BLOCKS_PER_TILE, blk = blockIdx.x - tile * BLOCKS_PER_TILE;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const TngProblem& q = p.pr[tng_problem_of(p, tile)];
    const int lt = tile - q.tile0;
    const int tm = lt / q.tiles_n, tn = lt - tm * q.tiles_n;
    const int ncoh = p.F * p.ntiles;
    // remainder contributors: the workgroups whose unit range meets [tile * R, (tile + 1) * R)
    int j0 = 0, j1 = -1;
    const long long units = (long long)p.ntiles * p.R;
    if (p.Wr > 0) {
        const long long u0 = (long long)tile * p.R, u1 = u0 + p.R;
        j0 = (int)(u0 * p.Wr / units);
        while (j0 > 0 && tng_bound(j0, units, p.Wr, p.R) > u0) --j0;
        while (j0 + 1 < p.Wr && tng_bound(j0 + 1, units, p.Wr, p.R) <= u0) ++j0;
        j1 = j0;
        while (j1 + 1 < p.Wr && tng_bound(j1 + 1, units, p.Wr, p.R) < u1) ++j1;
    }
    const int ncontrib = p.F + (j1 - j0 + 1);
    auto slot_of = [&](int i) -> long long {                     // contributor i in K order
        if (i < p.F) return (long long)i * p.ntiles + tile;
        const int j = j0 + (i - p.F);
        return (long long)ncoh + (long long)j * p.segs_r + (tile - (int)(tng_bound(j, units, p.Wr, p.R) / p.R));
    };

    const int c = blk * 64 + lane;                               // chunk of the tile: row c / 64, columns 4 (c % 64) ..
    const int row = c >> 6, col = (c & 63) << 2;
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    for (int i = g; i < ncontrib; i += 4)
        s += __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p.slots + slot_of(i) * (TILE * TILE) + row * TILE + col));
    part[g][lane] = s;
    __syncthreads();
    const int m = tm * TILE + row, n = tn * TILE + col;
    if (g == 0 && m < q.M && n < q.N) {                          // N % 4 == 0 (host): a chunk is inside or outside
        f32x4_t t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        float* dst = q.C + (long long)m * q.ldc + n;
        if (q.beta != 0.f) t += *reinterpret_cast<const f32x4_t*>(dst) * q.beta;
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4_t*>(dst));
    }
    if (blk == 0 && tn == 0 && q.dbias != nullptr) {
        const int mm = tm * TILE + threadIdx.x;
        if (mm < q.M) {
            float b = 0.f;
            for (int i = 0; i < ncontrib; ++i) b += p.bias_slots[slot_of(i) * TILE + threadIdx.x];
            q.dbias[mm] = q.dbias_beta != 0.f ? q.dbias_beta * q.dbias[mm] + b : b;
        }
    }
}

// C = beta * C + sum_s slab[s].  A block of 4 waves owns 64 consecutive 16-byte chunks of C; wave g sums the slabs
// s = g, g + 4, ... (4 independent loads in flight per lane), the four partial sums meet in LDS and are added in the fixed
// order ((p0 + p1) + (p2 + p3)): deterministic, and 4 x the loads in flight of one thread per chunk (the slabs of a ViT-S
// weight gradient are 25-85 x 0.6-2.4 MB: the kernel is latency-bound, not bandwidth-bound).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, long long slab_stride, int splits,
                                                            float* __restrict__ C, long long ldc, int M, int N, float beta,
                                                            const float* __restrict__ bias_ws, float* __restrict__ dbias, float dbias_beta) {
    __shared__ f32x4_t part[4][64];
    if (bias_ws != nullptr) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
            float s = 0.f;
            for (int k = 0; k < splits; ++k) s += bias_ws[(long long)k * M + i];
            dbias[i] = dbias_beta != 0.f ? dbias_beta * dbias[i] + s : s;
        }
    }
    if (slabs == nullptr) return;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long long n4 = N >> 2;
    const long long total = (long long)M * n4;
    for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {     // block-uniform
        const long long i = base + lane;
        const bool ok = i < total;
        const long long m = ok ? i / n4 : 0, c = ok ? (i - m * n4) * 4 : 0;
        const float* src = slabs + m * N + c;
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
        if (ok) {
            int k = g;
            for (; k + 12 < splits; k += 16) {
                s0 += __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src + (long long)k * slab_stride));
                s1 += __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src + (long long)(k + 4) * slab_stride));
                s2 += __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src + (long long)(k + 8) * slab_stride));
                s3 += __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src + (long long)(k + 12) * slab_stride));
            }
            for (; k < splits; k += 4) s0 += __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src + (long long)k * slab_stride));
        }
        part[g][lane] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (g == 0 && ok) {
            f32x4_t s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
            float* dst = C + m * ldc + c;
            if (beta != 0.f) s += *reinterpret_cast<const f32x4_t*>(dst) * beta;
            __builtin_nontemporal_store(s, reinterpret_cast<f32x4_t*>(dst));      // the gradient buffer is next read by the clip norm, after the whole backward
        }
        __syncthreads();
    }
}

template <typename KernelT>
int set_lds(KernelT k, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <typename C, int EPI, bool OUT_F32, bool AUX_F32, bool REMAP>
int launch_nt_cfg(GemmNTParams p, hipStream_t s) {
    static int attr = set_lds(gemm_nt_kernel<C, EPI, OUT_F32, AUX_F32, REMAP>, C::LDS);
    if (attr != 0) return attr;
    const int tiles_m = (int)nrv_cdiv(p.e.M, C::TBM), tiles_n = (int)nrv_cdiv(p.e.N, C::TBN);
    p.tiles_n = tiles_n;
    p.gn = 4;               // column groups of 4 tiles per XCD (swept 2, 3, 4, 6, 12, off in round 1: 8192^3 1111 -> 1336 TFLOP/s)
    hipLaunchKernelGGL((gemm_nt_kernel<C, EPI, OUT_F32, AUX_F32, REMAP>), dim3(tiles_m * tiles_n), dim3(C::THREADS), C::LDS, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}

int device_cus();

template <typename C, int EPI, bool OUT_F32, bool AUX_F32>
int launch_nt8_cfg(GemmNTParams p, hipStream_t s) {
    constexpr int LDS = nt8_lds_bytes<C>();
    static int attr = set_lds(gemm_nt8_kernel<C, EPI, OUT_F32, AUX_F32>, LDS);
    if (attr != 0) return attr;
    const int tiles_m = (int)nrv_cdiv(p.e.M, C::TBM), tiles_n = (int)nrv_cdiv(p.e.N, C::TBN);
    p.tiles_n = tiles_n;
    p.gn = 4;
    p.ntiles = tiles_m * tiles_n;
    // persistent: one workgroup per CU walks the tiles b, b + grid, ... (the grid stays a multiple of the 8 XCDs so that a
    // workgroup's tiles stay on its XCD's share of the tile order)
    const int cus = device_cus() & ~7;
    const int grid = NRV_TUNE_NT8_GRID(p.ntiles < cus || cus <= 0 ? p.ntiles : cus, p.ntiles);      // identity in the product (csrc/nrv_dev.hpp)
    hipLaunchKernelGGL((gemm_nt8_kernel<C, EPI, OUT_F32, AUX_F32>), dim3(grid), dim3(C::THREADS), LDS, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}

// tile selection.  device_cus(): the CUs the GEMM launches plan for = the device's CUs minus nrv_set_reserved_cus()
std::atomic<int> g_reserved_cus{0};
int physical_cus() {
    static int n = [] {
        int dev = 0, v = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) v = prop.multiProcessorCount;
        return v > 0 ? v : 256;
    }();
    return n;
}
int device_cus() {
    const int v = physical_cus() - g_reserved_cus.load(std::memory_order_relaxed);
    return v > 8 ? v : 8;
}

int nt_tile_choice(int64_t M, int64_t N, int64_t K, bool allow_384n) {
    (void)K;
    // One workgroup per CU: the kernel takes ceil(tiles / CUs) rounds of one tile time each, and a tile's time grows with
    // its height (MFMA work ~ rows; operand traffic ~ rows + 256).  Pick the height with the smallest rounds x cost;
    // e.g. [50432 x 768]: 591 tiles of 256 rows = 3 rounds, 474 tiles of 320 rows = 2 rounds (0.140 -> 0.121 ms measured);
    // [12544 x 768] (MAE encoder, 49 tokens): 147 tiles of 256 rows leave 43 % of the CUs idle, 198 tiles of 192 rows do not.
    const int64_t cus = device_cus();
    const int64_t tn = nrv_cdiv(N, 256);
    static const int heights[4] = {256, 320, 192, 128};          // ties go to the earlier entry
    // the 320-row tile's K-step is 1.41 x the 256-row tile's (3 280 vs 2 330 cycles: its fragment reads and three-piece A ops
    // fill the issue slots), its epilogue 1.25 x: 1.3 per tile, refitted on the sweep with the persistent kernel
    // (profiles/r03_nt_tile_sweep.txt: with the round-2 ratio 1.21 fc1 / dU of ViT-B and four ViT-S / MAE shapes took the 320-row tile
    // and lost 2 - 4 % to the 256-row one)
    static const double per_tile[4] = {256.0 + 64.0, (256.0 + 64.0) * 1.3, 192.0 + 64.0, 128.0 + 64.0};
    int best = 256;
    double best_cost = 0.0;
    for (int i = 0; i < 4; ++i) {
        const double c = (double)nrv_cdiv(nrv_cdiv(M, heights[i]) * tn, cus) * per_tile[i];
        if (i == 0 || c < best_cost * 0.999) { best = heights[i]; best_cost = c; }
    }
    // 384 x 128 tiles where 256-column tiles would compute a padding column block (N = 384: 3 x 128 instead of 2 x 256).
    // Cost 300 per tile from the ViT-S sweep (profiles/r02_nt_tile_sweep_128_column_tiles.txt: 52.5 vs 58.9 us on
    // [50432 x 384 x 1152], 67.0 vs 75.8 on [50432 x 384 x 1536]; both grids take 2 rounds)
    const int64_t tn128 = nrv_cdiv(N, 128);
    if (allow_384n && tn128 * 128 < tn * 256) {
        const double c = (double)nrv_cdiv(nrv_cdiv(M, 384) * tn128, cus) * 300.0;
        if (c < best_cost * 0.999) { best = 1384; best_cost = c; }
    }
    return best;
}

template <int EPI, bool OUT_F32, bool AUX_F32>
int launch_nt(const GemmNTParams& p, hipStream_t s) {
    // with the fp32 residual epilogue and a long K the 256 x 256 tile wins although it computes the padding columns
    // (profiles/r04_nt8_384x128_tile_phased.txt: [50432 x 384 x 1536] 91.5 vs 93.6 us with both tiles on the phased kernel)
    const bool allow_384n = !(EPI == NRV_EPI_BIAS_RESIDUAL && p.K >= 1024);
    const int tc = NRV_TUNE_NT_TILE(nt_tile_choice(p.e.M, p.e.N, p.K, allow_384n));      // identity in the product (csrc/nrv_dev.hpp)
    if (EPI == NRV_EPI_BIAS_RESIDUAL && (p.e.out_group > 0 || p.e.aux_row_mod > 0))       // row scatter / operand-row broadcast
        return launch_nt_cfg<Cfg256, NRV_EPI_BIAS_RESIDUAL, OUT_F32, AUX_F32, true>(p, s);    // one launch per step: 256-row tiles only
    if (tc == 1384) {
        if ((p.K & (BK - 1)) == 0 && p.K >= 3 * BK) return launch_nt8_cfg<Cfg384n, EPI, OUT_F32, AUX_F32>(p, s);
        return launch_nt_cfg<Cfg384n, EPI, OUT_F32, AUX_F32, false>(p, s);
    }
    // phased main loop: whole K-steps only, and at least three of them (its prologue issues 1.5 K-steps, its tail peels two)
    if ((p.K & (BK - 1)) == 0 && p.K >= 3 * BK) {
        if (tc == 320) return launch_nt8_cfg<Cfg320, EPI, OUT_F32, AUX_F32>(p, s);
        if (tc == 192) return launch_nt8_cfg<Cfg192, EPI, OUT_F32, AUX_F32>(p, s);
        if (tc == 128) return launch_nt8_cfg<Cfg128, EPI, OUT_F32, AUX_F32>(p, s);
        return launch_nt8_cfg<Cfg256, EPI, OUT_F32, AUX_F32>(p, s);
    }
    if (tc == 320) return launch_nt_cfg<Cfg320, EPI, OUT_F32, AUX_F32, false>(p, s);
    if (tc == 192) return launch_nt_cfg<Cfg192, EPI, OUT_F32, AUX_F32, false>(p, s);
    if (tc == 128) return launch_nt_cfg<Cfg128, EPI, OUT_F32, AUX_F32, false>(p, s);
    return launch_nt_cfg<Cfg256, EPI, OUT_F32, AUX_F32, false>(p, s);
}

// TN launch plan: 256 x 256 tiles and the number of token splits that fills the CUs
struct TnPlan { int tiles_m, tiles_n, splits; };
TnPlan tn_plan(int64_t M, int64_t N, int64_t T) {
    TnPlan pl;
    pl.tiles_m = (int)nrv_cdiv(M, 256);
    pl.tiles_n = (int)nrv_cdiv(N, 256);
    const int64_t tiles = (int64_t)pl.tiles_m * pl.tiles_n;
    const int64_t kt = nrv_cdiv(T, BK);
    int64_t s = device_cus() / tiles;
    if (s > kt) s = kt;
    if (s < 1) s = 1;
    pl.splits = (int)s;
    return pl;
}

}  // namespace

extern "C" int nrv_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb,
                                void* C, int c_dtype, int64_t ldc,
                                int64_t M, int64_t N, int64_t K,
                                int epilogue_id, const float* bias,
                                const void* aux, int aux_dtype, int64_t ld_aux, int64_t aux_row_mod,
                                void* aux_out, int64_t ld_aux_out,
                                int64_t out_group, int64_t out_group_stride, int64_t out_row_offset,
                                void* stream) {
    if (!A || !B || !C) return NRV_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0) return NRV_ERR_SHAPE;
    if (M > 0x7fffff00ll || N > 0x7fffff00ll || K > 0x7fffff00ll) return NRV_ERR_SHAPE;
    if ((K & 7) || (N & 7) || (lda & 7) || (ldb & 7) || lda < K || ldb < K || ldc < N) return NRV_ERR_SHAPE;
    if (c_dtype != NRV_F32 && c_dtype != NRV_BF16) return NRV_ERR_DTYPE;
    const int64_t csz = c_dtype == NRV_F32 ? 4 : 2;
    if (!nrv_aligned16(A) || !nrv_aligned16(B) || !nrv_aligned16(C) || ((ldc * csz) & 15)) return NRV_ERR_ALIGN;
    if (lda * 2 * 384 >= 0x7fffffffll || ldb * 2 * 256 >= 0x7fffffffll) return NRV_ERR_SHAPE;      // per-lane DMA row offsets within a tile
    if (out_group < 0 || (out_group > 0 && out_group_stride < out_group)) return NRV_ERR_SHAPE;
    if ((out_group > 0 || aux_row_mod > 0) && epilogue_id != NRV_EPI_BIAS_RESIDUAL) return NRV_ERR_EPILOGUE;   // nrv.h: remap rides on that epilogue
    // the epilogue addresses a wave's block (<= 160 rows) with 32-bit byte offsets
    if (ldc * 4 * 320 >= 0x7fffffffll || ld_aux * 4 * 320 >= 0x7fffffffll || ld_aux_out * 2 * 320 >= 0x7fffffffll) return NRV_ERR_SHAPE;
    if (epi_q8(epilogue_id) && c_dtype != NRV_BF16) return NRV_ERR_DTYPE;                    // the byte stream goes with bf16 outputs
    if (epi_q8(epilogue_id) && (N & 63)) return NRV_ERR_SHAPE;                                // ... in 64-column blocks of row pairs
    const bool need_aux = epilogue_id == NRV_EPI_BIAS_RESIDUAL || epi_dgelu(epilogue_id);
    if (need_aux) {
        if (!aux) return NRV_ERR_EPILOGUE;
        if (aux_dtype != NRV_F32 && aux_dtype != NRV_BF16 && aux_dtype != NRV_U8) return NRV_ERR_DTYPE;
        if (epilogue_id == NRV_EPI_DGELU && aux_dtype != NRV_BF16) return NRV_ERR_DTYPE;
        if ((epilogue_id == NRV_EPI_DGELU_Q8) != (aux_dtype == NRV_U8)) return NRV_ERR_DTYPE;
        const int64_t asz = aux_dtype == NRV_F32 ? 4 : aux_dtype == NRV_U8 ? 1 : 2;
        if (ld_aux < N || ((ld_aux * asz) & 15) || !nrv_aligned16(aux)) return NRV_ERR_ALIGN;
    }
    if ((epilogue_id == NRV_EPI_BIAS || epi_gelu(epilogue_id)) && !bias) return NRV_ERR_EPILOGUE;
    if (bias && !nrv_aligned16(bias)) return NRV_ERR_ALIGN;
    if (aux_out && (ld_aux_out < N || (ld_aux_out & (epilogue_id == NRV_EPI_BIAS_GELU_Q8 ? 15 : 7)) || !nrv_aligned16(aux_out))) return NRV_ERR_ALIGN;

    GemmNTParams p;
    p.A = static_cast<const bf16_t*>(A);
    p.B = static_cast<const bf16_t*>(B);
    p.lda = lda; p.ldb = ldb; p.K = (int)K;
    p.tiles_n = 0;
    p.e.C = C; p.e.bias = bias; p.e.aux = aux; p.e.aux_out = aux_out;
    p.e.ldc = ldc; p.e.ld_aux = ld_aux; p.e.ld_aux_out = ld_aux_out;
    p.e.M = (int)M; p.e.N = (int)N;
    p.e.aux_row_mod = (int)aux_row_mod;
    p.e.out_group = (int)out_group; p.e.out_group_stride = (int)out_group_stride; p.e.out_row_offset = (int)out_row_offset;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool of32 = c_dtype == NRV_F32;
    const bool af32 = aux_dtype == NRV_F32;
    switch (epilogue_id) {
        case NRV_EPI_NONE:
            return of32 ? launch_nt<NRV_EPI_NONE, true, true>(p, s) : launch_nt<NRV_EPI_NONE, false, true>(p, s);
        case NRV_EPI_BIAS:
            return of32 ? launch_nt<NRV_EPI_BIAS, true, true>(p, s) : launch_nt<NRV_EPI_BIAS, false, true>(p, s);
        case NRV_EPI_BIAS_GELU:
            return of32 ? launch_nt<NRV_EPI_BIAS_GELU, true, true>(p, s) : launch_nt<NRV_EPI_BIAS_GELU, false, true>(p, s);
        case NRV_EPI_BIAS_RESIDUAL:
            if (of32) return af32 ? launch_nt<NRV_EPI_BIAS_RESIDUAL, true, true>(p, s) : launch_nt<NRV_EPI_BIAS_RESIDUAL, true, false>(p, s);
            return af32 ? launch_nt<NRV_EPI_BIAS_RESIDUAL, false, true>(p, s) : launch_nt<NRV_EPI_BIAS_RESIDUAL, false, false>(p, s);
        case NRV_EPI_DGELU:
            return of32 ? launch_nt<NRV_EPI_DGELU, true, false>(p, s) : launch_nt<NRV_EPI_DGELU, false, false>(p, s);
        case NRV_EPI_BIAS_GELU_Q8:
            return launch_nt<NRV_EPI_BIAS_GELU_Q8, false, true>(p, s);
        case NRV_EPI_DGELU_Q8:
            return launch_nt<NRV_EPI_DGELU_Q8, false, false>(p, s);
        default:
            return NRV_ERR_EPILOGUE;
    }
}

extern "C" int nrv_set_reserved_cus(int n) {
    if (n < 0 || physical_cus() - n < 8) return NRV_ERR_SHAPE;
    return g_reserved_cus.exchange(n, std::memory_order_relaxed);
}

extern "C" size_t nrv_gemm_tn_workspace(int64_t M, int64_t N, int64_t T) {
    if (M <= 0 || N <= 0 || T <= 0) return 0;
    const int s = tn_plan(M, N, T).splits;
    return (size_t)s * (size_t)M * (size_t)N * 4 + (size_t)s * (size_t)M * 4;   // C slabs (also the beta == 1 single-split case) + bias slabs
}

extern "C" int nrv_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb,
                                float* C, int64_t ldc, int64_t M, int64_t N, int64_t T, float beta,
                                int64_t a_group, int64_t a_group_stride, int64_t a_row_offset,
                                float* dbias, float dbias_beta,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !B || !C) return NRV_ERR_NULL;
    if (M <= 0 || N <= 0 || T <= 0) return NRV_ERR_SHAPE;
    if (M > 0x7fffff00ll || N > 0x7fffff00ll || T > 0x7fffff00ll) return NRV_ERR_SHAPE;
    if ((M & 7) || (N & 7) || (lda & 7) || (ldb & 7) || lda < M || ldb < N || ldc < N || (ldc & 3)) return NRV_ERR_SHAPE;
    if (!nrv_aligned16(A) || !nrv_aligned16(B) || !nrv_aligned16(C)) return NRV_ERR_ALIGN;
    if (beta != 0.f && beta != 1.f) return NRV_ERR_SHAPE;
    if (a_group < 0 || (a_group > 0 && a_group_stride < a_group)) return NRV_ERR_SHAPE;
    const TnPlan pl = tn_plan(M, N, T);
    const int splits = pl.splits;
    const int64_t kt_total = nrv_cdiv(T, BK);
    const int kt_q = (int)(kt_total / splits), kt_r = (int)(kt_total % splits);
    const int kt_per_split = kt_q + (kt_r ? 1 : 0);
    // per-workgroup operand windows must stay below 2 GiB of byte offset
    const int64_t a_rows = a_group > 0 ? (int64_t)kt_per_split * BK * a_group_stride / a_group + a_group_stride : (int64_t)kt_per_split * BK;
    if (a_rows * lda * 2 >= 0x7fffffffll || (int64_t)kt_per_split * BK * ldb * 2 >= 0x7fffffffll) return NRV_ERR_SHAPE;
    const bool direct = splits == 1 && beta == 0.f;
    const size_t slab_bytes = direct ? 0 : (size_t)splits * (size_t)M * (size_t)N * 4;
    const size_t need = slab_bytes + (dbias ? (size_t)splits * (size_t)M * 4 : 0);
    if (need > 0 && (!workspace || workspace_bytes < need)) return NRV_ERR_WORKSPACE;
    if (need > 0 && !nrv_aligned16(workspace)) return NRV_ERR_ALIGN;
    if (dbias && dbias_beta != 0.f && dbias_beta != 1.f) return NRV_ERR_SHAPE;

    GemmTNParams p;
    p.A = static_cast<const bf16_t*>(A);
    p.B = static_cast<const bf16_t*>(B);
    p.lda = lda; p.ldb = ldb; p.T = (int)T;
    p.tiles_n = pl.tiles_n; p.tiles_mn = pl.tiles_m * pl.tiles_n;
    p.splits = splits; p.kt_q = kt_q; p.kt_r = kt_r;
    p.a_group = (int)a_group; p.a_group_stride = (int)a_group_stride; p.a_row_offset = (int)a_row_offset;
    p.e.bias = nullptr; p.e.aux = nullptr; p.e.aux_out = nullptr;
    p.e.ld_aux = 0; p.e.ld_aux_out = 0;
    p.e.M = (int)M; p.e.N = (int)N; p.e.aux_row_mod = 0;
    p.e.out_group = 0; p.e.out_group_stride = 0; p.e.out_row_offset = 0;
    if (direct) { p.e.C = C; p.e.ldc = ldc; p.slab_stride = 0; }
    else { p.e.C = workspace; p.e.ldc = N; p.slab_stride = (long long)M * N; }
    p.bias_ws = dbias ? reinterpret_cast<float*>(static_cast<char*>(workspace) + slab_bytes) : nullptr;

    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a_group == 0 && kt_q >= 3) {          // phased K loop: no row remap, at least three K-steps in every split
        static int attr = set_lds(gemm_tn8_kernel<TnCfg256>, TnCfg256::LDS);
        if (attr != 0) return attr;
        hipLaunchKernelGGL(gemm_tn8_kernel<TnCfg256>, dim3(p.tiles_mn * splits), dim3(GEMM_THREADS), TnCfg256::LDS, s, p);
    } else {
        static int attr = set_lds(gemm_tn_kernel<TnCfg256>, TnCfg256::LDS);
        if (attr != 0) return attr;
        hipLaunchKernelGGL(gemm_tn_kernel<TnCfg256>, dim3(p.tiles_mn * splits), dim3(GEMM_THREADS), TnCfg256::LDS, s, p);
    }
    NRV_CHECK_LAUNCH();
    if (!direct || dbias) {
        const long long total4 = direct ? (long long)nrv_cdiv(M, 4) : (long long)M * (N >> 2);
        int blocks = (int)((total4 + 63) / 64);            // 64 chunks of C per block
        if (blocks > 8192) blocks = 8192;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s,
                           direct ? nullptr : static_cast<const float*>(workspace), (long long)M * N, splits, C, (long long)ldc,
                           (int)M, (int)N, beta, static_cast<const float*>(p.bias_ws), dbias, dbias_beta);
        NRV_CHECK_LAUNCH();
    }
    return 0;
}

// ---- grouped stream-K form: all weight gradients of a layer in one launch (gemm_tng_kernel) ----------------------------------
namespace {
// launch plan of a group, or W == 0 when the grouped kernel does not take it (the caller then issues nrv_gemm_tn_bf16 per problem)
struct TngPlan { int W, nk, ntiles, Lc, F, Wr, R, segs_r; long long nslots; };
TngPlan tng_plan(const nrv_tn_problem* pr, int n, int64_t T) {
    TngPlan pl{};
    if (!pr || n < 1 || n > TNG_MAX || T <= 0 || T > 0x7fffff00ll) return pl;
    long long tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (pr[i].M <= 0 || pr[i].N <= 0 || pr[i].M > 0x7fffff00ll || pr[i].N > 0x7fffff00ll) return pl;
        tiles += nrv_cdiv(pr[i].M, 256) * nrv_cdiv(pr[i].N, 256);
    }
    const int W = device_cus();
    const long long nk = nrv_cdiv(T, BK);
    if (tiles > W || tiles * nk > 0x3fffffffll) return pl;
    long long Lc = nrv_cdiv(tiles * nk, W);                       // K-steps per workgroup when all do the same number
    if (Lc < 8) return pl;                                        // every segment needs >= 3 K-steps, and a prologue worth paying
    long long F = nk / Lc;
    if (F > W / tiles) F = W / tiles;
    if (F < 1) return pl;
    long long Wr = W - F * tiles, R = nk - F * Lc;
    int segs_r = 0;
    if (Wr > 0) {
        // A remainder workgroup runs several short segments, each with its own prologue, epilogue and partial store (SEG_COST
        // K-steps' worth, from the tile timeline: ~3 us + 5 - 8 us against 1.6 us per K-step): it gets fewer K-steps than a cohort
        // workgroup.  Pick the cohort length that levels the two times (with equal K-steps the remainder workgroups finished
        // last: the grouped launch measured 1 - 2 % behind four split-K launches).
        constexpr long long SEG_COST = 6;
        long long best = -1, best_t = 0;
        for (long long L = Lc; L <= Lc + 40 && F * L < nk; ++L) {
            const long long r = nk - F * L;
            if (r < 16) break;
            const long long per = nrv_cdiv(tiles * r, Wr);
            if (per < 8) break;
            const long long segs = 1 + nrv_cdiv(per, r);
            const long long t_rem = per + segs * SEG_COST, t_coh = L + SEG_COST;
            const long long t = t_rem > t_coh ? t_rem : t_coh;
            if (best < 0 || t < best_t) { best = L; best_t = t; }
        }
        if (best > 0) {
            Lc = best; R = nk - F * Lc;
            const long long per = nrv_cdiv(tiles * R, Wr);
            segs_r = (int)(2 + nrv_cdiv(per, R));
            if (segs_r > 12) Wr = 0;
        } else {
            Wr = 0;                                               // too fragmented: the last cohort takes the remainder instead
        }
    }
    if (Wr == 0) { R = 0; segs_r = 0; }
    pl.W = W; pl.nk = (int)nk; pl.ntiles = (int)tiles; pl.Lc = (int)Lc; pl.F = (int)F; pl.Wr = (int)Wr; pl.R = (int)R; pl.segs_r = segs_r;
    pl.nslots = F * tiles + Wr * segs_r;
    return pl;
}
}  // namespace

extern "C" size_t nrv_gemm_tn_grouped_workspace(const nrv_tn_problem* problems, int nprob, int64_t T) {
    const TngPlan pl = tng_plan(problems, nprob, T);
    if (pl.W == 0) return 0;
    return (size_t)pl.nslots * (256 * 256 + 256) * 4;
}

extern "C" int nrv_gemm_tn_grouped_bf16(const nrv_tn_problem* problems, int nprob, int64_t T,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!problems || !workspace) return NRV_ERR_NULL;
    const TngPlan pl = tng_plan(problems, nprob, T);
    if (pl.W == 0) return NRV_ERR_SHAPE;
    if (workspace_bytes < (size_t)pl.nslots * (256 * 256 + 256) * 4) return NRV_ERR_WORKSPACE;
    if (!nrv_aligned16(workspace)) return NRV_ERR_ALIGN;
    GemmTNGParams p;
    int tile0 = 0;
    for (int i = 0; i < TNG_MAX; ++i) {
        TngProblem& q = p.pr[i];
        if (i >= nprob) { q = p.pr[0]; q.tile0 = 0x7fffffff; continue; }
        const nrv_tn_problem& s = problems[i];
        if (!s.A || !s.B || !s.C) return NRV_ERR_NULL;
        if ((s.M & 7) || (s.N & 7) || (s.lda & 7) || (s.ldb & 7) || s.lda < s.M || s.ldb < s.N || s.ldc < s.N || (s.ldc & 3)) return NRV_ERR_SHAPE;
        if (!nrv_aligned16(s.A) || !nrv_aligned16(s.B) || !nrv_aligned16(s.C)) return NRV_ERR_ALIGN;
        if ((s.beta != 0.f && s.beta != 1.f) || (s.dbias && s.dbias_beta != 0.f && s.dbias_beta != 1.f)) return NRV_ERR_SHAPE;
        // a segment's operand window (at most one tile's token range) must stay below 2 GiB of byte offset
        if ((long long)pl.nk * BK * s.lda * 2 >= 0x7fffffffll || (long long)pl.nk * BK * s.ldb * 2 >= 0x7fffffffll) return NRV_ERR_SHAPE;
        q.A = static_cast<const bf16_t*>(s.A); q.B = static_cast<const bf16_t*>(s.B); q.C = s.C; q.dbias = s.dbias;
        q.lda = s.lda; q.ldb = s.ldb; q.ldc = s.ldc; q.M = (int)s.M; q.N = (int)s.N;
        q.tiles_n = (int)nrv_cdiv(s.N, 256); q.tile0 = tile0;
        q.beta = s.beta; q.dbias_beta = s.dbias_beta;
        tile0 += (int)(nrv_cdiv(s.M, 256) * nrv_cdiv(s.N, 256));
    }
    p.nprob = nprob; p.ntiles = pl.ntiles; p.nk = pl.nk; p.T = (int)T; p.W = pl.W;
    p.Lc = pl.Lc; p.F = pl.F; p.Wr = pl.Wr; p.R = pl.R; p.segs_r = pl.segs_r;
    p.slots = static_cast<float*>(workspace);
    p.bias_slots = p.slots + (size_t)pl.nslots * 256 * 256;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static int attr = set_lds(gemm_tng_kernel<TnCfg256>, TnCfg256::LDS);
    if (attr != 0) return attr;
    hipLaunchKernelGGL(gemm_tng_kernel<TnCfg256>, dim3(pl.W), dim3(GEMM_THREADS), TnCfg256::LDS, s, p);
    NRV_CHECK_LAUNCH();
    hipLaunchKernelGGL(tng_fixup_kernel, dim3(pl.ntiles * (256 * 256 / 4 / 64)), dim3(256), 0, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}
