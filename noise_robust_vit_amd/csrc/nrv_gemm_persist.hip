// Persistent MFMA bf16 "NT" GEMM for gfx950:  C[M,N] = A[M,K] . B[N,K]^T (+ fused epilogue).
//
// Why a second NT kernel.  With K = 768 (ViT-B) a 256x256 tile spends 12 K-steps in its main loop and then ~20-40 %
// of its time in the prologue (first tiles in flight, MFMA idle) and the epilogue (MFMA idle while the CU streams
// the tile to HBM); 591 tiles on 256 CUs also cost 3 rounds for 2.3 rounds of work (profiles/r01).
// This kernel removes the three losses structurally:
//   * persistent: gridDim = #CUs, every workgroup walks its share of the tiles (finer 192x128 tiles: the static
//     split is within one small tile of even);
//   * ONE continuous LDS-DMA stream over all (tile, k-step) pairs of the workgroup, 3-stage ring, the DMA of step
//     s+2 is issued at the top of step s and crosses tile boundaries -- no per-tile prologue;
//   * DEFERRED epilogue: when a tile's last k-step retires its accumulators move to a second register set and are
//     written out slab by slab (16 rows x 64 columns per k-step) underneath the next tile's MFMAs, so bias / GELU /
//     residual VALU work and the HBM stores overlap the matrix pipe instead of stalling it.
// Counted s_waitcnt vmcnt(N) keeps the ring's loads in flight across barriers: every VMEM instruction of the loop
// is unconditional (tails are out-of-range buffer offsets: loads return 0, stores are dropped), so the number of
// younger operations behind the tile a step needs is known exactly.
//
// Workgroup = 512 threads = 8 waves as 4 (M) x 2 (N); wave tile 48 x 64 = 3 x 4 MFMA 16x16x32 tiles.
// LDS: 3 x (24 KiB A + 16 KiB B) ring + 8 x 4352 B epilogue patches = 157,696 B of the CU's 163,840.
#include <stdlib.h>

#include "nrv_common.hpp"

namespace {

constexpr int PBM = 192, PBN = 128, PBK = 64;
constexpr int P_THREADS = 512;
constexpr int P_MI = 3, P_NI = 4;
constexpr int P_A_BYTES = PBM * 128, P_B_BYTES = PBN * 128, P_STAGE = P_A_BYTES + P_B_BYTES;
constexpr int P_NST = 3;
constexpr int P_PATCH_OFF = P_NST * P_STAGE;
constexpr int P_ROW_F32 = 68;
constexpr int P_PATCH_BYTES = 16 * P_ROW_F32 * 4;
constexpr int P_LDS = P_PATCH_OFF + 8 * P_PATCH_BYTES;
constexpr int P_CA = 3, P_CB = 2, P_ND = P_CA + P_CB;      // DMA instructions per thread per k-step

struct PParams {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    const float* bias;
    const void* aux;
    void* aux_out;
    long long lda, ldb, ldc, ld_aux, ld_aux_out;
    int M, N, K;
    int tiles_m, tiles_n;
    int aux_row_mod;
    int out_group, out_group_stride, out_row_offset;
    unsigned c_bytes, aux_bytes, auxo_bytes;      // buffer descriptor sizes (host guarantees < 2^31)
};

__device__ __forceinline__ int p_remap_row(int m, int group, int group_stride, int offset) {
    if (group <= 0) return m;
    const int g = m / group;
    return g * group_stride + (m - g * group) + offset;
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the immediate must be a literal)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define NRV_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        NRV_W(0) NRV_W(1) NRV_W(2) NRV_W(3) NRV_W(4) NRV_W(5) NRV_W(6) NRV_W(7) NRV_W(8) NRV_W(9)
        NRV_W(10) NRV_W(11) NRV_W(12) NRV_W(13) NRV_W(14) NRV_W(15) NRV_W(16) NRV_W(17) NRV_W(18) NRV_W(19)
        NRV_W(20) NRV_W(21) NRV_W(22) NRV_W(23) NRV_W(24) NRV_W(25) NRV_W(26) NRV_W(27) NRV_W(28) NRV_W(29)
        NRV_W(30) NRV_W(31) NRV_W(32) NRV_W(33) NRV_W(34) NRV_W(35) NRV_W(36) NRV_W(37) NRV_W(38) NRV_W(39)
        NRV_W(40) NRV_W(41) NRV_W(42) NRV_W(43) NRV_W(44) NRV_W(45) NRV_W(46) NRV_W(47) NRV_W(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef NRV_W
}

// Every VMEM instruction of this kernel goes through a compiler builtin, so hipcc's waitcnt pass counts the LDS-DMA
// ring, the epilogue operand loads and the stores together: the wait it places in front of the first use of an
// operand load (three k-steps after its issue) is an exact counted vmcnt(N) that leaves the ring in flight.
// (Inline-asm loads whose results are consumed steps later are unsafe: the compiler may copy their destination
// registers at the loop back-edge before the data lands.)
__device__ __forceinline__ void dma16b(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, unsigned voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_base), 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ u32x4_t ld_b128(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, 0, 0);
}
__device__ __forceinline__ u32x2_t ld_b64(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset) {
    return __builtin_amdgcn_raw_buffer_load_b64(rsrc, voffset, 0, 0);
}

template <int EPI, bool OUT_F32, bool AUX_F32>
struct EpiTraits {
    static constexpr int CW = OUT_F32 ? 4 : 8;          // columns per lane on the read side (16-byte global accesses)
    static constexpr int V = CW / 4;
    static constexpr int LPR = 64 / CW;
    static constexpr int RPI = 64 / LPR;
    static constexpr int NIT = 16 / RPI;
    static constexpr bool HAS_AUX = EPI == NRV_EPI_BIAS_RESIDUAL || EPI == NRV_EPI_DGELU;
    static constexpr bool AUX32 = EPI == NRV_EPI_BIAS_RESIDUAL && AUX_F32;
    // VMEM instructions one slab issues (all unconditional)
    static constexpr int STORES = NIT * (OUT_F32 ? V : 1) + (EPI == NRV_EPI_BIAS_GELU ? NIT : 0);
    static constexpr int AUX_LOADS = HAS_AUX ? NIT * (AUX32 ? V : 1) : 0;       // per slab
};

template <int EPI, bool OUT_F32, bool AUX_F32>
__global__ __launch_bounds__(P_THREADS, 2) void gemm_nt_persist_kernel(const PParams p) {
    using T = EpiTraits<EPI, OUT_F32, AUX_F32>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int M = p.M, N = p.N, K = p.K;
    const int nk = (K + PBK - 1) / PBK;
    const int total = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;

    // tile sequence of this workgroup: virtual id v = blockIdx.x + i * G; inside a full round the 8 XCDs (blocks
    // b, b+8, ... share one) get 32 consecutive tile ids each, so neighbouring tiles share their A panel in one L2
    auto tile_of = [&](int i) -> int {
        const int v = blockIdx.x + i * G;
        if (v >= total) return -1;
        const int round = v / G, pos = v - round * G;
        if ((round + 1) * G > total || (G & 7)) return v;
        return round * G + (pos & 7) * (G >> 3) + (pos >> 3);
    };
    int ntiles = 0;
    while (tile_of(ntiles) >= 0) ++ntiles;
    if (ntiles == 0) return;
    const int nsteps = ntiles * nk;

    const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, p.c_bytes);
    const __amdgpu_buffer_rsrc_t raux = make_rsrc(p.aux ? p.aux : p.C, p.aux ? p.aux_bytes : 0);
    const __amdgpu_buffer_rsrc_t rauxo = make_rsrc(p.aux_out ? p.aux_out : p.C, p.aux_out ? p.auxo_bytes : 0);

    // ---------------- DMA stream state (runs two steps ahead of the MFMA stream) ----------------
    int d_i = 0, d_kt = 0;                       // tile index / k-step of the next DMA group to issue
    int d_m0 = 0, d_n0 = 0;
    unsigned st_a[P_CA], st_b[P_CB];
    auto dma_tile_setup = [&]() {
        const int id = tile_of(d_i);
        const int tm = id / p.tiles_n, tn = id - tm * p.tiles_n;
        d_m0 = tm * PBM;
        d_n0 = tn * PBN;
#pragma unroll
        for (int i = 0; i < P_CA; ++i) {
            const int r = (i * 8 + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            st_a[i] = (d_m0 + r < M) ? (unsigned)((long long)r * p.lda * 2) + c * 16 : NRV_OOB;
        }
#pragma unroll
        for (int i = 0; i < P_CB; ++i) {
            const int r = (i * 8 + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            st_b[i] = (d_n0 + r < N) ? (unsigned)((long long)r * p.ldb * 2) + c * 16 : NRV_OOB;
        }
    };
    // issue the DMA group of step `s` (all P_ND instructions); returns the number of instructions issued
    auto dma_issue = [&](int s) -> int {
        if (s >= nsteps) return 0;
        if (d_kt == 0) dma_tile_setup();
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A + (long long)d_m0 * p.lda, (unsigned long long)(M - d_m0) * p.lda * 2ull);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B + (long long)d_n0 * p.ldb, (unsigned long long)(N - d_n0) * p.ldb * 2ull);
        char* base = smem + (s % P_NST) * P_STAGE;
        const int k0 = d_kt * PBK;
#pragma unroll
        for (int i = 0; i < P_CA; ++i) {
            const int r = (i * 8 + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            const bool ok = (k0 + c * 8 < K) && st_a[i] != NRV_OOB;
            dma16b(ra, base + (i * 8 + wave) * 1024, ok ? st_a[i] + k0 * 2 : NRV_OOB);
        }
#pragma unroll
        for (int i = 0; i < P_CB; ++i) {
            const int r = (i * 8 + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            const bool ok = (k0 + c * 8 < K) && st_b[i] != NRV_OOB;
            dma16b(rb, base + P_A_BYTES + (i * 8 + wave) * 1024, ok ? st_b[i] + k0 * 2 : NRV_OOB);
        }
        if (++d_kt == nk) { d_kt = 0; ++d_i; }
        return P_ND;
    };

    // ---------------- MFMA stream state ----------------
    const int fr = lane & 15, fg = lane >> 4;
    const int swz = (fg ^ ((fr >> 1) & 7)) << 4;
    const int a_rd = (wr * (P_MI * 16) + fr) * 128 + swz;
    const int b_rd = P_A_BYTES + (wc * (P_NI * 16) + fr) * 128 + swz;

    f32x4_t acc[P_MI][P_NI], accp[P_MI][P_NI];
#pragma unroll
    for (int mi = 0; mi < P_MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < P_NI; ++ni) {
            acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            accp[mi][ni] = acc[mi][ni];
        }

    // ---------------- deferred-epilogue state ----------------
    float* stg = reinterpret_cast<float*>(smem + P_PATCH_OFF + wave * P_PATCH_BYTES);
    const int wc_row = lane & 15, wg = lane >> 4;
    const int rcol = lane % T::LPR, rrow = lane / T::LPR;
    bool have_prev = false;
    int pm0 = 0, pn0 = 0;                        // origin of the finished tile whose accumulators sit in accp
    u32x4_t aux32[T::AUX32 ? P_MI : 1][T::NIT][T::V];
    u32x2_t aux16[(T::HAS_AUX && !T::AUX32) ? P_MI : 1][T::NIT][T::V];

    auto slab_offsets = [&](int m0, int n0, int mi, int i, bool& ok, long long& orow, long long& arow, int& ncol) {
        const int m = m0 + wr * (P_MI * 16) + mi * 16 + rrow + T::RPI * i;
        ncol = n0 + wc * 64 + rcol * T::CW;
        ok = m < M && ncol < N;
        orow = p_remap_row(m, p.out_group, p.out_group_stride, p.out_row_offset);
        arow = (EPI == NRV_EPI_BIAS_RESIDUAL && p.aux_row_mod > 0) ? (long long)(m % p.aux_row_mod) : orow;
    };
    // issue the epilogue operand loads of a whole finished tile; returns the number of VMEM instructions
    auto aux_issue = [&](int m0, int n0) -> int {
        if (!T::HAS_AUX) return 0;
#pragma unroll
        for (int mi = 0; mi < P_MI; ++mi)
#pragma unroll
            for (int i = 0; i < T::NIT; ++i) {
                bool ok; long long orow, arow; int ncol;
                slab_offsets(m0, n0, mi, i, ok, orow, arow, ncol);
                if (T::AUX32) {
#pragma unroll
                    for (int v = 0; v < T::V; ++v)
                        aux32[mi][i][v] = ld_b128(raux, ok ? (unsigned)((arow * p.ld_aux + ncol + 4 * v) * 4) : NRV_OOB);
                } else {
                    // bf16 operand: CW = 8 -> one 16-byte load per pass (kept as two u32x2 halves); CW = 4 -> 8 bytes
                    if (T::V == 2) {
                        const u32x4_t t4 = ld_b128(raux, ok ? (unsigned)((arow * p.ld_aux + ncol) * 2) : NRV_OOB);
                        aux16[mi][i][0] = u32x2_t{t4[0], t4[1]};
                        aux16[mi][i][T::V - 1] = u32x2_t{t4[2], t4[3]};
                    } else {
                        aux16[mi][i][0] = ld_b64(raux, ok ? (unsigned)((arow * p.ld_aux + ncol) * 2) : NRV_OOB);
                    }
                }
            }
        return P_MI * T::AUX_LOADS;
    };
    // bias for this lane's columns of a tile
    auto bias_of = [&](int n0, f32x4_t (&b4)[T::V]) {
#pragma unroll
        for (int v = 0; v < T::V; ++v) b4[v] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (EPI == NRV_EPI_BIAS || EPI == NRV_EPI_BIAS_GELU || EPI == NRV_EPI_BIAS_RESIDUAL) {
            const int ncol = n0 + wc * 64 + rcol * T::CW;
            if (p.bias != nullptr && ncol < N) {
#pragma unroll
                for (int v = 0; v < T::V; ++v) b4[v] = *reinterpret_cast<const f32x4_t*>(p.bias + ncol + 4 * v);
            }
        }
    };
    // write one 16 x 64 slab (row block mi) of a finished tile; all stores unconditional (OOB offsets are dropped)
    auto slab_store = [&](const f32x4_t (&a)[P_NI], int m0, int n0, int mi, const f32x4_t (&b4)[T::V]) {
#pragma unroll
        for (int ni = 0; ni < P_NI; ++ni)
            *reinterpret_cast<f32x4_t*>(stg + wc_row * P_ROW_F32 + ni * 16 + wg * 4) = a[ni];
#pragma unroll
        for (int i = 0; i < T::NIT; ++i) {
            const int r = rrow + T::RPI * i;
            bool ok; long long orow, arow; int ncol;
            slab_offsets(m0, n0, mi, i, ok, orow, arow, ncol);
            unsigned pk[2 * T::V], pku[2 * T::V];
#pragma unroll
            for (int v = 0; v < T::V; ++v) {
                f32x4_t x = *reinterpret_cast<const f32x4_t*>(stg + r * P_ROW_F32 + rcol * T::CW + 4 * v) + b4[v];
                if (EPI == NRV_EPI_BIAS_GELU) {
                    f32x4_t dg;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { float gv, dv; gelu_both(x[j], gv, dv); x[j] = gv; dg[j] = dv; }
                    pku[2 * v] = pack_bf16x2(dg[0], dg[1]);
                    pku[2 * v + 1] = pack_bf16x2(dg[2], dg[3]);
                }
                if (EPI == NRV_EPI_BIAS_RESIDUAL) {
                    if (T::AUX32) {
                        x += __builtin_bit_cast(f32x4_t, aux32[mi][i][v]);
                    } else {
                        const u32x2_t q = aux16[mi][i][v];
                        x[0] += bf16lo_to_f32(q[0]); x[1] += bf16hi_to_f32(q[0]);
                        x[2] += bf16lo_to_f32(q[1]); x[3] += bf16hi_to_f32(q[1]);
                    }
                }
                if (EPI == NRV_EPI_DGELU) {
                    const u32x2_t q = aux16[mi][i][v];
                    x[0] *= bf16lo_to_f32(q[0]); x[1] *= bf16hi_to_f32(q[0]);
                    x[2] *= bf16lo_to_f32(q[1]); x[3] *= bf16hi_to_f32(q[1]);
                }
                if (OUT_F32) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x), rc,
                                                           ok ? (unsigned)((orow * p.ldc + ncol + 4 * v) * 4) : NRV_OOB, 0, 0);
                } else {
                    pk[2 * v] = pack_bf16x2(x[0], x[1]);
                    pk[2 * v + 1] = pack_bf16x2(x[2], x[3]);
                }
            }
            if (!OUT_F32) {
                const u32x4_t o = {pk[0], pk[1], pk[2 * T::V - 2], pk[2 * T::V - 1]};
                __builtin_amdgcn_raw_buffer_store_b128(o, rc, ok ? (unsigned)((orow * p.ldc + ncol) * 2) : NRV_OOB, 0, 0);
            }
            if (EPI == NRV_EPI_BIAS_GELU) {
                const bool oku = ok && p.aux_out != nullptr;
                if (T::V == 2) {
                    const u32x4_t o = {pku[0], pku[1], pku[2 * T::V - 2], pku[2 * T::V - 1]};
                    __builtin_amdgcn_raw_buffer_store_b128(o, rauxo, oku ? (unsigned)((orow * p.ld_aux_out + ncol) * 2) : NRV_OOB, 0, 0);
                } else {
                    const u32x2_t o = {pku[0], pku[1]};
                    __builtin_amdgcn_raw_buffer_store_b64(o, rauxo, oku ? (unsigned)((orow * p.ld_aux_out + ncol) * 2) : NRV_OOB, 0, 0);
                }
            }
        }
    };

    // ---------------- prologue: two k-steps in flight ----------------
    int after_cur, after_next;                    // VMEM ops younger than the DMA group of step s / s+1
    {
        dma_issue(0);
        after_cur = dma_issue(1);
        after_next = 0;
    }
    int c_i = 0, c_kt = 0;                        // tile index / k-step of the MFMA stream
    int c_m0, c_n0;
    {
        const int id = tile_of(0);
        const int tm = id / p.tiles_n;
        c_m0 = tm * PBM;
        c_n0 = (id - tm * p.tiles_n) * PBN;
    }
    f32x4_t cbias[T::V], pbias[T::V];          // bias of the tile being accumulated / of the finished tile
    bias_of(c_n0, cbias);
#pragma unroll
    for (int v = 0; v < T::V; ++v) pbias[v] = cbias[v];

    for (int s = 0; s < nsteps; ++s) {
        wait_vmcnt(after_cur);                    // everything up to and including the DMA group of step s has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int issued = dma_issue(s + 2);            // ring slot (s + 2) % 3 == (s - 1) % 3 was last read in step s - 1
        after_next += issued;
        int a2 = 0;

        const char* sa = smem + (s % P_NST) * P_STAGE;
        bf16x8_t bfr[P_NI], af[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int ni = 0; ni < P_NI; ++ni) bfr[ni] = lds_read_b128(sa + ((b_rd ^ (ks << 6)) + ni * 2048));
            af[0] = lds_read_b128(sa + ((a_rd ^ (ks << 6))));
#pragma unroll
            for (int mi = 0; mi < P_MI; ++mi) {
                if (mi + 1 < P_MI) af[(mi + 1) & 1] = lds_read_b128(sa + ((a_rd ^ (ks << 6)) + (mi + 1) * 2048));
#pragma unroll
                for (int ni = 0; ni < P_NI; ++ni) acc[mi][ni] = mfma16(bfr[ni], af[mi & 1], acc[mi][ni]);
            }
        }

        // deferred epilogue of the previous tile: one 16-row slab per k-step, k-steps 2, 3, 4 (its operand loads were
        // issued at the tile switch, two top-of-step waits ago)
        if (have_prev && c_kt >= 2 && c_kt < 2 + P_MI) {
            if (c_kt == 2) slab_store(accp[0], pm0, pn0, 0, pbias);
            else if (c_kt == 3) slab_store(accp[1], pm0, pn0, 1, pbias);
            else slab_store(accp[2], pm0, pn0, 2, pbias);
            a2 += T::STORES;
        }

        // tile switch
        if (++c_kt == nk) {
            c_kt = 0;
#pragma unroll
            for (int mi = 0; mi < P_MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < P_NI; ++ni) {
                    accp[mi][ni] = acc[mi][ni];
                    acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
            pm0 = c_m0;
            pn0 = c_n0;
#pragma unroll
            for (int v = 0; v < T::V; ++v) pbias[v] = cbias[v];
            have_prev = true;
            a2 += aux_issue(pm0, pn0);
            ++c_i;
            const int id = tile_of(c_i);
            if (id >= 0) {
                const int tm = id / p.tiles_n;
                c_m0 = tm * PBM;
                c_n0 = (id - tm * p.tiles_n) * PBN;
                bias_of(c_n0, cbias);             // a (tiny) tracked load: counts as one VMEM op per float4
                a2 += (EPI == NRV_EPI_BIAS || EPI == NRV_EPI_BIAS_GELU || EPI == NRV_EPI_BIAS_RESIDUAL) ? T::V : 0;
            }
        }
        after_next += a2;
        after_cur = after_next;
        after_next = a2;
    }
    // last tile (or: tiles with fewer than 2 + P_MI k-steps): synchronous epilogue of whatever is still in accp
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (have_prev) {
#pragma unroll
        for (int mi = 0; mi < P_MI; ++mi) slab_store(accp[mi], pm0, pn0, mi, pbias);
    }
}

int num_cus() {
    static int n = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess) v = prop.multiProcessorCount;
        }
        return v > 0 ? v : 256;
    }();
    return n;
}

template <int EPI, bool OUT_F32, bool AUX_F32>
int launch_persist(const PParams& p, hipStream_t s) {
    static int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_persist_kernel<EPI, OUT_F32, AUX_F32>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
    if (attr != 0) return attr;
    int grid = num_cus();
    const int total = p.tiles_m * p.tiles_n;
    if (grid > total) grid = total;
    hipLaunchKernelGGL((gemm_nt_persist_kernel<EPI, OUT_F32, AUX_F32>), dim3(grid), dim3(P_THREADS), P_LDS, s, p);
    NRV_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Entry used by nrv_gemm_nt_bf16 (nrv_gemm.hip).  Returns -1000 when the shape is outside what this kernel takes
// (the caller then uses the one-tile-per-workgroup kernel); arguments have already been validated.
int nrv_gemm_nt_persist_try(const void* A, long long lda, const void* B, long long ldb, void* C, int c_dtype, long long ldc,
                            long long M, long long N, long long K, int epilogue_id, const float* bias,
                            const void* aux, int aux_dtype, long long ld_aux, long long aux_row_mod,
                            void* aux_out, long long ld_aux_out,
                            long long out_group, long long out_group_stride, long long out_row_offset, hipStream_t s) {
    const long long csz = c_dtype == NRV_F32 ? 4 : 2;
    const long long nk = (K + PBK - 1) / PBK;
    const long long tiles = nrv_cdiv(M, PBM) * nrv_cdiv(N, PBN);
    if (nk < 2 + P_MI + 1) return -1000;                                     // deferred epilogue needs k-steps 2..4 of the next tile
    static const int force = [] { const char* e = getenv("NRV_GEMM_PERSIST"); return e ? atoi(e) : 1; }();
    if (force < 2 && tiles < 2 * (long long)num_cus()) return -1000;         // too few tiles to amortise a persistent launch (2 = force, tests)
    const long long out_rows = out_group > 0 ? (M / out_group + 1) * out_group_stride + out_row_offset : M;
    const long long c_bytes = out_rows * ldc * csz;
    const long long asz = aux_dtype == NRV_F32 ? 4 : 2;
    const long long aux_rows = aux_row_mod > 0 ? aux_row_mod : out_rows;
    const long long aux_bytes = aux ? aux_rows * ld_aux * asz : 0;
    const long long auxo_bytes = aux_out ? out_rows * ld_aux_out * 2 : 0;
    if (c_bytes >= 0x7fffffffll || aux_bytes >= 0x7fffffffll || auxo_bytes >= 0x7fffffffll) return -1000;
    if (lda * 2 * PBM >= 0x7fffffffll || ldb * 2 * PBN >= 0x7fffffffll) return -1000;

    PParams p;
    p.A = static_cast<const bf16_t*>(A); p.B = static_cast<const bf16_t*>(B); p.C = C;
    p.bias = bias; p.aux = aux; p.aux_out = aux_out;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ld_aux = ld_aux; p.ld_aux_out = ld_aux_out;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.tiles_m = (int)nrv_cdiv(M, PBM); p.tiles_n = (int)nrv_cdiv(N, PBN);
    p.aux_row_mod = (int)aux_row_mod;
    p.out_group = (int)out_group; p.out_group_stride = (int)out_group_stride; p.out_row_offset = (int)out_row_offset;
    p.c_bytes = (unsigned)c_bytes; p.aux_bytes = (unsigned)aux_bytes; p.auxo_bytes = (unsigned)auxo_bytes;
    const bool of32 = c_dtype == NRV_F32;
    const bool af32 = aux_dtype == NRV_F32;
    switch (epilogue_id) {
        case NRV_EPI_NONE:
            return of32 ? launch_persist<NRV_EPI_NONE, true, true>(p, s) : launch_persist<NRV_EPI_NONE, false, true>(p, s);
        case NRV_EPI_BIAS:
            return of32 ? launch_persist<NRV_EPI_BIAS, true, true>(p, s) : launch_persist<NRV_EPI_BIAS, false, true>(p, s);
        case NRV_EPI_BIAS_GELU:
            return of32 ? launch_persist<NRV_EPI_BIAS_GELU, true, true>(p, s) : launch_persist<NRV_EPI_BIAS_GELU, false, true>(p, s);
        case NRV_EPI_BIAS_RESIDUAL:
            if (of32) return af32 ? launch_persist<NRV_EPI_BIAS_RESIDUAL, true, true>(p, s) : launch_persist<NRV_EPI_BIAS_RESIDUAL, true, false>(p, s);
            return af32 ? launch_persist<NRV_EPI_BIAS_RESIDUAL, false, true>(p, s) : launch_persist<NRV_EPI_BIAS_RESIDUAL, false, false>(p, s);
        case NRV_EPI_DGELU:
            return of32 ? launch_persist<NRV_EPI_DGELU, true, false>(p, s) : launch_persist<NRV_EPI_DGELU, false, false>(p, s);
        default:
            return -1000;
    }
}
