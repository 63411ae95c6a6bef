// Developer instrumentation hooks: the INSTRUMENTED version (tools/build_dev.py NAME --instrument).  Never part of
// libnrv_hip.so.  The including source defines NRV_DEV_TU (gemm / sinkhorn) first; each instrumented translation unit owns
// one stamp buffer and exports nrv_dev_stamps_enable_<TU>() and nrv_dev_read_stamps_<TU>(host_out, count).
//
//   workgroup stamps : buf[8 * blockIdx.x + i]                = s_memrealtime (100 MHz) of stamp i, i < 6; [6] = HW_ID, [7] = XCC_ID
//   wave accounting  : buf[2^19 + 16 * (nwaves * blockIdx.x + wave) + i] = shader cycles accumulated in section i < 16
//   sequential stamps: buf[16 * blockIdx.x + k]              = s_memtime of the k-th NRV_STAMP_SEQ() of thread 0
#pragma once
#include <hip/hip_runtime.h>
#ifndef NRV_DEV_TU
#error "define NRV_DEV_TU before including the instrumented nrv_dev.hpp"
#endif
#define NRV_DEV_CAT2(a, b) a##b
#define NRV_DEV_CAT(a, b) NRV_DEV_CAT2(a, b)

namespace {
__device__ unsigned long long* nrv_dev_buf = nullptr;
unsigned long long* nrv_dev_host_buf = nullptr;
constexpr size_t NRV_DEV_BYTES = 16u << 20;
}

extern "C" int NRV_DEV_CAT(nrv_dev_stamps_enable_, NRV_DEV_TU)() {
    if (!nrv_dev_host_buf) {
        void* p = nullptr;
        if (hipMalloc(&p, NRV_DEV_BYTES) != hipSuccess) return -1;
        if (hipMemset(p, 0, NRV_DEV_BYTES) != hipSuccess) return -1;
        nrv_dev_host_buf = static_cast<unsigned long long*>(p);
        if (hipMemcpyToSymbol(HIP_SYMBOL(nrv_dev_buf), &nrv_dev_host_buf, sizeof(nrv_dev_host_buf)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int NRV_DEV_CAT(nrv_dev_read_stamps_, NRV_DEV_TU)(unsigned long long* host_out, size_t count) {
    if (!nrv_dev_host_buf || !host_out || count * 8 > NRV_DEV_BYTES) return -1;
    return (int)hipMemcpy(host_out, nrv_dev_host_buf, count * 8, hipMemcpyDeviceToHost);
}

#ifdef NRV_DEV_NO_STAMPS        // tile-sweep builds: only the tile override below, the kernels stay the product's
#define NRV_STAMP_VARS(n)
#define NRV_STAMP(i)
#define NRV_STAMP_FLUSH_WG(n, tid)
#define NRV_WACC_VARS
#define NRV_WACC_MARK()
#define NRV_WACC(i)
#define NRV_WACC_FLUSH(nwaves, wave, lane)
#define NRV_STAMP_SEQ_VARS(tid)
#define NRV_STAMP_SEQ()
#define NRV_STAMP_SEQ_RESET()
#define NRV_TILE_STAMP_VARS(tid)
#define NRV_TILE_STAMP()
#else
#define NRV_STAMP_VARS(n) unsigned long long nrv_t_[n] = {}
#define NRV_STAMP(i) do { nrv_t_[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); } while (0)
#define NRV_STAMP_FLUSH_WG(n, tid)                                                                                      \
    do {                                                                                                                \
        if (nrv_dev_buf && (tid) == 0) {                                                                                \
            unsigned long long* o_ = nrv_dev_buf + (unsigned long long)blockIdx.x * 8;                                  \
            for (int i_ = 0; i_ < (n); ++i_) o_[i_] = nrv_t_[i_];                                                       \
            o_[6] = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4 /* HW_REG_HW_ID */);                                 \
            o_[7] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20 /* XCC_ID */);                                       \
        }                                                                                                               \
    } while (0)
#define NRV_WACC_VARS unsigned long long nrv_w_[16] = {}, nrv_wl_ = 0
#define NRV_WACC_MARK() do { nrv_wl_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); } while (0)
#define NRV_WACC(i)                                                                                                     \
    do {                                                                                                                \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();                                                     \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
        nrv_w_[i] += n_ - nrv_wl_;                                                                                      \
        nrv_wl_ = n_;                                                                                                   \
    } while (0)
#define NRV_WACC_FLUSH(nwaves, wave, lane)                                                                              \
    do {                                                                                                                \
        if (nrv_dev_buf && (lane) == 0) {                                                                               \
            unsigned long long* o_ = nrv_dev_buf + (1u << 19) + ((unsigned long long)blockIdx.x * (nwaves) + (wave)) * 16; \
            for (int i_ = 0; i_ < 16; ++i_) o_[i_] = nrv_w_[i_];                                                        \
        }                                                                                                               \
    } while (0)
#define NRV_STAMP_SEQ_VARS(tid) const bool nrv_sq_on_ = nrv_dev_buf && (tid) == 0; int nrv_sq_i_ = 0
#define NRV_STAMP_SEQ()                                                                                                 \
    do {                                                                                                                \
        if (nrv_sq_on_) nrv_dev_buf[(unsigned long long)blockIdx.x * 16 + nrv_sq_i_] = __builtin_amdgcn_s_memtime();    \
        ++nrv_sq_i_;                                                                                                    \
    } while (0)
#define NRV_STAMP_SEQ_RESET() do { nrv_sq_i_ = 0; } while (0)
#endif
// tile stamps (persistent kernels): buf[2^20 + 32 * blockIdx.x + k] = s_memrealtime (100 MHz) of the k-th NRV_TILE_STAMP() of wave 0, k < 32.
// `wave` is wave-uniform (readfirstlane) and all 64 lanes store the same value: a thread-0 branch in the tile loop made hipcc treat
// the loop-carried tile descriptors as divergent (vector registers, which the LDS-DMA asm cannot take)
#ifndef NRV_DEV_NO_STAMPS
#define NRV_TILE_STAMP_VARS(wave) const bool nrv_ts_on_ = nrv_dev_buf && (wave) == 0; int nrv_ts_i_ = 0
#define NRV_TILE_STAMP()                                                                                                \
    do {                                                                                                                \
        if (nrv_ts_on_ && nrv_ts_i_ < 32)                                                                               \
            nrv_dev_buf[(1u << 20) + (unsigned long long)blockIdx.x * 32 + nrv_ts_i_] = __builtin_amdgcn_s_memrealtime(); \
        ++nrv_ts_i_;                                                                                                    \
    } while (0)
#endif
// tile sweep builds: python tools/build_dev.py t256 --instrument -DNRV_DEV_NO_STAMPS -DNRV_FORCE_NT_TILE=256   (128 / 192 / 256 / 320 / 1384)
#ifdef NRV_FORCE_NT_TILE
#define NRV_TUNE_NT_TILE(choice) (NRV_FORCE_NT_TILE)
#else
#define NRV_TUNE_NT_TILE(choice) (choice)
#endif
// persistent NT kernel launched with one workgroup per tile (no hand-over between tiles): -DNRV_FORCE_NT8_ONE_TILE
#ifdef NRV_FORCE_NT8_ONE_TILE
#define NRV_TUNE_NT8_GRID(grid, ntiles) (ntiles)
#else
#define NRV_TUNE_NT8_GRID(grid, ntiles) (grid)
#endif
// phased K loops without the wave-group stagger (waves 4-7 in phase with waves 0-3): -DNRV_FORCE_NO_STAGGER
#ifdef NRV_FORCE_NO_STAGGER
#define NRV_TUNE_STAGGER(cond) (false)
#else
#define NRV_TUNE_STAGGER(cond) (cond)
#endif
