// Developer instrumentation hooks: the PRODUCT version.  Every hook is empty and every tuning override is off, so the
// product objects contain no clock reads, no stamp buffers and no experimental code path (tests/test_host_logic.py checks
// the built library for both).  tools/build_dev.py --instrument puts tools/dev/ in front of this directory on the include
// path, where a header of the same name defines the hooks for real.  Sources include it as <nrv_dev.hpp>.
#pragma once

#define NRV_STAMP_VARS(n)              // per-workgroup wall-clock stamps (s_memrealtime), kept in scalar registers
#define NRV_STAMP(i)
#define NRV_STAMP_FLUSH_WG(n, tid)
#define NRV_WACC_VARS                  // per-wave shader-cycle accounting of wait sections (s_memtime deltas)
#define NRV_WACC_MARK()
#define NRV_WACC(i)
#define NRV_WACC_FLUSH(nwaves, wave, lane)
#define NRV_STAMP_SEQ_VARS(tid)        // sequential phase stamps of one thread, written as they are taken
#define NRV_STAMP_SEQ()
#define NRV_STAMP_SEQ_RESET()          // persistent kernels: the sequence starts again with every unit of work
#define NRV_TILE_STAMP_VARS(wave)      // persistent kernels: wall-clock stamps of wave 0 along its tile sequence
#define NRV_TILE_STAMP()
#define NRV_TUNE_NT_TILE(choice) (choice)      // tile-height override of the instrumented header (tools/tile_sweep.py)
#define NRV_TUNE_NT8_GRID(grid, ntiles) (grid)  // workgroups of the persistent NT kernel (override: one tile per workgroup)
#define NRV_TUNE_STAGGER(cond) (cond)             // the wave-group stagger of the phased K loops (override: both groups in phase)
