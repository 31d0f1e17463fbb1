// common.h -- shared host/device helpers for libscanerf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "scanerf_hip.h"

#define SCANERF_API extern "C" __attribute__((visibility("default")))

namespace scanerf {

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kNumCU = 256;        // MI355X
constexpr int kNumXCD = 8;

// thread-local last error (scanerf_last_error)
void set_error(const char *fmt, ...);

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return 1;
    }
    return 0;
}

#define SCANERF_REQUIRE(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            scanerf::set_error(__VA_ARGS__);       \
            return 2;                              \
        }                                          \
    } while (0)

// Wait-state guards of rounds 1-2, COMPILED OUT (SCANERF_GUARDS = 0).  They were placed behind global stores
// (SCANERF_STORE_GUARD: 16 states), behind gathers (SCANERF_LOAD_GUARD: 32) and around matrix instructions (render_h3.h
// H3_REGIONS, render_t16.h T16_REGION_*, the operand guards of the splits) after launch-to-launch differences that came and went
// with the register allocation.  Round 3 found what those differences have in common: packed-f32 instructions
// (v_pk_mul/add/fma_f32, formed by the SLP vectoriser) in a kernel that also runs matrix instructions.  Without them no guard is
// needed (0 differing launches of 1 200 with every guard off, instruction caches swept or not; with them and no guards 199 of
// 199: tools/guard_probe.py), so the kernels are compiled with -fno-slp-vectorize (csrc/Makefile, tools/isa_audit.py) and carry no
// guards.  -DSCANERF_GUARDS=1 -DH3_REGIONS=1 rebuilds the guarded listings for that experiment matrix (tools/build_variant.py).
#ifndef SCANERF_GUARDS
#define SCANERF_GUARDS 0
#endif
#if SCANERF_GUARDS
#define SCANERF_LOAD_GUARD()                                                         \
    do {                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                           \
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");                    \
        __builtin_amdgcn_sched_barrier(0);                                           \
    } while (0)
#define SCANERF_STORE_GUARD()                    \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        asm volatile("s_nop 7\n\ts_nop 7");      \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
#else
#define SCANERF_LOAD_GUARD() do { } while (0)
#define SCANERF_STORE_GUARD() do { } while (0)
#endif

// Tuning switches of the A/B experiments (launch shapes, alternative producers): environment variables read at launch time ONLY in
// a library built with -DSCANERF_EXPERIMENTS (make EXP=1); the product build compiles every one of them to its default -- no
// getenv, no hidden global state on a launch path (thread-safe, re-entrant, no globals).
#ifdef SCANERF_EXPERIMENTS
#include <stdlib.h>
inline int tune_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
inline bool tune_set(const char *name) { return getenv(name) != nullptr; }
#else
inline int tune_int(const char *, int dflt) { return dflt; }
inline bool tune_set(const char *) { return false; }
#endif

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Grid for memory-bound 1-thread-per-item kernels: enough blocks to fill 256 CUs several
// times over, grid-stride the rest.
inline int stream_grid(int64_t items, int block, int max_blocks = kNumCU * 16)
{
    int64_t g = (items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

}  // namespace scanerf
