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

// After global stores in kernels that also run matrix instructions.  A vector-memory store reads its data registers for a few
// cycles after it issues.  Measured on MI355X / ROCm 7.2 (tools/step_determinism.py): in 1 of ~4e5 tiles the x-stash store of
// the forward kernel picked up, in its last quarter-wave, a register that later code (the next layer's MFMA results are
// allocated over the stored values) had already rewritten -- nothing in the generated code holds such writers back.  Sixteen
// wait states behind the stores, fenced for the scheduler, and no launch in 600 differed.
// After the last of a batch of gathers has been consumed, before other code reuses the loads' destination registers.  Measured on
// MI355X / ROCm 7.2 (tools/fwd_fault_rate.py): the forward kernel copies its 16 encoder outputs into the registers that the last
// level's eight corner loads had returned into (v_mov_b64, a dozen instructions behind the s_waitcnt that released the last
// load's consumer); in 5 of 59 first launches on cold caches ONE such copy came out wrong in its low register, lanes 48-63 --
// the value a late part of the load's return had written over it, as far as can be told.  With wait states between the
// encoder and whatever follows: 0 of 119 (sixteen; thirty-two now, and sixteen between the encoder's level groups, whose
// registers are reused the same way).  (Same family as SCANERF_STORE_GUARD: vector memory still touches a register a few
// cycles after the counters say it is done with it.)
#define SCANERF_LOAD_GUARD()                                                         \
    do {                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                           \
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");                    \
        __builtin_amdgcn_sched_barrier(0);                                           \
    } while (0)
// The same wait placed BEHIND the arrival of particular loads: the asm names the loaded registers, so the compiler's s_waitcnt
// for them comes first (a guard without operands would be scheduled in front of the wait it is meant to follow).
#define SCANERF_LOAD_GUARD_ON2(r0, r1)                                               \
    do {                                                                             \
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(r0), "+v"(r1)); \
    } while (0)

#define SCANERF_STORE_GUARD()                    \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        asm volatile("s_nop 7\n\ts_nop 7");      \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

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
