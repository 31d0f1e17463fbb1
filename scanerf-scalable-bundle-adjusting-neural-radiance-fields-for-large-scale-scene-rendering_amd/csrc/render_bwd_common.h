// render_bwd_common.h -- arguments and small helpers shared by the two fused backward kernels
// (render_bwd.hip: f32 MFMA; render_bwd_h3.hip: split-f16 MFMA).
#pragma once
#include "render_device.h"
#include "scatter_common.h"

namespace scanerf {

struct BwdArgs {
    RenderArgs f;              // forward inputs (out_ray = forward outputs, read-only here)
    const float *grad_out;     // [B,16] dL/d(out_ray)
    const float *tile_T;       // [B, ceil(S/16)] from the forward: transmittance entering each 16-sample tile
    float *dfeat;              // [16][B*S][2]; may be null when recs != nullptr
    float *dw_partial;         // [nwaves][SCANERF_PARAMSIZE], zero-filled by the host wrapper
    const float *xstash;       // optional [B*S][2][16]: the forward's encoder outputs (skips the re-gather)
    float *g_dnorm;            // optional [B, ntiles]: dL/d|d| partials (through delta = dist*|d|)
    float *g_rowsum;           // optional [B, 2, 64]: sum_s dL/d(dir layer-0 pre-activation), for dL/dSH
    float *g_raypos;           // optional [B, 6] (t16 kernel, needs f.jstash): dL/d(rays_o), dL/d(rays_d) through the sample
                               // positions (feature gradients x the forward's position Jacobians x the contraction's)
    // fused table-gradient producer (scatter.hip): when recs != nullptr the kernel appends the scatter
    // records itself (the stores hide under the MFMA work) and dfeat becomes optional
    BinGeom bins;
    const uint32_t *bin_rowprefix, *bin_starts;
    Rec *recs;
    uint32_t *maxbits;
    float *grad_features;      // only touched if the record workspace overflows
    int park;                  // t16s kernel: waves 4-7 park a tile's feature gradients in LDS and emit its records behind the NEXT
                               // tile's forward recompute (render_bwd_t16.hip "skewed emission"); set by its launcher when the LDS has room
};

__device__ __forceinline__ float dgauss(float u, float a) { return -100.0f * u * a; }  // d/du exp(-50 u^2)

// Hide a value's provenance from the optimiser.  The kernel RECOMPUTES cheap activations
// (exp(-50 u^2), SH of the ray) at each use instead of holding them; without this the compiler
// common-subexpression-eliminates the recomputation and keeps 32-64 extra registers alive across
// the phases, which is what pushes the wave into scratch.
__device__ __forceinline__ v16f opaque16(v16f v)
{
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float opaque1(float v)
{
    asm volatile("" : "+v"(v));
    return v;
}


// launchers of the two arithmetics (defined next to their kernels); lds_extra = bytes of record cursors
int launch_render_bwd_f32(const BwdArgs &a, int feat_dtype, int blocks, size_t lds_extra, hipStream_t st);
int launch_render_bwd_h3(const BwdArgs &a, int feat_dtype, int blocks, size_t lds_extra, hipStream_t st);
int launch_render_bwd_t16(const BwdArgs &a, int feat_dtype, int blocks, size_t lds_extra, hipStream_t st, bool split);

}  // namespace scanerf
