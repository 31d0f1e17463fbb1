// dda_device.h -- box clipping and the DDA grid walker shared by the training sampler (rays.hip)
// and the render-time samplers (render_time.hip).  Include only from files built with
// -ffp-contract=off: results must match the C oracle bit for bit.
// Reference behaviour: cuda/include/cuda_utils.h:564-613 (RayAABBIntersection),
// cuda/include/dda.h:206-268 (DDASatateScene_v2), cuda/include/cutil_math.h:913-926.
#pragma once
#include "common.h"

namespace scanerf {

struct F2 { float x, y; };

__device__ __forceinline__ float safe_div(float a, float b) { return b != 0.0f ? a / b : 100000000.0f; }

// slab test; interval starts as [0, 1e5]; miss -> (-1,-1)
__device__ __forceinline__ F2 clip_box(const float o[3], const float d[3], const float c[3], const float h[3])
{
    float lo_acc = 0.0f, hi_acc = 100000.0f;
    bool miss = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float inv = safe_div(1.0f, d[k]);
        float lo = (c[k] - h[k] - o[k]) * inv;
        float hi = (c[k] + h[k] - o[k]) * inv;
        if (hi < lo) { float t = lo; lo = hi; hi = t; }
        if (!miss) {
            if (hi < lo_acc || lo > hi_acc) miss = true;
            else {
                lo_acc = lo > lo_acc ? lo : lo_acc;
                hi_acc = hi < hi_acc ? hi : hi_acc;
                if (lo_acc > hi_acc) miss = true;
            }
        }
    }
    F2 r;
    r.x = miss ? -1.0f : lo_acc;
    r.y = miss ? -1.0f : hi_acc;
    return r;
}

// ------------------------------------------------------------------ DDA sampler
struct Walker {
    int step[3], cell[3], side[3];
    float tmax[3], tdelta[3];
    int mx, my, mz;
    float t0, t1;

    __device__ __forceinline__ void start(const float og[3], const float d[3], F2 span, const int side_[3],
                                          const float cs[3])
    {
        float p[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            side[k] = side_[k];
            p[k] = og[k] + span.x * d[k];
            int c = (int)(p[k] / cs[k]);
            c = c < 0 ? 0 : c;
            c = c > side_[k] - 1 ? side_[k] - 1 : c;
            cell[k] = c;
            step[k] = d[k] >= 0.0f ? 1 : -1;
        }
        t0 = span.x;
        t1 = span.y;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float nb = (float)(cell[k] + step[k]) * cs[k];
            if (step[k] < 0) nb += cs[k];
            tmax[k] = fmaxf(safe_div(nb - p[k], d[k]), 0.0f) + t0;
            tdelta[k] = fabsf(safe_div(cs[k], d[k]));
        }
    }
    __device__ __forceinline__ bool done() const
    {
        return cell[0] < 0 || cell[1] < 0 || cell[2] < 0 || cell[0] >= side[0] || cell[1] >= side[1] ||
               cell[2] >= side[2] || (tmax[0] <= 0 && tmax[1] <= 0 && tmax[2] <= 0);
    }
    __device__ __forceinline__ void pick()
    {
        mx = (tmax[0] < tmax[1]) & (tmax[0] <= tmax[2]);
        my = (tmax[1] < tmax[2]) & (tmax[1] <= tmax[0]);
        mz = !(mx | my);
        t1 = mx ? tmax[0] : (my ? tmax[1] : tmax[2]);
    }
    __device__ __forceinline__ void advance()
    {
        t0 = t1;
        tmax[0] += (float)mx * tdelta[0];
        tmax[1] += (float)my * tdelta[1];
        tmax[2] += (float)mz * tdelta[2];
        cell[0] += mx * step[0];
        cell[1] += my * step[1];
        cell[2] += mz * step[2];
    }
};


}  // namespace scanerf
