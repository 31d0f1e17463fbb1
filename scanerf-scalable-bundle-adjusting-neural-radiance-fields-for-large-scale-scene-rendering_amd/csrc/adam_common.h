// adam_common.h -- one element of the fused sparse Adam (cuda/adam_kernel.cu:24-69, :98-144), shared by the stand-alone
// kernel (adam.hip) and the table-gradient accumulate's epilogue (scatter.hip).  The same IEEE sequence as the C oracle:
// contraction is switched off locally, whatever the flags of the including file.
#pragma once
#include "common.h"

namespace scanerf {

struct AdamArgs {
    float lr, beta1, beta2, eps, bc1, bc2;  // bc = 1 - beta^t, computed once on the host (t = step + 1)
};
inline AdamArgs make_adam_args(float lr, float beta1, float beta2, float eps, int step)
{
    const float t = (float)(step + 1);
    return AdamArgs{ lr, beta1, beta2, eps, 1.0f - powf(beta1, t), 1.0f - powf(beta2, t) };
}

// one element; returns false when it is untouched (g == 0).  HALF_STATE: fp16 moments with LOSS_SCALE 128.
template <bool HALF_STATE>
__device__ __forceinline__ bool adam_update_one(float &p, float &mi, float &vi, float g_raw, const AdamArgs &a)
{
#pragma clang fp contract(off)
    constexpr float LS = 128.0f;
    const float g = HALF_STATE ? g_raw * LS : g_raw;
    if (g == 0.0f) return false;
    mi = a.beta1 * mi + (1.0f - a.beta1) * g;
    vi = a.beta2 * vi + (1.0f - a.beta2) * g * g;
    const float step_size = a.lr / a.bc1;
    float denom, upd;
    if (HALF_STATE) {
        denom = sqrtf(vi / (a.bc2 * LS * LS)) + a.eps;
        upd = step_size * mi / (denom * LS);
    } else {
        denom = sqrtf(vi / a.bc2) + a.eps;
        upd = step_size * mi / denom;
    }
    p = p - upd;
    return true;
}

}  // namespace scanerf
