// adam.hip -- fused sparse Adam step (gfx950).  Built with -ffp-contract=off so that the
// update is the same IEEE sequence as the C oracle (bit-exact parity).
//
// Reference behaviour: cuda/adam_kernel.cu:24-69 (fp32 moments), :98-144 (fp16 moments,
// LOSS_SCALE 128), host wrappers :72-94 / :147-169 (the kernel sees step+1).
//
// HBM-bound: 4 B/param grad scan + 28 B/param (fp32) for touched entries.  One thread owns
// one row of 8 parameters = two 16-B vectors per array; untouched rows cost the grad read
// only.  The bias corrections are the same for every element and are computed once on
// the host (powf), not per thread.
#include <hip/hip_fp16.h>
#include <math.h>

#include "common.h"

using namespace scanerf;

namespace {

struct AdamArgs {
    float lr, beta1, beta2, eps, bc1, bc2;
};

template <bool HALF_STATE>
struct Moments;
template <>
struct Moments<false> {
    using T = float;
    static __device__ __forceinline__ float get(const T *p, int64_t i) { return p[i]; }
    static __device__ __forceinline__ void put(T *p, int64_t i, float v) { p[i] = v; }
};
template <>
struct Moments<true> {
    using T = __half;
    static __device__ __forceinline__ float get(const T *p, int64_t i) { return __half2float(p[i]); }
    static __device__ __forceinline__ void put(T *p, int64_t i, float v) { p[i] = __float2half(v); }
};

template <bool HALF_STATE>
__device__ __forceinline__ void update_one(float *params, typename Moments<HALF_STATE>::T *m,
                                           typename Moments<HALF_STATE>::T *v, int64_t i, float g_raw,
                                           const AdamArgs &a)
{
    constexpr float LS = 128.0f;
    float g = HALF_STATE ? g_raw * LS : g_raw;
    if (g == 0.0f) return;
    float mi = a.beta1 * Moments<HALF_STATE>::get(m, i) + (1.0f - a.beta1) * g;
    float vi = a.beta2 * Moments<HALF_STATE>::get(v, i) + (1.0f - a.beta2) * g * g;
    float step_size = a.lr / a.bc1;
    float denom, upd;
    if (HALF_STATE) {
        denom = sqrtf(vi / (a.bc2 * LS * LS)) + a.eps;
        upd = step_size * mi / (denom * LS);
    } else {
        denom = sqrtf(vi / a.bc2) + a.eps;
        upd = step_size * mi / denom;
    }
    params[i] = params[i] - upd;
    Moments<HALF_STATE>::put(m, i, mi);
    Moments<HALF_STATE>::put(v, i, vi);
}

template <bool HALF_STATE>
__global__ void __launch_bounds__(256) k_adam(float *__restrict__ params, const float *__restrict__ grad,
                                              typename Moments<HALF_STATE>::T *__restrict__ m,
                                              typename Moments<HALF_STATE>::T *__restrict__ v, AdamArgs a, int64_t K,
                                              int param_dim)
{
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < K; r += (int64_t)gridDim.x * blockDim.x) {
        const float4 *g4 = reinterpret_cast<const float4 *>(grad + r * 8);
        float4 ga = g4[0], gb = g4[1];
        float g[8] = { ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w };
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < param_dim) update_one<HALF_STATE>(params, m, v, r * 8 + k, g[k], a);
    }
}

template <bool HALF_STATE>
int launch(float *params, const float *grad, void *m, void *v, float lr, float beta1, float beta2, float eps,
           int step, int64_t K, int param_dim, hipStream_t st, const char *what)
{
    SCANERF_REQUIRE(K >= 0 && param_dim >= 1 && param_dim <= 8, "%s: K=%lld param_dim=%d (rows are 8 wide)", what,
                    (long long)K, param_dim);
    if (K == 0) return 0;
    SCANERF_REQUIRE(params && grad && m && v, "%s: null pointer", what);
    SCANERF_REQUIRE(((uintptr_t)grad & 15) == 0, "%s: grad must be 16-byte aligned", what);
    float t = (float)(step + 1);
    AdamArgs a{ lr, beta1, beta2, eps, 1.0f - powf(beta1, t), 1.0f - powf(beta2, t) };
    using MT = typename Moments<HALF_STATE>::T;
    hipLaunchKernelGGL((k_adam<HALF_STATE>), dim3(stream_grid(K, 256)), dim3(256), 0, st, params, grad, (MT *)m,
                       (MT *)v, a, K, param_dim);
    return check_launch(what);
}

}  // namespace

SCANERF_API int scanerf_adam_step(float *params, const float *grad, float *exp_avg, float *exp_avg_sq, float lr,
                                  float beta1, float beta2, float eps, int step, int64_t K, int param_dim,
                                  scanerf_stream_t stream)
{
    return launch<false>(params, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, K, param_dim,
                         (hipStream_t)stream, "adam_step");
}

SCANERF_API int scanerf_adam_step_fp16(float *params, const float *grad, void *exp_avg, void *exp_avg_sq, float lr,
                                       float beta1, float beta2, float eps, int step, int64_t K, int param_dim,
                                       scanerf_stream_t stream)
{
    return launch<true>(params, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, K, param_dim,
                        (hipStream_t)stream, "adam_step_fp16");
}
