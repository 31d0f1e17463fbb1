// adam.hip -- fused sparse Adam step (gfx950).  Built with -ffp-contract=off so that the
// update is the same IEEE sequence as the C oracle (bit-exact parity).
//
// Reference behaviour: cuda/adam_kernel.cu:24-69 (fp32 moments), :98-144 (fp16 moments,
// LOSS_SCALE 128), host wrappers :72-94 / :147-169 (the kernel sees step+1).
//
// HBM-bound: 4 B/param grad scan + 28 B/param (fp32) for touched entries.  One thread owns
// half a row of 8 parameters = one 16-B vector per array; untouched vectors cost the grad read
// only.  The bias corrections are the same for every element and are computed once on
// the host (powf), not per thread.
#include <hip/hip_fp16.h>
#include <math.h>

#include "adam_common.h"

using namespace scanerf;

namespace {

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
__device__ __forceinline__ bool update_one(float &p, float &mi, float &vi, float g_raw, const AdamArgs &a)
{
    return adam_update_one<HALF_STATE>(p, mi, vi, g_raw, a);
}

// One thread owns HALF a row: 4 consecutive parameters = one 16-B vector of grad / params (and of the fp32 moments; 8 B of
// fp16 moments), so that every access of a wave is one contiguous 1 KB (the first version walked the 8 elements of a row
// with predicated 4-byte accesses at a 32-B lane stride: 3.4 TB/s on a 2 GB table against 5+ here).  A vector none of whose
// gradients is non-zero costs the gradient read only; in a touched vector the untouched elements are written back unchanged.
template <bool HALF_STATE>
__global__ void __launch_bounds__(256) k_adam(float *__restrict__ params, const float *__restrict__ grad,
                                              typename Moments<HALF_STATE>::T *__restrict__ m,
                                              typename Moments<HALF_STATE>::T *__restrict__ v, AdamArgs a, int64_t K,
                                              int param_dim)
{
    using MT = typename Moments<HALF_STATE>::T;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < 2 * K; q += (int64_t)gridDim.x * blockDim.x) {
        const int k0 = (int)(q & 1) * 4;  // first element of this half row
        if (k0 >= param_dim) continue;
        const float4 g4 = reinterpret_cast<const float4 *>(grad)[q];
        const float g[4] = { g4.x, g4.y, g4.z, g4.w };
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) any |= (k0 + j < param_dim) && g[j] != 0.0f;
        if (!any) continue;
        float4 p4 = reinterpret_cast<float4 *>(params)[q];
        float p[4] = { p4.x, p4.y, p4.z, p4.w }, mm[4], vv[4];
        if (HALF_STATE) {
            const uint2 mr = reinterpret_cast<const uint2 *>(m)[q], vr = reinterpret_cast<const uint2 *>(v)[q];
            const __half2 m01 = *reinterpret_cast<const __half2 *>(&mr.x), m23 = *reinterpret_cast<const __half2 *>(&mr.y);
            const __half2 v01 = *reinterpret_cast<const __half2 *>(&vr.x), v23 = *reinterpret_cast<const __half2 *>(&vr.y);
            mm[0] = __low2float(m01); mm[1] = __high2float(m01); mm[2] = __low2float(m23); mm[3] = __high2float(m23);
            vv[0] = __low2float(v01); vv[1] = __high2float(v01); vv[2] = __low2float(v23); vv[3] = __high2float(v23);
        } else {
            const float4 m4 = reinterpret_cast<const float4 *>(m)[q], v4 = reinterpret_cast<const float4 *>(v)[q];
            mm[0] = m4.x; mm[1] = m4.y; mm[2] = m4.z; mm[3] = m4.w;
            vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
        }
        bool touched[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) touched[j] = (k0 + j < param_dim) && update_one<HALF_STATE>(p[j], mm[j], vv[j], g[j], a);
        reinterpret_cast<float4 *>(params)[q] = make_float4(p[0], p[1], p[2], p[3]);
        if (HALF_STATE) {
            // an untouched element keeps its stored bits (a float round trip of a half is exact, so converting back is too)
            MT *mo = m + q * 4, *vo = v + q * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (touched[j]) {
                    Moments<HALF_STATE>::put(mo, j, mm[j]);
                    Moments<HALF_STATE>::put(vo, j, vv[j]);
                }
        } else {
            reinterpret_cast<float4 *>(m)[q] = make_float4(mm[0], mm[1], mm[2], mm[3]);
            reinterpret_cast<float4 *>(v)[q] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        }
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
    SCANERF_REQUIRE((((uintptr_t)grad | (uintptr_t)params) & 15) == 0 && (((uintptr_t)m | (uintptr_t)v) & (HALF_STATE ? 7 : 15)) == 0,
                    "%s: params, grad and moments must be 16-byte aligned (8 for fp16 moments)", what);
    const AdamArgs a = make_adam_args(lr, beta1, beta2, eps, step);
    using MT = typename Moments<HALF_STATE>::T;
    hipLaunchKernelGGL((k_adam<HALF_STATE>), dim3(stream_grid(2 * K, 256)), dim3(256), 0, st, params, grad, (MT *)m,
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
