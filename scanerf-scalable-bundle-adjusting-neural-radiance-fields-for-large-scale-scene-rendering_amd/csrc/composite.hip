// composite.hip -- alpha compositing of per-sample decoder outputs along rays, and its adjoint, as stand-alone ops (gfx950).
//
// What they replace on the op-by-op route (an unchanged caller of the binding surface): HashGrid.cal_integrate_weight
// (hashgrid/__init__.py:344-360), the four HashGrid.accumulate calls and the l2_reg_specular sum (:362-366, :564-574, :591-594) and
// their autograd -- ~35 torch kernels over [B,S]- and [B,S,3]-sized temporaries per step (3 ms at 65 536 x 128 on MI355X) against
// two launches that read every per-sample value once (0.35 ms).  The fused kernels (render.hip, render_bwd*.hip) carry the same
// arithmetic inside their tile loops; this file is the version for callers that keep the decoder outputs as tensors.
//
//   delta_s = dists_s |d|  (last sample 1e10 when `infinity`);  alpha_s = 1 - exp(-sigma_s delta_s);
//   T_s = prod_{j<s} (1 - alpha_j + 1e-6);  w_s = alpha_s T_s;  T_left = T_{S-1} (the transmittance BEFORE the last sample:
//   the reference's observable quirk, :358-360);  per ray: depth = sum w z, diffuse = sum w c_d, tint = sum w t,
//   specular = sum w t c_s, rgb = clamp(diffuse + specular, 0, 1), and sum w |c_s|^2 (the numerator of l2_reg_specular, whose
//   weights are DETACHED in the reference, :593) -- the 16 columns of out_ray as the fused forward writes them.
//
// One wave per ray, lane = sample, 64 samples per round with the running transmittance carried between rounds; products and
// suffix sums by wave shuffles (fixed order: bit-reproducible).
#include "common.h"

using namespace scanerf;

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float wave_sum64(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// inclusive prefix product over the lanes
__device__ __forceinline__ float scan_mul(float v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(v, off, 64);
        if (lane >= off) v *= t;
    }
    return v;
}
// inclusive SUFFIX sum over the lanes (lane l: sum of lanes >= l)
__device__ __forceinline__ float scan_add_down(float v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_down(v, off, 64);
        if (lane + off < 64) v += t;
    }
    return v;
}

struct Sample {
    float sigma, z, delta, alpha, u;   // u = 1 - alpha + 1e-6
    float cd[3], cs[3], tn[3];
};
__device__ __forceinline__ Sample load_sample(const float *sigma, const float *dif, const float *spec, const float *tint, const float *z_vals,
                                              const float *dists, size_t e, bool live, bool last, float dnorm, int infinity)
{
    Sample s = {};
    s.u = 1.0f;
    if (!live) return s;
    s.sigma = sigma[e];
    s.z = z_vals[e];
    s.delta = (infinity && last) ? 1e10f : dists[e] * dnorm;
    s.alpha = 1.0f - __expf(-s.sigma * s.delta);
    s.u = 1.0f - s.alpha + 1e-6f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        s.cd[c] = dif[3 * e + c];
        s.cs[c] = spec[3 * e + c];
        s.tn[c] = tint[3 * e + c];
    }
    return s;
}

__global__ void __launch_bounds__(kThreads) k_composite_fwd(const float *__restrict__ sigma, const float *__restrict__ dif,
                                                            const float *__restrict__ spec, const float *__restrict__ tint,
                                                            const float *__restrict__ z_vals, const float *__restrict__ dists,
                                                            const float *__restrict__ rays_d, float *__restrict__ out_ray,
                                                            float *__restrict__ weights, int B, int S, int infinity)
{
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * (kThreads >> 6);
    for (int i = blockIdx.x * (kThreads >> 6) + (threadIdx.x >> 6); i < B; i += nw) {
        const float dx = rays_d[3 * i], dy = rays_d[3 * i + 1], dz = rays_d[3 * i + 2];
        const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
        float T = 1.0f, T_left = 1.0f;
        float acc[11] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };   // depth, diffuse 3, tint 3, specular 3, w |c_s|^2
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + lane;
            const bool live = s < S;
            const size_t e = (size_t)i * S + (live ? s : 0);
            const Sample p = load_sample(sigma, dif, spec, tint, z_vals, dists, e, live, s == S - 1, dnorm, infinity);
            const float incl = scan_mul(p.u, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float Ts = T * excl, w = p.alpha * Ts;
            if (live) {
                if (weights) weights[e] = w;
                if (s == S - 1) T_left = Ts;
                acc[0] += w * p.z;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    acc[1 + c] += w * p.cd[c];
                    acc[4 + c] += w * p.tn[c];
                    acc[7 + c] += w * p.tn[c] * p.cs[c];
                    acc[10] += w * p.cs[c] * p.cs[c];
                }
            }
            T *= __shfl(incl, 63, 64);
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) acc[k] = wave_sum64(acc[k]);
        T_left = __shfl(T_left, (S - 1) & 63, 64);
        if (lane == 0) {
            float *o = out_ray + (size_t)i * SCANERF_RAY_OUT;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                o[c] = fminf(fmaxf(acc[1 + c] + acc[7 + c], 0.0f), 1.0f);
                o[5 + c] = acc[1 + c];
                o[8 + c] = acc[7 + c];
                o[11 + c] = acc[4 + c];
            }
            o[3] = acc[0];
            o[4] = T_left;
            o[14] = acc[10];
            o[15] = 0.0f;
        }
    }
}

// Adjoint.  g_out [B,16]: dL/d(out_ray) (column 14 = the gradient of sum w |c_s|^2, with w detached); g_w [B,S] (may be null):
// dL/d(weights).  Writes dL/d(sigma) [N], dL/d(diffuse / specular / tint) [N,3], and g_dnorm [B] = dL/d|d| through delta.
//   dL/dw_s   = g_depth z + g_dif . c_d + g_tint . t + g_spec . (t c_s) + g_w
//   dL/dalpha_s = dL/dw_s T_s - (sum_{k>s} dL/dw_k w_k + g_Tleft T_left [s < S-1]) / u_s ,   dalpha/dsigma = delta (1 - alpha)
__global__ void __launch_bounds__(kThreads) k_composite_bwd(const float *__restrict__ sigma, const float *__restrict__ dif,
                                                            const float *__restrict__ spec, const float *__restrict__ tint,
                                                            const float *__restrict__ z_vals, const float *__restrict__ dists,
                                                            const float *__restrict__ rays_d, const float *__restrict__ out_ray,
                                                            const float *__restrict__ g_out, const float *__restrict__ g_w,
                                                            float *__restrict__ g_sigma, float *__restrict__ g_dif,
                                                            float *__restrict__ g_spec, float *__restrict__ g_tint,
                                                            float *__restrict__ g_dnorm, int B, int S, int infinity)
{
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * (kThreads >> 6);
    const int rounds = (S + 63) / 64;
    for (int i = blockIdx.x * (kThreads >> 6) + (threadIdx.x >> 6); i < B; i += nw) {
        const float dx = rays_d[3 * i], dy = rays_d[3 * i + 1], dz = rays_d[3 * i + 2];
        const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
        const float *go = g_out + (size_t)i * SCANERF_RAY_OUT, *fo = out_ray + (size_t)i * SCANERF_RAY_OUT;
        float gd[3], gs[3], gt[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {   // rgb = clamp(diffuse + specular, 0, 1): its gradient passes where the sum is inside (torch.clamp: bounds included)
            const float sum = fo[5 + c] + fo[8 + c];
            const float grgb = (sum >= 0.0f && sum <= 1.0f) ? go[c] : 0.0f;
            gd[c] = go[5 + c] + grgb;
            gs[c] = go[8 + c] + grgb;
            gt[c] = go[11 + c];
        }
        const float gdepth = go[3], gTl = go[4], gl2 = go[14], T_left = fo[4];
        // transmittance entering each round of 64 samples (forward order), then the rounds last -> first with the suffix sum carried
        float Tin[8];   // (S <= 512)
        {
            float T = 1.0f;
            for (int r = 0; r < rounds; ++r) {
                Tin[r] = T;
                const int s = 64 * r + lane;
                const bool live = s < S;
                const size_t e = (size_t)i * S + (live ? s : 0);
                float u = 1.0f;
                if (live) {
                    const float delta = (infinity && s == S - 1) ? 1e10f : dists[e] * dnorm;
                    u = 1.0f - (1.0f - __expf(-sigma[e] * delta)) + 1e-6f;
                }
                T *= __shfl(scan_mul(u, lane), 63, 64);
            }
        }
        float R = 0.0f;   // sum over the samples of LATER rounds of dL/dw w (+ the T_left term, which every sample but the last sees)
        float gdn = 0.0f;
        for (int r = rounds - 1; r >= 0; --r) {
            const int s = 64 * r + lane;
            const bool live = s < S;
            const size_t e = (size_t)i * S + (live ? s : 0);
            const Sample p = load_sample(sigma, dif, spec, tint, z_vals, dists, e, live, s == S - 1, dnorm, infinity);
            const float incl = scan_mul(p.u, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float Ts = Tin[r] * excl, w = p.alpha * Ts;
            float gw = 0.0f;
            if (live) {
                gw = gdepth * p.z + (g_w ? g_w[e] : 0.0f);
#pragma unroll
                for (int c = 0; c < 3; ++c) gw += gd[c] * p.cd[c] + gt[c] * p.tn[c] + gs[c] * p.tn[c] * p.cs[c];
            }
            const float incl_suffix = scan_add_down(live ? gw * w : 0.0f, lane);   // lanes >= l of this round
            const float later = incl_suffix - (live ? gw * w : 0.0f) + R;          // samples after s
            if (live) {
                const float tl = s < S - 1 ? gTl * T_left : 0.0f;
                const float galpha = gw * Ts - (later + tl) / p.u;
                const float dads = p.delta * (1.0f - p.alpha);   // dalpha / dsigma; dalpha / ddelta = sigma (1 - alpha)
                g_sigma[e] = galpha * dads;
                if (!(infinity && s == S - 1)) gdn += galpha * p.sigma * (1.0f - p.alpha) * dists[e];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    g_dif[3 * e + c] = w * gd[c];
                    g_tint[3 * e + c] = w * (gt[c] + gs[c] * p.cs[c]);
                    g_spec[3 * e + c] = w * (gs[c] * p.tn[c] + 2.0f * gl2 * p.cs[c]);
                }
            }
            R += __shfl(incl_suffix, 0, 64);
        }
        gdn = wave_sum64(gdn);
        if (g_dnorm && lane == 0) g_dnorm[i] = gdn;
    }
}

}  // namespace

SCANERF_API int scanerf_composite_forward(const float *sigma, const float *diffuse, const float *specular, const float *tint,
                                          const float *z_vals, const float *dists, const float *rays_d, float *out_ray, float *weights,
                                          int B, int S, int infinity, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 1, "composite_forward: B=%d S=%d", B, S);
    if (B == 0) return 0;
    SCANERF_REQUIRE(sigma && diffuse && specular && tint && z_vals && dists && rays_d && out_ray, "composite_forward: null pointer");
    hipLaunchKernelGGL(k_composite_fwd, dim3(stream_grid((int64_t)B * 64, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, sigma, diffuse,
                       specular, tint, z_vals, dists, rays_d, out_ray, weights, B, S, infinity);
    return check_launch("composite_forward");
}

SCANERF_API int scanerf_composite_backward(const float *sigma, const float *diffuse, const float *specular, const float *tint,
                                           const float *z_vals, const float *dists, const float *rays_d, const float *out_ray,
                                           const float *grad_out, const float *grad_weights, float *g_sigma, float *g_diffuse,
                                           float *g_specular, float *g_tint, float *g_dnorm, int B, int S, int infinity,
                                           scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 1 && S <= 512, "composite_backward: B=%d S=%d (S <= 512)", B, S);
    if (B == 0) return 0;
    SCANERF_REQUIRE(sigma && diffuse && specular && tint && z_vals && dists && rays_d && out_ray && grad_out && g_sigma && g_diffuse &&
                    g_specular && g_tint, "composite_backward: null pointer");
    hipLaunchKernelGGL(k_composite_bwd, dim3(stream_grid((int64_t)B * 64, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, sigma, diffuse,
                       specular, tint, z_vals, dists, rays_d, out_ray, grad_out, grad_weights, g_sigma, g_diffuse, g_specular, g_tint,
                       g_dnorm, B, S, infinity);
    return check_launch("composite_backward");
}
