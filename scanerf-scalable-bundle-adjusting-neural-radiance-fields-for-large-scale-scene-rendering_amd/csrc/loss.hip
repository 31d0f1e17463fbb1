// loss.hip -- photometric loss of the training step and its gradient w.r.t. the per-ray outputs, two launches.
//
// Reference: criterions.py:90,142-144 (MSE over the valid rays' rgb) + tile.py:999 (0.01 * l2_reg_specular, the mean
// over valid rays x 3 channels of sum_i w_i |c_s,i|^2, hashgrid/__init__.py:593).  The reference builds this with
// ~20 torch ops on [B,3] tensors and autograd; at 65 536 rays the host-side launch gaps of those ops cost more than the
// arithmetic, so the fused training step computes loss and dL/d(out_ray) directly.
#include "common.h"

using namespace scanerf;

namespace {

constexpr int kBlocks = 256, kThreads = 256;

// per-block partial sums: [block][3] = (sum of squared rgb errors, sum of the w*|c_s|^2 column, number of valid rays)
__global__ void __launch_bounds__(kThreads) k_loss_partials(const float *__restrict__ out_ray, const float *__restrict__ target,
                                                            const uint8_t *__restrict__ valid, int B, float *__restrict__ partials)
{
    float se = 0.0f, w2 = 0.0f, n = 0.0f;
    for (int r = blockIdx.x * kThreads + threadIdx.x; r < B; r += kBlocks * kThreads) {
        if (valid && !valid[r]) continue;
        const float4 o = *reinterpret_cast<const float4 *>(out_ray + (size_t)r * SCANERF_RAY_OUT);
        const float dx = o.x - target[3 * r], dy = o.y - target[3 * r + 1], dz = o.z - target[3 * r + 2];
        se += dx * dx + dy * dy + dz * dz;
        w2 += out_ray[(size_t)r * SCANERF_RAY_OUT + 14];
        n += 1.0f;
    }
    __shared__ float red[3][kThreads];
    red[0][threadIdx.x] = se; red[1][threadIdx.x] = w2; red[2][threadIdx.x] = n;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {  // fixed tree: deterministic
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 3) partials[blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void __launch_bounds__(kThreads) k_loss_grad(const float *__restrict__ out_ray, const float *__restrict__ target,
                                                        const uint8_t *__restrict__ valid, int B, float reg,
                                                        const float *__restrict__ partials, float *__restrict__ grad_out,
                                                        float *__restrict__ loss)
{
    __shared__ float tot[3];
    if (threadIdx.x < 3) {  // every block adds the 256 partials in the same order
        float s = 0.0f;
        for (int b = 0; b < kBlocks; ++b) s += partials[b * 3 + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    const float nv3 = 3.0f * tot[2];
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = nv3 > 0.0f ? (tot[0] + reg * tot[1]) / nv3 : 0.0f;
    const float ginv = nv3 > 0.0f ? 1.0f / nv3 : 0.0f;
    for (int r = blockIdx.x * kThreads + threadIdx.x; r < B; r += gridDim.x * kThreads) {
        float4 g0 = make_float4(0, 0, 0, 0), g3 = make_float4(0, 0, 0, 0);
        if (!valid || valid[r]) {
            const float4 o = *reinterpret_cast<const float4 *>(out_ray + (size_t)r * SCANERF_RAY_OUT);
            g0.x = 2.0f * (o.x - target[3 * r]) * ginv;
            g0.y = 2.0f * (o.y - target[3 * r + 1]) * ginv;
            g0.z = 2.0f * (o.z - target[3 * r + 2]) * ginv;
            g3.z = reg * ginv;  // column 14
        }
        float4 *g = reinterpret_cast<float4 *>(grad_out + (size_t)r * SCANERF_RAY_OUT);
        g[0] = g0;
        g[1] = make_float4(0, 0, 0, 0);
        g[2] = make_float4(0, 0, 0, 0);
        g[3] = g3;
    }
}

// ---- the complete per-tile render: foreground + T_left * background (tile.py:666-690), loss of tile.py:880-1015 ----------
// pred = fg.rgb + fg.T_left * bg.rgb;  loss = mean over ALL rays x 3 of (pred - target)^2
//        + reg * (sum_{fg-valid} fg[:,14] / (3 n_fg) + sum_{bg-valid} bg[:,14] / (3 n_bg)).
// partials per block: [se, w2_fg, n_fg, w2_bg, n_bg]
__global__ void __launch_bounds__(kThreads) k_loss_partials_fgbg(const float *__restrict__ fg, const float *__restrict__ bg,
                                                                 const float *__restrict__ target, const uint8_t *__restrict__ vfg,
                                                                 const uint8_t *__restrict__ vbg, int B, float *__restrict__ partials)
{
    float acc[5] = { 0, 0, 0, 0, 0 };
    for (int r = blockIdx.x * kThreads + threadIdx.x; r < B; r += kBlocks * kThreads) {
        const float *f = fg + (size_t)r * SCANERF_RAY_OUT, *b = bg + (size_t)r * SCANERF_RAY_OUT;
        const float T = f[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float e = f[c] + T * b[c] - target[3 * r + c];
            acc[0] += e * e;
        }
        if (!vfg || vfg[r]) { acc[1] += f[14]; acc[2] += 1.0f; }
        if (!vbg || vbg[r]) { acc[3] += b[14]; acc[4] += 1.0f; }
    }
    __shared__ float red[5][kThreads];
#pragma unroll
    for (int k = 0; k < 5; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {  // fixed tree: deterministic
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 5; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 5) partials[blockIdx.x * 5 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void __launch_bounds__(kThreads) k_loss_grad_fgbg(const float *__restrict__ fg, const float *__restrict__ bg,
                                                             const float *__restrict__ target, const uint8_t *__restrict__ vfg,
                                                             const uint8_t *__restrict__ vbg, int B, float reg,
                                                             const float *__restrict__ partials, float *__restrict__ gfg,
                                                             float *__restrict__ gbg, float *__restrict__ loss)
{
    __shared__ float tot[5];
    if (threadIdx.x < 5) {
        float s = 0.0f;
        for (int b = 0; b < kBlocks; ++b) s += partials[b * 5 + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    const float n3 = 3.0f * (float)B;
    const float l2f = tot[2] > 0.0f ? 1.0f / (3.0f * tot[2]) : 0.0f, l2b = tot[4] > 0.0f ? 1.0f / (3.0f * tot[4]) : 0.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = tot[0] / n3 + reg * (tot[1] * l2f + tot[3] * l2b);
    for (int r = blockIdx.x * kThreads + threadIdx.x; r < B; r += gridDim.x * kThreads) {
        const float *f = fg + (size_t)r * SCANERF_RAY_OUT, *b = bg + (size_t)r * SCANERF_RAY_OUT;
        const float T = f[4];
        float gp[3], gT = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gp[c] = 2.0f * (f[c] + T * b[c] - target[3 * r + c]) / n3;
            gT += gp[c] * b[c];
        }
        float4 *g = reinterpret_cast<float4 *>(gfg + (size_t)r * SCANERF_RAY_OUT), *h = reinterpret_cast<float4 *>(gbg + (size_t)r * SCANERF_RAY_OUT);
        g[0] = make_float4(gp[0], gp[1], gp[2], 0.0f);
        g[1] = make_float4(gT, 0, 0, 0);   // column 4 = T_left
        g[2] = make_float4(0, 0, 0, 0);
        g[3] = make_float4(0, 0, (!vfg || vfg[r]) ? reg * l2f : 0.0f, 0);
        h[0] = make_float4(gp[0] * T, gp[1] * T, gp[2] * T, 0.0f);
        h[1] = make_float4(0, 0, 0, 0);
        h[2] = make_float4(0, 0, 0, 0);
        h[3] = make_float4(0, 0, (!vbg || vbg[r]) ? reg * l2b : 0.0f, 0);
    }
}

}  // namespace

SCANERF_API int scanerf_photometric_loss_scratch_floats(void) { return kBlocks * 5; }

// Loss of the complete per-tile render and its gradients w.r.t. the two branches' per-ray outputs, two launches:
// pred = fg.rgb + fg.T_left * bg.rgb (tile.py:666-690), MSE over all rays (criterions.py:142-144) + reg * the two branches'
// l2_reg_specular (tile.py:999).  Rays a branch does not render hold (0, .., T_left = 1) (render_forward's invalid-ray output).
SCANERF_API int scanerf_photometric_loss_grad_fgbg(const float *out_fg, const float *out_bg, const float *target,
                                                   const uint8_t *valid_fg, const uint8_t *valid_bg, float reg_weight,
                                                   float *grad_fg, float *grad_bg, float *loss, float *scratch, int B,
                                                   scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 1, "photometric_loss_grad_fgbg: B=%d", B);
    SCANERF_REQUIRE(out_fg && out_bg && target && grad_fg && grad_bg && loss && scratch, "photometric_loss_grad_fgbg: null pointer");
    SCANERF_REQUIRE((((uintptr_t)grad_fg | (uintptr_t)grad_bg) & 15) == 0, "photometric_loss_grad_fgbg: gradients must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_loss_partials_fgbg, dim3(kBlocks), dim3(kThreads), 0, st, out_fg, out_bg, target, valid_fg, valid_bg, B, scratch);
    hipLaunchKernelGGL(k_loss_grad_fgbg, dim3(stream_grid(B, kThreads, kBlocks)), dim3(kThreads), 0, st, out_fg, out_bg, target,
                       valid_fg, valid_bg, B, reg_weight, scratch, grad_fg, grad_bg, loss);
    return check_launch("photometric_loss_grad_fgbg");
}


SCANERF_API int scanerf_photometric_loss_grad(const float *out_ray, const float *target, const uint8_t *ray_valid,
                                              float reg_weight, float *grad_out, float *loss, float *scratch, int B,
                                              scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0, "photometric_loss_grad: B=%d", B);
    SCANERF_REQUIRE(out_ray && target && grad_out && loss && scratch, "photometric_loss_grad: null pointer");
    SCANERF_REQUIRE(((uintptr_t)out_ray & 15) == 0 && ((uintptr_t)grad_out & 15) == 0,
                    "photometric_loss_grad: out_ray / grad_out must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_loss_partials, dim3(kBlocks), dim3(kThreads), 0, st, out_ray, target, ray_valid, B, scratch);
    hipLaunchKernelGGL(k_loss_grad, dim3(stream_grid(B > 0 ? B : 1, kThreads, kBlocks)), dim3(kThreads), 0, st, out_ray, target,
                       ray_valid, B, reg_weight, scratch, grad_out, loss);
    return check_launch("photometric_loss_grad");
}
