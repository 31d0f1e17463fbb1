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
// pred = fg.rgb + fg.T_left * bg.rgb;  loss = mean over the rays VALID IN EITHER BRANCH (criterions.py:121-138:
//        input[valid], target[valid] with valid = fore_valid | bg_valid) x 3 of (pred - target)^2
//        + reg * (sum_{fg-valid} fg[:,14] / (3 n_fg) + sum_{bg-valid} bg[:,14] / (3 n_bg)).
// A ray invalid in both branches (under-ground invalidation, occlusion masks) contributes nothing and gets a zero gradient.
// partials per block: [se, w2_fg, n_fg, w2_bg, n_bg, n_union]
__global__ void __launch_bounds__(kThreads) k_loss_partials_fgbg(const float *__restrict__ fg, const float *__restrict__ bg,
                                                                 const float *__restrict__ target, const uint8_t *__restrict__ vfg,
                                                                 const uint8_t *__restrict__ vbg, int B, float *__restrict__ partials)
{
    float acc[6] = { 0, 0, 0, 0, 0, 0 };
    for (int r = blockIdx.x * kThreads + threadIdx.x; r < B; r += kBlocks * kThreads) {
        const float *f = fg + (size_t)r * SCANERF_RAY_OUT, *b = bg + (size_t)r * SCANERF_RAY_OUT;
        const float T = f[4];
        const bool in_fg = !vfg || vfg[r], in_bg = !vbg || vbg[r];
        if (in_fg || in_bg) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float e = f[c] + T * b[c] - target[3 * r + c];
                acc[0] += e * e;
            }
            acc[5] += 1.0f;
        }
        if (in_fg) { acc[1] += f[14]; acc[2] += 1.0f; }
        if (in_bg) { acc[3] += b[14]; acc[4] += 1.0f; }
    }
    __shared__ float red[6][kThreads];
#pragma unroll
    for (int k = 0; k < 6; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {  // fixed tree: deterministic
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 6) partials[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void __launch_bounds__(kThreads) k_loss_grad_fgbg(const float *__restrict__ fg, const float *__restrict__ bg,
                                                             const float *__restrict__ target, const uint8_t *__restrict__ vfg,
                                                             const uint8_t *__restrict__ vbg, int B, float reg,
                                                             const float *__restrict__ partials, float *__restrict__ gfg,
                                                             float *__restrict__ gbg, float *__restrict__ loss)
{
    __shared__ float tot[6];
    if (threadIdx.x < 6) {
        float s = 0.0f;
        for (int b = 0; b < kBlocks; ++b) s += partials[b * 6 + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    const float n3 = 3.0f * tot[5];   // rays valid in either branch
    const float in3 = n3 > 0.0f ? 1.0f / n3 : 0.0f;
    const float l2f = tot[2] > 0.0f ? 1.0f / (3.0f * tot[2]) : 0.0f, l2b = tot[4] > 0.0f ? 1.0f / (3.0f * tot[4]) : 0.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = tot[0] * in3 + reg * (tot[1] * l2f + tot[3] * l2b);
    for (int r = blockIdx.x * kThreads + threadIdx.x; r < B; r += gridDim.x * kThreads) {
        const float *f = fg + (size_t)r * SCANERF_RAY_OUT, *b = bg + (size_t)r * SCANERF_RAY_OUT;
        const float T = f[4];
        const bool in_fg = !vfg || vfg[r], in_bg = !vbg || vbg[r];
        float gp[3], gT = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gp[c] = (in_fg || in_bg) ? 2.0f * (f[c] + T * b[c] - target[3 * r + c]) * in3 : 0.0f;
            gT += gp[c] * b[c];
        }
        float4 *g = reinterpret_cast<float4 *>(gfg + (size_t)r * SCANERF_RAY_OUT), *h = reinterpret_cast<float4 *>(gbg + (size_t)r * SCANERF_RAY_OUT);
        g[0] = make_float4(gp[0], gp[1], gp[2], 0.0f);
        g[1] = make_float4(gT, 0, 0, 0);   // column 4 = T_left
        g[2] = make_float4(0, 0, 0, 0);
        g[3] = make_float4(0, 0, in_fg ? reg * l2f : 0.0f, 0);
        h[0] = make_float4(gp[0] * T, gp[1] * T, gp[2] * T, 0.0f);
        h[1] = make_float4(0, 0, 0, 0);
        h[2] = make_float4(0, 0, 0, 0);
        h[3] = make_float4(0, 0, in_bg ? reg * l2b : 0.0f, 0);
    }
}

// ---- ray-gradient epilogue of the fused backward with the in-kernel position path ------------------------------------------
// dL/d(rays_o) = g_raypos[:, 0:3]; dL/d(rays_d) = g_raypos[:, 3:6] + the two per-ray paths the backward kernel leaves as sums:
// |d| (delta = dist |d|: g_dnorm summed over the tiles) and SH(d / (|d| + 1e-8)) of the decoder's directional layer
// (g_rowsum [B,2,64] = the row sums of its layer-0 pre-activation gradient -> times W[:, 32:48]^T -> through the degree-3
// harmonics and the normalisation).  One wave per ray; what render.ray_gradients_fused did with ~150 torch kernels of
// autograd on [B]-sized tensors (network.py:38-77, 177 for the harmonics and the normalisation).
__global__ void __launch_bounds__(256) k_ray_grad_epilogue(const float *__restrict__ rays_d, const float *__restrict__ blob,
                                                           const float *__restrict__ g_raypos, const float *__restrict__ g_dnorm,
                                                           const float *__restrict__ g_rowsum, const uint8_t *__restrict__ ray_valid,
                                                           float *__restrict__ g_o, float *__restrict__ g_d, int B, int ntile)
{
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * (blockDim.x >> 6);
    constexpr int kWsh = 6503 + 64 + 32 * 64;  // Directional_MLP.mlp.0: [bias 64][W^T 48 x 64]; rows 32..47 take the harmonics
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = blob[kWsh + i * 64 + lane];
    for (int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); b < B; b += nw) {
        const bool keep = !ray_valid || ray_valid[b];
        const float r = g_rowsum[(size_t)b * 128 + lane] + g_rowsum[(size_t)b * 128 + 64 + lane];
        float gsh[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = r * w[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            gsh[i] = v;
        }
        float gdn = 0.0f;
        for (int t = lane; t < ntile; t += 64) gdn += g_dnorm[(size_t)b * ntile + t];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gdn += __shfl_xor(gdn, off, 64);
        if (lane != 0) continue;
        if (!keep) {
#pragma unroll
            for (int k = 0; k < 3; ++k) g_o[3 * b + k] = g_d[3 * b + k] = 0.0f;
            continue;
        }
        const float d[3] = { rays_d[3 * b], rays_d[3 * b + 1], rays_d[3 * b + 2] };
        const float dn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), inv = 1.0f / (dn + 1e-8f);
        const float x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        constexpr float C1 = 0.4886025119029199f, C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f,
                        C23 = -1.0925484305920792f, C24 = 0.5462742152960396f, C30 = -0.5900435899266435f, C31 = 2.890611442640554f,
                        C32 = -0.4570457994644658f, C33 = 0.3731763325901154f, C34 = -0.4570457994644658f, C35 = 1.445305721320277f,
                        C36 = -0.5900435899266435f;
        // g_u = J^T g_sh, J = d(harmonics)/d(x, y, z) in render_device.h ray_sh's order
        const float gx = gsh[3] * C1 + gsh[4] * C20 * y - gsh[6] * 2.0f * C22 * x + gsh[7] * C23 * z + gsh[8] * 2.0f * C24 * x +
                         gsh[9] * 6.0f * C30 * xy + gsh[10] * C31 * yz - gsh[11] * 2.0f * C32 * xy - gsh[12] * 6.0f * C33 * xz +
                         gsh[13] * C34 * (4.0f * zz - 3.0f * xx - yy) + gsh[14] * 2.0f * C35 * xz + gsh[15] * C36 * (3.0f * xx - 3.0f * yy);
        const float gy = gsh[1] * C1 + gsh[4] * C20 * x + gsh[5] * C21 * z - gsh[6] * 2.0f * C22 * y - gsh[8] * 2.0f * C24 * y +
                         gsh[9] * C30 * (3.0f * xx - 3.0f * yy) + gsh[10] * C31 * xz + gsh[11] * C32 * (4.0f * zz - xx - 3.0f * yy) -
                         gsh[12] * 6.0f * C33 * yz - gsh[13] * 2.0f * C34 * xy - gsh[14] * 2.0f * C35 * yz - gsh[15] * 6.0f * C36 * xy;
        const float gz = gsh[2] * C1 + gsh[5] * C21 * y + gsh[6] * 4.0f * C22 * z + gsh[7] * C23 * x + gsh[10] * C31 * xy +
                         gsh[11] * 8.0f * C32 * yz + gsh[12] * C33 * (6.0f * zz - 3.0f * xx - 3.0f * yy) + gsh[13] * 8.0f * C34 * xz +
                         gsh[14] * C35 * (xx - yy);
        // u = d / (|d| + eps): g_d = g_u inv - (g_u . d) inv^2 d / |d|;  |d|: + g_dn d / |d|
        const float gu[3] = { gx, gy, gz };
        const float dot = gu[0] * d[0] + gu[1] * d[1] + gu[2] * d[2];
        const float coef = dn > 0.0f ? (gdn - dot * inv * inv) / dn : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            g_o[3 * b + k] = g_raypos[6 * b + k];
            g_d[3 * b + k] = gu[k] * inv + coef * d[k] + g_raypos[6 * b + 3 + k];
        }
    }
}

}  // namespace

SCANERF_API int scanerf_ray_grad_epilogue(const float *rays_d, const float *mlp_blob, const float *g_raypos, const float *g_dnorm,
                                          const float *g_rowsum, const uint8_t *ray_valid, float *g_o, float *g_d, int B, int S,
                                          scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 1, "ray_grad_epilogue: B=%d S=%d", B, S);
    if (B == 0) return 0;
    SCANERF_REQUIRE(rays_d && mlp_blob && g_raypos && g_dnorm && g_rowsum && g_o && g_d, "ray_grad_epilogue: null pointer");
    const int blocks = (B + 3) / 4 < 1024 ? (B + 3) / 4 : 1024;
    hipLaunchKernelGGL(k_ray_grad_epilogue, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rays_d, mlp_blob, g_raypos, g_dnorm,
                       g_rowsum, ray_valid, g_o, g_d, B, (S + 31) / 32);
    return check_launch("ray_grad_epilogue");
}

SCANERF_API int scanerf_photometric_loss_scratch_floats(void) { return kBlocks * 6; }

// Loss of the complete per-tile render and its gradients w.r.t. the two branches' per-ray outputs, two launches:
// pred = fg.rgb + fg.T_left * bg.rgb (tile.py:666-690), MSE over the rays valid in either branch (criterions.py:121-138,142-144) + reg * the two branches'
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
