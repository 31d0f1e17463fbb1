/*
 * scanerf_oracle.c -- CPU restatement (plain C, fp32) of the ScaNeRF per-tile
 * volume-rendering hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * Pinning status
 *   - functions restating the reference's CUDA sources (.cu/.h) cannot be
 *     checked against a build of the reference here (CUDA only, no nvcc, the
 *     reference ships no tests or golden vectors): PARITY UNPINNED for those;
 *     they are pinned only by known-answer tests derived from the source text
 *     (tests/test_oracle_kats.py).
 *   - the Python half of the path (MLP, compositing, contraction, consensus)
 *     is restated in oracle/oracle.py and in orc_decoder_inference below, and
 *     IS pinned by golden vectors captured from the importable reference
 *     Python (tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off makes every a*b+c two IEEE roundings, the same arithmetic
 * the HIP sampler is compiled with, so z_vals/dists compare bit for bit.
 *
 * All citations are file:line under /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ helpers */

/* cuda/include/cutil_math.h:924-926 */
static inline float safe_divide(float a, float b) { return b != 0.0f ? a / b : 100000000.0f; }
/* cuda/include/cutil_math.h:913-916 */
static inline int signf_i(float a) { return a >= 0.0f ? 1 : -1; }

typedef struct { float x, y; } f2;

/* cuda/include/cuda_utils.h:564-613 RayAABBIntersection */
static f2 ray_aabb(const float o[3], const float d[3], const float c[3], const float h[3])
{
    float f_low = 0.0f, f_high = 100000.0f;
    f2 miss = { -1.0f, -1.0f };
    for (int k = 0; k < 3; ++k) {
        float inv = safe_divide(1.0f, d[k]);
        float lo = (c[k] - h[k] - o[k]) * inv;
        float hi = (c[k] + h[k] - o[k]) * inv;
        if (hi < lo) { float t = lo; lo = hi; hi = t; }
        if (hi < f_low) return miss;
        if (lo > f_high) return miss;
        f_low = lo > f_low ? lo : f_low;
        f_high = hi < f_high ? hi : f_high;
        if (f_low > f_high) return miss;
    }
    f2 r = { f_low, f_high };
    return r;
}

/* ------------------------------------------------------------ a1/a2: rays */

/* cuda/compute_ray_kernel.cu:18-43, cuda/include/cuda_utils.h:143-155 */
ORC_API void orc_compute_ray_forward(const int32_t *locs, const float *Ks, const float *C2Ws,
                                     float *rays_o, float *rays_d, int B)
{
    for (int i = 0; i < B; ++i) {
        int v = locs[3 * i], px = locs[3 * i + 1], py = locs[3 * i + 2];
        const float *K = Ks + 9 * v, *M = C2Ws + 12 * v;
        float x = (1.0f * px + 0.5f - K[2]) / K[0];
        float y = (1.0f * py + 0.5f - K[5]) / K[4];
        rays_d[3 * i + 0] = M[0] * x + M[1] * y + M[2];
        rays_d[3 * i + 1] = M[4] * x + M[5] * y + M[6];
        rays_d[3 * i + 2] = M[8] * x + M[9] * y + M[10];
        rays_o[3 * i + 0] = M[3];
        rays_o[3 * i + 1] = M[7];
        rays_o[3 * i + 2] = M[11];
    }
}

/* cuda/compute_ray_kernel.cu:46-92.  The reference reads grad_rays_*[view_idx]
 * (lines 71-72) where the ray index is meant; ref_bug=1 reproduces that read,
 * ref_bug=0 is the mathematically correct adjoint of orc_compute_ray_forward
 * (the one the product implements; divergence documented in DESIGN.md). */
ORC_API void orc_compute_ray_backward(const float *g_o, const float *g_d, const float *Ks,
                                      const int32_t *locs, float *grad_C2Ws, int B, int ref_bug)
{
    for (int i = 0; i < B; ++i) {
        int v = locs[3 * i], px = locs[3 * i + 1], py = locs[3 * i + 2];
        const float *K = Ks + 9 * v;
        float x = (1.0f * px + 0.5f - K[2]) / K[0];
        float y = (1.0f * py + 0.5f - K[5]) / K[4];
        int src = ref_bug ? v : i;
        const float *go = g_o + 3 * src, *gd = g_d + 3 * src;
        float *g = grad_C2Ws + 12 * v;
        g[3] += go[0]; g[7] += go[1]; g[11] += go[2];
        for (int r = 0; r < 3; ++r) {
            g[4 * r + 0] += gd[r] * x;
            g[4 * r + 1] += gd[r] * y;
            g[4 * r + 2] += gd[r];
        }
    }
}

/* ------------------------------------------------------------------ a3: aabb */

/* cuda/helper_kernel.cu:108-148 (K==1) and :153-197 (v2, bounds [B,K,2]) */
ORC_API void orc_ray_aabb_intersection(const float *rays_o, const float *rays_d,
                                       const float *center, const float *size,
                                       float *bounds, int B, int K)
{
    for (int i = 0; i < B; ++i)
        for (int k = 0; k < K; ++k) {
            float h[3] = { size[3 * k] / 2.0f, size[3 * k + 1] / 2.0f, size[3 * k + 2] / 2.0f };
            f2 r = ray_aabb(rays_o + 3 * i, rays_d + 3 * i, center + 3 * k, h);
            bounds[2 * (i * K + k)] = r.x;
            bounds[2 * (i * K + k) + 1] = r.y;
        }
}

/* ------------------------------------------------- a4: occupancy-grid sampler */

/* cuda/include/dda.h:206-268 DDASatateScene_v2 */
typedef struct {
    int tstep[3], tile[3], mask[3], side[3];
    float tMax[3], tDelta[3], tsize[3];
    float tx, ty;
} dda_t;

static void dda_init(dda_t *s, const float origin_in[3], const float dir[3], f2 t_start,
                     const int side[3], const float tsize[3])
{
    float origin[3];
    for (int k = 0; k < 3; ++k) {
        s->side[k] = side[k];
        s->tsize[k] = tsize[k];
        origin[k] = origin_in[k] + t_start.x * dir[k];
        int c = (int)(origin[k] / tsize[k]);
        if (c < 0) c = 0;
        if (c > side[k] - 1) c = side[k] - 1;
        s->tile[k] = c;
        s->tstep[k] = signf_i(dir[k]);
    }
    s->tx = t_start.x;
    s->ty = t_start.y;
    for (int k = 0; k < 3; ++k) {
        float nb = (float)(s->tile[k] + s->tstep[k]) * tsize[k];
        if (s->tstep[k] < 0) nb += tsize[k];
        s->tMax[k] = fmaxf(safe_divide(nb - origin[k], dir[k]), 0.0f) + s->tx;
        s->tDelta[k] = fabsf(safe_divide(tsize[k], dir[k]));
    }
}
static inline void dda_next(dda_t *s)
{
    s->mask[0] = (s->tMax[0] < s->tMax[1]) & (s->tMax[0] <= s->tMax[2]);
    s->mask[1] = (s->tMax[1] < s->tMax[2]) & (s->tMax[1] <= s->tMax[0]);
    s->mask[2] = !(s->mask[0] | s->mask[1]);
    s->ty = s->mask[0] ? s->tMax[0] : (s->mask[1] ? s->tMax[1] : s->tMax[2]);
}
static inline void dda_step(dda_t *s)
{
    s->tx = s->ty;
    for (int k = 0; k < 3; ++k) {
        s->tMax[k] += (float)s->mask[k] * s->tDelta[k];
        s->tile[k] += s->mask[k] * s->tstep[k];
    }
}
static inline int dda_terminate(const dda_t *s)
{
    return s->tile[0] < 0 || s->tile[1] < 0 || s->tile[2] < 0 ||
           s->tile[0] >= s->side[0] || s->tile[1] >= s->side[1] || s->tile[2] >= s->side[2] ||
           (s->tMax[0] <= 0 && s->tMax[1] <= 0 && s->tMax[2] <= 0);
}

/* cuda/helper_kernel.cu:540-615 sample_points_sparse_single_ray,
 * cuda/include/cuda_utils.h:101-113 uniform_sample_bound_v3 */
static void sample_sparse_ray(const float o[3], const float d[3], int S, float *z, float *dist,
                              const float corner[3], const float size[3], const uint8_t *occ,
                              const int l2d[3])
{
    float c[3], h[3], og[3], tsize[3];
    int side[3];
    for (int k = 0; k < 3; ++k) {
        h[k] = size[k] / 2.0f;
        c[k] = corner[k] + h[k];
        side[k] = 1 << l2d[k];
        tsize[k] = size[k] / (float)side[k];
        og[k] = o[k] - corner[k];
    }
    f2 bound = ray_aabb(o, d, c, h);
    if (bound.x == -1.0f) return;

    dda_t s;
    dda_init(&s, og, d, bound, side, tsize);
    float total = 0.0f;
    int count = 0;
    while (!dda_terminate(&s)) {
        dda_next(&s);
        uint32_t n = ((uint32_t)s.tile[0] << (l2d[1] + l2d[2])) | ((uint32_t)s.tile[1] << l2d[2]) |
                     (uint32_t)s.tile[2];
        if (occ[n]) {
            float len = s.ty - s.tx;
            if (len > 0) { total += len; count++; }
        }
        dda_step(&s);
    }
    if (count == 0) return;

    dda_init(&s, og, d, bound, side, tsize);
    int left = S, seg = 0;
    while (!dda_terminate(&s)) {
        dda_next(&s);
        uint32_t n = ((uint32_t)s.tile[0] << (l2d[1] + l2d[2])) | ((uint32_t)s.tile[1] << l2d[2]) |
                     (uint32_t)s.tile[2];
        if (occ[n]) {
            float len = s.ty - s.tx;
            if (len > 0) {
                int num = (int)((float)S * len / total);
                if (num < 1) num = 1;
                if (num > left) num = left;
                if (seg == count - 1) num = left;
                float interval = (s.ty - s.tx) / (float)num;
                float *zz = z + S - left, *dd = dist + S - left;
                for (int i = 0; i < num; ++i) {
                    zz[i] = s.tx + (float)i * interval;
                    dd[i] = interval;
                }
                left -= num;
                seg++;
            }
        }
        dda_step(&s);
    }
}

/* cuda/helper_kernel.cu:618-671.  z_vals/dists are pre-filled by the caller (-1). */
ORC_API void orc_sample_points_grid(const float *rays_o, const float *rays_d, float *z_vals,
                                    float *dists, const float *corner, const float *size,
                                    const uint8_t *occ, const int32_t *log2dim, int B, int S)
{
    int l2d[3] = { log2dim[0], log2dim[1], log2dim[2] };
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < B; ++i)
        sample_sparse_ray(rays_o + 3 * i, rays_d + 3 * i, S, z_vals + (size_t)i * S,
                          dists + (size_t)i * S, corner, size, occ, l2d);
}

/* --------------------------------------------------------- a5: other samplers */

/* cuda/sample_kernel.cu:71-100; cuda_utils.h:61-87.  Returns the number of rays
 * that miss the box (the reference asserts on those); their rows are untouched. */
ORC_API int orc_sample_insideout_block(const float *rays_o, const float *rays_d, int S, int S_bg,
                                       const float *center, const float *size, float far_,
                                       float *z_vals, float *z_vals_bg, int B)
{
    int missed = 0;
    float h[3] = { size[0] / 2.0f, size[1] / 2.0f, size[2] / 2.0f };
    for (int i = 0; i < B; ++i) {
        f2 b = ray_aabb(rays_o + 3 * i, rays_d + 3 * i, center, h);
        if (b.x == -1.0f || b.y == -1.0f) { missed++; continue; }
        float interval = (b.y - b.x) / (float)(S - 1);
        for (int k = 0; k < S; ++k) z_vals[(size_t)i * S + k] = b.x + (float)k * interval;
        float inv_near = 1.0f / b.y, inv_far = 1.0f / far_;
        float inv_bound = inv_far - inv_near;
        float step = 1.0f / (float)(S_bg - 1);
        for (int k = 0; k < S_bg; ++k)
            z_vals_bg[(size_t)i * S_bg + k] = 1.0f / (step * (float)k * inv_bound + inv_near);
    }
    return missed;
}

/* cuda/sample_kernel.cu:18-44 */
ORC_API void orc_background_sampling(const float *starts, const float *bg_depth, float *z_vals,
                                     int S, float sample_range, int B)
{
    for (int i = 0; i < B; ++i) {
        float near_ = fmaxf(starts[i] + 0.00001f, bg_depth[i] - sample_range * 0.5f);
        float far_ = near_ + sample_range;
        float interval = (far_ - near_) / (float)(S - 1);
        for (int k = 0; k < S; ++k) z_vals[(size_t)i * S + k] = near_ + (float)k * interval;
    }
}

/* ------------------------------------------------------ a6/a7: hash encoder */

/* hashgrid/src/hashgrid_bg_kernel.cu:14-24 */
ORC_API uint32_t orc_hash(int x, int y, int z, int hashmap_size)
{
    uint32_t r = 0;
    r ^= (uint32_t)x * 1u;
    r ^= (uint32_t)y * 2654435761u;
    r ^= (uint32_t)z * 805459861u;
    return (uint32_t)(hashmap_size - 1) & r;
}

/* hashgrid_bg_kernel.cu:27-38 (corner order 000,001,...,111; z fastest) */
static void linear_weight(float w[8], const float t[3])
{
    w[0] = (1 - t[0]) * (1 - t[1]) * (1 - t[2]);
    w[1] = (1 - t[0]) * (1 - t[1]) * t[2];
    w[2] = (1 - t[0]) * t[1] * (1 - t[2]);
    w[3] = (1 - t[0]) * t[1] * t[2];
    w[4] = t[0] * (1 - t[1]) * (1 - t[2]);
    w[5] = t[0] * (1 - t[1]) * t[2];
    w[6] = t[0] * t[1] * (1 - t[2]);
    w[7] = t[0] * t[1] * t[2];
}
/* hashgrid_bg_kernel.cu:40-77 */
static void dweights(float dx[8], float dy[8], float dz[8], const float t[3])
{
    float x = t[0], y = t[1], z = t[2];
    dx[0] = (-1.0f) * (1 - y) * (1 - z); dx[1] = (-1.0f) * (1 - y) * z;
    dx[2] = (-1.0f) * y * (1 - z);       dx[3] = (-1.0f) * y * z;
    dx[4] = (1 - y) * (1 - z);           dx[5] = (1 - y) * z;
    dx[6] = y * (1 - z);                 dx[7] = y * z;
    dy[0] = (1 - x) * (-1.0f) * (1 - z); dy[1] = (1 - x) * (-1.0f) * z;
    dy[2] = (1 - x) * (1 - z);           dy[3] = (1 - x) * z;
    dy[4] = x * (-1.0f) * (1 - z);       dy[5] = x * (-1.0f) * z;
    dy[6] = x * (1 - z);                 dy[7] = x * z;
    dz[0] = (1 - x) * (1 - y) * (-1.0f); dz[1] = (1 - x) * (1 - y);
    dz[2] = (1 - x) * y * (-1.0f);       dz[3] = (1 - x) * y;
    dz[4] = x * (1 - y) * (-1.0f);       dz[5] = x * (1 - y);
    dz[6] = x * y * (-1.0f);             dz[7] = x * y;
}
static void corner_indices(uint32_t idx[8], const int b[3], int T)
{
    for (int c = 0; c < 8; ++c)
        idx[c] = orc_hash(b[0] + ((c >> 2) & 1), b[1] + ((c >> 1) & 1), b[2] + (c & 1), T);
}

/* Cell + offsets for the contracted-space ("bg") variant: hashgrid_bg_kernel.cu:124-130.
 * Returns d(offset)/d(point) per axis in scale[] (:182). */
static void locate_bg(const float p[3], const int32_t res[3], int b[3], float t[3], float scale[3])
{
    for (int k = 0; k < 3; ++k) {
        float p01 = (p[k] + 2.0f) / 4.0f;
        float v = p01 * (float)(res[k] - 1);
        b[k] = (int)v;
        t[k] = v - (float)b[k];
        scale[k] = (float)(res[k] - 1) / 4.0f;
    }
}
/* World-space box variant: hashgrid/src/hashgrid_kernel.cu:127-143, :237-239 */
static void locate_box(const float p_in[3], const int32_t res[3], const float corner[3],
                       const float size[3], int b[3], float t[3], float scale[3])
{
    for (int k = 0; k < 3; ++k) {
        float p = fmaxf(corner[k], fminf(p_in[k], corner[k] + size[k]));
        float g = size[k] / (float)(res[k] - 1);
        b[k] = (int)((p - corner[k]) / g);
        float vmin = (float)b[k] * g + corner[k];
        t[k] = (p - vmin) / g;
        scale[k] = 1.0f / g;
    }
}

static void embed_fwd(const float *points, float *out, const float *features,
                      const int32_t *res, int N, int L, int T, const float *corner, const float *size)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i)
        for (int l = 0; l < L; ++l) {
            int b[3]; float t[3], sc[3], w[8]; uint32_t idx[8];
            if (corner) locate_box(points + 3 * i, res + 3 * l, corner, size, b, t, sc);
            else locate_bg(points + 3 * i, res + 3 * l, b, t, sc);
            linear_weight(w, t);
            corner_indices(idx, b, T);
            const float *lf = features + (size_t)l * T * 2;
            float ax = 0.0f, ay = 0.0f;
            for (int c = 0; c < 8; ++c) {
                ax = ax + w[c] * lf[2 * idx[c]];
                ay = ay + w[c] * lf[2 * idx[c] + 1];
            }
            out[((size_t)i * L + l) * 2] = ax;
            out[((size_t)i * L + l) * 2 + 1] = ay;
        }
}

static void embed_bwd(const float *points, const float *grad_in, float *grad_points,
                      float *grad_features, const float *features, const int32_t *res,
                      int N, int L, int T, const float *corner, const float *size)
{
    /* sequential accumulation order (the reference uses atomics: order unspecified) */
    for (int l = 0; l < L; ++l)
        for (int i = 0; i < N; ++i) {
            int b[3]; float t[3], sc[3], w[8], dx[8], dy[8], dz[8]; uint32_t idx[8];
            if (corner) locate_box(points + 3 * i, res + 3 * l, corner, size, b, t, sc);
            else locate_bg(points + 3 * i, res + 3 * l, b, t, sc);
            linear_weight(w, t);
            corner_indices(idx, b, T);
            dweights(dx, dy, dz, t);
            const float *lf = features + (size_t)l * T * 2;
            float *lg = grad_features + (size_t)l * T * 2;
            float gx = grad_in[((size_t)i * L + l) * 2], gy = grad_in[((size_t)i * L + l) * 2 + 1];
            float ox[2] = { 0, 0 }, oy[2] = { 0, 0 }, oz[2] = { 0, 0 };
            for (int c = 0; c < 8; ++c) {
                lg[2 * idx[c]] += w[c] * gx;
                lg[2 * idx[c] + 1] += w[c] * gy;
                float fx = lf[2 * idx[c]], fy = lf[2 * idx[c] + 1];
                ox[0] = ox[0] + fx * dx[c]; ox[1] = ox[1] + fy * dx[c];
                oy[0] = oy[0] + fx * dy[c]; oy[1] = oy[1] + fy * dy[c];
                oz[0] = oz[0] + fx * dz[c]; oz[1] = oz[1] + fy * dz[c];
            }
            grad_points[3 * i + 0] += sc[0] * (gx * ox[0] + gy * ox[1]);
            grad_points[3 * i + 1] += sc[1] * (gx * oy[0] + gy * oy[1]);
            grad_points[3 * i + 2] += sc[2] * (gx * oz[0] + gy * oz[1]);
        }
}

/* hashgrid/src/hashgrid_bg_kernel.cu:107-150, :229-249 */
ORC_API void orc_embedding_bg_forward(const float *points, float *out, const float *features,
                                      const int32_t *res, int N, int L, int T)
{ embed_fwd(points, out, features, res, N, L, T, NULL, NULL); }
/* hashgrid/src/hashgrid_bg_kernel.cu:152-226, :251-275 */
ORC_API void orc_embedding_bg_backward(const float *points, const float *grad_in, float *grad_points,
                                       float *grad_features, const float *features,
                                       const int32_t *res, int N, int L, int T)
{ embed_bwd(points, grad_in, grad_points, grad_features, features, res, N, L, T, NULL, NULL); }
/* hashgrid/src/hashgrid_kernel.cu:106-158, :246-270 */
ORC_API void orc_embedding_forward(const float *points, float *out, const float *features,
                                   const float *corner, const float *size, const int32_t *res,
                                   int N, int L, int T)
{ embed_fwd(points, out, features, res, N, L, T, corner, size); }
/* hashgrid/src/hashgrid_kernel.cu:160-243, :272-300 */
ORC_API void orc_embedding_backward(const float *points, const float *grad_in, float *grad_points,
                                    float *grad_features, const float *features, const float *corner,
                                    const float *size, const int32_t *res, int N, int L, int T)
{ embed_bwd(points, grad_in, grad_points, grad_features, features, res, N, L, T, corner, size); }

/* ----------------------------------------------------------- a14: sparse Adam */

/* IEEE binary16 <-> binary32, round to nearest even (what __float2half does) */
static uint16_t f2h(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t e = (x >> 23) & 0xffu, m = x & 0x7fffffu;
    if (e == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u | (m >> 13) : 0));
    int32_t ne = (int32_t)e - 127 + 15;
    if (ne >= 31) return (uint16_t)(sign | 0x7c00u);
    if (ne <= 0) {
        if (ne < -10) return (uint16_t)sign;
        m |= 0x800000u;
        uint32_t shift = (uint32_t)(14 - ne);
        uint32_t hm = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1))) hm++;
        return (uint16_t)(sign | hm);
    }
    uint32_t h = sign | ((uint32_t)ne << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)h;
}
static float h2f(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; sh++; }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e - 15 + 127) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4);
    return f;
}
ORC_API uint16_t orc_float2half(float f) { return f2h(f); }
ORC_API float orc_half2float(uint16_t h) { return h2f(h); }

/* cuda/adam_kernel.cu:24-69 (kernel), :72-94 (host: uses step+1).  `step` here is
 * the value the caller passes to adam_step_cuda, i.e. the PREVIOUS step count.
 * Index = row*8 + dim with dim < param_dim (the 8 is hard-coded at :42). */
ORC_API void orc_adam_step(float *params, const float *grad, float *m, float *v, float lr,
                           float beta1, float beta2, float eps, int step, int64_t K, int param_dim)
{
    float t = (float)(step + 1);
    /* the reference evaluates powf per element (adam_kernel.cu:53-54); same value for all, hoisted */
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    for (int64_t r = 0; r < K; ++r)
        for (int dcol = 0; dcol < param_dim; ++dcol) {
            int64_t i = r * 8 + dcol;
            float g = grad[i];
            if (g == 0.0f) continue;
            float mi = beta1 * m[i] + (1.0f - beta1) * g;
            float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
            float denom = sqrtf(vi / bc2) + eps;
            float step_size = lr / bc1;
            params[i] = params[i] - step_size * mi / denom;
            m[i] = mi;
            v[i] = vi;
        }
}

/* cuda/adam_kernel.cu:98-144, :147-169 (LOSS_SCALE 128, moments stored as half) */
ORC_API void orc_adam_step_fp16(float *params, const float *grad, uint16_t *m, uint16_t *v, float lr,
                                float beta1, float beta2, float eps, int step, int64_t K, int param_dim)
{
    const float LS = 128.0f;
    float t = (float)(step + 1);
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    for (int64_t r = 0; r < K; ++r)
        for (int dcol = 0; dcol < param_dim; ++dcol) {
            int64_t i = r * 8 + dcol;
            float g = grad[i] * LS;
            if (g == 0.0f) continue;
            float mi = beta1 * h2f(m[i]) + (1.0f - beta1) * g;
            float vi = beta2 * h2f(v[i]) + (1.0f - beta2) * g * g;
            float denom = sqrtf(vi / (bc2 * LS * LS)) + eps;
            float step_size = lr / bc1;
            params[i] = params[i] - step_size * mi / (denom * LS);
            m[i] = f2h(mi);
            v[i] = f2h(vi);
        }
}

/* ------------------------------------------------- a15: blob decoder (render) */

/* hashgrid/include/decoder.h:84-117 */
static void sh_deg3(const float d[3], float *o)
{
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f,
                C23 = -1.0925484305920792f, C24 = 0.5462742152960396f;
    const float C30 = -0.5900435899266435f, C31 = 2.890611442640554f, C32 = -0.4570457994644658f,
                C33 = 0.3731763325901154f, C34 = -0.4570457994644658f, C35 = 1.445305721320277f,
                C36 = -0.5900435899266435f;
    float x = d[0], y = d[1], z = d[2];
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    o[0] = C0; o[1] = C1 * y; o[2] = C1 * z; o[3] = C1 * x;
    o[4] = C20 * xy; o[5] = C21 * yz; o[6] = (float)(C22 * (2.0 * zz - xx - yy)); /* 2.0 is double at :104 */
    o[7] = C23 * xz; o[8] = C24 * (xx - yy);
    o[9] = C30 * y * (3 * xx - yy); o[10] = C31 * xy * z; o[11] = C32 * y * (4 * zz - xx - yy);
    o[12] = C33 * z * (2 * zz - 3 * xx - 3 * yy); o[13] = C34 * x * (4 * zz - xx - yy);
    o[14] = C35 * z * (xx - yy); o[15] = C36 * x * (xx - 3 * yy);
}

/* decoder.h:149-167 Linear: blob = [bias(out), W^T (in-major, out fastest)] */
static const float *linear(const float *p, const float *in, int n_in, float *out, int n_out)
{
    for (int j = 0; j < n_out; ++j) out[j] = *p++;
    for (int i = 0; i < n_in; ++i)
        for (int j = 0; j < n_out; ++j) out[j] += in[i] * *p++;
    return p;
}
static inline float gauss(float x) { return expf(x * x / -0.02f); }   /* decoder.h:126 */
static inline float softplus_raw(float x) { return logf(1.0f + expf(x)); } /* decoder.h:131-135 */
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-1.0f * x)); }

#define ORC_PARAMSIZE 13994

/* decoder.h:169-218 Decoder::inference, one sample.  specular is returned tinted (tint * sigmoid). */
static void decoder_one(const float *params, const float *feat, const float *d, float *sigma, float *diffuse,
                        float *specular, float *tint_out)
{
    float h0[64], h1[64], in48[48], d0[64], d1[64], o3[3], o1[1], tint[3];
    const float *p = params;
    p = linear(p, feat, 32, h0, 64);
    for (int j = 0; j < 64; ++j) h0[j] = gauss(h0[j]);
    p = linear(p, h0, 64, h1, 64);
    p = linear(p, h1, 32, o1, 1);
    *sigma = softplus_raw(o1[0]);
    p = linear(p, h1, 32, o3, 3);
    for (int k = 0; k < 3; ++k) diffuse[k] = sigmoidf_(o3[k]);
    p = linear(p, h1, 32, o3, 3);
    for (int k = 0; k < 3; ++k) tint[k] = sigmoidf_(o3[k]);
    float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); /* normalize(): rsqrt, no eps */
    float dn[3] = { d[0] * inv, d[1] * inv, d[2] * inv };
    for (int j = 0; j < 32; ++j) in48[j] = h1[32 + j];
    sh_deg3(dn, in48 + 32);
    p = linear(p, in48, 48, d0, 64);
    for (int j = 0; j < 64; ++j) d0[j] = gauss(d0[j]);
    p = linear(p, d0, 64, d1, 64);
    for (int j = 0; j < 64; ++j) d1[j] = gauss(d1[j]);
    p = linear(p, d1, 64, o3, 3);
    for (int k = 0; k < 3; ++k) {
        specular[k] = tint[k] * sigmoidf_(o3[k]);
        if (tint_out) tint_out[k] = tint[k];
    }
}

/* feat [n,32], dirs [n,3] (un-normalised), params [13994] ->
 * sigma[n], diffuse[n,3], specular[n,3] (= tint*sigmoid(...)), tint[n,3] (extra output) */
ORC_API void orc_decoder_inference(const float *params, const float *feat, const float *dirs,
                                   float *sigma, float *diffuse, float *specular, float *tint_out, int n)
{
#pragma omp parallel for schedule(static)
    for (int s = 0; s < n; ++s)
        decoder_one(params, feat + 32 * (size_t)s, dirs + 3 * (size_t)s, sigma + s, diffuse + 3 * s, specular + 3 * s,
                    tint_out ? tint_out + 3 * s : NULL);
}

ORC_API int orc_param_size(void) { return ORC_PARAMSIZE; }


/* ===================================================================================
 * a16: render-time kernels, hashgrid/src/rendering_kernel.cu (multi-tile novel-view render)
 * =================================================================================== */
#define ORC_MAX_PTS_BLOCKS 4
#define ORC_INF_INTERSECTION 10000000.0f

/* rendering_kernel.cu:126-174 */
ORC_API void orc_ray_block_intersection(const float *rays_o, const float *rays_d, const float *corners,
                                        const float *sizes, float *inter, int B, int nb)
{
    for (int i = 0; i < B; ++i)
        for (int b = 0; b < nb; ++b) {
            float h[3], c[3];
            for (int k = 0; k < 3; ++k) { h[k] = sizes[3 * b + k] / 2.0f; c[k] = corners[3 * b + k] + h[k]; }
            f2 r = ray_aabb(rays_o + 3 * i, rays_d + 3 * i, c, h);
            if (r.x == -1.0f) { r.x = ORC_INF_INTERSECTION; r.y = ORC_INF_INTERSECTION; }
            inter[2 * ((size_t)i * nb + b)] = r.x;
            inter[2 * ((size_t)i * nb + b) + 1] = r.y;
        }
}

static inline uint32_t cell_offset(const int t[3], const int l2d[3])
{
    return ((uint32_t)t[0] << (l2d[1] + l2d[2])) | ((uint32_t)t[1] << l2d[2]) | (uint32_t)t[2];
}

/* rendering_kernel.cu:179-344: one tracing step of samplepoints_kernel for one ray */
ORC_API void orc_render_sample_points(const float *rays_o, const float *rays_d, const float *corners,
                                      const float *sizes, const uint8_t *occ, const int64_t *grid_starts,
                                      const int32_t *log2dim, int S, int nb, const int32_t *tracing_blocks,
                                      const float *inter, int32_t *tracing_idx, float *z_start, float *z_vals,
                                      float *dists, int B)
{
    for (int i = 0; i < B; ++i) {
        const float *o = rays_o + 3 * i, *d = rays_d + 3 * i;
        const int32_t *tb = tracing_blocks + (size_t)i * nb;
        const float *ci = inter + 2 * (size_t)i * nb;
        float *cz = z_vals + (size_t)i * S, *cd = dists + (size_t)i * S;
        int step = tracing_idx[i];
        float tsx = z_start[i];
        while (step < nb) {
            int b = tb[step];
            float bx = ci[2 * b], by = ci[2 * b + 1];
            if (bx == ORC_INF_INTERSECTION) break;
            if (tsx >= by) { step++; continue; }
            if (step == 0) tsx = bx;
            int l2d[3] = { log2dim[3 * b], log2dim[3 * b + 1], log2dim[3 * b + 2] };
            int side[3]; float tsize[3], og[3];
            for (int k = 0; k < 3; ++k) {
                side[k] = 1 << l2d[k];
                tsize[k] = sizes[3 * b + k] / (float)side[k];
                og[k] = o[k] - corners[3 * b + k];
            }
            const uint8_t *g = occ + grid_starts[b];
            f2 ts = { tsx, 0.0f };
            dda_t s;
            dda_init(&s, og, d, ts, side, tsize);
            int num_seg = 0; float total = 0.0f;
            while (!dda_terminate(&s)) {
                dda_next(&s);
                if (g[cell_offset(s.tile, l2d)]) {
                    float len = s.ty - s.tx;
                    if (len > 0) { total += len; num_seg++; }
                }
                dda_step(&s);
            }
            if (num_seg == 0) { tsx = by; step++; continue; }
            int num = 0, count = 0;
            dda_init(&s, og, d, ts, side, tsize);
            while (!dda_terminate(&s)) {
                dda_next(&s);
                if (g[cell_offset(s.tile, l2d)]) {
                    float len = s.ty - s.tx;
                    if (len > 0) {
                        int n = (int)(len / total * (float)S);
                        if (n < 1) n = 1;
                        if (n > S - num) n = S - num;
                        if (count == num_seg - 1) n = S - num;
                        if (n > 0) {
                            float interval = (s.ty - s.tx) / (float)n;
                            for (int k = 0; k < n; ++k) { cz[num + k] = s.tx + (float)k * interval; cd[num + k] = interval; }
                        }
                        num += n;
                        count++;
                    }
                }
                dda_step(&s);
            }
            tsx = by;
            step++;
            break;
        }
        tracing_idx[i] = step;
        z_start[i] = tsx;
    }
}

/* rendering_kernel.cu:391-449 (the reference can overrun the 4 slots when >4 tiles overlap; clamped here) */
ORC_API void orc_prepare_points(const float *z_vals, const uint8_t *running, int16_t *block_idxs, const float *inter,
                                int S, int nb, int B)
{
    for (int i = 0; i < B; ++i) {
        if (!running[i]) continue;
        for (int s = 0; s < S; ++s) {
            float z = z_vals[(size_t)i * S + s];
            if (z == -1.0f) continue;
            int16_t *bi = block_idxs + ((size_t)i * S + s) * ORC_MAX_PTS_BLOCKS;
            int idx = 0;
            for (int b = 0; b < nb && idx < ORC_MAX_PTS_BLOCKS; ++b) {
                float bx = inter[2 * ((size_t)i * nb + b)], by = inter[2 * ((size_t)i * nb + b) + 1];
                if (z >= bx && z <= by) bi[idx++] = (int16_t)b;
            }
        }
    }
}

/* rendering_kernel.cu:79-114 get_multilevel_features: p01 in [0,1] directly, fp16 tables */
static void multilevel_features_h(const float p01[3], const int32_t *res, const uint16_t *tables, int T, float *feat)
{
    for (int l = 0; l < 16; ++l) {
        int b[3]; float t[3], w[8]; uint32_t idx[8];
        for (int k = 0; k < 3; ++k) {
            float v = p01[k] * (float)(res[3 * l + k] - 1);
            b[k] = (int)v;
            t[k] = v - (float)b[k];
        }
        linear_weight(w, t);
        corner_indices(idx, b, T);
        const uint16_t *lf = tables + (size_t)l * T * 2;
        float ax = 0.0f, ay = 0.0f;
        for (int c = 0; c < 8; ++c) {
            ax += w[c] * h2f(lf[2 * idx[c]]);
            ay += w[c] * h2f(lf[2 * idx[c] + 1]);
        }
        feat[2 * l] = ax;
        feat[2 * l + 1] = ay;
    }
}

/* xz-distance blend weight, rendering_kernel.cu:523-537 / :1335-1351 */
static float xz_weight(float dx, float dz)
{
    if (dx != 0 && dz != 0) return dx * dz;
    if (dx != 0) return dx;
    if (dz != 0) return dz;
    return 0.0f;
}

/* rendering_kernel.cu:467-621 */
ORC_API void orc_pts_inference(const float *rays_o, const float *rays_d, const float *z_vals, const float *dists,
                               const int16_t *block_idxs, const uint16_t *tables, const float *params,
                               const int32_t *res, const uint8_t *occ, const int64_t *grid_starts,
                               const int32_t *log2dim, int T, const float *corners, const float *sizes,
                               float *out_dif, float *out_spec, float *out_alpha, int B, int S)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t t = 0; t < (int64_t)B * S; ++t) {
        int i = (int)(t / S);
        const int16_t *bi = block_idxs + t * ORC_MAX_PTS_BLOCKS;
        float dif[3] = { 0, 0, 0 }, spc[3] = { 0, 0, 0 }, alpha = 0.0f, weight = 0.0f;
        for (int k = 0; k < ORC_MAX_PTS_BLOCKS; ++k) {
            int b = bi[k];
            if (b == -1) break;
            const float *o = rays_o + 3 * i, *d = rays_d + 3 * i;
            float z = z_vals[t], dist = dists[t];
            int l2d[3] = { log2dim[3 * b], log2dim[3 * b + 1], log2dim[3 * b + 2] };
            float pts[3], dis[3];
            int loc[3];
            for (int a = 0; a < 3; ++a) {
                float w_ = o[a] + z * d[a];
                pts[a] = (w_ - corners[3 * b + a]) / sizes[3 * b + a];
                dis[a] = (0.5f - fabsf(pts[a] - 0.5f)) * sizes[3 * b + a];
                int r = 1 << l2d[a];
                int c = (int)(pts[a] * (float)r);
                loc[a] = c < 0 ? 0 : (c > r - 1 ? r - 1 : c);
            }
            float w = xz_weight(dis[0], dis[2]);
            if (occ[grid_starts[b] + cell_offset(loc, l2d)]) {
                float p01[3] = { pts[0] / 2.0f + 0.25f, pts[1] / 2.0f + 0.25f, pts[2] / 2.0f + 0.25f };
                float feat[32], sg, pd[3], ps[3];
                multilevel_features_h(p01, res + (size_t)b * 48, tables + (size_t)b * 16 * T * 2, T, feat);
                decoder_one(params + (size_t)b * ORC_PARAMSIZE, feat, d, &sg, pd, ps, NULL);
                float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                float pa = 1.0f - expf(-1.0f * sg * dist * nrm);
                for (int a = 0; a < 3; ++a) { dif[a] += w * pa * pd[a]; spc[a] += w * pa * ps[a]; }
                alpha += w * pa;
            }
            weight += w;
        }
        if (weight > 0) {
            for (int a = 0; a < 3; ++a) { dif[a] /= weight; spc[a] /= weight; }
            alpha /= weight;
        }
        for (int a = 0; a < 3; ++a) { out_dif[3 * t + a] = dif[a]; out_spec[3 * t + a] = spc[a]; }
        out_alpha[t] = alpha;
    }
}

/* rendering_kernel.cu:624-702 */
ORC_API void orc_accumulate_color(const float *pts_dif, const float *pts_spec, const float *pts_alpha, float *transp,
                                  const float *z_vals, float *dif, float *spec, float *depth, int B, int S)
{
    for (int i = 0; i < B; ++i) {
        float T = transp[i];
        if (T < 0.00001f) continue;
        for (int s = 0; s < S; ++s) {
            size_t t = (size_t)i * S + s;
            for (int a = 0; a < 3; ++a) { dif[3 * i + a] += T * pts_dif[3 * t + a]; spec[3 * i + a] += T * pts_spec[3 * t + a]; }
            depth[i] += T * pts_alpha[t] * z_vals[t];
            T = T * (1 - pts_alpha[t]);
        }
        transp[i] = T;
    }
}

/* rendering_kernel.cu:816-868 */
ORC_API void orc_render_inverse_z_sampling(const float *inter, const int16_t *related, int S, int nb, float range,
                                           float *z_vals, int B)
{
    for (int i = 0; i < B; ++i) {
        int b = related[i];
        if (b == -1) continue;
        float bx = inter[2 * ((size_t)i * nb + b)], by = inter[2 * ((size_t)i * nb + b) + 1];
        if (bx == ORC_INF_INTERSECTION) continue;
        float near_ = by, far_ = near_ + range;
        float inv_near = 1.0f / near_, inv_far = 1.0f / far_, inv_bound = inv_far - inv_near;
        float step = 1.0f / (float)(S - 1);
        for (int k = 0; k < S; ++k) z_vals[(size_t)i * S + k] = 1.0f / (step * (float)k * inv_bound + inv_near);
    }
}

/* rendering_kernel.cu:1012-1171 */
ORC_API void orc_bg_pts_inference_v2(const float *rays_o, const float *rays_d, const float *z_vals,
                                     const uint16_t *tables, const float *params, const float *corners,
                                     const float *sizes, const int32_t *res, const int16_t *bg_idxs, int step,
                                     float *out_dif, float *out_spec, float *out_alpha, int T, int B, int S)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t t = 0; t < (int64_t)B * S; ++t) {
        int i = (int)(t / S), s = (int)(t % S);
        int b = bg_idxs[i * ORC_MAX_PTS_BLOCKS + step];
        if (b == -1) continue;
        const float *o = rays_o + 3 * i, *d = rays_d + 3 * i;
        float z = z_vals[t];
        float sample_step = (s == S - 1) ? 10000000.0f : z_vals[t + 1] - z;
        float pts[3], inf_norm = 0.0f;
        for (int a = 0; a < 3; ++a) {
            float w_ = o[a] + z * d[a];
            pts[a] = 2.0f * (w_ - corners[3 * b + a]) / sizes[3 * b + a] - 1.0f;
        }
        inf_norm = fabsf(pts[0]);
        for (int a = 1; a < 3; ++a) if (fabsf(pts[a]) > inf_norm) inf_norm = fabsf(pts[a]);
        float temp = 2.0f - 1.0f / inf_norm;
        float ratio = temp / inf_norm;
        float p01[3];
        for (int a = 0; a < 3; ++a) { pts[a] *= ratio; p01[a] = (pts[a] + 2.0f) / 4.0f; }
        float feat[32], sg, pd[3], ps[3];
        multilevel_features_h(p01, res + (size_t)b * 48, tables + (size_t)b * 16 * T * 2, T, feat);
        decoder_one(params + (size_t)b * ORC_PARAMSIZE, feat, d, &sg, pd, ps, NULL);
        float pa = 1.0f - expf(-1.0f * sg * sample_step);
        for (int a = 0; a < 3; ++a) { out_dif[3 * t + a] = pa * pd[a]; out_spec[3 * t + a] = pa * ps[a]; }
        out_alpha[t] = pa;
    }
}

/* rendering_kernel.cu:872-1008 (bg_pts_inference, v1; host entry :1176-1208): per sample, over the ray's outgoing blocks until the
 * first -1 (the loop BREAKS there), the v2 per-block inference on the SAME z_vals, blended with the ray's blend weights:
 * diffuse = sum w a c_d / sum w, specular = sum w a (tint c_s) / sum w, alpha = sum w a / sum w (untouched sums where sum w <= 0). */
ORC_API void orc_bg_pts_inference(const float *rays_o, const float *rays_d, const float *z_vals, const uint16_t *tables,
                                  const float *params, const float *corners, const float *sizes, const int32_t *res,
                                  const int16_t *outgoing_bidxs, const float *blend_weights, float *out_dif, float *out_spec,
                                  float *out_alpha, int T, int B, int S)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t t = 0; t < (int64_t)B * S; ++t) {
        int i = (int)(t / S), s = (int)(t % S);
        const float *o = rays_o + 3 * i, *d = rays_d + 3 * i;
        float z = z_vals[t];
        float sample_step = (s == S - 1) ? 10000000.0f : z_vals[t + 1] - z;
        float dif[3] = { 0, 0, 0 }, spec[3] = { 0, 0, 0 }, alpha = 0.0f, weight = 0.0f;
        for (int k = 0; k < ORC_MAX_PTS_BLOCKS; ++k) {
            int b = outgoing_bidxs[i * ORC_MAX_PTS_BLOCKS + k];
            if (b == -1) break;
            float pts[3];
            for (int a = 0; a < 3; ++a) {
                float w_ = o[a] + z * d[a];
                pts[a] = 2.0f * (w_ - corners[3 * b + a]) / sizes[3 * b + a] - 1.0f;
            }
            float w = blend_weights[i * ORC_MAX_PTS_BLOCKS + k];
            float inf_norm = fabsf(pts[0]);
            for (int a = 1; a < 3; ++a) if (fabsf(pts[a]) > inf_norm) inf_norm = fabsf(pts[a]);
            float temp = 2.0f - 1.0f / inf_norm;
            float ratio = temp / inf_norm;
            float p01[3];
            for (int a = 0; a < 3; ++a) { pts[a] *= ratio; p01[a] = (pts[a] + 2.0f) / 4.0f; }
            float feat[32], sg, pd[3], ps[3];
            multilevel_features_h(p01, res + (size_t)b * 48, tables + (size_t)b * 16 * T * 2, T, feat);
            decoder_one(params + (size_t)b * ORC_PARAMSIZE, feat, d, &sg, pd, ps, NULL);
            float pa = 1.0f - expf(-1.0f * sg * sample_step);
            for (int a = 0; a < 3; ++a) { dif[a] = dif[a] + w * pa * pd[a]; spec[a] = spec[a] + w * pa * ps[a]; }
            alpha = alpha + w * pa;
            weight += w;
        }
        if (weight > 0) {
            for (int a = 0; a < 3; ++a) { dif[a] /= weight; spec[a] /= weight; }
            alpha /= weight;
        }
        for (int a = 0; a < 3; ++a) { out_dif[3 * t + a] = dif[a]; out_spec[3 * t + a] = spec[a]; }
        out_alpha[t] = alpha;
    }
}

/* rendering_kernel.cu:1263-1401 */
ORC_API void orc_update_outgoing_bidx(const float *rays_o, const float *rays_d, const float *corners, const float *sizes,
                                      const int32_t *tracing_blocks, const float *inter, int16_t *out_bidx,
                                      float *blend, float ratio, int skip, int nb, int B)
{
    (void)ratio;
    for (int i = 0; i < B; ++i) {
        const int32_t *tb = tracing_blocks + (size_t)i * nb;
        const float *ci = inter + 2 * (size_t)i * nb;
        float far_ = -1.0f;
        int index = 0;
        int16_t outb[ORC_MAX_PTS_BLOCKS] = { -1, -1, -1, -1 };
        for (int k = 0; k < nb; ++k) {
            int b = tb[k];
            float bx = ci[2 * b], by = ci[2 * b + 1];
            if (bx == ORC_INF_INTERSECTION) break;
            if (!skip && (bx > far_ && far_ != -1.0f)) break;
            if (by > far_) {
                far_ = by;
                for (int j = 0; j < ORC_MAX_PTS_BLOCKS; ++j) outb[j] = -1;
                index = 0;
                outb[index++] = (int16_t)b;
            } else if (by == far_) {
                if (index < ORC_MAX_PTS_BLOCKS) outb[index++] = (int16_t)b; /* reference: unchecked */
            }
        }
        if (far_ == -1.0f) continue;
        if (index == 1) {
            blend[i * ORC_MAX_PTS_BLOCKS] = 1.0f;
            out_bidx[i * ORC_MAX_PTS_BLOCKS] = outb[0];
            continue;
        }
        for (int k = 0; k < ORC_MAX_PTS_BLOCKS; ++k) {
            int b = outb[k];
            if (b == -1) break;
            float dis[3];
            for (int a = 0; a < 3; ++a) {
                float pw = rays_o[3 * i + a] + far_ * rays_d[3 * i + a];
                float p = (pw - corners[3 * b + a]) / sizes[3 * b + a];
                p = p < 0.0f ? 0.0f : (p > 1.0f ? 1.0f : p);
                dis[a] = (0.5f - fabsf(p - 0.5f)) * sizes[3 * b + a];
            }
            blend[i * ORC_MAX_PTS_BLOCKS + k] = xz_weight(dis[0], dis[2]);
            out_bidx[i * ORC_MAX_PTS_BLOCKS + k] = (int16_t)b;
        }
    }
}

/* rendering_kernel.cu:1406-1447 */
ORC_API void orc_update_outgoing_bidx_v2(const float *rays_o, const float *corners, const float *sizes,
                                         int16_t *inside, float *blend, int nb, int B)
{
    for (int i = 0; i < B; ++i) {
        int index = 0;
        for (int b = 0; b < nb && index < ORC_MAX_PTS_BLOCKS; ++b) {
            float loc[3], dis[3];
            int in = 1;
            for (int a = 0; a < 3; ++a) {
                loc[a] = (rays_o[3 * i + a] - corners[3 * b + a]) / sizes[3 * b + a];
                if (!(loc[a] >= 0 && loc[a] <= 1)) in = 0;
                dis[a] = (0.5f - fabsf(loc[a] - 0.5f)) * sizes[3 * b + a];
            }
            if (in) {
                inside[i * ORC_MAX_PTS_BLOCKS + index] = (int16_t)b;
                blend[i * ORC_MAX_PTS_BLOCKS + index] = dis[0] * dis[1] * dis[2];
                index++;
            }
        }
    }
}

/* rendering_kernel.cu:1212-1260 */
ORC_API void orc_get_last_block(const int32_t *tracing_blocks, int32_t *bidxs, const float *inter, int nb, int B)
{
    for (int i = 0; i < B; ++i) {
        int idx = -1;
        for (int k = 0; k < nb; ++k) {
            int b = tracing_blocks[(size_t)i * nb + k];
            if (inter[2 * ((size_t)i * nb + b)] == ORC_INF_INTERSECTION) break;
            idx = b;
        }
        bidxs[i] = idx;
    }
}

/* rendering_kernel.cu:705-782 (`tracing` :705-732 + ray_firsthit_block_kernel :735-782): per ray, walk its tiles in
 * tracing order until the first untouched one (near == INF); a tile "hits" when the DDA over [near, far] meets an occupied
 * cell (no length test, unlike the samplers); among hitting tiles the one with the smallest FAR bound wins (`dis > bound.y`,
 * strict: the earlier tile wins ties); a ray that hits nothing keeps the last tile it crosses; `hit` is read back for the
 * -1 test, so it must be pre-filled with -1 by the caller. */
ORC_API void orc_ray_firsthit_block(const float *rays_o, const float *rays_d, const float *corners, const float *sizes,
                                    const uint8_t *occ, const int64_t *grid_starts, const int32_t *log2dim,
                                    const int32_t *tracing_blocks, const float *inter, int16_t *hit, int nb, int B)
{
    for (int i = 0; i < B; ++i) {
        const float *o = rays_o + 3 * i, *d = rays_d + 3 * i;
        float dis = 10000000.0f;
        int last = -1;
        for (int k = 0; k < nb; ++k) {
            const int b = tracing_blocks[(size_t)i * nb + k];
            const f2 bound = { inter[2 * ((size_t)i * nb + b)], inter[2 * ((size_t)i * nb + b) + 1] };
            if (bound.x == ORC_INF_INTERSECTION) break;
            const int l2d[3] = { log2dim[3 * b], log2dim[3 * b + 1], log2dim[3 * b + 2] };
            int side[3]; float tsize[3], og[3];
            for (int a = 0; a < 3; ++a) {
                side[a] = 1 << l2d[a];
                tsize[a] = sizes[3 * b + a] / (float)side[a];
                og[a] = o[a] - corners[3 * b + a];
            }
            const uint8_t *g = occ + grid_starts[b];
            dda_t s;
            dda_init(&s, og, d, bound, side, tsize);
            int found = 0;
            while (!dda_terminate(&s)) {
                dda_next(&s);
                if (g[cell_offset(s.tile, l2d)]) { found = 1; break; }
                dda_step(&s);
            }
            if (found && dis > bound.y) { hit[i] = (int16_t)b; dis = bound.y; }
            last = b;
        }
        if (last != -1 && hit[i] == -1) hit[i] = (int16_t)last;
    }
}

/* rendering_kernel.cu:1479-1564: dilate tile bidx's occupancy into the grids of the tiles it overlaps */
ORC_API void orc_process_occupied_grid(int bidx, int total_grid, const float *corners, const float *sizes,
                                       const uint8_t *occ, const int64_t *grid_starts, const int32_t *log2dim,
                                       uint8_t *tgt, int nb)
{
    const int l0[3] = { log2dim[3 * bidx], log2dim[3 * bidx + 1], log2dim[3 * bidx + 2] };
    const int r0[3] = { 1 << l0[0], 1 << l0[1], 1 << l0[2] };
    float gs[3];
    for (int a = 0; a < 3; ++a) gs[a] = sizes[3 * bidx + a] / (float)r0[a];
    static const float vtx[8][3] = { {0,0,0},{0,0,1},{0,1,0},{1,0,0},{0,1,1},{1,0,1},{1,1,0},{1,1,1} };
    for (int t = 0; t < total_grid; ++t) {
        if (!occ[grid_starts[bidx] + t]) continue;
        int x = t / (r0[1] * r0[2]);
        int y = (t - x * (r0[1] * r0[2])) / r0[2];
        int z = (t - x * (r0[1] * r0[2])) % r0[2];
        int loc[3] = { x, y, z };
        float pts[3];
        for (int a = 0; a < 3; ++a) pts[a] = (float)loc[a] * gs[a] + corners[3 * bidx + a];
        for (int b = 0; b < nb; ++b) {
            if (b == bidx) continue;
            for (int j = 0; j < 8; ++j) {
                float p[3];
                int in = 1;
                for (int a = 0; a < 3; ++a) {
                    p[a] = (pts[a] + vtx[j][a] * gs[a] - corners[3 * b + a]) / sizes[3 * b + a];
                    if (!(p[a] >= 0 && p[a] < 1)) in = 0;
                }
                if (in) {
                    int l2d[3] = { log2dim[3 * b], log2dim[3 * b + 1], log2dim[3 * b + 2] };
                    int ijk[3];
                    for (int a = 0; a < 3; ++a) ijk[a] = (int)(p[a] * (float)(1 << l2d[a]));
                    tgt[grid_starts[b] + cell_offset(ijk, l2d)] = 1;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ voxelize_mesh
 * cuda/include/voxelize.h:12-119 after read_plyFile: vertices [V,3], faces [F,3]; vis / outside byte grids
 * [2^lx,2^ly,2^lz] (caller zero-fills, as hashgrid/__init__.py:71-78 does). */
ORC_API void orc_voxelize_mesh(const float *vertices, const int32_t *faces, int F, const int32_t *log2dim,
                               const float *block_corner, const float *block_size, uint8_t *vis, int init_out,
                               uint8_t *outside)
{
    const int res[3] = { 1 << log2dim[0], 1 << log2dim[1], 1 << log2dim[2] };
    float gs[3], bmax[3];
    for (int c = 0; c < 3; ++c) {
        gs[c] = block_size[c] / (float)res[c];            /* :28 */
        bmax[c] = block_corner[c] + block_size[c];        /* :32 */
    }
    float geo_min[3] = { 100000000.0f, 100000000.0f, 100000000.0f };   /* :43-44 */
    float geo_max[3] = { -1.0f * 100000000.0f, -1.0f * 100000000.0f, -1.0f * 100000000.0f };
    for (int f = 0; f < F; ++f) {
        float mn[3], mx[3];
        int lo[3], hi[3], skip = 0;
        for (int c = 0; c < 3; ++c) {
            const float A = vertices[3 * faces[3 * f] + c], B = vertices[3 * faces[3 * f + 1] + c],
                        C = vertices[3 * faces[3 * f + 2] + c];
            const float mnc = fminf(fminf(A, B), C), mxc = fmaxf(fmaxf(A, B), C);   /* :52-53 */
            const float center = (mnc + mxc) / 2.0f;                               /* :55 */
            const float half = ((mxc - mnc) * 1.5f) / 2.0f;                         /* :56-57 */
            mn[c] = center - half;                                                  /* :58-59 */
            mx[c] = center + half;
            if (mx[c] <= block_corner[c] || mn[c] >= bmax[c]) skip = 1;             /* :61-62 */
        }
        if (skip) continue;
        for (int c = 0; c < 3; ++c) {
            geo_min[c] = fminf(mn[c], geo_min[c]);                                  /* :64-65 */
            geo_max[c] = fmaxf(mx[c], geo_max[c]);
            int l = (int)((mn[c] - block_corner[c]) / gs[c]);                       /* :67-71 */
            int h = (int)((mx[c] - block_corner[c]) / gs[c]);
            lo[c] = l < 0 ? 0 : (l > res[c] - 1 ? res[c] - 1 : l);                  /* :73-74 */
            hi[c] = h < 0 ? 0 : (h > res[c] - 1 ? res[c] - 1 : h);
        }
        for (int x = lo[0]; x <= hi[0]; ++x)
            for (int y = lo[1]; y <= hi[1]; ++y)
                for (int z = lo[2]; z <= hi[2]; ++z)
                    vis[((uint32_t)x << (log2dim[1] + log2dim[2])) | ((uint32_t)y << log2dim[2]) | (uint32_t)z] = 1;  /* :82-83 */
    }
    if (init_out) {                                                                 /* :89-109 */
        for (int x = 0; x < res[0]; ++x)
            for (int y = 0; y < res[1]; ++y)
                for (int z = 0; z < res[2]; ++z) {
                    const int idx[3] = { x, y, z };
                    int out = 0;
                    for (int c = 0; c < 3; ++c) {
                        const float loc = block_corner[c] + (float)idx[c] * gs[c] + gs[c] / 2.0f;
                        if (loc < geo_min[c] || loc > geo_max[c]) out = 1;
                    }
                    if (out) {
                        const uint32_t n = ((uint32_t)x << (log2dim[1] + log2dim[2])) | ((uint32_t)y << log2dim[2]) | (uint32_t)z;
                        vis[n] = 1;
                        outside[n] = 1;
                    }
                }
    }
}
