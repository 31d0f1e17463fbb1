// render_time.hip -- the multi-tile novel-view render kernels (gfx950).
//
// Reference behaviour: hashgrid/src/rendering_kernel.cu (file:line per kernel below), driven by
// RenderingHashGrid.render_rays_base (rendering.py:286-544):
//   ray_block_intersection -> argsort(near) -> per tracing step { sample_points -> prepare_points ->
//   pts_inference -> accumulate_color } -> update_outgoing_bidx -> per blended background
//   { inverse_z_sampling -> bg_pts_inference_v2 -> accumulate_color }.
//
// Built with -ffp-contract=off (the samplers and box tests are bit-exact against the oracle).
//
// pts_inference / bg_pts_inference_v2 are the render hot loop.  The reference runs the whole
// 13 994-MAC decoder serially in one thread per sample, weights from global memory.  Here a wave
// takes 32 consecutive samples and runs the decoder on the matrix cores.  Default: chunk-major
// with the split-f16 decoder image of each tile a chunk touches staged in LDS
// (k_pts_inference_chunks: the rate of the training forward -- table gathers bound it).  Kept for comparison
// (SCANERF_INFER_F32 in sample_major): a single pass on the fp32 matrix pipe that reads the packed image of
// whichever tile the samples reference through L2 and loops over the distinct tiles of a wave
// (1.5e9 samples/s).
#include <hip/hip_fp16.h>

#include <stdlib.h>

#include "dda_device.h"
#include "render_device.h"
#define H3_OPAQUE_ADDR 1  // (render_h3.h: one base register for the decoder image reads; measured clean on this kernel, tools/render_soak.py)
#include "render_h3.h"
#include "render_t16.h"

using namespace scanerf;

namespace {

constexpr int kMaxPtsBlocks = 4;          // MAX_PTS_BLOCKS, rendering_kernel.cu:25
constexpr float kInf = 10000000.0f;       // INF_INTERSECTION, :26

struct Tiles {
    const float *corners, *sizes;         // [nb,3]
    const uint8_t *occ;                   // concatenated bool grids
    const int64_t *grid_starts;           // [nb]
    const int32_t *log2dim;               // [nb,3]
    int nb;
};

__device__ __forceinline__ uint32_t cell_offset(const int c[3], int ly, int lz)
{
    return ((uint32_t)c[0] << (ly + lz)) | ((uint32_t)c[1] << lz) | (uint32_t)c[2];
}

// Layout of the per-sample arrays of the render-time ops (scanerf_hip.h `sample_major`): element (ray i, sample s) of B x S
//   0  [B][S]          the reference's
//   1  [S][B]          sample-major
//   2  [B/32][S][32]   ray blocks: 32 neighbouring rays side by side, their samples in order (B a multiple of 32)
// In 1 and 2 a wave's 32 samples are one depth index of 32 neighbouring rays (neighbouring pixels share their cells down to
// the fine levels: the gathers of a wave fall on a few lines); in 2 consecutive groups of a wave also walk ALONG those rays, and
// the chip is spread over all depths at any time (in 1 every CU works on the same depth slab and the same few lines of the
// coarse levels -- measured 1.5-2x slower than 0).
__device__ __forceinline__ size_t pt_index(int i, int s, int B, int S, int lay)
{
    return lay == 0 ? (size_t)i * S + s : lay == 1 ? (size_t)s * B + i : ((size_t)(i >> 5) * S + s) * 32 + (i & 31);
}
__device__ __forceinline__ void pt_decompose(uint32_t e, uint32_t B, uint32_t S, int lay, int &i, int &s)
{
    if (lay == 0) { i = (int)(e / S); s = (int)(e - (uint32_t)i * S); }
    else if (lay == 1) { s = (int)(e / B); i = (int)(e - (uint32_t)s * B); }
    else { const uint32_t g = e >> 5, rb = g / S; s = (int)(g - rb * S); i = (int)(rb * 32 + (e & 31u)); }
}
__device__ __forceinline__ size_t pt_sample_stride(int B, int lay) { return lay == 0 ? 1 : lay == 1 ? (size_t)B : 32; }

// ---- rendering_kernel.cu:126-174 ---------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ray_block_intersection(const float *__restrict__ rays_o,
                                                                const float *__restrict__ rays_d, Tiles t,
                                                                float *__restrict__ inter, int B)
{
    const int64_t total = (int64_t)B * t.nb;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / t.nb), b = (int)(e % t.nb);
        float o[3], d[3], c[3], h[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = rays_o[3 * i + k];
            d[k] = rays_d[3 * i + k];
            h[k] = t.sizes[3 * b + k] / 2.0f;
            c[k] = t.corners[3 * b + k] + h[k];
        }
        F2 r = clip_box(o, d, c, h);
        if (r.x == -1.0f) r.x = r.y = kInf;
        reinterpret_cast<float2 *>(inter)[e] = make_float2(r.x, r.y);
    }
}

// ---- rendering_kernel.cu:179-382: one tracing step per ray ---------------------------------------
__global__ void __launch_bounds__(64) k_render_sample_points(const float *__restrict__ rays_o,
                                                             const float *__restrict__ rays_d, Tiles t, int S,
                                                             const int32_t *__restrict__ tracing_blocks,
                                                             const float *__restrict__ inter,
                                                             int32_t *__restrict__ tracing_idx, float *__restrict__ z_start,
                                                             float *__restrict__ z_vals, float *__restrict__ dists, int B, int sm)
{
    const size_t ks = pt_sample_stride(B, sm);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        const float o[3] = { rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2] };
        const float d[3] = { rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2] };
        const int32_t *tb = tracing_blocks + (size_t)i * t.nb;
        const float2 *ci = reinterpret_cast<const float2 *>(inter) + (size_t)i * t.nb;
        float *cz = z_vals + pt_index(i, 0, B, S, sm), *cd = dists + pt_index(i, 0, B, S, sm);
        int step = tracing_idx[i];
        float tsx = z_start[i];
        while (step < t.nb) {
            const int b = tb[step];
            const float2 bound = ci[b];
            if (bound.x == kInf) break;
            if (tsx >= bound.y) { ++step; continue; }
            if (step == 0) tsx = bound.x;
            const int l2d[3] = { t.log2dim[3 * b], t.log2dim[3 * b + 1], t.log2dim[3 * b + 2] };
            int side[3];
            float cs[3], og[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                side[k] = 1 << l2d[k];
                cs[k] = t.sizes[3 * b + k] / (float)side[k];
                og[k] = o[k] - t.corners[3 * b + k];
            }
            const uint8_t *g = t.occ + t.grid_starts[b];
            F2 ts;
            ts.x = tsx;
            ts.y = 0.0f;
            Walker w;
            w.start(og, d, ts, side, cs);
            int num_seg = 0;
            float total = 0.0f;
            while (!w.done()) {
                w.pick();
                if (g[cell_offset(w.cell, l2d[1], l2d[2])]) {
                    const float len = w.t1 - w.t0;
                    if (len > 0) { total += len; ++num_seg; }
                }
                w.advance();
            }
            if (num_seg == 0) { tsx = bound.y; ++step; continue; }
            int num = 0, count = 0;
            w.start(og, d, ts, side, cs);
            while (!w.done()) {
                w.pick();
                if (g[cell_offset(w.cell, l2d[1], l2d[2])]) {
                    const float len = w.t1 - w.t0;
                    if (len > 0) {
                        int n = (int)(len / total * (float)S);
                        n = n < 1 ? 1 : n;
                        n = n > S - num ? S - num : n;
                        if (count == num_seg - 1) n = S - num;
                        if (n > 0) {
                            const float interval = (w.t1 - w.t0) / (float)n;
                            for (int k = 0; k < n; ++k) {
                                cz[(num + k) * ks] = w.t0 + (float)k * interval;
                                cd[(num + k) * ks] = interval;
                            }
                        }
                        num += n;
                        ++count;
                    }
                }
                w.advance();
            }
            tsx = bound.y;
            ++step;
            break;
        }
        tracing_idx[i] = step;
        z_start[i] = tsx;
    }
}

// ---- rendering_kernel.cu:391-449 (the reference overruns its 4 slots when >4 tiles overlap; clamped) --
__global__ void __launch_bounds__(256) k_prepare_points(const float *__restrict__ z_vals,
                                                        const uint8_t *__restrict__ running,
                                                        int16_t *__restrict__ block_idxs,
                                                        const float *__restrict__ inter, int S, int nb, int B, int sm)
{
    const int64_t total = (int64_t)B * S;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int i, s_;
        pt_decompose((uint32_t)e, (uint32_t)B, (uint32_t)S, sm, i, s_);
        if (!running[i]) continue;
        const float z = z_vals[e];
        if (z == -1.0f) continue;
        const float2 *ci = reinterpret_cast<const float2 *>(inter) + (size_t)i * nb;
        int16_t out[kMaxPtsBlocks] = { -1, -1, -1, -1 };
        int idx = 0;
        bool any = false;
        for (int b = 0; b < nb && idx < kMaxPtsBlocks; ++b) {
            const float2 bd = ci[b];
            if (z >= bd.x && z <= bd.y) { out[idx++] = (int16_t)b; any = true; }
        }
        if (any) {  // the reference leaves untouched slots as the caller filled them (-1)
            int16_t *dst = block_idxs + e * kMaxPtsBlocks;
            for (int k = 0; k < idx; ++k) dst[k] = out[k];
        }
    }
}

// xz-distance blend weight (rendering_kernel.cu:523-537, :1335-1351)
__device__ __forceinline__ float xz_weight(float dx, float dz)
{
    if (dx != 0 && dz != 0) return dx * dz;
    if (dx != 0) return dx;
    if (dz != 0) return dz;
    return 0.0f;
}

struct InferArgs {
    const float *rays_o, *rays_d, *z_vals, *dists;
    const int16_t *block_idxs;    // fg: [B,S,4]; bg: [B,4] (+ step)
    const void *tables;           // [nb,16,T,2] f16
    const float *images;          // [nb, PK_TOTAL] packed decoders (weight_feature == 1)
    const int32_t *res;           // [nb,16,3]
    Tiles t;
    float *out_dif, *out_spec, *out_alpha;
    int T, B, S, step;
    int sm;   // layout of the per-sample arrays (pt_index)
    const uint8_t *running;   // scanerf_pts_inference_tracing: the slot lists are derived in the kernel from the running mask,
    const float *inter;       // the samples' depths and the rays' [nb] (near, far) intervals (= prepare_points, :391-449)
    int skip_unsampled;       // (tracing only) rays whose first depth is -1 hold no sample: their outputs are left unwritten
    int dbg;  // timing experiments only (-DSCANERF_RT_EXPERIMENTS, SCANERF_DEBUG_RT): 1 = no decoder, 2 = no table gathers
};

// ---- rendering_kernel.cu:467-621 (BG == false) and :1012-1171 (BG == true) -------------------------
template <bool BG>
__global__ void __launch_bounds__(256, 2) k_pts_inference(InferArgs a)
{
    const int lane = threadIdx.x & 63, sl = lane & 31, h = lane >> 5;
    const int64_t total = (int64_t)a.B * a.S;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32; base < total; base += nwaves * 32) {
        const int64_t e = base + sl;
        const bool in_range = e < total;
        const int64_t ec = in_range ? e : total - 1;
        const int i = (int)(ec / a.S), s = (int)(ec % a.S);
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.rays_o[3 * i + k];
            d[k] = a.rays_d[3 * i + k];
        }
        const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float z = a.z_vals[ec];
        float delta;  // what multiplies sigma in alpha = 1 - exp(-sigma*delta)
        if (BG) delta = (s == a.S - 1) ? 10000000.0f : a.z_vals[ec + 1] - z;   // :1045-1047: raw depth step
        else delta = a.dists[ec] * dnorm;                                       // :557
        float dif[3] = { 0, 0, 0 }, spc[3] = { 0, 0, 0 }, alpha = 0.0f, weight = 0.0f;

        const int nslots = BG ? 1 : kMaxPtsBlocks;
        bool ended = !in_range;  // fg: the slot list stops at the first -1 (:499)
        for (int k = 0; k < nslots; ++k) {
            int b_lane = -1;
            if (!ended) b_lane = BG ? a.block_idxs[i * kMaxPtsBlocks + a.step] : a.block_idxs[ec * kMaxPtsBlocks + k];
            if (b_lane == -1) ended = true;
            unsigned long long pending = __ballot(b_lane != -1);
            while (pending) {
                const int leader = __ffsll((long long)pending) - 1;
                const int b = __shfl(b_lane, leader, 64);  // wave-uniform tile
                const bool mine = b_lane == b;
                pending &= ~__ballot(mine);
                // tile-space position, blend weight, occupancy
                float p01[3], w = 0.0f;
                bool run = mine;
                if (BG) {
                    // L-infinity contraction of the 2x-box coordinates (:1056-1096), then [-2,2] -> [0,1]
                    float q[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) q[c] = 2.0f * ((o[c] + z * d[c]) - a.t.corners[3 * b + c]) / a.t.sizes[3 * b + c] - 1.0f;
                    const float linf = fmaxf(fabsf(q[0]), fmaxf(fabsf(q[1]), fabsf(q[2])));
                    const float ratio = (2.0f - 1.0f / linf) / linf;
#pragma unroll
                    for (int c = 0; c < 3; ++c) p01[c] = (q[c] * ratio + 2.0f) / 4.0f;
                } else {
                    float pt[3], dis[3];
                    int loc[3];
                    const int l2d[3] = { a.t.log2dim[3 * b], a.t.log2dim[3 * b + 1], a.t.log2dim[3 * b + 2] };
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        pt[c] = ((o[c] + z * d[c]) - a.t.corners[3 * b + c]) / a.t.sizes[3 * b + c];
                        dis[c] = (0.5f - fabsf(pt[c] - 0.5f)) * a.t.sizes[3 * b + c];
                        const int r = 1 << l2d[c];
                        int cc = (int)(pt[c] * (float)r);
                        loc[c] = cc < 0 ? 0 : (cc > r - 1 ? r - 1 : cc);
                        p01[c] = pt[c] / 2.0f + 0.25f;  // tile -> the middle half of the 2x box (:548)
                    }
                    w = xz_weight(dis[0], dis[2]);
                    if (mine) {
                        weight += w;
                        run = a.t.occ[a.t.grid_starts[b] + cell_offset(loc, l2d[1], l2d[2])] != 0;
                    }
                }
                if (!__any(run)) continue;  // wave-uniform
                v16f x;
                encode8_01<SCANERF_F16>((const char *)a.tables + (size_t)b * 16 * a.T * 4, a.res + (size_t)b * 48, a.T, h, p01,
                                        run, x);
                SampleOut so = decode_tile_dir(a.images + (size_t)b * WS_FLOATS, lane, x, d, dnorm, 0.0f);
                if (run) {
                    const float pa = 1.0f - expf(-1.0f * so.sigma * delta);
                    if (BG) {
                        alpha = pa;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            dif[c] = pa * so.dif[c];
                            spc[c] = pa * (so.tint[c] * so.spec[c]);
                        }
                    } else {
                        alpha += w * pa;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            dif[c] += w * pa * so.dif[c];
                            spc[c] += w * pa * (so.tint[c] * so.spec[c]);
                        }
                    }
                }
            }
        }
        if (!BG && weight > 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { dif[c] /= weight; spc[c] /= weight; }
            alpha /= weight;
        }
        // every sample is written: fg zeros when no tile applies (:569-571); bg rays without a tile at this blend step -- which
        // the reference leaves as its caller cleared them (:1032-1036, rendering.py:493-495) -- get their zeros here
        const bool wr = in_range && h == 0;
        if (wr) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.out_dif[3 * e + c] = dif[c];
                a.out_spec[3 * e + c] = spc[c];
            }
            a.out_alpha[e] = alpha;
        }
    }
}


// ---- pieces of the software-pipelined group loop of k_pts_inference_chunks ------------------------------------------------
// acc += w * (float)(low / high half of an f16 pair): one instruction instead of a conversion and an fma, the same value.  The
// result stays in acc's register, which ordinary instructions have written before (render_h3.h, h3_residual_lo: inline asm is
// invisible to the hazard recogniser, so it must not be handed a register that a matrix instruction in flight may own).
__device__ __forceinline__ float fma_mix_lo(uint32_t h2, float w, float acc)
{
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h2), "v"(w));
    return acc;
}
__device__ __forceinline__ float fma_mix_hi(uint32_t h2, float w, float acc)
{
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h2), "v"(w));
    return acc;
}

// what one 32-sample group reads from the per-sample arrays (loaded one group ahead of its use)
struct GroupIn {
    float z, aux;     // depth; fg: dists, bg: the next sample's depth
    uint32_t s0, s1;  // fg: the four 16-bit slots; bg: s0 = the ray's background tile at this blend step (sign-extended)
    uint32_t e;       // element index (clamped into range)
    int i;            // ray
    bool in_range, last;
};
// a group found to have samples in the staged tile: what its decoder and output stage needs
struct GroupPrep {
    uint32_t e;
    bool run;
};

template <bool BG>
__device__ __forceinline__ void group_load(const InferArgs &a, int64_t total, int64_t base, int sl, GroupIn &in)
{
    const int64_t e = base + sl;
    in.in_range = e < total;
    const uint32_t ec = (uint32_t)(in.in_range ? e : total - 1);
    int s;
    pt_decompose(ec, (uint32_t)a.B, (uint32_t)a.S, a.sm, in.i, s);
    in.e = ec;
    in.z = a.z_vals[ec];
    if (BG) {
        in.last = s == a.S - 1;
        in.aux = a.z_vals[in.last ? ec : ec + (uint32_t)pt_sample_stride(a.B, a.sm)];
        in.s0 = (uint32_t)(int)a.block_idxs[in.i * kMaxPtsBlocks + a.step];
        in.s1 = 0;
    } else {
        in.last = false;
        in.aux = a.dists[ec];
        const uint2 raw = *reinterpret_cast<const uint2 *>(a.block_idxs + (size_t)ec * kMaxPtsBlocks);
        in.s0 = raw.x;
        in.s1 = raw.y;
    }
}

// Same arithmetic as the group loop below (rendering_kernel.cu:499-557 / :1040-1060).  Returns whether any sample of the
// group runs the decoder of tile b (wave-uniform).
template <bool BG>
__device__ __forceinline__ bool group_prep(const InferArgs &a, int b, const float cb[3], const float sb[3], const GroupIn &in,
                                           GroupPrep &P, float p01[3], float (*park)[64], int lane)
{
    // (direction, depth step, blend weight and 1/sum of weights wait in LDS for the output stage -- registers the compiler
    // would otherwise spill to scratch, whose reloads wait behind the gathers in flight)
    float d[3], delta;
    int16_t slot[kMaxPtsBlocks] = { -1, -1, -1, -1 };
    bool mine = false;
    if (BG) {
        mine = in.in_range && (int)in.s0 == b;
    } else if (in.in_range) {
        slot[0] = (int16_t)(in.s0 & 0xffffu); slot[1] = (int16_t)(in.s0 >> 16);
        slot[2] = (int16_t)(in.s1 & 0xffffu); slot[3] = (int16_t)(in.s1 >> 16);
        bool ended = false;
#pragma unroll
        for (int k = 0; k < kMaxPtsBlocks; ++k) {
            ended |= slot[k] == -1;
            if (ended) slot[k] = -1;
            mine |= slot[k] == b;
        }
    }
    P.run = false;
    if (!__any(mine)) return false;
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = a.rays_o[3 * in.i + k];
        d[k] = a.rays_d[3 * in.i + k];
    }
    const float z = in.z;
    float w_b = 0.0f, weight = 0.0f;
    bool run = mine;
    if (BG) {
        delta = in.last ? 10000000.0f : in.aux - z;  // :1045-1047: raw depth step
        float q[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = 2.0f * ((o[c] + z * d[c]) - cb[c]) / sb[c] - 1.0f;
        const float linf = fmaxf(fabsf(q[0]), fmaxf(fabsf(q[1]), fabsf(q[2])));
        const float ratio = (2.0f - 1.0f / linf) / linf;
#pragma unroll
        for (int c = 0; c < 3; ++c) p01[c] = (q[c] * ratio + 2.0f) / 4.0f;
    } else {
        const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        delta = in.aux * dnorm;  // :557
#pragma unroll
        for (int k = 0; k < kMaxPtsBlocks; ++k) {
            const int bk = slot[k];
            if (bk == -1) continue;
            float dis[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float sz = a.t.sizes[3 * bk + c];
                const float pt = ((o[c] + z * d[c]) - a.t.corners[3 * bk + c]) / sz;
                dis[c] = (0.5f - fabsf(pt - 0.5f)) * sz;
            }
            const float w = xz_weight(dis[0], dis[2]);
            weight += w;
            if (bk == b) w_b = w;
        }
        int loc[3];
        const int l2d[3] = { a.t.log2dim[3 * b], a.t.log2dim[3 * b + 1], a.t.log2dim[3 * b + 2] };
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float pt = ((o[c] + z * d[c]) - cb[c]) / sb[c];
            const int r = 1 << l2d[c];
            const int cc = (int)(pt * (float)r);
            loc[c] = cc < 0 ? 0 : (cc > r - 1 ? r - 1 : cc);
            p01[c] = pt / 2.0f + 0.25f;
        }
        if (mine) run = a.t.occ[a.t.grid_starts[b] + cell_offset(loc, l2d[1], l2d[2])] != 0;
    }
    P.e = in.e;
    P.run = run;
    park[0][lane] = d[0]; park[1][lane] = d[1]; park[2][lane] = d[2];
    park[3][lane] = delta; park[4][lane] = w_b; park[5][lane] = weight > 0 ? 1.0f / weight : 1.0f;
    return __any(run);
}

// First half of encode8_01<F16, 8, STRAIGHT>: the 64 corner loads of a group go out (raw f16 pairs; uniform level base + a
// 32-bit lane offset) and the interpolation offsets are parked in LDS; nothing here waits for memory.  Level by level (fenced),
// so that only one level's addresses are live beside the 64 destinations.
__device__ __forceinline__ void gather_issue(const char *table, const float *rs, int T, int h, const float p01[3],
                                             uint32_t raw[64], float (*tp)[64], int lane)
{
    const uint32_t mask = (uint32_t)T - 1u, hoff = (uint32_t)(2 * h) * (uint32_t)T;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int lu = 4 * (j >> 1) + (j & 1);  // level lu + 2h
        int bc[3];
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float v = p01[k] * rs[3 * lu + k];  // rs = (float)(res - 1) of this half's levels (staged in LDS with the tile's image)
                bc[k] = (int)v;
                tp[3 * j + k][lane] = v - (float)bc[k];
            }
        }
        uint32_t idx[8];
        corner_indices(idx, bc[0], bc[1], bc[2], mask);
        const char *base = table + (size_t)lu * T * 4;
#pragma unroll
        for (int c = 0; c < 8; ++c) raw[8 * j + c] = *reinterpret_cast<const uint32_t *>(base + (size_t)((hoff + idx[c]) * 4u));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Second half: trilinear interpolation of the loaded corners (same sums, in the same order, as encode8_01)
__device__ __forceinline__ void gather_finish(const uint32_t raw[64], float (*tp)[64], int lane, v16f &x)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float w[8];
        trilinear_weights(w, tp[3 * j][lane], tp[3 * j + 1][lane], tp[3 * j + 2][lane]);
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            ax = fma_mix_lo(raw[8 * j + c], w[c], ax);
            ay = fma_mix_hi(raw[8 * j + c], w[c], ay);
        }
        x[2 * j] = ax;  // (lanes without a live sample decode whatever their valid-address loads returned; nothing of theirs is written)
        x[2 * j + 1] = ay;
    }
    SCANERF_LOAD_GUARD();
}

// ---- pts_inference / bg_pts_inference_v2, chunk-major with the tile's decoder in LDS (default) ---------------------------
// The kernel above reads every MFMA operand of whichever tile a sample references from global memory and multiplies on
// the f32 matrix pipe (1.5e9 samples/s).  Here a workgroup takes a chunk of 64 consecutive 32-sample groups (16 rays at
// 128 samples), finds the set of tiles its samples list (one 8-byte slot load per lane and group), and for each tile of the
// set in ascending order stages that tile's split-f16 decoder image (render_h3.h, the arithmetic of the training kernels;
// 70 KB, from L2) in LDS and runs the groups that list it.  Neighbouring rays see the same one or two tiles, so a chunk
// stages one or two images for 2048 samples; the cost does not grow with the number of tiles of the scene (a first version
// made one launch per tile: every pass re-read all slot lists, ~1 ms per tile the view does not even see).  Samples in
// the overlap of several tiles are blended across the chunk's tile steps: each adds w_b * pa * colour / sum_k w_k into the
// (zero-filled) outputs; a sample belongs to one workgroup and the steps are sequential, so the read-modify-write is
// race-free and its order (ascending tile index) is fixed.
// 8 waves share one staged image (two per SIMD; with 4, the 104 KB image left one wave per SIMD and the decoder's dependent
// MFMA chains exposed: 3.3e9 samples/s whatever the gathers did)
#ifndef RT_GATHER_BATCH
#define RT_GATHER_BATCH 8
#endif
#ifndef RT_STRAIGHT
#define RT_STRAIGHT true
#endif
constexpr int kChunkThreads = 512, kChunkWaves = kChunkThreads / 64, kChunkWaveGroups = 16;
// NT threads, WPS waves per SIMD: <512, 1> = the software-pipelined form (PIPE; two waves per SIMD by its 142 KB of LDS);
// <768, 3> without the pipeline = three waves per SIMD (one workgroup of 12 waves per CU around one staged image)
#ifndef RT_W3_GROUPS
#define RT_W3_GROUPS 16
#endif
template <bool BG, bool PIPE, int NT = kChunkThreads, int WPS = 1, int kWG = kChunkWaveGroups>
__global__ void __launch_bounds__(NT, WPS) k_pts_inference_chunks(InferArgs a)
{
    constexpr int kChunkThreads = NT, kChunkWaves = NT / 64, kChunkWaveGroups = kWG;
    static_assert(!PIPE || (NT == 512 && WPS == 1), "the pipelined form is the 512-thread one");
    // One block of LDS, the decoder image first: its reads then are `lane base + 16-bit immediate` (with the image behind the
    // other arrays every read past 64 KB took an address register of its own, ~25 live across the group loop).
    // PIPE: tpark = interpolation offsets of the group whose gathers are in flight; ppark = direction, depth step and blend
    // weights of the two groups in the pipeline; rscale = (float)(res - 1) of the staged tile's 16 levels
    constexpr int kImg = (H3_BYTES + 15) & ~15, kTp = PIPE ? kChunkWaves * 24 * 64 * 4 : 0, kPp = PIPE ? kChunkWaves * 2 * 6 * 64 * 4 : 0;
    __shared__ __attribute__((aligned(16))) char smem[kImg + kTp + kPp + 48 * 4 + 8];
    char *const lds = smem;
    float (*const tpark)[24][64] = reinterpret_cast<float (*)[24][64]>(smem + kImg);
    float (*const ppark)[2][6][64] = reinterpret_cast<float (*)[2][6][64]>(smem + kImg + kTp);
    float *const rscale = reinterpret_cast<float *>(smem + kImg + kTp + kPp);
    uint32_t *const tileset = reinterpret_cast<uint32_t *>(smem + kImg + kTp + kPp + 48 * 4);
    const int lane = threadIdx.x & 63, sl = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    const int64_t total = (int64_t)a.B * a.S;
    constexpr int kWaveGroups = kChunkWaveGroups, kChunkGroups = kChunkWaves * kWaveGroups;
    const int64_t ngroups = (total + 31) / 32, nchunks = (ngroups + kChunkGroups - 1) / kChunkGroups;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (threadIdx.x < 2) tileset[threadIdx.x] = 0;
        __syncthreads();  // (also: the previous chunk's last tile step is complete)
        const int64_t wbase = (chunk * kChunkGroups + (int64_t)wave * kWaveGroups) * 32;
        {   // 1. the tiles this chunk's samples list: all slot loads of the wave in flight together
            uint32_t mlo = 0, mhi = 0;
            auto mark = [&](int t) {
                if (t >= 0) {
                    if (t < 32) mlo |= 1u << t;
                    else mhi |= 1u << (t - 32);
                }
            };
#pragma unroll
            for (int g = 0; g < kWaveGroups; ++g) {
                const int64_t e = wbase + g * 32 + sl;
                if (e >= total || h != 0) continue;
                const uint32_t e32 = (uint32_t)e;
                if (BG) {
                    int ri, rs;
                    pt_decompose(e32, (uint32_t)a.B, (uint32_t)a.S, a.sm, ri, rs);
                    const int tb = a.block_idxs[ri * kMaxPtsBlocks + a.step];
                    mark(tb);
                    if (tb < 0) {  // no background tile at this blend step: the sample's outputs are zero (the caller need not clear them)
                        a.out_alpha[e] = 0.0f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            a.out_dif[3 * e + c] = 0.0f;
                            a.out_spec[3 * e + c] = 0.0f;
                        }
                    }
                } else {
                    const uint2 raw = *reinterpret_cast<const uint2 *>(a.block_idxs + (size_t)e32 * kMaxPtsBlocks);
                    const int s0 = (int16_t)(raw.x & 0xffffu), s1 = (int16_t)(raw.x >> 16), s2 = (int16_t)(raw.y & 0xffffu),
                              s3 = (int16_t)(raw.y >> 16);
                    mark(s0);  // the list stops at the first -1 (rendering_kernel.cu:499)
                    if (s0 != -1) { mark(s1); if (s1 != -1) { mark(s2); if (s2 != -1) mark(s3); } }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                mlo |= __shfl_xor(mlo, off, 64);
                mhi |= __shfl_xor(mhi, off, 64);
            }
            if (lane == 0) {
                if (mlo) atomicOr(&tileset[0], mlo);
                if (mhi) atomicOr(&tileset[1], mhi);
            }
        }
        __syncthreads();
        uint64_t todo = (uint64_t)tileset[0] | ((uint64_t)tileset[1] << 32);
      while (todo) {  // 2. one step per listed tile, ascending
        const int b = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        __syncthreads();  // every wave is done with the previous image
        {
            const float4 *src = reinterpret_cast<const float4 *>(a.images + (size_t)b * WS_FLOATS + PK_TOTAL);
            float4 *dst = reinterpret_cast<float4 *>(lds);
            for (int i = threadIdx.x; i < H3_BYTES / 16; i += kChunkThreads) dst[i] = src[i];
            if (PIPE && threadIdx.x < 48) rscale[threadIdx.x] = (float)(a.res[(size_t)b * 48 + threadIdx.x] - 1);
        }
        __syncthreads();
        float cb[3], sb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            cb[c] = a.t.corners[3 * b + c];
            sb[c] = a.t.sizes[3 * b + c];
        }
#ifdef RT_SKEW   // timing experiment (tools/build_variant.py render_time="-ffp-contract=off -DRT_SKEW=n"): the second wave of every SIMD starts
        // a tile step n x 64 cycles late, so that one wave's matrix work meets the other's vector work instead of both competing for the same pipe
        if (wave >= 4) __builtin_amdgcn_s_sleep(RT_SKEW);
#endif
      if constexpr (PIPE) {
        // Software pipeline over the wave's groups that run this tile's decoder: while group g is decoded, the 64 corner
        // loads of the next running group and the per-sample inputs of the group after it are in flight (the kernel is bound
        // by the round trips of these loads, not by their number -- DESIGN.md 4.8).
        const int64_t rem = total - wbase;
        const int ng = rem <= 0 ? 0 : (int)(rem >= (int64_t)kWaveGroups * 32 ? kWaveGroups : (rem + 31) / 32);
        const char *table = (const char *)a.tables + (size_t)b * 16 * a.T * 4;
        int rso = 6 * h;  // (opaque, or every rscale address becomes its own lane-dependent register instead of base + immediate)
        asm volatile("" : "+v"(rso));
        const float *rsh = rscale + rso;
        GroupIn in;
        GroupPrep cur, nxt;
        uint32_t raw[64];
        float p01[3];
        int g_in = 0, slot_cur = 0;
        if (ng > 0) group_load<BG>(a, total, wbase, sl, in);
        auto advance = [&](GroupPrep &P, int sl_) -> bool {  // the next group with samples to decode; keeps one group of inputs ahead
            while (g_in < ng) {
                const bool r = group_prep<BG>(a, b, cb, sb, in, P, p01, ppark[wave][sl_], lane);
                ++g_in;
                if (g_in < ng) group_load<BG>(a, total, wbase + (int64_t)g_in * 32, sl, in);
                if (r) return true;
            }
            return false;
        };
        bool have = advance(cur, 0);
        if (have) gather_issue(table, rsh, a.T, h, p01, raw, tpark[wave], lane);
        while (have) {
            v16f x;
            gather_finish(raw, tpark[wave], lane, x);
            __builtin_amdgcn_sched_barrier(0);  // the next group's loads go out after this group's corners are consumed ...
            const bool more = advance(nxt, slot_cur ^ 1);
            if (more) gather_issue(table, rsh, a.T, h, p01, raw, tpark[wave], lane);
            __builtin_amdgcn_sched_barrier(0);  // ... and before its decoder starts
            float (*pk)[64] = ppark[wave][slot_cur];
            v16f dinit[2];
            {
                const float d[3] = { pk[0][lane], pk[1][lane], pk[2][lane] };
                const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                float sh[16];
                ray_sh(d, dnorm, sh, 0.0f);
                h3_dinit(lds, lane, sh, dinit);
            }
            const SampleOut so = decode_tile_h3(lds, lane, x, dinit);
            if (cur.run && h == 0) {
                const uint32_t e = cur.e;
                const float delta = pk[3][lane], w_b = pk[4][lane], inv = pk[5][lane];
                const float pa = 1.0f - expf(-1.0f * so.sigma * delta);
                if (BG) {
                    a.out_alpha[e] = pa;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        a.out_dif[3 * (size_t)e + c] = pa * so.dif[c];
                        a.out_spec[3 * (size_t)e + c] = pa * (so.tint[c] * so.spec[c]);
                    }
                } else {
                    a.out_alpha[e] += (w_b * pa) * inv;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        a.out_dif[3 * (size_t)e + c] += (w_b * pa * so.dif[c]) * inv;
                        a.out_spec[3 * (size_t)e + c] += (w_b * pa * (so.tint[c] * so.spec[c])) * inv;
                    }
                }
            }
            SCANERF_STORE_GUARD();
            cur = nxt;
            slot_cur ^= 1;
            have = more;
        }
      } else {
#pragma unroll 1
      for (int g = 0; g < kWaveGroups; ++g) {
        const int64_t base = wbase + g * 32;
        if (base >= total) break;
        const int64_t e = base + sl;
        const bool in_range = e < total;
        const int64_t ec = in_range ? e : total - 1;
        // (32-bit division: the host keeps B*S below 2^31 for this kernel; a 64-bit one costs ~100 instructions per group)
        int i, s;
        pt_decompose((uint32_t)ec, (uint32_t)a.B, (uint32_t)a.S, a.sm, i, s);
        // does this sample list tile b?  (fg: the slot list stops at the first -1, rendering_kernel.cu:499)
        int16_t slot[kMaxPtsBlocks] = { -1, -1, -1, -1 };
        bool mine = false;
        if (BG) {
            mine = in_range && a.block_idxs[i * kMaxPtsBlocks + a.step] == b;
        } else if (in_range) {
            const uint2 raw = *reinterpret_cast<const uint2 *>(a.block_idxs + ec * kMaxPtsBlocks);
            slot[0] = (int16_t)(raw.x & 0xffffu); slot[1] = (int16_t)(raw.x >> 16);
            slot[2] = (int16_t)(raw.y & 0xffffu); slot[3] = (int16_t)(raw.y >> 16);
            bool ended = false;
#pragma unroll
            for (int k = 0; k < kMaxPtsBlocks; ++k) {
                ended |= slot[k] == -1;
                if (ended) slot[k] = -1;
                mine |= slot[k] == b;
            }
        }
        if (!__any(mine)) continue;  // wave-uniform
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.rays_o[3 * i + k];
            d[k] = a.rays_d[3 * i + k];
        }
        const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float z = a.z_vals[ec];
        float delta;
        if (BG) delta = (s == a.S - 1) ? 10000000.0f : a.z_vals[ec + pt_sample_stride(a.B, a.sm)] - z;   // :1045-1047: raw depth step
        else delta = a.dists[ec] * dnorm;                                       // :557
        float p01[3], w_b = 0.0f, weight = 0.0f;
        bool run = mine;
        if (BG) {
            float q[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) q[c] = 2.0f * ((o[c] + z * d[c]) - cb[c]) / sb[c] - 1.0f;
            const float linf = fmaxf(fabsf(q[0]), fmaxf(fabsf(q[1]), fabsf(q[2])));
            const float ratio = (2.0f - 1.0f / linf) / linf;
#pragma unroll
            for (int c = 0; c < 3; ++c) p01[c] = (q[c] * ratio + 2.0f) / 4.0f;
        } else {
            // blend weights of every listed tile (occupied or not, :523-541), this tile's cell and position
#pragma unroll
            for (int k = 0; k < kMaxPtsBlocks; ++k) {
                const int bk = slot[k];
                if (bk == -1) continue;
                float dis[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float sz = a.t.sizes[3 * bk + c];
                    const float pt = ((o[c] + z * d[c]) - a.t.corners[3 * bk + c]) / sz;
                    dis[c] = (0.5f - fabsf(pt - 0.5f)) * sz;
                }
                const float w = xz_weight(dis[0], dis[2]);
                weight += w;
                if (bk == b) w_b = w;
            }
            int loc[3];
            const int l2d[3] = { a.t.log2dim[3 * b], a.t.log2dim[3 * b + 1], a.t.log2dim[3 * b + 2] };
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float pt = ((o[c] + z * d[c]) - cb[c]) / sb[c];
                const int r = 1 << l2d[c];
                const int cc = (int)(pt * (float)r);
                loc[c] = cc < 0 ? 0 : (cc > r - 1 ? r - 1 : cc);
                p01[c] = pt / 2.0f + 0.25f;  // tile -> the middle half of the 2x box (:548)
            }
            if (mine) run = a.t.occ[a.t.grid_starts[b] + cell_offset(loc, l2d[1], l2d[2])] != 0;
        }
        if (!__any(run)) continue;  // wave-uniform: nothing of this group is occupied (the outputs stay as they are)
        v16f x;
#ifdef SCANERF_RT_EXPERIMENTS
        if (a.dbg == 2) {
#pragma unroll
            for (int g2 = 0; g2 < 16; ++g2) x[g2] = p01[g2 % 3] * (0.01f * g2);
        } else if (a.dbg == 3 || a.dbg == 4) {  // only one lane in 8 (3) / 4 (4) gathers: what would fewer lane-loads buy?
            encode8_01<SCANERF_F16, RT_GATHER_BATCH>((const char *)a.tables + (size_t)b * 16 * a.T * 4, a.res + (size_t)b * 48, a.T, h, p01,
                                                      run && ((lane & (a.dbg == 3 ? 7 : 3)) == 0), x);
        } else
#endif
        encode8_01<SCANERF_F16, RT_GATHER_BATCH, RT_STRAIGHT>((const char *)a.tables + (size_t)b * 16 * a.T * 4, a.res + (size_t)b * 48, a.T, h, p01, run, x);
        SampleOut so;
#ifdef SCANERF_RT_EXPERIMENTS
        if (a.dbg == 1) {
            so.sigma = x[0] + x[5] + x[10] + x[15];
#pragma unroll
            for (int c = 0; c < 3; ++c) { so.dif[c] = x[1 + c] + x[12 + c]; so.tint[c] = x[4 + c] + x[9 + c]; so.spec[c] = x[7 + c] + x[6 + c]; }
        } else
#endif
            so = decode_tile_h3<true>(lds, lane, x, nullptr, d, 0.0f);
        if (run && h == 0) {
            const float pa = 1.0f - expf(-1.0f * so.sigma * delta);
            if (BG) {
                a.out_alpha[e] = pa;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a.out_dif[3 * e + c] = pa * so.dif[c];
                    a.out_spec[3 * e + c] = pa * (so.tint[c] * so.spec[c]);
                }
            } else {
                const float inv = weight > 0 ? 1.0f / weight : 1.0f;
                a.out_alpha[e] += (w_b * pa) * inv;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a.out_dif[3 * e + c] += (w_b * pa * so.dif[c]) * inv;
                    a.out_spec[3 * e + c] += (w_b * pa * (so.tint[c] * so.spec[c])) * inv;
                }
            }
        }
        SCANERF_STORE_GUARD();
      }
      }
      }
    }
}

// ---- the same chunk-major kernel on 16-SAMPLE tiles at four waves per SIMD (default) ------------------------------------------
// k_pts_inference_chunks above runs two waves per SIMD: ~250 registers (64 gather destinations of the software pipeline beside the
// 32-sample decoder) and 142 KB of LDS per workgroup, and its decoder's dependent MFMA -> activation -> split chains have one
// other wave to hide behind (de-phasing the two waves changes nothing: profiles/r05_render_pmc.txt).  Here the decoder is
// decode_tile_s16 (render_t16.h: v_mfma_f32_16x16x32_f16 on the t16s image, the arithmetic of the training backward's recompute,
// ~95 registers) and a lane gathers 4 levels instead of 8 (32 destinations, dead before the decoder starts): <= 128 registers,
// no software pipeline, 77 KB of LDS -> two workgroups = 16 waves per CU, and the gathers of one wave wait behind the matrix and
// vector work of three others.  A wave takes 64 consecutive samples at a time: lane l prepares sample l (position, blend weights,
// occupancy), then each of the four 16-sample tiles with a live sample is decoded with lane (c, q) = sample 16 t + c, quarter q
// (inputs by ds_bpermute from the preparing lane).
#ifndef T16_WAVE_GROUPS
#define T16_WAVE_GROUPS 8   // 8 x 64 = the 512 samples per wave and chunk of the kernel above (4 / 16: chunk-size experiments)
#endif
constexpr int kT16Threads = 512, kT16Waves = kT16Threads / 64, kT16WaveGroups = T16_WAVE_GROUPS;
constexpr int kT16ChunkGroups32 = kT16Waves * kT16WaveGroups * 2;   // a chunk in 32-sample groups

// lane (c, q): levels l0 + {0, 1, 4, 5}, l0 = 8 (q & 1) + 2 (q >> 1)  =  decoder inputs 2 l0 + {0..3} (xa) and + 8 (xb)
__device__ __forceinline__ void encode4_t16(const char *table, const float *rs, int T, int q, const float p01[3], v4f &xa, v4f &xb)
{
    const uint32_t mask = (uint32_t)T - 1u, l0 = (uint32_t)(8 * (q & 1) + 2 * (q >> 1)), hoff = l0 * (uint32_t)T;
    uint32_t raw[32];
    float tf[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lu = (k & 1) + 4 * (k >> 1);  // level l0 + lu
        int bc[3];
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float v = p01[j] * rs[3 * lu + j];  // rs = (float)(res - 1) from level l0 on (staged in LDS with the tile's image)
                bc[j] = (int)v;
                tf[k][j] = __builtin_amdgcn_fractf(v);    // = v - (float)(int)v for the v >= 0 of a live sample, one instruction
            }
        }
        uint32_t idx[8];
        corner_indices(idx, bc[0], bc[1], bc[2], mask);
        const char *base = table + (size_t)lu * T * 4;
#pragma unroll
        for (int c = 0; c < 8; ++c) raw[8 * k + c] = *reinterpret_cast<const uint32_t *>(base + (size_t)((hoff + idx[c]) * 4u));
    }
    float x[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float w[8];
        trilinear_weights(w, tf[k][0], tf[k][1], tf[k][2]);
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            ax = fma_mix_lo(raw[8 * k + c], w[c], ax);
            ay = fma_mix_hi(raw[8 * k + c], w[c], ay);
        }
        x[2 * k] = ax;  // (lanes without a live sample decode whatever their valid-address loads returned; nothing of theirs is written)
        x[2 * k + 1] = ay;
    }
    xa = v4f{ x[0], x[1], x[2], x[3] };
    xb = v4f{ x[4], x[5], x[6], x[7] };
    SCANERF_LOAD_GUARD();
}

// the slot list of one sample -- what prepare_points (rendering_kernel.cu:391-449) would have stored for it: the first
// kMaxPtsBlocks tiles (ascending) whose (near, far) interval along the ray holds its depth; none for a stopped ray or an unset depth
__device__ __forceinline__ uint2 tracing_slots(const InferArgs &a, int i, float z)
{
    int sl[kMaxPtsBlocks] = { -1, -1, -1, -1 };
    if (a.running[i] && z != -1.0f) {
        const float2 *ci = reinterpret_cast<const float2 *>(a.inter) + (size_t)i * a.t.nb;
        int n = 0;
        for (int b = 0; b < a.t.nb; ++b) {
            const float2 bd = ci[b];
            const bool hit = z >= bd.x && z <= bd.y;
#pragma unroll
            for (int k = 0; k < kMaxPtsBlocks; ++k) sl[k] = hit && n == k ? b : sl[k];
            n += hit;
        }
    }
    return make_uint2(((uint32_t)sl[0] & 0xffffu) | ((uint32_t)sl[1] << 16), ((uint32_t)sl[2] & 0xffffu) | ((uint32_t)sl[3] << 16));
}

// SHT: the split SH operands of the chunk's rays wait in LDS (render_t16.h s16_sh_row), one row per ray of the chunk's contiguous
// ray range -- 32 rays for 4096 samples at S = 128 in the layouts 0 and 2 -- written once per chunk, read per tile; the host picks
// SHT when that range fits (t16_sh_rows_fit), otherwise every tile evaluates the harmonics of its 16 samples' directions.
constexpr int kShRows = (S16_BIAS - T16_FWD_BYTES) / 64 - 1;   // rows in the unused 12 KB of the image's footprint, less the row of zeros
__host__ __device__ inline void t16_chunk_rays(int64_t e0, int64_t e1, int B, int S, int sm, int &r0, int &r1)
{
    if (sm == 0) { r0 = (int)(e0 / S); r1 = (int)(e1 / S); }
    else if (sm == 2) { r0 = (int)((e0 >> 5) / S) * 32; r1 = (int)((e1 >> 5) / S) * 32 + 31; }
    else { r0 = 0; r1 = B - 1; }
    if (r1 > B - 1) r1 = B - 1;
}
inline bool t16_sh_rows_fit(int B, int S, int sm)
{
    constexpr int64_t kChunkSamples = kT16Waves * kT16WaveGroups * 64;
    if (sm == 1) return B <= kShRows;
    // the widest range any chunk can see: ceil(chunk / (samples per ray or ray block)) + 1 units
    const int64_t per = sm == 0 ? S : (int64_t)S * 32, units = (kChunkSamples + per - 1) / per + 1;
    return units * (sm == 0 ? 1 : 32) <= kShRows;
}
// TR (fg): no block_idxs array -- every use derives the sample's slot list (tracing_slots)
// FOLD: the images hold the activation constant in the three Gaussian layers (SCANERF_INFER_FOLDED; decode_tile_s16<.., FOLD>)
template <bool BG, bool SHT, bool TR = false, bool FOLD = false>
__global__ void __launch_bounds__(kT16Threads, 4) k_pts_inference_t16(InferArgs a)
{
    // the t16s image at its own offsets (decode_tile_s16 reads the forward sub-images and the f32 tail; the transposed narrow
    // sub-images between them are the backward's and are not staged: their 12 KB hold the chunk's SH rows)
    __shared__ __attribute__((aligned(16))) char smem[S16_BYTES + 48 * 4 + 8];
    char *const lds = smem;
    float *const rscale = reinterpret_cast<float *>(smem + S16_BYTES);
    uint32_t *const tileset = reinterpret_cast<uint32_t *>(smem + S16_BYTES + 48 * 4);
    char *const shrows = smem + T16_FWD_BYTES;
    const int lane = threadIdx.x & 63, c16 = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const int64_t total = (int64_t)a.B * a.S;
    constexpr int kChunkSamples = kT16Waves * kT16WaveGroups * 64;

    const int64_t nchunks = (total + kChunkSamples - 1) / kChunkSamples;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (threadIdx.x < 2) tileset[threadIdx.x] = 0;
        __syncthreads();  // (also: the previous chunk's last tile step is complete)
        const int64_t wbase = chunk * kChunkSamples + (int64_t)wave * (kT16WaveGroups * 64);
        int r0 = 0;
        if constexpr (SHT) {   // the chunk's rays' SH operands (visible after the barriers below, before any tile step)
            int r1;
            const int64_t e0 = chunk * kChunkSamples, e1 = e0 + kChunkSamples - 1 < total ? e0 + kChunkSamples - 1 : total - 1;
            t16_chunk_rays(e0, e1, a.B, a.S, a.sm, r0, r1);
            const int nr = r1 - r0 + 1 < kShRows ? r1 - r0 + 1 : kShRows;
            for (int r = threadIdx.x; r <= nr; r += kT16Threads) {
                if (r < nr) {
                    const float d[3] = { a.rays_d[3 * (size_t)(r0 + r)], a.rays_d[3 * (size_t)(r0 + r) + 1], a.rays_d[3 * (size_t)(r0 + r) + 2] };
                    s16_sh_row(shrows + 64 * r, d, 0.0f);
                }
            }
            if (threadIdx.x < 16) reinterpret_cast<float *>(shrows + 64 * kShRows)[threadIdx.x] = 0.0f;   // the row of the lanes with q >= 2
        }
        {   // 1. the tiles this chunk's samples list
            uint32_t mlo = 0, mhi = 0;
            auto mark = [&](int t) {
                if (t >= 0) {
                    if (t < 32) mlo |= 1u << t;
                    else mhi |= 1u << (t - 32);
                }
            };
#pragma unroll
            for (int g = 0; g < kT16WaveGroups; ++g) {
                const int64_t e = wbase + g * 64 + lane;
                if (e >= total) continue;
                const uint32_t e32 = (uint32_t)e;
                if (BG) {
                    int ri, rs;
                    pt_decompose(e32, (uint32_t)a.B, (uint32_t)a.S, a.sm, ri, rs);
                    const int tb = a.block_idxs[ri * kMaxPtsBlocks + a.step];
                    mark(tb);
                    if (tb < 0) {  // no background tile at this blend step: the sample's outputs are zero (the caller need not clear them)
                        a.out_alpha[e] = 0.0f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            a.out_dif[3 * e + c] = 0.0f;
                            a.out_spec[3 * e + c] = 0.0f;
                        }
                    }
                } else {
                    uint2 raw;
                    bool unsampled = false;
                    if constexpr (TR) {
                        int ri, rs;
                        pt_decompose(e32, (uint32_t)a.B, (uint32_t)a.S, a.sm, ri, rs);
                        raw = tracing_slots(a, ri, a.z_vals[e]);
                        // (sample_points fills a ray's depths from index 0: a first depth of -1 = a ray without samples in this pass,
                        // which the accumulation under the same flag does not read)
                        unsampled = a.skip_unsampled && a.z_vals[pt_index(ri, 0, a.B, a.S, a.sm)] == -1.0f;
                    } else {
                        raw = *reinterpret_cast<const uint2 *>(a.block_idxs + (size_t)e32 * kMaxPtsBlocks);
                    }
                    const int s0 = (int16_t)(raw.x & 0xffffu), s1 = (int16_t)(raw.x >> 16), s2 = (int16_t)(raw.y & 0xffffu),
                              s3 = (int16_t)(raw.y >> 16);
                    mark(s0);  // the list stops at the first -1 (rendering_kernel.cu:499)
                    if (s0 != -1) { mark(s1); if (s1 != -1) { mark(s2); if (s2 != -1) mark(s3); } }
                    else if (!unsampled) {   // no tile: zeros (:569-571).  Every other sample is WRITTEN by the step of its first listed tile and added
                             // to by the later ones, so the caller's arrays need no clearing pass (7.4 GB per launch at 1920x1080x128)
                        a.out_alpha[e] = 0.0f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            a.out_dif[3 * e + c] = 0.0f;
                            a.out_spec[3 * e + c] = 0.0f;
                        }
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                mlo |= __shfl_xor(mlo, off, 64);
                mhi |= __shfl_xor(mhi, off, 64);
            }
            if (lane == 0) {
                if (mlo) atomicOr(&tileset[0], mlo);
                if (mhi) atomicOr(&tileset[1], mhi);
            }
        }
        __syncthreads();
        uint64_t todo = (uint64_t)tileset[0] | ((uint64_t)tileset[1] << 32);
        while (todo) {  // 2. one step per listed tile, ascending
            const int b = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            __syncthreads();  // every wave is done with the previous image
            {
                const float4 *src = reinterpret_cast<const float4 *>(a.images + (size_t)b * WS_FLOATS + WS_S16);
                float4 *dst = reinterpret_cast<float4 *>(lds);
                for (int i = threadIdx.x; i < T16_FWD_BYTES / 16; i += kT16Threads) dst[i] = src[i];
                for (int i = S16_BIAS / 16 + threadIdx.x; i < S16_BYTES / 16; i += kT16Threads) dst[i] = src[i];
                if (threadIdx.x < 48) rscale[threadIdx.x] = (float)(a.res[(size_t)b * 48 + threadIdx.x] - 1);
            }
            __syncthreads();
            float cb[3], sb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                cb[c] = a.t.corners[3 * b + c];
                sb[c] = a.t.sizes[3 * b + c];
            }
            const char *table = (const char *)a.tables + (size_t)b * 16 * a.T * 4;
            int rso = 3 * (8 * (q & 1) + 2 * (q >> 1));
            asm volatile("" : "+v"(rso));  // (opaque: one lane-dependent base register + immediates)
            const float *rsq = rscale + rso;
#pragma unroll 1
            for (int g = 0; g < kT16WaveGroups; ++g) {
                const int64_t base = wbase + g * 64;
                if (base >= total) break;
                const int64_t e = base + lane;
                const bool in_range = e < total;
                const uint32_t ec = (uint32_t)(in_range ? e : total - 1);
                // (32-bit index arithmetic: the host keeps B*S below 2^31 for this kernel)
                int i, s;
                pt_decompose(ec, (uint32_t)a.B, (uint32_t)a.S, a.sm, i, s);
                // does this sample list tile b?  (fg: the slot list stops at the first -1, rendering_kernel.cu:499)
                int16_t slot[kMaxPtsBlocks] = { -1, -1, -1, -1 };
                bool mine = false, first = false;   // first: b is the sample's first listed tile (fg): this step writes, later ones add
                if (BG) {
                    mine = in_range && a.block_idxs[i * kMaxPtsBlocks + a.step] == b;
                } else if (in_range) {
                    const uint2 raw = TR ? tracing_slots(a, i, a.z_vals[ec]) : *reinterpret_cast<const uint2 *>(a.block_idxs + (size_t)ec * kMaxPtsBlocks);
                    slot[0] = (int16_t)(raw.x & 0xffffu); slot[1] = (int16_t)(raw.x >> 16);
                    slot[2] = (int16_t)(raw.y & 0xffffu); slot[3] = (int16_t)(raw.y >> 16);
                    first = slot[0] == b;
                    bool ended = false;
#pragma unroll
                    for (int k = 0; k < kMaxPtsBlocks; ++k) {
                        ended |= slot[k] == -1;
                        if (ended) slot[k] = -1;
                        mine |= slot[k] == b;
                    }
                }
                if (!__any(mine)) continue;  // wave-uniform
                float o[3], d[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    o[k] = a.rays_o[3 * i + k];
                    d[k] = a.rays_d[3 * i + k];
                }
                const float z = a.z_vals[ec];
                float delta;
                if (BG) delta = (s == a.S - 1) ? 10000000.0f : a.z_vals[ec + (uint32_t)pt_sample_stride(a.B, a.sm)] - z;   // :1045-1047: raw depth step
                else delta = a.dists[ec] * sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);                               // :557
                float p01[3], w_b = 0.0f, weight = 0.0f;
                bool run = mine;
                if (BG) {
                    float qq[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) qq[c] = 2.0f * ((o[c] + z * d[c]) - cb[c]) / sb[c] - 1.0f;
                    const float linf = fmaxf(fabsf(qq[0]), fmaxf(fabsf(qq[1]), fabsf(qq[2])));
                    const float ratio = (2.0f - 1.0f / linf) / linf;
#pragma unroll
                    for (int c = 0; c < 3; ++c) p01[c] = (qq[c] * ratio + 2.0f) / 4.0f;
                } else {
                    // blend weights of every listed tile (occupied or not, :523-541), this tile's cell and position
#pragma unroll
                    for (int k = 0; k < kMaxPtsBlocks; ++k) {
                        const int bk = slot[k];
                        if (bk == -1) continue;
                        float dis[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float sz = a.t.sizes[3 * bk + c];
                            const float pt = ((o[c] + z * d[c]) - a.t.corners[3 * bk + c]) / sz;
                            dis[c] = (0.5f - fabsf(pt - 0.5f)) * sz;
                        }
                        const float w = xz_weight(dis[0], dis[2]);
                        weight += w;
                        if (bk == b) w_b = w;
                    }
                    int loc[3];
                    const int l2d[3] = { a.t.log2dim[3 * b], a.t.log2dim[3 * b + 1], a.t.log2dim[3 * b + 2] };
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float pt = ((o[c] + z * d[c]) - cb[c]) / sb[c];
                        const int r = 1 << l2d[c];
                        const int cc = (int)(pt * (float)r);
                        loc[c] = cc < 0 ? 0 : (cc > r - 1 ? r - 1 : cc);
                        p01[c] = pt / 2.0f + 0.25f;  // tile -> the middle half of the 2x box (:548)
                    }
                    if (mine) run = a.t.occ[a.t.grid_starts[b] + cell_offset(loc, l2d[1], l2d[2])] != 0;
                }
                if (!BG && first && !run) {   // listed but not occupied here: the sample's value starts at zero
                    a.out_alpha[ec] = 0.0f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        a.out_dif[3 * (size_t)ec + c] = 0.0f;
                        a.out_spec[3 * (size_t)ec + c] = 0.0f;
                    }
                }
                const uint64_t rm = __ballot(run);
                if (!rm) continue;  // wave-uniform: nothing of this group is occupied
                const float inv = weight > 0 ? 1.0f / weight : 1.0f;
                int shrow = i - r0;   // this sample's ray's row (bytes); the host launches SHT only where every chunk's range fits
                shrow = 64 * (shrow < kShRows ? shrow : kShRows);
#pragma unroll 1
                for (int t = 0; t < 4; ++t) {
                    if (((rm >> (16 * t)) & 0xffffu) == 0) continue;  // wave-uniform
                    const int src = 16 * t + c16;
                    const float pt[3] = { __shfl(p01[0], src, 64), __shfl(p01[1], src, 64), __shfl(p01[2], src, 64) };
                    const bool act = (rm >> src) & 1u;
                    v4f xa, xb;
                    encode4_t16(table, rsq, a.T, q, pt, xa, xb);
                    const uint32_t es = (uint32_t)__shfl((int)ec, src, 64);
                    const float dl = __shfl(delta, src, 64);
                    // no live sample of the tile with a non-zero opacity -> the directional layers are skipped (decode_tile_s16)
                    // (the opacity is evaluated ONCE, here, from the sigma the decoder hands the gate, and kept for the outputs below:
                    // the same expression gave the same bits twice, at ~20 vector instructions per tile for the second one)
                    float pa_keep = 0.0f;
                    auto gate = [&](float sigma) {
                        // (folded form: the exponential on v_exp_f32, ~1 ulp of e -- 1.2e-7 absolute on the opacity instead of 6e-8 --
                        // in place of the library's expf, ~15 vector instructions per tile)
                        pa_keep = FOLD ? 1.0f - __builtin_amdgcn_exp2f(-1.4426950408889634f * (sigma * dl)) : 1.0f - expf(-1.0f * sigma * dl);
                        return __any(act && q == 0 && pa_keep != 0.0f) != 0;
                    };
                    SampleOut so;
                    if constexpr (SHT) {
                        const int row = __shfl(shrow, src, 64);
                        so = decode_tile_s16<true, decltype(gate), FOLD>(lds, lane, xa, xb, nullptr, 0.0f, shrows + (q < 2 ? row + 16 * q : 64 * kShRows), gate);
                    } else {
                        const float dd[3] = { __shfl(d[0], src, 64), __shfl(d[1], src, 64), __shfl(d[2], src, 64) };
                        so = decode_tile_s16<false, decltype(gate), FOLD>(lds, lane, xa, xb, dd, 0.0f, nullptr, gate);
                    }
                    if (BG) {
                        if (act && q == 0) {
                            const float pa = pa_keep;
                            a.out_alpha[es] = pa;
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                a.out_dif[3 * (size_t)es + c] = pa * so.dif[c];
                                a.out_spec[3 * (size_t)es + c] = pa * (so.tint[c] * so.spec[c]);
                            }
                        }
                    } else {
                        const float wb = __shfl(w_b, src, 64), iv = __shfl(inv, src, 64);
                        const bool fst = __shfl((int)first, src, 64) != 0;
                        if (act && q == 0) {
                            const float pa = pa_keep;
                            // (0 + x == x exactly: writing x where the cleared array held 0 gives the bits the += gave)
                            a.out_alpha[es] = (fst ? 0.0f : a.out_alpha[es]) + (wb * pa) * iv;
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                a.out_dif[3 * (size_t)es + c] = (fst ? 0.0f : a.out_dif[3 * (size_t)es + c]) + (wb * pa * so.dif[c]) * iv;
                                a.out_spec[3 * (size_t)es + c] = (fst ? 0.0f : a.out_spec[3 * (size_t)es + c]) + (wb * pa * (so.tint[c] * so.spec[c])) * iv;
                            }
                        }
                    }
                    SCANERF_STORE_GUARD();
                }
            }
        }
    }
}

// Decoder arithmetic of the inference entry points: flag bits OR-ed into their `sample_major` argument (scanerf_hip.h):
// none = the 16-sample-tile kernel at four waves per SIMD (default); SCANERF_INFER_H3 = the 32-sample-tile kernel at two
// (round 4's); SCANERF_INFER_F32 = the f32-MFMA single-pass kernel (exact f32; comparison / debugging).
// (+ 4: SCANERF_INFER_FOLDED, the images carry the activation constant: the 16-sample-tile kernel only)
inline int infer_arith_of(int &sample_major)
{
    const int ar = ((sample_major & SCANERF_INFER_F32) ? 2 : ((sample_major & SCANERF_INFER_H3) ? 1 : 0)) | ((sample_major & SCANERF_INFER_FOLDED) ? 4 : 0);
    sample_major &= ~(SCANERF_INFER_F32 | SCANERF_INFER_H3 | SCANERF_INFER_FOLDED);
    return ar;
}
inline bool render_single_pass(int64_t total, int nb, int arith)
{
    // the chunk-major kernel indexes samples in 32 bits and keeps a chunk's tile set in 64 bits
    return (arith & 3) == 2 || total >= ((int64_t)1 << 31) || nb > 64;
}

// SCANERF_INFER_H3 in sample_major: the 32-sample-tile kernel at two waves per SIMD (k_pts_inference_chunks; comparison) instead of the
// 16-sample-tile one at four (k_pts_inference_t16, default)

// SCANERF_RENDER_PIPE=0 (experiments build): the group loop without the software pipeline (comparison; the two give the same bits)
inline bool render_pipelined() { return tune_int("SCANERF_RENDER_PIPE", 1) != 0; }
inline bool render_t16_tiles(int arith) { return (arith & 3) == 0 && tune_int("SCANERF_RENDER_H3_WAVES", 0) == 0; }
// SCANERF_RENDER_H3_WAVES=3 / 4 (experiments build): the 32-sample-tile kernel without the software pipeline at three / four waves per SIMD
inline int render_h3_waves() { return tune_int("SCANERF_RENDER_H3_WAVES", 0); }
template <bool BG>
inline void launch_chunks(const InferArgs &a, int64_t tiles32, int arith, hipStream_t stream)
{
    // One workgroup per chunk (up to 2^20).  A foreground chunk's cost is anything between nothing (rays that miss) and 4096 decoded
    // samples; with 8 workgroups per CU walking ~30 chunks each at a fixed stride the busiest workgroup had ~1.6x the mean share
    // of live chunks and the launch waited for it.  The dispatcher hands a finished workgroup's slot to the next chunk instead
    // (same box, ms per frame: 8 per CU 71.2, 64 per CU 69.2, one per chunk 67.5; the background launch gains its tail too).
    const int cap_cu = tune_int("SCANERF_RENDER_GRID_CAP", 0);   // workgroups per CU (comparison; experiments build)
    const int64_t cap = cap_cu > 0 ? (int64_t)kNumCU * cap_cu : (int64_t)1 << 20;
    auto nblocks = [&](int64_t per_chunk) {
        const int64_t nchunks = (tiles32 + per_chunk - 1) / per_chunk;
        return (int)(nchunks < cap ? nchunks : cap);
    };
    const int w = render_h3_waves();
    (void)w;
    if (render_t16_tiles(arith)) {
        // SCANERF_RENDER_SH_ROWS=0 (experiments build): every tile evaluates its samples' harmonics (comparison; the same bits)
        const bool rows = t16_sh_rows_fit(a.B, a.S, a.sm) && tune_int("SCANERF_RENDER_SH_ROWS", 1) != 0;
        const dim3 grid(nblocks(kT16ChunkGroups32));
        const bool fold = (arith & 4) != 0;
        if constexpr (!BG) {
            if (a.running) {   // scanerf_pts_inference_tracing
                if (rows && fold) hipLaunchKernelGGL((k_pts_inference_t16<false, true, true, true>), grid, dim3(kT16Threads), 0, stream, a);
                else if (rows) hipLaunchKernelGGL((k_pts_inference_t16<false, true, true>), grid, dim3(kT16Threads), 0, stream, a);
                else if (fold) hipLaunchKernelGGL((k_pts_inference_t16<false, false, true, true>), grid, dim3(kT16Threads), 0, stream, a);
                else hipLaunchKernelGGL((k_pts_inference_t16<false, false, true>), grid, dim3(kT16Threads), 0, stream, a);
                return;
            }
        }
        if (rows && fold) hipLaunchKernelGGL((k_pts_inference_t16<BG, true, false, true>), grid, dim3(kT16Threads), 0, stream, a);
        else if (rows) hipLaunchKernelGGL((k_pts_inference_t16<BG, true>), grid, dim3(kT16Threads), 0, stream, a);
        else if (fold) hipLaunchKernelGGL((k_pts_inference_t16<BG, false, false, true>), grid, dim3(kT16Threads), 0, stream, a);
        else hipLaunchKernelGGL((k_pts_inference_t16<BG, false>), grid, dim3(kT16Threads), 0, stream, a);
    }
#ifdef RT_H3_WAVES_EXPERIMENT   // (tools/build_variant.py render_time="-ffp-contract=off -DRT_H3_WAVES_EXPERIMENT": 23 / 70 spilled registers;
    // 85.0 / 91.2 ms per frame against 89.8 for the pipelined form and 82.8 for the first 16-sample-tile kernel on the same box)
    else if (w == 3) hipLaunchKernelGGL((k_pts_inference_chunks<BG, false, 768, 3, RT_W3_GROUPS>), dim3(nblocks(12 * RT_W3_GROUPS)), dim3(768), 0, stream, a);
    else if (w == 4) hipLaunchKernelGGL((k_pts_inference_chunks<BG, false, 512, 4, 16>), dim3(nblocks(8 * 16)), dim3(512), 0, stream, a);
#endif
#ifdef SCANERF_EXPERIMENTS
    else if (!render_pipelined()) hipLaunchKernelGGL((k_pts_inference_chunks<BG, false>), dim3(nblocks(kChunkWaves * kChunkWaveGroups)), dim3(kChunkThreads), 0, stream, a);
#endif
    else hipLaunchKernelGGL((k_pts_inference_chunks<BG, true>), dim3(nblocks(kChunkWaves * kChunkWaveGroups)), dim3(kChunkThreads), 0, stream, a);
}

// ---- rendering_kernel.cu:624-702: front-to-back accumulation, one wave per ray ------------------------
__global__ void __launch_bounds__(256) k_accumulate_color(const float *__restrict__ pts_dif,
                                                          const float *__restrict__ pts_spec,
                                                          const float *__restrict__ pts_alpha, float *__restrict__ transp,
                                                          const float *__restrict__ z_vals, float *__restrict__ dif,
                                                          float *__restrict__ spec, float *__restrict__ depth, int B, int S, int skip)
{
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * (blockDim.x >> 6);
    for (int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < B; i += nw) {
        float T = transp[i];
        if (T < 0.00001f) continue;  // wave-uniform
        if (skip && z_vals[(size_t)i * S] == -1.0f) continue;   // SCANERF_SKIP_UNSAMPLED: a ray without samples in this pass
        float acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + lane;
            const bool live = s < S;
            const size_t e = (size_t)i * S + (live ? s : 0);
            const float al = live ? pts_alpha[e] : 0.0f;
            float incl = 1.0f - al;  // T_k = T * prod_{j<k} (1 - alpha_j)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                float t = __shfl_up(incl, off, 64);
                if (lane >= off) incl *= t;
            }
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float Tk = T * excl;
            if (live) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    acc[c] += Tk * pts_dif[3 * e + c];
                    acc[3 + c] += Tk * pts_spec[3 * e + c];
                }
                acc[6] += Tk * al * z_vals[e];
            }
            T *= __shfl(incl, 63, 64);
        }
#pragma unroll
        for (int c = 0; c < 7; ++c)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
        if (lane == 0) {
            transp[i] = T;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dif[3 * i + c] += acc[c];
                spec[3 * i + c] += acc[3 + c];
            }
            depth[i] += acc[6];
        }
    }
}

// the same over layouts 1 and 2: one LANE per ray, its samples in order (neighbouring lanes = neighbouring rays read neighbouring words)
__global__ void __launch_bounds__(256) k_accumulate_color_sm(const float *__restrict__ pts_dif, const float *__restrict__ pts_spec,
                                                             const float *__restrict__ pts_alpha, float *__restrict__ transp,
                                                             const float *__restrict__ z_vals, float *__restrict__ dif,
                                                             float *__restrict__ spec, float *__restrict__ depth, int B, int S, int lay,
                                                             int skip)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        float T = transp[i];
        if (T < 0.00001f) continue;
        if (skip && z_vals[pt_index(i, 0, B, S, lay)] == -1.0f) continue;   // SCANERF_SKIP_UNSAMPLED: a ray without samples in this pass
        float acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
        for (int s = 0; s < S; ++s) {
            const size_t e = pt_index(i, s, B, S, lay);
            const float al = pts_alpha[e];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[c] += T * pts_dif[3 * e + c];
                acc[3 + c] += T * pts_spec[3 * e + c];
            }
            acc[6] += T * al * z_vals[e];
            T *= 1.0f - al;
        }
        transp[i] = T;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dif[3 * i + c] += acc[c];
            spec[3 * i + c] += acc[3 + c];
        }
        depth[i] += acc[6];
    }
}

// ---- rendering_kernel.cu:816-868 -------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_render_inverse_z(const float *__restrict__ inter, const int16_t *__restrict__ related,
                                                          int S, int nb, float range, float *__restrict__ z_vals, int B, int sm)
{
    // a thread keeps ONE ray and walks its depths: the ray's constants (three divisions, the interval load) once instead of per
    // sample, no index division.  Threads of a wave hold neighbouring rays (layouts 1 and 2: one depth of 64 / 32 rays is
    // contiguous) or, for the reference's [B,S], the 64 lanes of a wave share a ray and take every 64th depth.
    if (sm == 0) {
        const int lane = threadIdx.x & 63;
        const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
        for (int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < B; i += nw) {
            const int b = related[i];
            if (b == -1) continue;
            const float2 bd = reinterpret_cast<const float2 *>(inter)[(size_t)i * nb + b];
            if (bd.x == kInf) continue;
            const float near_ = bd.y, far_ = near_ + range;
            const float inv_near = 1.0f / near_, inv_far = 1.0f / far_, inv_bound = inv_far - inv_near;
            const float stp = 1.0f / (float)(S - 1);
            for (int k = lane; k < S; k += 64) z_vals[(size_t)i * S + k] = 1.0f / (stp * (float)k * inv_bound + inv_near);
        }
        return;
    }
    // layouts 1 / 2: thread = (ray i, depth phase); a ray's depths k = phase, phase + P, ...
    const int P = 4;   // depth phases per ray: 4 x B threads' worth of parallelism
    const int64_t total = (int64_t)B * P;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        // consecutive threads = consecutive rays of one 32-ray block (layout 2) / of the batch (layout 1); the phase is the slow index
        const int i = (int)(t % B), ph = (int)(t / B);
        const int b = related[i];
        if (b == -1) continue;
        const float2 bd = reinterpret_cast<const float2 *>(inter)[(size_t)i * nb + b];
        if (bd.x == kInf) continue;
        const float near_ = bd.y, far_ = near_ + range;
        const float inv_near = 1.0f / near_, inv_far = 1.0f / far_, inv_bound = inv_far - inv_near;
        const float stp = 1.0f / (float)(S - 1);
        float *col = z_vals + pt_index(i, 0, B, S, sm);
        const size_t ks = pt_sample_stride(B, sm);
        for (int k = ph; k < S; k += P) col[(size_t)k * ks] = 1.0f / (stp * (float)k * inv_bound + inv_near);
    }
}

// ---- rendering_kernel.cu:1263-1401 -------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_update_outgoing(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                         Tiles t, const int32_t *__restrict__ tracing_blocks,
                                                         const float *__restrict__ inter, int16_t *__restrict__ out_bidx,
                                                         float *__restrict__ blend, int skip, int B)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        const int32_t *tb = tracing_blocks + (size_t)i * t.nb;
        const float2 *ci = reinterpret_cast<const float2 *>(inter) + (size_t)i * t.nb;
        float far_ = -1.0f;
        int index = 0;
        int outb[kMaxPtsBlocks] = { -1, -1, -1, -1 };
        for (int k = 0; k < t.nb; ++k) {
            const int b = tb[k];
            const float2 bd = ci[b];
            if (bd.x == kInf) break;
            if (!skip && (bd.x > far_ && far_ != -1.0f)) break;
            if (bd.y > far_) {
                far_ = bd.y;
                outb[0] = b; outb[1] = outb[2] = outb[3] = -1;
                index = 1;
            } else if (bd.y == far_) {
                if (index < kMaxPtsBlocks) {  // the reference writes unchecked
                    if (index == 1) outb[1] = b; else if (index == 2) outb[2] = b; else outb[3] = b;
                    ++index;
                }
            }
        }
        if (far_ == -1.0f) continue;
        if (index == 1) {
            blend[i * kMaxPtsBlocks] = 1.0f;
            out_bidx[i * kMaxPtsBlocks] = (int16_t)outb[0];
            continue;
        }
#pragma unroll
        for (int k = 0; k < kMaxPtsBlocks; ++k) {
            const int b = outb[k];
            if (b == -1) break;
            float dis[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float pw = rays_o[3 * i + c] + far_ * rays_d[3 * i + c];
                float p = (pw - t.corners[3 * b + c]) / t.sizes[3 * b + c];
                p = p < 0.0f ? 0.0f : (p > 1.0f ? 1.0f : p);
                dis[c] = (0.5f - fabsf(p - 0.5f)) * t.sizes[3 * b + c];
            }
            blend[i * kMaxPtsBlocks + k] = xz_weight(dis[0], dis[2]);
            out_bidx[i * kMaxPtsBlocks + k] = (int16_t)b;
        }
    }
}

// ---- rendering_kernel.cu:1406-1447 -----------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_update_outgoing_v2(const float *__restrict__ rays_o, Tiles t,
                                                            int16_t *__restrict__ inside, float *__restrict__ blend, int B)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        int index = 0;
        for (int b = 0; b < t.nb && index < kMaxPtsBlocks; ++b) {
            float loc[3], dis[3];
            bool in = true;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                loc[c] = (rays_o[3 * i + c] - t.corners[3 * b + c]) / t.sizes[3 * b + c];
                in = in && (loc[c] >= 0 && loc[c] <= 1);
                dis[c] = (0.5f - fabsf(loc[c] - 0.5f)) * t.sizes[3 * b + c];
            }
            if (in) {
                inside[i * kMaxPtsBlocks + index] = (int16_t)b;
                blend[i * kMaxPtsBlocks + index] = dis[0] * dis[1] * dis[2];
                ++index;
            }
        }
    }
}

// ---- rendering_kernel.cu:1212-1260 -------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_get_last_block(const int32_t *__restrict__ tracing_blocks, int32_t *__restrict__ bidxs,
                                                        const float *__restrict__ inter, int nb, int B)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        int idx = -1;
        for (int k = 0; k < nb; ++k) {
            const int b = tracing_blocks[(size_t)i * nb + k];
            if (inter[2 * ((size_t)i * nb + b)] == kInf) break;
            idx = b;
        }
        bidxs[i] = idx;
    }
}

// ---- rendering_kernel.cu:705-813: first tile along the ray whose occupancy the ray touches ----------------
__global__ void __launch_bounds__(64) k_ray_firsthit_block(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                           Tiles t, const int32_t *__restrict__ tracing_blocks,
                                                           const float *__restrict__ inter, int16_t *__restrict__ hit, int B)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        const float o[3] = { rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2] };
        const float d[3] = { rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2] };
        float dis = 10000000.0f;
        int last = -1;
        for (int k = 0; k < t.nb; ++k) {
            const int b = tracing_blocks[(size_t)i * t.nb + k];
            const float2 bd = reinterpret_cast<const float2 *>(inter)[(size_t)i * t.nb + b];
            if (bd.x == kInf) break;
            const int l2d[3] = { t.log2dim[3 * b], t.log2dim[3 * b + 1], t.log2dim[3 * b + 2] };
            int side[3];
            float cs[3], og[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                side[c] = 1 << l2d[c];
                cs[c] = t.sizes[3 * b + c] / (float)side[c];
                og[c] = o[c] - t.corners[3 * b + c];
            }
            const uint8_t *g = t.occ + t.grid_starts[b];
            F2 ts;
            ts.x = bd.x;
            ts.y = bd.y;
            Walker w;
            w.start(og, d, ts, side, cs);
            bool found = false;
            while (!w.done()) {
                w.pick();
                if (g[cell_offset(w.cell, l2d[1], l2d[2])]) { found = true; break; }
                w.advance();
            }
            if (found && dis > bd.y) {
                hit[i] = (int16_t)b;
                dis = bd.y;
            }
            last = b;
        }
        if (last != -1 && hit[i] == -1) hit[i] = (int16_t)last;
    }
}

// ---- rendering_kernel.cu:1479-1564: dilate tile `bidx`'s occupancy into the tiles it overlaps ------------------
__global__ void __launch_bounds__(256) k_process_occupied_grid(int bidx, Tiles t, uint8_t *__restrict__ tgt, int total_grid)
{
    const int l0[3] = { t.log2dim[3 * bidx], t.log2dim[3 * bidx + 1], t.log2dim[3 * bidx + 2] };
    const int r0[3] = { 1 << l0[0], 1 << l0[1], 1 << l0[2] };
    float gs[3], c0[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gs[c] = t.sizes[3 * bidx + c] / (float)r0[c];
        c0[c] = t.corners[3 * bidx + c];
    }
    const uint8_t *g = t.occ + t.grid_starts[bidx];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total_grid; e += gridDim.x * blockDim.x) {
        if (!g[e]) continue;
        const int x = e / (r0[1] * r0[2]);
        const int y = (e - x * (r0[1] * r0[2])) / r0[2];
        const int z = (e - x * (r0[1] * r0[2])) % r0[2];
        const float pts[3] = { (float)x * gs[0] + c0[0], (float)y * gs[1] + c0[1], (float)z * gs[2] + c0[2] };
        for (int b = 0; b < t.nb; ++b) {
            if (b == bidx) continue;
            const int l2d[3] = { t.log2dim[3 * b], t.log2dim[3 * b + 1], t.log2dim[3 * b + 2] };
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // vertex order of the reference: 000 001 010 100 011 101 110 111
                const int vx = (j == 3 || j == 5 || j == 6 || j == 7), vy = (j == 2 || j == 4 || j == 6 || j == 7),
                          vz = (j == 1 || j == 4 || j == 5 || j == 7);
                const float v[3] = { (float)vx, (float)vy, (float)vz };
                float p[3];
                bool in = true;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    p[c] = (pts[c] + v[c] * gs[c] - t.corners[3 * b + c]) / t.sizes[3 * b + c];
                    in = in && (p[c] >= 0 && p[c] < 1);
                }
                if (in) {
                    int ijk[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) ijk[c] = (int)(p[c] * (float)(1 << l2d[c]));
                    tgt[t.grid_starts[b] + cell_offset(ijk, l2d[1], l2d[2])] = 1;
                }
            }
        }
    }
}

Tiles make_tiles(const float *corners, const float *sizes, const uint8_t *occ, const int64_t *gst, const int32_t *l2d, int nb)
{
    Tiles t;
    t.corners = corners; t.sizes = sizes; t.occ = occ; t.grid_starts = gst; t.log2dim = l2d; t.nb = nb;
    return t;
}

}  // namespace

// ---------------------------------------------------------------------------- C ABI
#define RT_REQ(cond, name) SCANERF_REQUIRE(cond, name ": bad argument (null pointer or negative size)")

SCANERF_API int scanerf_ray_block_intersection(const float *rays_o, const float *rays_d, const float *corners,
                                               const float *sizes, float *inter, int B, int nb, scanerf_stream_t stream)
{
    RT_REQ(B >= 0 && nb >= 1, "ray_block_intersection");
    if (B == 0) return 0;
    RT_REQ(rays_o && rays_d && corners && sizes && inter, "ray_block_intersection");
    hipLaunchKernelGGL(k_ray_block_intersection, dim3(stream_grid((int64_t)B * nb, 256)), dim3(256), 0, (hipStream_t)stream,
                       rays_o, rays_d, make_tiles(corners, sizes, nullptr, nullptr, nullptr, nb), inter, B);
    return check_launch("ray_block_intersection");
}

SCANERF_API int scanerf_render_sample_points(const float *rays_o, const float *rays_d, const float *corners,
                                             const float *sizes, const uint8_t *occ, const int64_t *grid_starts,
                                             const int32_t *log2dim, const int32_t *tracing_blocks, const float *inter,
                                             int32_t *tracing_idx, float *z_start, float *z_vals, float *dists, int B, int S,
                                             int nb, int sample_major, scanerf_stream_t stream)
{
    RT_REQ(B >= 0 && S >= 1 && nb >= 1, "sample_points");
    SCANERF_REQUIRE(sample_major >= 0 && sample_major <= 2 && (sample_major != 2 || B % 32 == 0),
                    "sample_points" ": sample_major=%d (0, 1, or 2 with B a multiple of 32; B=%d)", sample_major, B);
    if (B == 0) return 0;
    RT_REQ(rays_o && rays_d && corners && sizes && occ && grid_starts && log2dim && tracing_blocks && inter && tracing_idx &&
               z_start && z_vals && dists, "sample_points");
    hipLaunchKernelGGL(k_render_sample_points, dim3(stream_grid(B, 64, kNumCU * 64)), dim3(64), 0, (hipStream_t)stream, rays_o,
                       rays_d, make_tiles(corners, sizes, occ, grid_starts, log2dim, nb), S, tracing_blocks, inter, tracing_idx,
                       z_start, z_vals, dists, B, sample_major);
    return check_launch("sample_points");
}

SCANERF_API int scanerf_prepare_points(const float *z_vals, const uint8_t *running_mask, const float *inter,
                                       int16_t *block_idxs, int B, int S, int nb, int sample_major, scanerf_stream_t stream)
{
    RT_REQ(B >= 0 && S >= 1 && nb >= 1, "prepare_points");
    SCANERF_REQUIRE(sample_major >= 0 && sample_major <= 2 && (sample_major != 2 || B % 32 == 0),
                    "prepare_points" ": sample_major=%d (0, 1, or 2 with B a multiple of 32; B=%d)", sample_major, B);
    if (B == 0) return 0;
    RT_REQ(z_vals && running_mask && inter && block_idxs, "prepare_points");
    hipLaunchKernelGGL(k_prepare_points, dim3(stream_grid((int64_t)B * S, 256)), dim3(256), 0, (hipStream_t)stream, z_vals,
                       running_mask, block_idxs, inter, S, nb, B, sample_major);
    return check_launch("prepare_points");
}

// images: [nb][scanerf_render_workspace_floats()] decoders packed by scanerf_pack_decoder with weight_feature == 1
constexpr int kTracingMaxTiles = 8;
static int pts_inference_impl(const float *rays_o, const float *rays_d, const float *z_vals, const float *dists,
                              const int16_t *block_idxs, const uint8_t *running, const float *inter, const void *tables_f16, const float *images,
                                      const int32_t *res, const uint8_t *occ, const int64_t *grid_starts,
                                      const int32_t *log2dim, const float *corners, const float *sizes, float *out_dif,
                                      float *out_spec, float *out_alpha, int B, int S, int T, int nb, int sample_major,
                                      scanerf_stream_t stream)
{
    RT_REQ(B >= 0 && S >= 1 && nb >= 1, "pts_inference");
    const int skip_unsampled = !block_idxs && (sample_major & SCANERF_SKIP_UNSAMPLED);   // (the tracing entry point only)
    sample_major &= ~SCANERF_SKIP_UNSAMPLED;
    const int arith = infer_arith_of(sample_major);
    SCANERF_REQUIRE(sample_major >= 0 && sample_major <= 2 && (sample_major != 2 || B % 32 == 0),
                    "pts_inference" ": sample_major=%d (0, 1, or 2 with B a multiple of 32; B=%d)", sample_major, B);
    SCANERF_REQUIRE(T >= 2 && (T & (T - 1)) == 0, "pts_inference: T=%d must be a power of two", T);
    if (B == 0) return 0;
    RT_REQ(rays_o && rays_d && z_vals && dists && (block_idxs || (running && inter)) && tables_f16 && images && res && occ && grid_starts && log2dim &&
               corners && sizes && out_dif && out_spec && out_alpha, "pts_inference");
    InferArgs a;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z_vals = z_vals; a.dists = dists; a.block_idxs = block_idxs;
    a.tables = tables_f16; a.images = images; a.res = res; a.t = make_tiles(corners, sizes, occ, grid_starts, log2dim, nb);
    a.out_dif = out_dif; a.out_spec = out_spec; a.out_alpha = out_alpha; a.T = T; a.B = B; a.S = S; a.step = 0;
    a.sm = sample_major; a.running = block_idxs ? nullptr : running; a.inter = inter; a.skip_unsampled = skip_unsampled;
    a.dbg = tune_int("SCANERF_DEBUG_RT", 0);
    const int64_t tiles32 = ((int64_t)B * S + 31) / 32;
    int blocks = (int)((tiles32 + 3) / 4 < kNumCU * 4 ? (tiles32 + 3) / 4 : kNumCU * 4);
    SCANERF_REQUIRE(!(arith & 4) || (render_t16_tiles(arith) && !render_single_pass((int64_t)B * S, nb, arith)),
                    "pts_inference: SCANERF_INFER_FOLDED images are for the 16-sample-tile kernel only (nb <= 64, no SCANERF_INFER_H3 / _F32)");
    SCANERF_REQUIRE(!sample_major || !render_single_pass((int64_t)B * S, nb, arith), "pts_inference: sample-major arrays need the chunk kernel");
    SCANERF_REQUIRE(block_idxs || (render_t16_tiles(arith) && !render_single_pass((int64_t)B * S, nb, arith) && nb <= kTracingMaxTiles),
                    "pts_inference_tracing: needs the 16-sample-tile kernel and nb <= %d tiles (nb=%d); use prepare_points + pts_inference", kTracingMaxTiles, nb);
    if (render_single_pass((int64_t)B * S, nb, arith)) {
        hipLaunchKernelGGL((k_pts_inference<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        return check_launch("pts_inference");
    }
    // every sample is written (zeros where no tile applies, :569-571).  The 32-sample-tile kernels add into cleared arrays; the
    // 16-sample-tile kernel writes each sample at its first listed tile's step and needs no clearing pass
    if (!render_t16_tiles(arith) || tune_int("SCANERF_RENDER_CLEAR", 0) == 1) {   // (=1: clear anyway: timing comparison, experiments build)
        const size_t n = (size_t)B * S;
        const hipError_t ce[3] = { hipMemsetAsync(out_dif, 0, n * 12, (hipStream_t)stream), hipMemsetAsync(out_spec, 0, n * 12, (hipStream_t)stream),
                                   hipMemsetAsync(out_alpha, 0, n * 4, (hipStream_t)stream) };
        for (hipError_t e : ce) SCANERF_REQUIRE(e == hipSuccess, "pts_inference: clearing the outputs failed: %s", hipGetErrorString(e));
    }
    launch_chunks<false>(a, tiles32, arith, (hipStream_t)stream);
    return check_launch("pts_inference");
}

SCANERF_API int scanerf_pts_inference(const float *rays_o, const float *rays_d, const float *z_vals, const float *dists,
                                      const int16_t *block_idxs, const void *tables_f16, const float *images,
                                      const int32_t *res, const uint8_t *occ, const int64_t *grid_starts,
                                      const int32_t *log2dim, const float *corners, const float *sizes, float *out_dif,
                                      float *out_spec, float *out_alpha, int B, int S, int T, int nb, int sample_major,
                                      scanerf_stream_t stream)
{
    SCANERF_REQUIRE(block_idxs, "pts_inference: null block_idxs");
    return pts_inference_impl(rays_o, rays_d, z_vals, dists, block_idxs, nullptr, nullptr, tables_f16, images, res, occ, grid_starts, log2dim,
                              corners, sizes, out_dif, out_spec, out_alpha, B, S, T, nb, sample_major, stream);
}

// prepare_points + pts_inference in one launch: the slot lists (8 bytes per sample written, then read once per tile step) never
// exist; every use re-derives them from the ray's intervals (nb comparisons per sample, nb <= 8).  Same values as the two ops.
SCANERF_API int scanerf_pts_inference_tracing(const float *rays_o, const float *rays_d, const float *z_vals, const float *dists,
                                              const uint8_t *running_mask, const float *intersections, const void *tables_f16,
                                              const float *images, const int32_t *res, const uint8_t *occ, const int64_t *grid_starts,
                                              const int32_t *log2dim, const float *corners, const float *sizes, float *out_dif,
                                              float *out_spec, float *out_alpha, int B, int S, int T, int nb, int sample_major,
                                              scanerf_stream_t stream)
{
    SCANERF_REQUIRE(running_mask && intersections, "pts_inference_tracing: null running_mask / intersections");
    return pts_inference_impl(rays_o, rays_d, z_vals, dists, nullptr, running_mask, intersections, tables_f16, images, res, occ, grid_starts,
                              log2dim, corners, sizes, out_dif, out_spec, out_alpha, B, S, T, nb, sample_major, stream);
}

SCANERF_API int scanerf_bg_pts_inference_v2(const float *rays_o, const float *rays_d, const float *z_vals,
                                            const int16_t *bg_idxs, int step, const float *corners, const float *sizes,
                                            const int32_t *res, const void *tables_f16, const float *images, float *out_dif,
                                            float *out_spec, float *out_alpha, int B, int S, int T, int nb, int sample_major,
                                            scanerf_stream_t stream)
{
    RT_REQ(B >= 0 && S >= 1 && nb >= 1 && step >= 0 && step < kMaxPtsBlocks, "bg_pts_inference_v2");
    const int arith = infer_arith_of(sample_major);
    SCANERF_REQUIRE(sample_major >= 0 && sample_major <= 2 && (sample_major != 2 || B % 32 == 0),
                    "bg_pts_inference_v2" ": sample_major=%d (0, 1, or 2 with B a multiple of 32; B=%d)", sample_major, B);
    SCANERF_REQUIRE(T >= 2 && (T & (T - 1)) == 0, "bg_pts_inference_v2: T=%d must be a power of two", T);
    if (B == 0) return 0;
    RT_REQ(rays_o && rays_d && z_vals && bg_idxs && tables_f16 && images && res && corners && sizes && out_dif && out_spec &&
               out_alpha, "bg_pts_inference_v2");
    InferArgs a;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z_vals = z_vals; a.dists = nullptr; a.block_idxs = bg_idxs;
    a.tables = tables_f16; a.images = images; a.res = res; a.t = make_tiles(corners, sizes, nullptr, nullptr, nullptr, nb);
    a.out_dif = out_dif; a.out_spec = out_spec; a.out_alpha = out_alpha; a.T = T; a.B = B; a.S = S; a.step = step;
    a.sm = sample_major; a.running = nullptr; a.inter = nullptr; a.skip_unsampled = 0;
    a.dbg = tune_int("SCANERF_DEBUG_RT", 0);
    const int64_t tiles32 = ((int64_t)B * S + 31) / 32;
    int blocks = (int)((tiles32 + 3) / 4 < kNumCU * 4 ? (tiles32 + 3) / 4 : kNumCU * 4);
    SCANERF_REQUIRE(!(arith & 4) || (render_t16_tiles(arith) && !render_single_pass((int64_t)B * S, nb, arith)),
                    "bg_pts_inference_v2: SCANERF_INFER_FOLDED images are for the 16-sample-tile kernel only (nb <= 64, no SCANERF_INFER_H3 / _F32)");
    SCANERF_REQUIRE(!sample_major || !render_single_pass((int64_t)B * S, nb, arith), "bg_pts_inference_v2: sample-major arrays need the chunk kernel");
    if (render_single_pass((int64_t)B * S, nb, arith)) {
        hipLaunchKernelGGL((k_pts_inference<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        return check_launch("bg_pts_inference_v2");
    }
    launch_chunks<true>(a, tiles32, arith, (hipStream_t)stream);
    return check_launch("bg_pts_inference_v2");
}

SCANERF_API int scanerf_accumulate_color(const float *pts_dif, const float *pts_spec, const float *pts_alpha, float *transp,
                                         const float *z_vals, float *dif, float *spec, float *depth, int B, int S,
                                         int sample_major, scanerf_stream_t stream)
{
    RT_REQ(B >= 0 && S >= 1, "accumulate_color");
    const int skip = (sample_major & SCANERF_SKIP_UNSAMPLED) != 0;
    sample_major &= ~SCANERF_SKIP_UNSAMPLED;
    SCANERF_REQUIRE(sample_major >= 0 && sample_major <= 2 && (sample_major != 2 || B % 32 == 0),
                    "accumulate_color" ": sample_major=%d (0, 1, or 2 with B a multiple of 32; B=%d)", sample_major, B);
    if (B == 0) return 0;
    RT_REQ(pts_dif && pts_spec && pts_alpha && transp && z_vals && dif && spec && depth, "accumulate_color");
    if (sample_major)
        hipLaunchKernelGGL(k_accumulate_color_sm, dim3(stream_grid(B, 256)), dim3(256), 0, (hipStream_t)stream, pts_dif, pts_spec,
                           pts_alpha, transp, z_vals, dif, spec, depth, B, S, sample_major, skip);
    else
        hipLaunchKernelGGL(k_accumulate_color, dim3(stream_grid((int64_t)B * 64, 256)), dim3(256), 0, (hipStream_t)stream, pts_dif,
                           pts_spec, pts_alpha, transp, z_vals, dif, spec, depth, B, S, skip);
    return check_launch("accumulate_color");
}

SCANERF_API int scanerf_render_inverse_z_sampling(const float *inter, const int16_t *related_bidx, float *z_vals,
                                                  float sample_range, int B, int S, int nb, int sample_major,
                                                  scanerf_stream_t stream)
{
    RT_REQ(B >= 0 && S >= 2 && nb >= 1, "inverse_z_sampling");
    SCANERF_REQUIRE(sample_major >= 0 && sample_major <= 2 && (sample_major != 2 || B % 32 == 0),
                    "inverse_z_sampling" ": sample_major=%d (0, 1, or 2 with B a multiple of 32; B=%d)", sample_major, B);
    if (B == 0) return 0;
    RT_REQ(inter && related_bidx && z_vals, "inverse_z_sampling");
    hipLaunchKernelGGL(k_render_inverse_z, dim3(stream_grid(sample_major == 0 ? (int64_t)B * 64 : (int64_t)B * 4, 256)), dim3(256), 0,
                       (hipStream_t)stream, inter, related_bidx, S, nb, sample_range, z_vals, B, sample_major);
    return check_launch("inverse_z_sampling");
}

SCANERF_API int scanerf_update_outgoing_bidx(const float *rays_o, const float *rays_d, const float *corners,
                                             const float *sizes, const int32_t *tracing_blocks, const float *inter,
                                             int16_t *out_bidx, float *blend, float ratio, int skip, int B, int nb,
                                             scanerf_stream_t stream)
{
    (void)ratio;  // unused by the reference kernel as well (its only use is commented out, :1323-1331)
    RT_REQ(B >= 0 && nb >= 1, "update_outgoing_bidx");
    if (B == 0) return 0;
    RT_REQ(rays_o && rays_d && corners && sizes && tracing_blocks && inter && out_bidx && blend, "update_outgoing_bidx");
    hipLaunchKernelGGL(k_update_outgoing, dim3(stream_grid(B, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d,
                       make_tiles(corners, sizes, nullptr, nullptr, nullptr, nb), tracing_blocks, inter, out_bidx, blend, skip, B);
    return check_launch("update_outgoing_bidx");
}

SCANERF_API int scanerf_update_outgoing_bidx_v2(const float *rays_o, const float *corners, const float *sizes,
                                                int16_t *inside_bidx, float *blend, int B, int nb, scanerf_stream_t stream)
{
    RT_REQ(B >= 0 && nb >= 1, "update_outgoing_bidx_v2");
    if (B == 0) return 0;
    RT_REQ(rays_o && corners && sizes && inside_bidx && blend, "update_outgoing_bidx_v2");
    hipLaunchKernelGGL(k_update_outgoing_v2, dim3(stream_grid(B, 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                       make_tiles(corners, sizes, nullptr, nullptr, nullptr, nb), inside_bidx, blend, B);
    return check_launch("update_outgoing_bidx_v2");
}

// Tile order per ray = torch.argsort(intersections[..., 0], dim=-1, stable=True) (rendering.py:301 sorts the tiles a ray
// meets by their entry distance; misses hold 1e7): one thread per ray, stable insertion sort of its nb <= 64 entries.
namespace {
__global__ void __launch_bounds__(256) k_sort_tracing_blocks(const float *__restrict__ inter, int32_t *__restrict__ order, int nb, int B)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        const float2 *ci = reinterpret_cast<const float2 *>(inter) + (size_t)i * nb;
        int32_t *out = order + (size_t)i * nb;
        for (int b = 0; b < nb; ++b) {
            const float key = ci[b].x;
            int pos = b;
            while (pos > 0 && ci[out[pos - 1]].x > key) {   // strictly greater: equal keys keep their index order
                out[pos] = out[pos - 1];
                --pos;
            }
            out[pos] = b;
        }
    }
}
}  // namespace

SCANERF_API int scanerf_sort_tracing_blocks(const float *inter, int32_t *order, int B, int nb, scanerf_stream_t stream)
{
    RT_REQ(B >= 0 && nb >= 1 && nb <= 64, "sort_tracing_blocks");
    if (B == 0) return 0;
    RT_REQ(inter && order, "sort_tracing_blocks");
    hipLaunchKernelGGL(k_sort_tracing_blocks, dim3(stream_grid(B, 256)), dim3(256), 0, (hipStream_t)stream, inter, order, nb, B);
    return check_launch("sort_tracing_blocks");
}

SCANERF_API int scanerf_get_last_block(const int32_t *tracing_blocks, int32_t *bidxs, const float *inter, int B, int nb,
                                       scanerf_stream_t stream)
{
    RT_REQ(B >= 0 && nb >= 1, "get_last_block");
    if (B == 0) return 0;
    RT_REQ(tracing_blocks && bidxs && inter, "get_last_block");
    hipLaunchKernelGGL(k_get_last_block, dim3(stream_grid(B, 256)), dim3(256), 0, (hipStream_t)stream, tracing_blocks, bidxs,
                       inter, nb, B);
    return check_launch("get_last_block");
}

SCANERF_API int scanerf_ray_firsthit_block(const float *rays_o, const float *rays_d, const float *corners, const float *sizes,
                                           const uint8_t *occ, const int64_t *grid_starts, const int32_t *log2dim,
                                           const int32_t *tracing_blocks, const float *inter, int16_t *hit, int B, int nb,
                                           scanerf_stream_t stream)
{
    RT_REQ(B >= 0 && nb >= 1, "ray_firsthit_block");
    if (B == 0) return 0;
    RT_REQ(rays_o && rays_d && corners && sizes && occ && grid_starts && log2dim && tracing_blocks && inter && hit,
           "ray_firsthit_block");
    hipLaunchKernelGGL(k_ray_firsthit_block, dim3(stream_grid(B, 64, kNumCU * 64)), dim3(64), 0, (hipStream_t)stream, rays_o,
                       rays_d, make_tiles(corners, sizes, occ, grid_starts, log2dim, nb), tracing_blocks, inter, hit, B);
    return check_launch("ray_firsthit_block");
}

SCANERF_API int scanerf_process_occupied_grid(int bidx, int total_grid, const float *corners, const float *sizes,
                                              const uint8_t *occ, const int64_t *grid_starts, const int32_t *log2dim,
                                              uint8_t *tgt_occ, int nb, scanerf_stream_t stream)
{
    RT_REQ(total_grid >= 0 && nb >= 1 && bidx >= 0 && bidx < nb, "process_occupied_grid");
    if (total_grid == 0) return 0;
    RT_REQ(corners && sizes && occ && grid_starts && log2dim && tgt_occ, "process_occupied_grid");
    hipLaunchKernelGGL(k_process_occupied_grid, dim3(stream_grid(total_grid, 256)), dim3(256), 0, (hipStream_t)stream, bidx,
                       make_tiles(corners, sizes, occ, grid_starts, log2dim, nb), tgt_occ, total_grid);
    return check_launch("process_occupied_grid");
}
