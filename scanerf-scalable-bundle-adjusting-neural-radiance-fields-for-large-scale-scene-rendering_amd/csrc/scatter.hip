// scatter.hip -- table-gradient scatter of the hash encoder WITHOUT global atomics (gfx950).
//
// What it replaces: hashgrid/src/hashgrid_bg_kernel.cu:196-201 -- 16 atomicAdd per (point,
// level) into grad_features.  On MI355X device-scope float atomics execute at the memory
// side (the 8 XCD L2s are not coherent): scattered 4-byte adds run at ~2e10/s chip-wide, so
// the 2.1e9 adds of one 65 536 x 128 batch take ~110 ms (measured, profiles/).
//
// Design: a radix partition by table bucket, then LDS accumulation.
//   * The hash  idx = x ^ y*P1 ^ z*P2  keeps the x bits in place: the two x-neighbours of a
//     (y,z) corner pair land in the same aligned 2048-entry bucket (idx>>11 depends on x only
//     through bits >= 11).  One 16-byte record carries both: {local0 | local1<<16, tx, gx, gy}
//     with (gx,gy) = w_yz * dL/dout; entry0 += (1-tx) g, entry1 += tx g.
//   * pass 1 (count):   per workgroup LDS histogram over the L*NB bins of its sample range.
//   * scan:             exclusive offsets per (bin, workgroup); bins are contiguous in HBM.
//   * pass 2 (scatter): same walk, LDS cursors, one 16-B store per record.  A workgroup's
//                       records of one bin are contiguous, so lines fill up in L2.
//   * accumulate:       one workgroup per bin streams its records (coalesced 16-B loads) and adds
//                       them into a 64-bit fixed-point LDS image of the bucket (integer LDS
//                       atomics; see k_bin_accumulate), then adds the image to grad_features
//                       with plain stores.  No global atomics anywhere; bit-reproducible.
// HBM traffic: 64 B written + 64 B read per (point, level) instead of 16 memory-side atomics.
#include "adam_common.h"
#include "render_device.h"
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include "scatter_common.h"

using namespace scanerf;

namespace {

constexpr int kThreads = 256;
constexpr int kRun = 8;  // consecutive points per lane in the stand-alone count / scatter walks
// i over [lo, hi): thread t takes points lo + (j*blockDim.x + t)*kRun + k, k < kRun
#define SCANERF_RUN_WALK(i, lo, hi)                                                     \
    for (int c_ = (lo) + threadIdx.x * kRun; c_ < (hi); c_ += (int)blockDim.x * kRun)    \
        for (int i = c_; i < c_ + kRun && i < (hi); ++i)

// Where the points of a stand-alone scatter come from: an array of contracted points with its gradient rows (the binding
// surface), or -- RAYS -- the samples of up to two render branches over the same B rays (a tile's foreground and background,
// tile.py:639-692): point i < N1 is sample i % S[0] of ray i / S[0] of branch 0, the rest belong to branch 1; positions are formed
// as the render kernels form them (contract_point_box), gradients are the backward's level-major dfeat [16][B * S[k]][2] of each
// branch.  Round 6: replaces torch's construction of the contracted points (a dozen elementwise launches) and the
// concatenation of both branches' points and gradients (0.5 GB copied per iteration at T = 2^24, 16 384 rays).
struct PointSrc {
    const float *rays_o, *rays_d;
    const float *z[2];
    const uint8_t *valid[2];   // per ray, may be null
    const float2 *grad[2];
    int S[2], mode[2], N1;
    float min_bbox[3], bbox_size[3];
};
// point i of a ray source: position and branch / index inside the branch; false = its ray is masked out (no records)
__device__ __forceinline__ bool src_point(const PointSrc &s, int i, float p[3], int &br, int &j)
{
    br = i >= s.N1 ? 1 : 0;
    j = br ? i - s.N1 : i;
    const int ray = j / s.S[br];
    if (s.valid[br] && !s.valid[br][ray]) return false;
    const float o[3] = { s.rays_o[3 * ray], s.rays_o[3 * ray + 1], s.rays_o[3 * ray + 2] };
    const float d[3] = { s.rays_d[3 * ray], s.rays_d[3 * ray + 1], s.rays_d[3 * ray + 2] };
    contract_point_box(s.min_bbox, s.bbox_size, s.mode[br], o, d, s.z[br][j], p);
    return true;
}

__global__ void __launch_bounds__(256) k_src_points(PointSrc src, float *__restrict__ pts, int N)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        float p[3];
        int br, j;
        if (!src_point(src, i, p, br, j)) p[0] = p[1] = p[2] = __builtin_nanf("");
        pts[3 * (size_t)i] = p[0];
        pts[3 * (size_t)i + 1] = p[1];
        pts[3 * (size_t)i + 2] = p[2];
    }
}

// ---- pass 1: count ---------------------------------------------------------------------
// PER_LEVEL: the LDS holds one level's NB counters at a time (large tables: L*NB counters do not fit); the points are
// walked once per level (re-read from L2).
// MASKED: points of rays that are masked out carry a NaN x coordinate (k_src_points) and leave no records.
template <bool PER_LEVEL, bool MASKED = false>
__global__ void __launch_bounds__(1024) k_bin_count(const float *__restrict__ points,
                                                        const int32_t *__restrict__ resolutions, BinGeom g,
                                                        uint32_t *__restrict__ counts, uint32_t *__restrict__ maxbits,
                                                        uint32_t *__restrict__ overflow)
{
    extern __shared__ uint32_t hist[];  // [L*NB], or [NB] if PER_LEVEL
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *maxbits = 0;
        *overflow = 0;
    }
    if (PER_LEVEL) {
        const uint32_t mask = (uint32_t)g.T - 1u;
        const int lo = blockIdx.x * g.per_wg, hi = min(g.N, lo + g.per_wg);
        for (int l = 0; l < g.L; ++l) {
            for (int i = threadIdx.x; i < g.NB; i += (int)blockDim.x) hist[i] = 0;
            __syncthreads();
            // Coarse levels: lane-owned runs of consecutive points (neighbouring samples of a ray share cells there and would meet
            // in one histogram word from neighbouring lanes: see below); from level 6 on the cells are smaller than a ray's sample
            // spacing and a lane-interleaved walk reads the points coalesced (T = 2^24, 4.2e6 points: 0.39 -> 0.32 ms; all
            // levels interleaved 0.37, from level 8 on 0.33).  Counts do not depend on the order.
            if (l >= 6) {
                for (int i = lo + threadIdx.x; i < hi; i += (int)blockDim.x) {
                    const float p[3] = { points[3 * i], points[3 * i + 1], points[3 * i + 2] };
                    Pairs pr;
                    make_pairs(p, resolutions + 3 * l, mask, pr);
                    if (!MASKED || p[0] == p[0]) count_pairs(pr, hist, g.bucket_log);
                }
            } else
            SCANERF_RUN_WALK(i, lo, hi) {
                const float p[3] = { points[3 * i], points[3 * i + 1], points[3 * i + 2] };
                Pairs pr;
                make_pairs(p, resolutions + 3 * l, mask, pr);   // (a NaN position gives finite indices: the conversions saturate)
                if (!MASKED || p[0] == p[0]) count_pairs(pr, hist, g.bucket_log);
            }
            __syncthreads();
            // (format 3: the range of this (bucket, workgroup) is counted in 64-byte segments of kSegRecs records)
            for (int i = threadIdx.x; i < g.NB; i += (int)blockDim.x)
                counts[((size_t)l * g.NB + i) * g.W + blockIdx.x] = g.rec8 == 3 ? (hist[i] + kSegRecs - 1) / kSegRecs : hist[i];
            __syncthreads();
        }
        return;
    }
    const int nbins = g.L * g.NB;
    for (int i = threadIdx.x; i < nbins; i += (int)blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint32_t mask = (uint32_t)g.T - 1u;
    const int lo = blockIdx.x * g.per_wg, hi = min(g.N, lo + g.per_wg);
    // lane-owned runs of kRun consecutive points: neighbouring points (samples of one ray) share cells at the coarse levels
    // and would hit the same LDS word from neighbouring lanes (serialised: tools/lds_atomic_bench.hip); a wave still walks a
    // contiguous window of 64*kRun points, so its loads reuse the lines they touch
    SCANERF_RUN_WALK(i, lo, hi) {
        const float p[3] = { points[3 * i], points[3 * i + 1], points[3 * i + 2] };
        for (int l = 0; l < g.L; ++l) {
            Pairs pr;
            make_pairs(p, resolutions + 3 * l, mask, pr);
            count_pairs(pr, hist + l * g.NB, g.bucket_log);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += (int)blockDim.x) counts[(size_t)i * g.W + blockIdx.x] = hist[i];
}

// ---- scan: counts[bin][wg] -> in-row exclusive prefix (in place) + bin totals ------------
__global__ void __launch_bounds__(kThreads) k_bin_rowscan(uint32_t *__restrict__ counts, uint32_t *__restrict__ totals,
                                                          int W)
{
    __shared__ uint32_t part[kThreads];
    uint32_t *row = counts + (size_t)blockIdx.x * W;
    const int per = (W + kThreads - 1) / kThreads;
    const int lo = threadIdx.x * per, hi = min(W, lo + per);
    uint32_t s = 0;
    for (int i = lo; i < hi; ++i) s += row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    // exclusive scan of 256 partials (Hillis-Steele in LDS)
    for (int off = 1; off < kThreads; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        uint32_t c = row[i];
        row[i] = run;
        run += c;
    }
    if (threadIdx.x == kThreads - 1) totals[blockIdx.x] = part[kThreads - 1];
}

// exclusive scan of the bin totals (one workgroup; nbins <= a few 100k) -> starts[nbins+1]
__global__ void __launch_bounds__(1024) k_bin_starts(const uint32_t *__restrict__ totals, uint32_t *__restrict__ starts,
                                                     int nbins)
{
    __shared__ uint32_t part[1024];
    const int per = (nbins + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(nbins, lo + per);
    uint32_t s = 0;
    for (int i = lo; i < hi; ++i) s += totals[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        starts[i] = run;
        run += totals[i];
    }
    if (threadIdx.x == 1023) starts[nbins] = part[1023];
}

// ---- pass 2: scatter records ---------------------------------------------------------------
// LEVEL_MAJOR_GRAD: grad_in is [L][N][2] (two-kernel render path) instead of [N][L][2].
// REC: record format (scatter_common.h): 0 = Rec (16 bytes), 1 = Rec8 (gradients that come out of the t16 backward's f16 products
// anyway), 2 = Rec12 (f32-grade in 12 bytes: behind the t16s backward)
template <bool LEVEL_MAJOR_GRAD, bool PER_LEVEL = false, int REC = 0>
__global__ void __launch_bounds__(1024) k_bin_scatter(const float *__restrict__ points,
                                                          const float2 *__restrict__ grad_in,
                                                          const int32_t *__restrict__ resolutions, BinGeom g,
                                                          const uint32_t *__restrict__ rowprefix,
                                                          const uint32_t *__restrict__ starts, Rec *__restrict__ recs,
                                                          float *__restrict__ grad_features,
                                                          uint32_t *__restrict__ maxbits)
{
    extern __shared__ uint32_t cursor[];  // [L*NB], or one level's [NB] at a time if PER_LEVEL
    float gmax = 0.0f;
    const int nbins = g.L * g.NB;
    if (!PER_LEVEL) {
        for (int i = threadIdx.x; i < nbins; i += (int)blockDim.x)
            cursor[i] = starts[i] + rowprefix[(size_t)i * g.W + blockIdx.x];
        __syncthreads();
    }
    const uint32_t mask = (uint32_t)g.T - 1u;
    const int lo = blockIdx.x * g.per_wg, hi = min(g.N, lo + g.per_wg);
    // One (sample, level): 4 records.
    auto one_g = [&](int l, const float p[3], const float2 gi) {
        gmax = fmaxf(gmax, fmaxf(fabsf(gi.x), fabsf(gi.y)));
        Pairs pr;
        make_pairs(p, resolutions + 3 * l, mask, pr);
        if (REC == 1)
            emit_pairs8(pr, gi.x, gi.y, cursor + l * g.NB, g.bucket_log, rec_capacity(g.capacity, 1), recs, grad_features + (size_t)l * g.T * 2);
        else if (REC == 2)
            emit_pairs12(pr, gi.x, gi.y, cursor + l * g.NB, g.bucket_log, rec_capacity(g.capacity, 2), recs, grad_features + (size_t)l * g.T * 2);
        else
            emit_pairs(pr, gi.x, gi.y, cursor + l * g.NB, g.bucket_log, g.capacity, recs, grad_features + (size_t)l * g.T * 2);
    };
    auto one = [&](int i, int l, const float p[3]) {
        const float2 gi = LEVEL_MAJOR_GRAD ? grad_in[(size_t)l * g.N + i] : grad_in[(size_t)i * g.L + l];
        gmax = fmaxf(gmax, fmaxf(fabsf(gi.x), fabsf(gi.y)));
        Pairs pr;
        make_pairs(p, resolutions + 3 * l, mask, pr);
        if (REC == 1)
            emit_pairs8(pr, gi.x, gi.y, PER_LEVEL ? cursor : cursor + l * g.NB, g.bucket_log, rec_capacity(g.capacity, 1), recs,
                        grad_features + (size_t)l * g.T * 2);
        else if (REC == 2)
            emit_pairs12(pr, gi.x, gi.y, PER_LEVEL ? cursor : cursor + l * g.NB, g.bucket_log, rec_capacity(g.capacity, 2), recs,
                         grad_features + (size_t)l * g.T * 2);
        else
            emit_pairs(pr, gi.x, gi.y, PER_LEVEL ? cursor : cursor + l * g.NB, g.bucket_log, g.capacity, recs,
                       grad_features + (size_t)l * g.T * 2);
    };
    if (PER_LEVEL) {
        for (int l = 0; l < g.L; ++l) {
            for (int i = threadIdx.x; i < g.NB; i += (int)blockDim.x) {
                const int bin = l * g.NB + i;
                cursor[i] = starts[bin] + rowprefix[(size_t)bin * g.W + blockIdx.x];
            }
            __syncthreads();
            for (int i = lo + threadIdx.x; i < hi; i += (int)blockDim.x) {
                const float p[3] = { points[3 * i], points[3 * i + 1], points[3 * i + 2] };
                one(i, l, p);
            }
            __syncthreads();
        }
    } else if (LEVEL_MAJOR_GRAD) {
        // level-major walk: a workgroup appends to only NB bins at a time, so the partially written
        // lines of its ranges (one per bin) stay in L2 until complete (full-line write-backs); the
        // gradient reads are contiguous per level and the points are re-read from L2.
        // (lane-interleaved on purpose: with lane-owned runs as in k_bin_count the cursor conflicts go away but the 8-byte
        // gradient and 12-byte point loads and the record stores lose their coalescing -- measured 4.7 -> 7.6 ms)
        for (int l = 0; l < g.L; ++l)
            for (int i = lo + threadIdx.x; i < hi; i += (int)blockDim.x) {
                const float p[3] = { points[3 * i], points[3 * i + 1], points[3 * i + 2] };
                one(i, l, p);
            }
    } else if (g.rows16) {
        // Point-major gradients [N][16][2] (the binding surface: what autograd hands embedding_bg_backward_cuda), round 5.
        // A thread reads its point's row ONCE (128 contiguous bytes, eight 16-byte loads: the wave consumes whole lines) and
        // keeps it in registers; the workgroup then emits LEVEL BY LEVEL, in step (one barrier per level): while it is on a
        // level it appends to that level's NB ranges only, blockDim * 4 records of them, so the lines of a range fill up
        // within one pass instead of being revisited 16 levels later (counters at 65 536 x 128 samples, 2048-entry buckets,
        // 1024 workgroups of 256 threads, a gradient load per level: 2.8e8 fabric write requests for 8.6 GB of records -- twice
        // what full lines need --, 8.4 GB fetched for a 1 GB gradient, 4.1e8 fabric requests in 7.7 ms = the 54 G/s ceiling
        // of DESIGN.md 4.11).
        for (int i0 = lo; i0 < hi; i0 += (int)blockDim.x) {
            const int i = i0 + (int)threadIdx.x;
            const bool live = i < hi;
            float p[3] = { 0.0f, 0.0f, 0.0f };
            float4 gr[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) gr[k] = make_float4(0, 0, 0, 0);
            if (live) {
                p[0] = points[3 * (size_t)i]; p[1] = points[3 * (size_t)i + 1]; p[2] = points[3 * (size_t)i + 2];
                const float4 *row = reinterpret_cast<const float4 *>(grad_in + (size_t)i * 16);
#pragma unroll
                for (int k = 0; k < 8; ++k) gr[k] = row[k];
            }
#pragma unroll
            for (int l = 0; l < 16; ++l) {
                if (live) one_g(l, p, (l & 1) ? make_float2(gr[l >> 1].z, gr[l >> 1].w) : make_float2(gr[l >> 1].x, gr[l >> 1].y));
                __syncthreads();
            }
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += (int)blockDim.x) {
            const float p[3] = { points[3 * i], points[3 * i + 1], points[3 * i + 2] };
            for (int l = 0; l < g.L; ++l) one(i, l, p);
        }
    }
    // launch-wide max |dL/dout| for the fixed-point scale of the accumulate pass
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
    if ((threadIdx.x & 63) == 0 && gmax > 0.0f) atomicMax(maxbits, __float_as_uint(gmax));
}

// ---- pass 2 for LARGE tables (round 6): records leave the CU as full 64-byte segments -------------------------------------
// k_bin_scatter<.., PER_LEVEL> stores every 12-byte record on its own into one of NB = 2048 open ranges per workgroup: 256
// workgroups x 2048 ranges x 128-byte lines = 67 MB of partially written lines against 32 MB of L2 that turn over every ~13 us,
// and a range receives two records per batch -- the lines reach HBM a few records at a time (T = 2^24, 4.2e6 samples: 6.3 GB
// written for 3.2 GB of records, 2.55 ms; sub-passes and fewer workgroups did not help).  Lines fill only with
// temporal density, so the density is created where nothing is evicted: the workgroup keeps ONE 60-byte slot per bucket in its
// LDS (2048 x 60 B = 120 KB), a record goes to slot position (ordinal mod 5) of its bucket, and a slot that holds its five
// records is written as one aligned 64-byte store to the range's next segment (format 3 of scatter_common.h).
//   per batch of 1024 points (one level at a time): reserve ordinals (LDS atomics on cnt[bucket]) | barrier | records of the
//   slot's segment go into it, records of segments the batch completes on its own are stored directly | barrier | slots whose
//   segment is complete are flushed and move on | barrier | records of the new current segment go into the slot (see place()).
//   A first version repeated "fill the slot, flush" until no record waited: the coarse levels put ~100 records of a batch into
//   one bucket = 20 rounds, 7.0 ms per launch against the old producer's 2.4.
//   end of level: the partly filled slots are flushed with zero words in the unused positions.
// A range's ordinal -> address map is exact (counted by k_bin_count in segments), so results do not depend on timing beyond
// the order of records inside a bucket, which the integer accumulate does not see.
// PTS: points per lane and batch (a batch = 1024 * PTS points: the three barriers are paid per batch).
template <bool LEVEL_MAJOR_GRAD, bool RAYS, int PTS>
__global__ void __launch_bounds__(1024) k_bin_scatter_seg(const float *__restrict__ points, const float2 *__restrict__ grad_in,
                                                          const int32_t *__restrict__ resolutions, BinGeom g,
                                                          const uint32_t *__restrict__ rowprefix, const uint32_t *__restrict__ starts,
                                                          Rec *__restrict__ recs, float *__restrict__ grad_features,
                                                          uint32_t *__restrict__ maxbits, PointSrc src)
{
    extern __shared__ uint32_t lds[];
    constexpr int R = 4 * PTS;   // records per lane and batch
    const int NB = g.NB;
    uint32_t *cnt = lds, *segn = lds + NB, *gbase = lds + 2 * NB, *slot = lds + 3 * NB;   // slot[NB][15]
    const uint32_t cap_segs = rec_capacity(g.capacity, 3);
    uint4 *segs = reinterpret_cast<uint4 *>(recs);
    const uint32_t mask = (uint32_t)g.T - 1u, lmask = (1u << g.bucket_log) - 1u;
    const int lo = blockIdx.x * g.per_wg, hi = min(g.N, lo + g.per_wg);
    float gmax = 0.0f;
    for (int l = 0; l < g.L; ++l) {
        for (int i = threadIdx.x; i < NB; i += 1024) {
            const int bin = l * NB + i;
            cnt[i] = 0;
            segn[i] = 0;
            gbase[i] = starts[bin] + rowprefix[(size_t)bin * g.W + blockIdx.x];
        }
        float *grad_level = grad_features + (size_t)l * g.T * 2;
        auto to_overflow_table = [&](uint32_t b, uint32_t w0, uint32_t w1, uint32_t w2) {
            *overflow_flag(recs) = 1u;
            float *gsb = grad_level + ((size_t)b << g.bucket_log) * 2;
            const Rec12Fields f = unpack_rec12(w0, w1, w2);
            unsafeAtomicAdd(gsb + 2 * f.l0, (1.0f - f.w1) * f.gx);
            unsafeAtomicAdd(gsb + 2 * f.l0 + 1, (1.0f - f.w1) * f.gy);
            if (f.l1 <= lmask) {
                unsafeAtomicAdd(gsb + 2 * f.l1, f.w1 * f.gx);
                unsafeAtomicAdd(gsb + 2 * f.l1 + 1, f.w1 * f.gy);
            }
        };
        // n (<= 5) records of bucket b's slot -> segment segn[b] of the range; beyond the workspace: the overflow table (atomics)
        auto flush = [&](int b, uint32_t n) {
            const uint32_t *sl = slot + b * 15;
            const uint32_t gs = gbase[b] + segn[b];
            if (gs < cap_segs) {
                uint32_t w[16];
#pragma unroll
                for (int j = 0; j < 15; ++j) w[j] = (uint32_t)j < 3u * n ? sl[j] : 0u;
                w[15] = 0u;
                uint4 *d = segs + (size_t)gs * 4;
                d[0] = make_uint4(w[0], w[1], w[2], w[3]);
                d[1] = make_uint4(w[4], w[5], w[6], w[7]);
                d[2] = make_uint4(w[8], w[9], w[10], w[11]);
                d[3] = make_uint4(w[12], w[13], w[14], w[15]);
            } else {
                for (uint32_t r = 0; r < n; ++r) to_overflow_table((uint32_t)b, sl[3 * r], sl[3 * r + 1], sl[3 * r + 2]);
            }
        };
        // one record straight to its final place (a segment that this batch completes without the slot)
        auto direct = [&](uint32_t b, uint32_t sg, uint32_t within, const uint32_t (&w)[3]) {
            const uint32_t gs = gbase[b] + sg;
            if (gs < cap_segs) {
                struct __attribute__((aligned(4))) W3 { uint32_t a, b, c; };
                *reinterpret_cast<W3 *>(reinterpret_cast<uint32_t *>(segs + (size_t)gs * 4) + within * 3u) = W3{ w[0], w[1], w[2] };
            } else {
                to_overflow_table(b, w[0], w[1], w[2]);
            }
        };
        // A batch's records of one bucket have the ordinals [c0, c1).  With cur = the slot's segment (c0 / 5) and full = c1 / 5:
        //   segment cur           -> the slot (it is flushed below once full > cur)
        //   cur < segment < full  -> completed by this batch alone: its five records are stored directly, 12 bytes each, by
        //                            their lanes at the same moment (coarse levels put a hundred samples of a batch into one
        //                            bucket: those lines fill in L2 without help)
        //   segment full (> cur)  -> the slot's NEXT segment: written after the flush
        // Three barriers per batch, whatever the distribution.
        auto place = [&](const uint32_t (&bk)[R], const uint32_t (&rw)[R][3], uint32_t have) {   // have: bit r = record r exists
            uint32_t pos[R];
            uint32_t late = 0u;
#pragma unroll
            for (int r = 0; r < R; ++r) pos[r] = (have >> r) & 1u ? atomicAdd(&cnt[bk[r]], 1u) : 0u;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!((have >> r) & 1u)) continue;
                const uint32_t b = bk[r], sg = pos[r] / (uint32_t)kSegRecs, within = pos[r] - sg * kSegRecs;
                const uint32_t cur = segn[b], full = cnt[b] / (uint32_t)kSegRecs;
                if (sg == cur) {
                    uint32_t *d = slot + b * 15u + within * 3u;
                    d[0] = rw[r][0];
                    d[1] = rw[r][1];
                    d[2] = rw[r][2];
                } else if (sg < full) {
#ifdef SCANERF_EXPERIMENTS
                    if (g.dbg & 4) { if (rw[r][0] == 0x12345u) segs[0] = make_uint4(rw[r][0], rw[r][1], rw[r][2], 0u); } else
#endif
                    direct(b, sg, within, rw[r]);
                } else {
                    late |= 1u << r;
                }
            }
            __syncthreads();
            // four neighbouring lanes write one segment, 16 bytes each: a store instruction covers 16 whole 64-byte segments
            // (one lane per segment and four stores each = 64 separate 16-byte requests per instruction)
            for (int b = threadIdx.x >> 2; b < NB; b += 256) {
                const uint32_t full = cnt[b] / (uint32_t)kSegRecs, cur = segn[b], qd = threadIdx.x & 3u;
                if (full > cur) {
                    const uint32_t *sl = slot + b * 15 + qd * 4;
                    const uint4 w = make_uint4(sl[0], sl[1], sl[2], qd == 3u ? 0u : sl[3]);   // (slot[b][15] does not exist: the spare word)
                    const uint32_t gs = gbase[b] + cur;
#ifdef SCANERF_EXPERIMENTS
                    if (g.dbg & 4) { if (w.x == 0x12345u) segs[0] = w; } else
#endif
                    if (gs < cap_segs) segs[(size_t)gs * 4 + qd] = w;
                    else if (qd == 0u) flush(b, (uint32_t)kSegRecs);   // (beyond the workspace: the overflow table)
                    if (qd == 0u) segn[b] = full;   // (the four lanes read `cur` in the same instruction above)
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r)
                if ((late >> r) & 1u) {
                    uint32_t *d = slot + bk[r] * 15u + (pos[r] % (uint32_t)kSegRecs) * 3u;
                    d[0] = rw[r][0];
                    d[1] = rw[r][1];
                    d[2] = rw[r][2];
                }
        };
        __syncthreads();
        for (int i0 = lo; i0 < hi; i0 += 1024 * PTS) {
            uint32_t bk[R], rw[R][3];
            uint32_t have = 0u, strad = 0u;
            Pairs pr[PTS];
            float2 gi[PTS];
#pragma unroll
            for (int u = 0; u < PTS; ++u) {
                const int i = i0 + u * 1024 + (int)threadIdx.x;   // (lane-interleaved: a wave's loads of one u are contiguous)
                gi[u] = make_float2(0.0f, 0.0f);
#pragma unroll
                for (int q = 0; q < 4; ++q) bk[4 * u + q] = 0u;
                if (i >= hi) continue;
                const float p[3] = { points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2] };
                const bool masked = RAYS && p[0] != p[0];   // (positions by k_src_points: NaN = the ray is masked out)
                if (RAYS) {   // (gradients from the branch's own dfeat)
                    gi[u] = i >= src.N1 ? src.grad[1][(size_t)l * (g.N - src.N1) + (i - src.N1)] : src.grad[0][(size_t)l * src.N1 + i];
                } else {
                    gi[u] = LEVEL_MAJOR_GRAD ? grad_in[(size_t)l * g.N + i] : grad_in[(size_t)i * g.L + l];
                }
                if (!masked) gmax = fmaxf(gmax, fmaxf(fabsf(gi[u].x), fabsf(gi[u].y)));
                make_pairs(p, resolutions + 3 * l, mask, pr[u]);
                const bool straddle = (pr[u].xm >> g.bucket_log) != 0u;
                const uint32_t k = straddle ? 15u : (uint32_t)(31 - __clz((int)pr[u].xm));
                const uint32_t t = straddle ? 0u : (uint32_t)min(__float2int_rn(pr[u].tx * 8388608.0f), 8388607);
                const float a0 = 1.0f - pr[u].tx;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gx = pr[u].wyz[q] * gi[u].x, gy = pr[u].wyz[q] * gi[u].y;
                    bk[4 * u + q] = pr[u].idx0[q] >> g.bucket_log;
                    rw[4 * u + q][0] = (pr[u].idx0[q] & lmask) | (k << 13) | ((t >> 8) << 17);
                    rw[4 * u + q][1] = rec12_round(straddle ? a0 * gx : gx) | ((t >> 4) & 15u);
                    rw[4 * u + q][2] = rec12_round(straddle ? a0 * gy : gy) | (t & 15u);
                }
                if (!masked) {
                    have |= 15u << (4 * u);
                    if (straddle) strad |= 15u << (4 * u);
                }
            }
#ifdef SCANERF_EXPERIMENTS
            if (g.dbg & 8) { if (have == 0x12345u) segs[0] = make_uint4(rw[0][0], rw[1][1], rw[2][2], bk[3]); continue; }
#endif
            place(bk, rw, have);
            // x-neighbours in different buckets (only where a level's resolution exceeds the bucket size): the second entries
            if (__syncthreads_or(strad != 0u)) {
#pragma unroll
                for (int u = 0; u < PTS; ++u) {
                    if (!((strad >> (4 * u)) & 1u)) continue;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t i1 = pr[u].idx0[q] ^ pr[u].xm;
                        const float gx = pr[u].wyz[q] * gi[u].x, gy = pr[u].wyz[q] * gi[u].y;
                        bk[4 * u + q] = i1 >> g.bucket_log;
                        rw[4 * u + q][0] = (i1 & lmask) | (15u << 13);
                        rw[4 * u + q][1] = rec12_round(pr[u].tx * gx);
                        rw[4 * u + q][2] = rec12_round(pr[u].tx * gy);
                    }
                }
                place(bk, rw, strad);
            }
        }
        // (after place() every slot holds the 0..4 records of its range's last, incomplete segment)
        __syncthreads();
        for (int b = threadIdx.x; b < NB; b += 1024) {
            const uint32_t n = cnt[b] - (uint32_t)kSegRecs * segn[b];
            if (n) flush(b, n);
        }
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
    if ((threadIdx.x & 63) == 0 && gmax > 0.0f) atomicMax(maxbits, __float_as_uint(gmax));
}

// ---- accumulate: one workgroup per bin -------------------------------------------------------
// One workgroup per bin.  The bucket image is accumulated in 64-bit FIXED POINT with integer
// LDS atomics: on gfx950 ds_add_f32 is ~12x slower than ds_add_u32/u64 (measured: 2.1e9 float
// adds 9.5 ms, the same adds as u64 1.7 ms).  With M = max|dL/dout| of the launch (found by the
// scatter pass) and n records in the bin, every partial sum is < n*M, so values are scaled by
// 2^k, k = min(51, 62 - ceil(log2(n+1))) - ceil(log2 M): no overflow, resolution M*2^-51 .. M*n*2^-62.  Integer
// addition is associative, so the table gradient is bit-reproducible run to run (the
// reference's atomics are not) and closer to the exact sum than an fp32 running sum.
// ADAM: the epilogue applies the fused sparse Adam (adam_common.h: the IEEE sequence of adam.hip / the oracle) to the bucket's
// entries instead of adding the image to grad_features.  The image IS the list of touched entries, so the gradient table, its
// zero-fill and the optimiser's scan of it disappear (T = 2^24: 0.4 + 1.4 ms of a 7.7 ms step).  Optionally the half-precision
// gather table is refreshed in the same pass.  ad.overflow_grad (may be null): the table the backward's workspace-overflow
// path adds to -- read (and re-zeroed) only if the overflow flag is set.
struct AdamEpilogue {
    // a SECOND record set accumulated into the same image before the epilogue (the background branch of a tile's iteration:
    // both branches' gradients must meet in ONE Adam step); null = none.  Same bin geometry as the first.
    const Rec *recs2;
    const uint32_t *starts2, *maxbits2;
    uint32_t capacity2;
    float *params, *exp_avg, *exp_avg_sq;
    void *half_table;        // optional f16 / bf16 copy of params (same [L][T][2] layout)
    int half_dtype;          // SCANERF_F16 / SCANERF_BF16
    float *overflow_grad;    // optional
    AdamArgs a;
    int half_state;          // OPT-IN: exp_avg / exp_avg_sq are __half arrays, adam_step_cuda_fp16 semantics (cuda/adam_kernel.cu:98-144:
                             // loss scale 128, moments stored in half); the default -- what the reference's live code runs, torch.optim.Adam -- is 0
};
template <int kThreads, int U, bool LANE_OWNS_RUN = false, bool ADAM = false>
__global__ void __launch_bounds__(kThreads) k_bin_accumulate(const Rec *__restrict__ recs,
                                                             const uint32_t *__restrict__ starts,
                                                             const uint32_t *__restrict__ maxbits, BinGeom g,
                                                             float *__restrict__ grad_features, AdamEpilogue ad)
{
    extern __shared__ long long acc64[];  // [2 << window_log]
    // Buckets larger than the LDS image (tables above 2^21 entries: bucket = T/256 so that the producer's cursors fit its
    // LDS) are accumulated in windows of 2^13 entries: every window pass re-reads the bucket's records -- about 2 MB,
    // they stay in L2 -- and applies the entries that fall inside it.
    const int wl = g.bucket_log < 13 ? g.bucket_log : 13, ws = 1 << wl;
    // record format (uniform): the fused plan wrote it next to the records; both sets of a two-branch step share it
    const int fmt = g.rec8 < 0 ? (int)*format_word(const_cast<Rec *>(recs)) : g.rec8;   // 0 = Rec, 1 = Rec8, 2 = Rec12
    const bool rec8 = fmt == 1;
    const uint32_t cap1 = rec_capacity(g.capacity, fmt), cap2 = rec_capacity(ad.capacity2, fmt);
    const uint32_t lo1 = min(starts[blockIdx.x], cap1), hi1 = min(starts[blockIdx.x + 1], cap1);
    const bool two = ADAM && ad.recs2 != nullptr;
    const uint32_t lo2 = two ? min(ad.starts2[blockIdx.x], cap2) : 0u, hi2 = two ? min(ad.starts2[blockIdx.x + 1], cap2) : 0u;
    float M = two ? fmaxf(__uint_as_float(*maxbits), __uint_as_float(*ad.maxbits2)) : __uint_as_float(*maxbits);
    // Rec12 components are ROUNDED to 19 mantissa bits after the maximum was taken (scatter_common.h rec12_round): a value just
    // below 2^eM may have become 2^eM.  One 19-bit unit on top of the maximum keeps "every |v| < 2^eM" true for the stored values.
    if ((fmt == 2 || fmt == 3) && M > 0.0f && M < 3.0e38f) M = __uint_as_float(__float_as_uint(M) + 16u);
    const bool overflowed = ADAM && ad.overflow_grad &&
                            (*overflow_flag(const_cast<Rec *>(recs)) != 0u || (two && *overflow_flag(const_cast<Rec *>(ad.recs2)) != 0u));  // uniform
    const uint32_t nrec = ((hi1 - lo1) + (hi2 - lo2)) * (fmt == 3 ? (uint32_t)kSegRecs : 1u);
    if ((nrec == 0 || !(M > 0.0f)) && !overflowed) return;  // nothing to add (uniform per workgroup)
    int eM;
    frexpf(M, &eM);  // M < 2^eM
    // float -> fixed point through the double "magic number": d = v*2^k + 1.5*2^52 holds round(v*2^k) in
    // its low mantissa bits for |v*2^k| < 2^51, so bits(d) - bits(magic) is the integer (one cvt, one
    // fma, one 64-bit subtract instead of the ~20-instruction f32 -> i64 software conversion).
    // |v| <= M < 2^eM: one value must stay below 2^51 after scaling (the magic-number conversion), the sum of the bin's n
    // values below 2^62: k = min(51 - eM, 62 - eM - ceil(log2(n+1))).  At configs[1] (5e5 records per bin) the image resolves
    // 2^-42 of the launch's largest gradient; contributions smaller than that vanish (the reference's f32 atomics keep them,
    // and its sparse Adam then moves such an entry by ~lr: a few 1e-4 of the entries of a foreground + background step).
    const int kb = 62 - eM - (32 - __clz(nrec));
    const int k = kb < 51 - eM ? kb : 51 - eM;
    const double scale = ldexp(1.0, k), magic = 6755399441055744.0;  // 1.5 * 2^52
    auto fx = [&](float v) {
        return (unsigned long long)(__double_as_longlong(fma((double)v, scale, magic)) - __double_as_longlong(magic));
    };
    const int level = blockIdx.x / g.NB, bucket = blockIdx.x % g.NB;
#ifdef SCANERF_EXPERIMENTS   // timing only (wrong results): bins of levels below N = dbg bits 8..15 do nothing (bit 16: the others do nothing)
    if ((g.dbg >> 8) & 0xff) {
        const bool below = level < ((g.dbg >> 8) & 0xff);
        if (below != (((g.dbg >> 16) & 1) != 0)) return;
    }
#endif
    for (int win = 0; win < (1 << (g.bucket_log - wl)); ++win) {
        for (int i = threadIdx.x; i < 2 * ws; i += kThreads) acc64[i] = 0;
        __syncthreads();
        const uint32_t wbase = (uint32_t)win << wl;
        auto apply = [&](const float4 &r) {
            const uint32_t hdr = __float_as_uint(r.x);
            const uint32_t e0 = (hdr & 0xffffu) - wbase, e1 = (hdr >> 16) - wbase;  // unsigned: outside the window = huge
            const float w1 = r.y, w0 = 1.0f - w1;
            unsigned long long *a = reinterpret_cast<unsigned long long *>(acc64);
            const uint32_t oa = threadIdx.x & 1u, ob = oa ^ 1u;   // (odd lanes start with the y word: see apply8)
            const float ga = oa ? r.w : r.z, gb = oa ? r.z : r.w;
            if (e0 < (uint32_t)ws) {
                atomicAdd(&a[2 * e0 + oa], fx(w0 * ga));
                atomicAdd(&a[2 * e0 + ob], fx(w0 * gb));
            }
            if (e1 < (uint32_t)ws) {
                atomicAdd(&a[2 * e1 + oa], fx(w1 * ga));
                atomicAdd(&a[2 * e1 + ob], fx(w1 * gb));
            }
        };
        // Rec8 (scatter_common.h): integers end to end.  p = m * weight < 2^25 in magnitude, value = p * 2^(E - 25), on the
        // image's grid p * 2^(E - 25 + k) < 2^51.  k = 15 records have l1 outside the bucket: one entry only.
        auto apply8 = [&](uint32_t w0, uint32_t w1) {
            const Rec8Fields f = unpack_rec8(w0, w1);
            const uint32_t l0 = f.l0, l1 = f.l1;
            const int t = f.t, mx = f.mx, my = f.my;
            const int sh = f.e25 + k;   // <= 26 (E <= eM, k <= 51 - eM)
            const int rs = 32 - sh > 63 ? 63 : 32 - sh;
            // p * 2^sh on the image's grid: one 64-bit shift of p * 2^32 (values below the grid are floored: records
            // 2^-18 of the launch's largest gradient and smaller, by less than 2^-43 of it each)
            auto fi = [&](int p) { return (unsigned long long)(((long long)p << 32) >> rs); };
            unsigned long long *a = reinterpret_cast<unsigned long long *>(acc64);
            // Odd lanes add their y component first, even lanes their x: the x words of all entries (8-byte word 2 l) lie on
            // the LDS banks 0,1 mod 4 and the y words on 2,3 mod 4, so an instruction in which every lane adds an x word has
            // half of the banks to spread over (measured: three quarters of the LDS's active cycles were bank conflicts).
            const uint32_t oa = threadIdx.x & 1u, ob = oa ^ 1u;
            const int ma = oa ? my : mx, mb = oa ? mx : my;
            atomicAdd(&a[2 * l0 + oa], fi(ma * (8192 - t)));
            atomicAdd(&a[2 * l0 + ob], fi(mb * (8192 - t)));
            if (l1 < (uint32_t)ws) {
                atomicAdd(&a[2 * l1 + oa], fi(ma * t));
                atomicAdd(&a[2 * l1 + ob], fi(mb * t));
            }
        };
        // Rec12 (scatter_common.h): f32 components, 23-bit weight; through the same fixed-point conversion as the 16-byte records
        auto apply12 = [&](uint32_t w0, uint32_t w1, uint32_t w2) {
            const Rec12Fields f = unpack_rec12(w0, w1, w2);
            const float wa = 1.0f - f.w1;
            unsigned long long *a = reinterpret_cast<unsigned long long *>(acc64);
            const uint32_t oa = threadIdx.x & 1u, ob = oa ^ 1u;   // (odd lanes start with the y word: see apply8)
            const float ga = oa ? f.gy : f.gx, gb = oa ? f.gx : f.gy;
            atomicAdd(&a[2 * f.l0 + oa], fx(wa * ga));
            atomicAdd(&a[2 * f.l0 + ob], fx(wa * gb));
            if (f.l1 < (uint32_t)ws) {
                atomicAdd(&a[2 * f.l1 + oa], fx(f.w1 * ga));
                atomicAdd(&a[2 * f.l1 + ob], fx(f.w1 * gb));
            }
        };
#ifdef SCANERF_EXPERIMENTS
        if (!(g.dbg & 1))
#endif
        for (int set = 0; set < (two ? 2 : 1); ++set) {   // (one copy of the streaming code for both record sets)
        const float4 *r4 = reinterpret_cast<const float4 *>(set ? ad.recs2 : recs);
        const uint32_t lo = set ? lo2 : lo1, hi = set ? hi2 : hi1;
        if (fmt == 3) {
            // 64-byte segments of five records (k_bin_scatter_seg): a lane takes SEGS consecutive segments, four 16-byte loads
            // each; all-zero record slots (a range's padding) and records with an exactly zero gradient add nothing and are skipped
            const uint4 *u4 = reinterpret_cast<const uint4 *>(r4);
            // (ONE segment per lane and round: a large table's bucket holds ~1 700 segments, which 1 024 lanes share best one at a
            // time -- same-box A/B at T = 2^24, interleaved runs: two per lane 2.57 / 2.59 / 2.58 ms, one per lane 2.40 / 2.42 / 2.39 / 2.40)
            constexpr int SEGS = 1;
            auto apply12z = [&](uint32_t w0, uint32_t w1, uint32_t w2) {
                if (((w1 | w2) & ~15u) != 0u) apply12(w0, w1, w2);
            };
            for (uint32_t c = lo + threadIdx.x * SEGS; c < hi; c += SEGS * kThreads) {
                uint4 r[4 * SEGS];
#pragma unroll
                for (int u = 0; u < SEGS; ++u)
                    if (c + u < hi) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[4 * u + j] = u4[(size_t)(c + u) * 4 + j];
                    }
#pragma unroll
                for (int u = 0; u < SEGS; ++u)
                    if (c + u < hi) {
                        const uint4 &a0 = r[4 * u], &a1 = r[4 * u + 1], &a2 = r[4 * u + 2], &a3 = r[4 * u + 3];
                        apply12z(a0.x, a0.y, a0.z);
                        apply12z(a0.w, a1.x, a1.y);
                        apply12z(a1.z, a1.w, a2.x);
                        apply12z(a2.y, a2.z, a2.w);
                        apply12z(a3.x, a3.y, a3.z);
                    }
            }
            continue;
        }
        if (fmt == 2) {
            // each lane takes runs of 4 consecutive records = three aligned 16-byte loads, U / 4 runs in flight (the bin's range
            // may start anywhere: runs start at multiples of 4 records, partial runs go one record at a time)
            const uint32_t *r1 = reinterpret_cast<const uint32_t *>(r4);
            const uint4 *u4 = reinterpret_cast<const uint4 *>(r4);
            constexpr int RUNS = U / 4 > 0 ? U / 4 : 1;
            for (uint32_t c = (lo & ~3u) + threadIdx.x * 4 * RUNS; c < hi; c += 4 * RUNS * kThreads) {
                if (c >= lo && c + 4 * RUNS <= hi) {
                    uint4 r[3 * RUNS];
#pragma unroll
                    for (int u = 0; u < 3 * RUNS; ++u) r[u] = u4[(size_t)(c >> 2) * 3 + u];
#pragma unroll
                    for (int u = 0; u < RUNS; ++u) {
                        const uint4 &a0 = r[3 * u], &a1 = r[3 * u + 1], &a2 = r[3 * u + 2];
                        apply12(a0.x, a0.y, a0.z);
                        apply12(a0.w, a1.x, a1.y);
                        apply12(a1.z, a1.w, a2.x);
                        apply12(a2.y, a2.z, a2.w);
                    }
                } else {
                    for (uint32_t j = c < lo ? lo : c; j < hi && j < c + 4 * RUNS; ++j) apply12(r1[(size_t)j * 3], r1[(size_t)j * 3 + 1], r1[(size_t)j * 3 + 2]);
                }
            }
            continue;
        }
        if (rec8) {
            const uint2 *r2 = reinterpret_cast<const uint2 *>(r4);
            if (LANE_OWNS_RUN) {
                // as below: each lane takes a run of consecutive records, 2 * U of them through U 16-byte loads; runs start
                // at even record indices (the bin's range may not), partial runs go one record at a time
                const uint4 *u4 = reinterpret_cast<const uint4 *>(r4);
                for (uint32_t c = (lo & ~1u) + threadIdx.x * 2 * U; c < hi; c += 2 * U * kThreads) {
                    if (c >= lo && c + 2 * U <= hi) {
                        uint4 r[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) r[u] = u4[(c >> 1) + u];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            apply8(r[u].x, r[u].y);
                            apply8(r[u].z, r[u].w);
                        }
                    } else {
                        for (uint32_t j = c < lo ? lo : c; j < hi && j < c + 2 * U; ++j) {
                            const uint2 r = r2[j];
                            apply8(r.x, r.y);
                        }
                    }
                }
            } else {
                for (uint32_t j = lo + threadIdx.x; j < hi; j += kThreads) {
                    const uint2 r = r2[j];
                    apply8(r.x, r.y);
                }
            }
            continue;
        }
        // U independent 16-B loads in flight per lane (the records are read once from HBM; window passes re-read them from L2)
        uint32_t i = lo + threadIdx.x;
        if (LANE_OWNS_RUN) {
            // Each lane takes U CONSECUTIVE records.  Records of consecutive samples of a ray that share a cell (coarse
            // levels: up to ~8 per cell) sit next to each other in the bin; handled by neighbouring lanes they hit the same
            // LDS address in one instruction, which the LDS serialises at ~6 clocks per duplicate (tools/lds_atomic_bench:
            // 64 equal addresses 0.15 lane-ops/clk/CU against 5.8 for distinct ones).  Inside one lane they are just
            // successive instructions.  A lane reads 16*U contiguous bytes, so the wave still consumes whole lines.
            uint32_t c = lo + threadIdx.x * U;
            for (; c + U <= hi; c += U * kThreads) {
                float4 r[U];
#pragma unroll
                for (int u = 0; u < U; ++u) r[u] = r4[c + u];  // (not nontemporal: a line serves 8 of these loads through L1 -- 1.75 vs 4.5 ms)
#pragma unroll
                for (int u = 0; u < U; ++u) apply(r[u]);
            }
            for (uint32_t j = c; j < hi && j < c + U; ++j) apply(r4[j]);  // at most one lane has a partial run
            i = hi;
        }
        for (; i + (U - 1) * kThreads < hi; i += U * kThreads) {
            float4 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f *src = reinterpret_cast<const v4f *>(r4 + i + u * kThreads);
                const v4f t = g.bucket_log <= 13 ? __builtin_nontemporal_load(src) : *src;  // windowed: keep them cached
                r[u] = make_float4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) apply(r[u]);
        }
        for (; i < hi; i += kThreads) apply(r4[i]);
        }
        __syncthreads();
        const size_t ebase = (size_t)level * g.T + ((size_t)bucket << g.bucket_log) + wbase;  // first entry of the window
#ifdef SCANERF_EXPERIMENTS
        if (g.dbg & 2) continue;
#endif
        if (ADAM) {
            float2 *P = reinterpret_cast<float2 *>(ad.params) + ebase, *Mo = reinterpret_cast<float2 *>(ad.exp_avg) + ebase,
                   *Vo = reinterpret_cast<float2 *>(ad.exp_avg_sq) + ebase;
            float2 *og = overflowed ? reinterpret_cast<float2 *>(ad.overflow_grad) + ebase : nullptr;
            // (measured, round 6: issuing the loads of four entry PAIRS per thread before the first use -- 16-byte accesses, no
            // dependent round trips -- is SLOWER at T = 2^24, 2.85 vs 2.46 ms: the 16 waves already keep enough loads in flight,
            // and pairs touch 98 % of the fine levels' entries where single entries touch 86 %)
            for (int j = threadIdx.x; j < ws; j += kThreads) {
                const long long qx = acc64[2 * j], qy = acc64[2 * j + 1];
                float gx = 0.0f + (float)ldexp((double)qx, -k), gy = 0.0f + (float)ldexp((double)qy, -k);  // as grad_features would hold them
                if (og) {
                    const float2 e = og[j];
                    if (e.x != 0.0f || e.y != 0.0f) {
                        gx += e.x;
                        gy += e.y;
                        og[j] = make_float2(0.0f, 0.0f);
                    }
                }
                if ((gx != 0.0f || gy != 0.0f) && ad.half_state) {   // (uniform) fp16 moments: 8 + 4 + 4 bytes per entry each way
                    __half2 *Mh = reinterpret_cast<__half2 *>(ad.exp_avg) + ebase, *Vh = reinterpret_cast<__half2 *>(ad.exp_avg_sq) + ebase;
                    float2 p = P[j];
                    const __half2 mh = Mh[j], vh = Vh[j];
                    float mx = __low2float(mh), my = __high2float(mh), vx = __low2float(vh), vy = __high2float(vh);
                    // (an untouched component keeps its stored bits: a float round trip of a half is exact)
                    adam_update_one<true>(p.x, mx, vx, gx, ad.a);
                    adam_update_one<true>(p.y, my, vy, gy, ad.a);
                    P[j] = p;
                    Mh[j] = __halves2half2(__float2half(mx), __float2half(my));
                    Vh[j] = __halves2half2(__float2half(vx), __float2half(vy));
                    if (ad.half_table) {
                        if (ad.half_dtype == SCANERF_F16)
                            reinterpret_cast<__half2 *>(ad.half_table)[ebase + j] = __floats2half2_rn(p.x, p.y);
                        else
                            reinterpret_cast<__hip_bfloat162 *>(ad.half_table)[ebase + j] =
                                __hip_bfloat162{ __float2bfloat16(p.x), __float2bfloat16(p.y) };
                    }
                } else if (gx != 0.0f || gy != 0.0f) {
                    float2 p = P[j], m = Mo[j], v = Vo[j];
                    adam_update_one<false>(p.x, m.x, v.x, gx, ad.a);
                    adam_update_one<false>(p.y, m.y, v.y, gy, ad.a);
                    P[j] = p;
                    Mo[j] = m;
                    Vo[j] = v;
                    if (ad.half_table) {
                        if (ad.half_dtype == SCANERF_F16)
                            reinterpret_cast<__half2 *>(ad.half_table)[ebase + j] = __floats2half2_rn(p.x, p.y);
                        else
                            reinterpret_cast<__hip_bfloat162 *>(ad.half_table)[ebase + j] =
                                __hip_bfloat162{ __float2bfloat16(p.x), __float2bfloat16(p.y) };  // round to nearest even, as torch
                    }
                }
            }
        } else {
            float2 *dst = reinterpret_cast<float2 *>(grad_features) + ebase;
            for (int j = threadIdx.x; j < ws; j += kThreads) {
                const long long qx = acc64[2 * j], qy = acc64[2 * j + 1];
                if (qx | qy) {
                    float2 v = dst[j];
                    v.x += (float)ldexp((double)qx, -k);
                    v.y += (float)ldexp((double)qy, -k);
                    dst[j] = v;
                }
            }
        }
        __syncthreads();
    }
}

// ---- count for the FUSED producer (k_render_bwd emits the records) ------------------------------
// Same ray -> workgroup map as k_render_bwd (ray = blockIdx.x + i * gridDim.x), same point arithmetic
// (contract_point) and the same make_pairs/count_pairs as its emit_pairs, so the ranges it reserves
// are exactly the ones the backward kernel fills.
__global__ void __launch_bounds__(1024) k_bin_count_rays(RenderArgs f, BinGeom g, uint32_t *__restrict__ counts,
                                                         uint32_t *__restrict__ maxbits, uint32_t *__restrict__ overflow)
{
    extern __shared__ uint32_t hist[];  // [16*NB]
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *maxbits = 0;
        *overflow = 0;
        overflow[-1] = (uint32_t)g.rec8;  // format_word(): what the backward will emit and the accumulate must decode
        overflow[-2] = f.skip_levels;     // skip_word(): the levels left out below
    }
    const int nbins = 16 * g.NB;
    for (int i = threadIdx.x; i < nbins; i += 1024) hist[i] = 0;
    __syncthreads();
    const uint32_t mask = (uint32_t)f.T - 1u;
    const int G = gridDim.x, w = blockIdx.x, R = g.rpg;
    const int ngroups_all = (f.B + R - 1) / R;                          // ray groups of the launch
    const int ngroups = w < ngroups_all ? (ngroups_all - w + G - 1) / G : 0;  // ... visited by this workgroup
    // ray-fastest walk: the 64 lanes of a wave count 64 DIFFERENT rays at one sample index.  Sample-fastest, neighbouring
    // lanes are neighbouring samples of a ray, share cells at the coarse levels and so hit the same histogram word in one
    // instruction, which the LDS serialises (tools/lds_atomic_bench: ~6 clocks per duplicate address).
    const int nlr = ngroups * R;
    for (int idx = threadIdx.x; idx < nlr * f.S; idx += 1024) {
        const int lr = idx % nlr, s = idx / nlr;
        const int ray = (w + (lr / R) * G) * R + lr % R;
        if (ray >= f.B) continue;
        if (f.ray_valid && !f.ray_valid[ray]) continue;
        float o[3], d[3], p[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = f.rays_o[3 * ray + k];
            d[k] = f.rays_d[3 * ray + k];
        }
        contract_point(f, o, d, f.z_vals[(size_t)ray * f.S + s], p);
        for (int l = 0; l < 16; ++l) {
            if ((f.skip_levels >> l) & 1u) continue;   // coarse-to-fine: a masked level's gradients are exactly zero, no records
            Pairs pr;
            make_pairs(p, f.resolutions + 3 * l, mask, pr);
            count_pairs(pr, hist + l * g.NB, g.bucket_log);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += 1024) counts[(size_t)i * g.W + blockIdx.x] = hist[i];
}

// bin geometry of the fused producer: W = the backward kernel's grid, buckets sized so that the
// 16*NB cursors fit next to the backward kernel's LDS image (NB <= 256)
bool fused_geom(int B, int S, int T, BinGeom &g, int arith = SCANERF_ARITH_F32)
{
    if (B < 1 || S < 1 || T < 2 || (T & (T - 1))) return false;
    g.bucket_log = fused_bucket_log(T);
    if (g.bucket_log > 16) return false;  // local entry indices are 16-bit; buckets above 2^13 entries are accumulated in windows
    if ((int64_t)B * S * 16 * 4 + (1 << 20) >= (int64_t)1 << 31) return false;  // 32-bit record offsets
    g.N = B * S; g.L = 16; g.T = T;
    g.dbg = tune_int("SCANERF_ACC_DBG", 0) & ~0xff;   // (experiments build: the accumulate's level switches; 0 in the product build)
    g.rows16 = 0;
    g.NB = T >> g.bucket_log;
    g.W = scanerf_render_backward_grid(B);
    g.per_wg = 0;
    g.rpg = (arith == SCANERF_ARITH_T16 || arith == SCANERF_ARITH_T16S) ? 8 : (arith == SCANERF_ARITH_H3 ? 4 : 1);
    g.capacity = 0;
    g.rec8 = fused_rec8(arith, g.bucket_log);
    return true;
}

}  // namespace

// bucket size of the stand-alone path: 2^11 entries, growing to at most 2^13 (the accumulate's LDS image) so that a level
// has at most 2048 buckets
static int standalone_bucket_log(int T)
{
    const int lt = bin_ilog2(T);
    int bl = lt - 11 > kBucketLog ? lt - 11 : kBucketLog;
    bl = tune_int("SCANERF_STANDALONE_BUCKET_LOG", bl);  // tuning experiments only
    if (bl > 13) bl = 13;
    return lt < bl ? lt : bl;
}

// Segments the large-table producer may need: the records (4 per (point, level) + the straddle slack of the record budget) in
// fives, plus one partly filled segment per (bucket, producer workgroup) range.
constexpr int kSegProducers = 256;
static int seg_route_producers(int N) { return N >= kSegProducers * 256 ? kSegProducers : (N + 255) / 256; }   // (as binned_backward's g.W)
static size_t seg_route_segments(int N, int L, int64_t nbins)
{
    return ((size_t)N * L * 4 + (size_t)N * L / 8) / kSegRecs + (size_t)nbins * seg_route_producers(N) + 4096;
}

// Workspace bytes for a binned backward of N points.  0 => shape unsupported by the binned path
// (the atomics kernel is used instead).
SCANERF_API size_t scanerf_embedding_bwd_workspace_bytes(int N, int L, int T)
{
    if (N <= 0 || L < 1 || T < 2 || (T & (T - 1))) return 0;
    const int bl = standalone_bucket_log(T);
    const int64_t nbins = (int64_t)L * (T >> bl);
    if ((T >> bl) * 4 > 64 * 1024) return 0;                   // one level's LDS counters
    if ((int64_t)N * L * 4 + (1 << 20) >= (int64_t)1 << 31) return 0;  // 32-bit record offsets
    const int W = 1024;
    size_t recs = ((size_t)N * L * 4 + (size_t)N * L / 8 + 4096) * sizeof(Rec);
    // large tables (one level's counters in LDS at a time): records in 64-byte segments of five, every (bucket, workgroup)
    // range rounded up to whole segments (k_bin_scatter_seg)
    if ((size_t)nbins * 4 > 64 * 1024) recs = std::max(recs, seg_route_segments(N, L, nbins) * 64) + (((size_t)N * 12 + 255) & ~(size_t)255);   // (+ the contracted points of scanerf_table_grad_scatter_adam_rays)
    return recs + (size_t)nbins * W * 4 + (size_t)(2 * nbins + 6) * 4 + 256;
}

// grad_features += scatter(grad_in) through the binned path.  grad_layout: 0 = [N][L][2], 1 = [L][N][2].
// grad_features: the table the image is added to -- or, with an Adam epilogue, the overflow table (ad->overflow_grad)
static int binned_backward(const float *points, const float *grad_in, float *grad_features, const int32_t *resolutions, int N,
                           int L, int T, int grad_layout, void *workspace, size_t workspace_bytes, const AdamEpilogue *ad,
                           scanerf_stream_t stream, int compact_records = -1, const PointSrc *rays = nullptr)
{
    // compact_records: -1 = the layout's default (point-major rows of 16 levels: 12-byte records; level-major: 16-byte),
    // 0 = 16-byte records wherever a 16-byte producer exists, 1 = 8-byte (level-major only), 2 = 12-byte
    const bool want16 = compact_records == 0;
    if (compact_records < 0) compact_records = 0;
    SCANERF_REQUIRE(N >= 0 && L >= 1, "embedding_bg_backward_binned: N=%d L=%d", N, L);
    if (N == 0) return 0;
    const size_t need = scanerf_embedding_bwd_workspace_bytes(N, L, T);
    SCANERF_REQUIRE(need != 0, "embedding_bg_backward_binned: shape N=%d L=%d T=%d not supported by the binned path", N, L,
                    T);
    SCANERF_REQUIRE((rays || (points && grad_in)) && grad_features && resolutions && workspace,
                    "embedding_bg_backward_binned: null pointer");
    SCANERF_REQUIRE(((uintptr_t)workspace & 15) == 0, "embedding_bg_backward_binned: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    BinGeom g;
    g.N = N; g.L = L; g.T = T;
    g.rpg = 1;
    g.rows16 = 0;
    g.dbg = tune_int("SCANERF_ACC_DBG", 0);
    g.bucket_log = standalone_bucket_log(T);
    // Point-major gradients of 16 levels (round 5): the level-by-level producer (k_bin_scatter) with the fused path's bucket size
    // (2^13 entries: 64 ranges per level at T = 2^19 instead of 256), 256 persistent-size workgroups of 1024 threads (fewer
    // open ranges per L2) and 12-byte records.  Measured on the op-by-op training step (tools/ops_path_profile.py, backward
    // section): 2^11 / 1024 workgroups 16.2 ms, 2^13 / 256 workgroups 12.6 ms, with the row in registers and Rec12: see DESIGN.md.
    bool rows16 = grad_layout == 0 && L == 16 && ((uintptr_t)grad_in & 15) == 0 && !want16 && !tune_set("SCANERF_SCATTER_OLD");
    if (rows16) {
        const int lt = bin_ilog2(T);
        int bl = lt < 13 ? lt : 13;
        { const int v = tune_int("SCANERF_STANDALONE_BUCKET_LOG", bl); bl = v < bl ? v : bl; }
        if ((size_t)L * (T >> bl) * 4 > 64 * 1024) rows16 = false;   // (all levels' cursors must fit the producer's LDS: T <= 2^23)
        else g.bucket_log = bl;
    }
    // 16-byte records, or the 8- / 12-byte ones: on request for level-major gradients (out of the 16-sample-tile backward
    // kernels), 12-byte ones by default for the point-major rows (f32 components with 19-bit mantissas, 23-bit weights:
    // scatter_common.h Rec12; compact_records = 0 keeps the 16-byte records)
    if (rows16 && compact_records == 0) compact_records = 2;
    g.rows16 = rows16 ? 1 : 0;
    g.rec8 = (compact_records >= 1 && compact_records <= 2 && (grad_layout == 1 || (rows16 && compact_records == 2)) &&
              g.bucket_log <= kRec8MaxBucketLog && !tune_set("SCANERF_REC16")) ? compact_records : 0;
    g.NB = T >> g.bucket_log;
    // Large tables (round 6): 12-byte records leave the producer as full 64-byte segments (k_bin_scatter_seg, format 3) -- for the
    // level-major gradients of the t16s backward and for the point-major rows of the binding surface (whose default is Rec12
    // as well, see rows16); compact_records = 0, a smaller workspace, or SCANERF_REC16 / SCANERF_SCATTER_OLD of an experiments build keep the record-at-a-time producer
    const bool seg_route = (size_t)L * g.NB * 4 > 64 * 1024 && g.bucket_log <= 13 && g.NB <= 2048 &&
                           (compact_records == 2 || (grad_layout == 0 && compact_records == 0 && !want16)) &&
                           !tune_set("SCANERF_REC16") && !tune_set("SCANERF_SCATTER_OLD");
    if (seg_route) g.rec8 = 3;
    // producer workgroups: every one of them writes and reads a counter per bin, so with the tens of thousands of bins of a
    // large table fewer, longer-running workgroups are cheaper (T = 2^24, 2.1 M points: count 0.62 -> see DESIGN.md)
    g.W = ((size_t)L * g.NB * 4 > 64 * 1024 || rows16) ? 256 : 1024;
    { const int v = tune_int("SCANERF_SCATTER_W", 0); if (v >= 1 && v <= 1024) g.W = v; }   // tuning experiments
    if (g.W > (N + kThreads - 1) / kThreads) g.W = (N + kThreads - 1) / kThreads;
    g.per_wg = (N + g.W - 1) / g.W;
    const int nbins = L * g.NB;
    BinWorkspace w;
    if (rays) workspace_bytes &= ~(size_t)255;   // (the contracted points sit at the end: keep them aligned whatever size the caller passes)
    const size_t pts_tail = rays ? (((size_t)N * 12 + 255) & ~(size_t)255) : 0;
    SCANERF_REQUIRE(workspace_bytes > pts_tail && bin_workspace_carve(workspace, workspace_bytes - pts_tail, nbins, g.W, w),
                    "embedding_bg_backward_binned: workspace too small (%zu B)", workspace_bytes);
    g.capacity = w.capacity;
    if (g.rec8 == 3 && (g.W > seg_route_producers(N) || rec_capacity(g.capacity, 3) < seg_route_segments(N, L, L * g.NB)))
        g.rec8 = (compact_records == 2 && g.bucket_log <= kRec8MaxBucketLog) ? 2 : 0;   // (a caller's smaller workspace: the old producer)
    SCANERF_REQUIRE(!rays || g.rec8 == 3, "table_grad_scatter_adam_rays: T=%d N=%d needs the large-table producer (T >= 2^22, L = 16, "
                    "12-byte records, workspace of scanerf_embedding_bwd_workspace_bytes)", T, N);
    uint32_t *counts = w.counts, *totals = w.totals, *starts = w.starts, *maxbits = w.maxbits;
    Rec *recs = w.recs;

    // all levels' counters in LDS when they fit in 64 KB, else one level's at a time (large tables)
    const bool per_level = (size_t)nbins * 4 > 64 * 1024;
    const size_t lds_bins = per_level ? (size_t)g.NB * 4 : (size_t)nbins * 4;
    const float2 *gi = reinterpret_cast<const float2 *>(grad_in);
    if (rays) {
        // the contracted points of both branches, once, into the workspace's tail (read 2 x 16 times from the L2s afterwards)
        // (measured and not kept: 16-byte point rows read with one load per level -- the count kernel takes 0.75 instead of 0.39 ms)
        float *pts = reinterpret_cast<float *>(static_cast<char *>(workspace) + workspace_bytes - pts_tail);
        hipLaunchKernelGGL(k_src_points, dim3(stream_grid(N, 256)), dim3(256), 0, st, *rays, pts, N);
        points = pts;
        hipLaunchKernelGGL((k_bin_count<true, true>), dim3(g.W), dim3(1024), lds_bins, st, points, resolutions, g, counts, maxbits, overflow_flag(recs));
    } else if (per_level)
        hipLaunchKernelGGL((k_bin_count<true>), dim3(g.W), dim3(1024), lds_bins, st, points, resolutions, g, counts, maxbits, overflow_flag(recs));
    else
        hipLaunchKernelGGL((k_bin_count<false>), dim3(g.W), dim3(rows16 ? 1024 : kThreads), lds_bins, st, points, resolutions, g, counts, maxbits, overflow_flag(recs));   // (256 producer workgroups: give them all 16 waves)
    hipLaunchKernelGGL(k_bin_rowscan, dim3(nbins), dim3(kThreads), 0, st, counts, totals, g.W);
    hipLaunchKernelGGL(k_bin_starts, dim3(1), dim3(1024), 0, st, totals, starts, nbins);
    if (g.rec8 == 3) {
        const size_t lds_seg = (size_t)g.NB * 18 * 4;   // cnt, segn, gbase, 15-word slots
        const PointSrc none{};
#define SCANERF_LAUNCH_SEG(LM, RY, PT)                                                                                          \
    {                                                                                                                           \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_scatter_seg<LM, RY, PT>),                        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_seg);                          \
        SCANERF_REQUIRE(e == hipSuccess, "embedding_bg_backward_binned: cannot reserve %zu B of LDS: %s", lds_seg, hipGetErrorString(e)); \
        hipLaunchKernelGGL((k_bin_scatter_seg<LM, RY, PT>), dim3(g.W), dim3(1024), lds_seg, st, points, gi, resolutions, g, counts, \
                           starts, recs, grad_features, maxbits, rays ? *rays : none);                                          \
    }
        // two points per lane and batch (measured at T = 2^24, 4.2e6 points: 1 / 2 points 1.94 / 1.91 ms; 4 spill registers)
        if (rays) SCANERF_LAUNCH_SEG(true, true, 2)
        else if (grad_layout == 1) SCANERF_LAUNCH_SEG(true, false, 2)
        else SCANERF_LAUNCH_SEG(false, false, 2)
#undef SCANERF_LAUNCH_SEG
    } else if (rows16 && g.rec8 == 2)
        hipLaunchKernelGGL((k_bin_scatter<false, false, 2>), dim3(g.W), dim3(1024), lds_bins, st, points, gi, resolutions, g,
                           counts, starts, recs, grad_features, maxbits);
    else if (rows16)
        hipLaunchKernelGGL((k_bin_scatter<false>), dim3(g.W), dim3(1024), lds_bins, st, points, gi, resolutions, g,
                           counts, starts, recs, grad_features, maxbits);
    else if (g.rec8 == 1 && per_level)
        hipLaunchKernelGGL((k_bin_scatter<true, true, 1>), dim3(g.W), dim3(1024), lds_bins, st, points, gi, resolutions, g,
                           counts, starts, recs, grad_features, maxbits);
    else if (g.rec8 == 1)
        hipLaunchKernelGGL((k_bin_scatter<true, false, 1>), dim3(g.W), dim3(kThreads), lds_bins, st, points, gi, resolutions, g,
                           counts, starts, recs, grad_features, maxbits);
    else if (g.rec8 == 2 && per_level)
        hipLaunchKernelGGL((k_bin_scatter<true, true, 2>), dim3(g.W), dim3(1024), lds_bins, st, points, gi, resolutions, g,
                           counts, starts, recs, grad_features, maxbits);
    else if (g.rec8 == 2)
        hipLaunchKernelGGL((k_bin_scatter<true, false, 2>), dim3(g.W), dim3(kThreads), lds_bins, st, points, gi, resolutions, g,
                           counts, starts, recs, grad_features, maxbits);
    else if (per_level && grad_layout == 0)
        hipLaunchKernelGGL((k_bin_scatter<false, true>), dim3(g.W), dim3(1024), lds_bins, st, points, gi, resolutions, g,
                           counts, starts, recs, grad_features, maxbits);

    else if (per_level)
        hipLaunchKernelGGL((k_bin_scatter<true, true>), dim3(g.W), dim3(1024), lds_bins, st, points, gi, resolutions, g,
                           counts, starts, recs, grad_features, maxbits);
    else if (grad_layout == 0)
        hipLaunchKernelGGL((k_bin_scatter<false>), dim3(g.W), dim3(kThreads), lds_bins, st, points, gi, resolutions, g,
                           counts, starts, recs, grad_features, maxbits);
    else
        hipLaunchKernelGGL((k_bin_scatter<true>), dim3(g.W), dim3(kThreads), lds_bins, st, points, gi, resolutions, g,
                           counts, starts, recs, grad_features, maxbits);
    const size_t lds_acc = (size_t)(2 << g.bucket_log) * 8;
    if (lds_acc > 64 * 1024) {  // one image per CU: give the workgroup all 16 waves
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_accumulate<1024, 16, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_acc);
        SCANERF_REQUIRE(e == hipSuccess, "embedding_bg_backward_binned: cannot reserve %zu B of LDS: %s", lds_acc, hipGetErrorString(e));
        if (ad) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_accumulate<1024, 16, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_acc);
            SCANERF_REQUIRE(e == hipSuccess, "embedding_bg_backward_binned: cannot reserve %zu B of LDS: %s", lds_acc, hipGetErrorString(e));
            hipLaunchKernelGGL((k_bin_accumulate<1024, 16, true, true>), dim3(nbins), dim3(1024), lds_acc, st, recs, starts, maxbits,
                               g, (float *)nullptr, *ad);
        } else if (g.rec8 == 2 && nbins <= 4096) {
            // few, long ranges of 12-byte records (small tables: 1 024 buckets of 5e5 records at configs[1]): the fused accumulate's
            // shape (512 threads x 16 records per lane 1.43 ms against 1.53 for 1024 x 16, scanerf_render_scatter_accumulate_adam)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_accumulate<512, 16, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_acc);
            SCANERF_REQUIRE(e == hipSuccess, "embedding_bg_backward_binned: cannot reserve %zu B of LDS: %s", lds_acc, hipGetErrorString(e));
            hipLaunchKernelGGL((k_bin_accumulate<512, 16, true>), dim3(nbins), dim3(512), lds_acc, st, recs, starts, maxbits, g, grad_features,
                               AdamEpilogue{});
        } else
            hipLaunchKernelGGL((k_bin_accumulate<1024, 16, true>), dim3(nbins), dim3(1024), lds_acc, st, recs, starts, maxbits, g,
                               grad_features, AdamEpilogue{});
    } else if (ad) {
        hipLaunchKernelGGL((k_bin_accumulate<256, 32, true, true>), dim3(nbins), dim3(256), lds_acc, st, recs, starts, maxbits, g,
                           (float *)nullptr, *ad);
    } else {
        hipLaunchKernelGGL((k_bin_accumulate<256, 32, true>), dim3(nbins), dim3(256), lds_acc, st, recs, starts, maxbits, g,
                           grad_features, AdamEpilogue{});
    }
    return check_launch("embedding_bg_backward_binned");
}

SCANERF_API int scanerf_embedding_bg_backward_binned(const float *points, const float *grad_in, float *grad_features,
                                                     const int32_t *resolutions, int N, int L, int T, int grad_layout,
                                                     void *workspace, size_t workspace_bytes, int compact_records,
                                                     scanerf_stream_t stream)
{
    SCANERF_REQUIRE(compact_records >= -1 && compact_records <= 2, "embedding_bg_backward_binned: compact_records=%d", compact_records);
    return binned_backward(points, grad_in, grad_features, resolutions, N, L, T, grad_layout, workspace, workspace_bytes, nullptr, stream,
                           compact_records);
}

// The binned scatter ending in the fused sparse Adam (see scanerf_render_scatter_accumulate_adam): the table-gradient path of
// tables too large for the backward kernel's own record emission (the reference's default T = 2^24), without a gradient table,
// its zero-fill or the optimiser's scan of 2 GB.  overflow_grad: zero [L][T][2] f32 table, written only if the workspace overflows.
SCANERF_API int scanerf_embedding_bg_backward_binned_adam(const float *points, const float *grad_in, const int32_t *resolutions,
                                                          int N, int L, int T, int grad_layout, void *workspace,
                                                          size_t workspace_bytes, float *params, float *exp_avg,
                                                          float *exp_avg_sq, void *half_table, int half_dtype,
                                                          float *overflow_grad, float lr, float beta1, float beta2, float eps,
                                                          int step, int compact_records, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(params && exp_avg && exp_avg_sq && overflow_grad, "embedding_bg_backward_binned_adam: null pointer");
    SCANERF_REQUIRE(!half_table || half_dtype == SCANERF_F16 || half_dtype == SCANERF_BF16,
                    "embedding_bg_backward_binned_adam: half_dtype=%d", half_dtype);
    const AdamEpilogue ad{ nullptr, nullptr, nullptr, 0u, params, exp_avg, exp_avg_sq, half_table, half_dtype, overflow_grad,
                           make_adam_args(lr, beta1, beta2, eps, step), 0 };
    SCANERF_REQUIRE(compact_records >= -1 && compact_records <= 2, "embedding_bg_backward_binned_adam: compact_records=%d", compact_records);
    return binned_backward(points, grad_in, overflow_grad, resolutions, N, L, T, grad_layout, workspace, workspace_bytes, &ad, stream,
                           compact_records);
}

// The same for the samples of up to two render branches over the same rays (PointSrc above): positions from rays and depths,
// gradients = each branch's level-major dfeat; both branches' gradients meet in ONE Adam step (tile.py:639-692, :1010).  Tables of
// at least 2^22 entries per level (below, the backward kernel emits its own records: scanerf_render_scatter_*).
SCANERF_API int scanerf_table_grad_scatter_adam_rays(const float *rays_o, const float *rays_d, int B, const float *z1, const float *dfeat1,
                                                     const uint8_t *valid1, int S1, int contract_mode1, const float *z2,
                                                     const float *dfeat2, const uint8_t *valid2, int S2, int contract_mode2,
                                                     const float *min_bbox, const float *bbox_size, const int32_t *resolutions, int T,
                                                     void *workspace, size_t workspace_bytes, float *params, float *exp_avg,
                                                     float *exp_avg_sq, void *half_table, int half_dtype, float *overflow_grad, float lr,
                                                     float beta1, float beta2, float eps, int step, int fp16_moments,
                                                     scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S1 >= 1 && (!z2 || S2 >= 1), "table_grad_scatter_adam_rays: B=%d S1=%d S2=%d", B, S1, S2);
    if (B == 0) return 0;
    SCANERF_REQUIRE(rays_o && rays_d && z1 && dfeat1 && (!z2 || dfeat2) && min_bbox && bbox_size && params && exp_avg && exp_avg_sq &&
                    overflow_grad, "table_grad_scatter_adam_rays: null pointer");
    SCANERF_REQUIRE(!half_table || half_dtype == SCANERF_F16 || half_dtype == SCANERF_BF16, "table_grad_scatter_adam_rays: half_dtype=%d", half_dtype);
    const int64_t N = (int64_t)B * S1 + (z2 ? (int64_t)B * S2 : 0);
    SCANERF_REQUIRE(N < ((int64_t)1 << 31) / 64, "table_grad_scatter_adam_rays: %lld points", (long long)N);
    PointSrc src{};
    src.rays_o = rays_o; src.rays_d = rays_d;
    src.z[0] = z1; src.z[1] = z2;
    src.valid[0] = valid1; src.valid[1] = valid2;
    src.grad[0] = reinterpret_cast<const float2 *>(dfeat1); src.grad[1] = reinterpret_cast<const float2 *>(dfeat2);
    src.S[0] = S1; src.S[1] = z2 ? S2 : 1;
    src.mode[0] = contract_mode1; src.mode[1] = contract_mode2;
    src.N1 = B * S1;
    for (int k = 0; k < 3; ++k) { src.min_bbox[k] = min_bbox[k]; src.bbox_size[k] = bbox_size[k]; }
    const AdamEpilogue ad{ nullptr, nullptr, nullptr, 0u, params, exp_avg, exp_avg_sq, half_table, half_dtype, overflow_grad,
                           make_adam_args(lr, beta1, beta2, eps, step), fp16_moments ? 1 : 0 };
    return binned_backward(nullptr, nullptr, overflow_grad, resolutions, (int)N, 16, T, 1, workspace, workspace_bytes, &ad, stream, 2, &src);
}

// Launch-shape hint for the accumulate: which record format the last plan on a workspace chose.  The kernel decodes by the
// format word IN the workspace; this host-side note only picks the faster of two equally correct launch shapes.
static std::mutex g_hint_mutex;
static std::unordered_map<const void *, int> g_plan_rec8;
static void note_plan_format(const void *workspace, int rec8)
{
    std::lock_guard<std::mutex> lock(g_hint_mutex);
    if (g_plan_rec8.size() > 64) g_plan_rec8.clear();
    g_plan_rec8[workspace] = rec8;
}
static int plan_format(const void *workspace)
{
    std::lock_guard<std::mutex> lock(g_hint_mutex);
    auto it = g_plan_rec8.find(workspace);
    return it != g_plan_rec8.end() ? it->second : 0;
}

namespace {
// One workgroup per COARSE bucket (2^bucket_log entries, 16-byte records with 16-bit local entries): its records go, grouped by
// window w = local entry >> 13, to the same index range of the fine area as 12-byte records (scatter_common.h Rec12) with 13-bit
// local entries; starts_f[8 c + w] = where window w of bucket c begins.  The order inside a fine range depends on timing; the
// accumulate sums in integers, so its result does not.
// A pair whose two entries lie in DIFFERENT windows (x + 1 carries into bit 13: only where a level's resolution exceeds 8 192)
// keeps its first entry as a single-entry record; the second entry's share goes to `overflow_table` by atomics and the overflow
// flag is set, exactly as a record that did not fit the workspace (emit_pairs' fallback): correct, and slow only there.
__global__ void __launch_bounds__(512) k_bin_split(const Rec *__restrict__ recs_c, const uint32_t *__restrict__ starts_c, uint32_t cap_c,
                                                   Rec *__restrict__ recs_f, uint32_t *__restrict__ starts_f, int nbins_c, int wshift,
                                                   float *__restrict__ overflow_table, int NB_c, int bucket_log, int T)
{
    __shared__ uint32_t cnt[8], base[8];
    const int c = blockIdx.x, nwin = 1 << wshift;   // (wshift <= 3)
    const uint32_t lo = min(starts_c[c], cap_c), hi = min(starts_c[c + 1], cap_c);
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
    if (c == 0 && threadIdx.x == 0) {   // the words in front of the fine records: levels left out, format = Rec12, overflow flag
        *skip_word(recs_f) = *skip_word(const_cast<Rec *>(recs_c));
        *format_word(recs_f) = 2u;
        atomicOr(overflow_flag(recs_f), *overflow_flag(const_cast<Rec *>(recs_c)));   // (zeroed by the host before this launch; other workgroups may set it)
    }
    __syncthreads();
    const float4 *r4 = reinterpret_cast<const float4 *>(recs_c);
    uint32_t cn[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 512) {
        const uint32_t w = (__float_as_uint(r4[i].x) & 0xffffu) >> 13;
#pragma unroll
        for (int k = 0; k < 8; ++k) cn[k] += w == (uint32_t)k ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t v = cn[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&cnt[k], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = lo;
        for (int k = 0; k < nwin; ++k) {
            base[k] = run;
            starts_f[(size_t)c * nwin + k] = run;
            run += cnt[k];
        }
        if (c == nbins_c - 1) starts_f[(size_t)nbins_c * nwin] = hi;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i0 = lo + (threadIdx.x & ~63u); i0 < hi; i0 += 512) {   // (whole waves iterate together: the ballots need every lane)
        const uint32_t i = i0 + lane;
        const bool live = i < hi;
        float4 r = make_float4(0, 0, 0, 0);
        if (live) r = r4[i];
        const uint32_t hdr = __float_as_uint(r.x), l0 = hdr & 0xffffu, l1 = hdr >> 16;
        const uint32_t w = l0 >> 13;
        uint32_t pos = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint64_t m = __ballot(live && w == (uint32_t)k);
            if (m == 0) continue;   // (wave-uniform)
            uint32_t b = 0;
            if (lane == (uint32_t)__builtin_ctzll(m)) b = atomicAdd(&base[k], (uint32_t)__builtin_popcountll(m));
            b = (uint32_t)__shfl((int)b, __builtin_ctzll(m), 64);
            if (live && w == (uint32_t)k) pos = b + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        }
        if (live) {
            const uint32_t xm = l0 ^ l1;   // 2^(k+1) - 1 for an x-neighbour pair, 0 for a single entry
            if (xm >> 13) {   // the pair crosses a window: first entry here with its weight, second entry through the overflow table
                const float w1 = r.y, w0 = 1.0f - w1;
                store_rec12(recs_f, pos, l0 & 0x1fffu, 15u, 0u, w0 * r.z, w0 * r.w);
                float *gs = overflow_table + ((size_t)(c / NB_c) * T + ((size_t)(c % NB_c) << bucket_log)) * 2;
                unsafeAtomicAdd(gs + 2 * l1, w1 * r.z);
                unsafeAtomicAdd(gs + 2 * l1 + 1, w1 * r.w);
                atomicOr(overflow_flag(recs_f), 1u);
            } else {
                const uint32_t kk = xm ? (uint32_t)(31 - __clz((int)xm)) : 15u;
                const uint32_t t = xm ? (uint32_t)min(__float2int_rn(r.y * 8388608.0f), 8388607) : 0u;
                store_rec12(recs_f, pos, l0 & 0x1fffu, kk, t, r.z, r.w);
            }
        }
    }
}
}  // namespace

// ---- fused producer: plan (count + scan) before k_render_bwd, accumulate after it ---------------
// Workspace bytes of the fused table-gradient path of scanerf_render_backward; 0 => shape unsupported
// (use dfeat + scanerf_embedding_bg_backward_binned instead).
// ---- large tables (buckets above 2^13 entries): the SPLIT pass ------------------------------------------------------------
// The backward's LDS holds 256 cursors per level, so above 2^21 entries per level its buckets (T / 256 entries) outgrow the
// accumulate's LDS image (2^13 entries).  Round 1 accumulated such a bucket in windows, every window pass re-reading ALL of the
// bucket's records (T = 2^24: 8 passes, 6.9 ms per 16 384-ray step), and the training step went through dfeat + the stand-alone
// binned scatter instead.  Round 4: one pass over the coarse records partitions every bucket's records by their window
// (local entry >> 13) into a second record area -- as 12-byte records with 13-bit local entries, i.e. exactly the stream the
// accumulate's fast path reads -- and writes the fine ranges' starts; the accumulate then runs on 2^13-entry buckets as for
// small tables.  The second area lives in the same workspace, behind the budget of coarse records
// (scanerf_render_scatter_workspace_bytes sizes both); a workspace without it (a caller's smaller buffer) keeps the windows.
struct SplitLayout {
    size_t coarse_bytes;    // head + budget of 16-byte records
    size_t fine_off;        // offset of the fine area: [starts_f (nbins_f + 1)] ... [skip, format, overflow flag][records, 12 B each]
    size_t fine_recs_off;   // offset of its records
    size_t total_bytes;
    uint32_t budget;        // records
    int nbins_f;
};
static bool split_layout(int B, int S, const BinGeom &g, SplitLayout &L)
{
    if (g.bucket_log <= 13) return false;
    L.budget = (uint32_t)fused_record_budget(B, S);
    L.coarse_bytes = bin_workspace_head(16 * g.NB, g.W) + (size_t)L.budget * sizeof(Rec);
    L.nbins_f = 16 * (g.NB << (g.bucket_log - 13));
    L.fine_off = (L.coarse_bytes + 255) & ~(size_t)255;
    const size_t head_f = (((size_t)L.nbins_f + 1 + 3) * 4 + 255) & ~(size_t)255;
    L.fine_recs_off = L.fine_off + head_f;
    L.total_bytes = L.fine_recs_off + (size_t)L.budget * 12 + 256;
    return true;
}

SCANERF_API size_t scanerf_render_scatter_workspace_bytes(int B, int S, int T)
{
    BinGeom g;
    if (!fused_geom(B, S, T, g)) return 0;
    SplitLayout L;
    if (split_layout(B, S, g, L)) return L.total_bytes;
    return bin_workspace_head(16 * g.NB, g.W) + fused_record_budget(B, S) * sizeof(Rec);
}

SCANERF_API int scanerf_render_scatter_plan(const float *rays_o, const float *rays_d, const float *z_vals,
                                            const int32_t *resolutions, const scanerf_render_cfg *cfg,
                                            const uint8_t *ray_valid, int B, int S, int T, void *workspace,
                                            size_t workspace_bytes, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 1, "render_scatter_plan: B=%d S=%d", B, S);
    if (B == 0) return 0;
    BinGeom g;
    SCANERF_REQUIRE(cfg, "render_scatter_plan: cfg is null");
    SCANERF_REQUIRE(fused_geom(B, S, T, g, cfg->arith), "render_scatter_plan: shape B=%d S=%d T=%d not supported", B, S, T);
    SCANERF_REQUIRE(rays_o && rays_d && z_vals && resolutions && workspace, "render_scatter_plan: null pointer");
    SCANERF_REQUIRE(((uintptr_t)workspace & 15) == 0, "render_scatter_plan: workspace must be 16-byte aligned");
    const int nbins = 16 * g.NB;
    BinWorkspace w;
    SCANERF_REQUIRE(bin_workspace_carve(workspace, workspace_bytes, nbins, g.W, w),
                    "render_scatter_plan: workspace too small (%zu B)", workspace_bytes);
    RenderArgs f = {};
    f.rays_o = rays_o; f.rays_d = rays_d; f.z_vals = z_vals; f.resolutions = resolutions; f.ray_valid = ray_valid;
    f.B = B; f.S = S; f.T = T;
    f.contract_mode = cfg->contract_mode; f.infinity = cfg->infinity;
    // (only the t16 backward leaves masked levels' records out; the other two emit every level)
    f.skip_levels = ((cfg->arith == SCANERF_ARITH_T16 || cfg->arith == SCANERF_ARITH_T16S) && !tune_set("SCANERF_NO_LEVEL_SKIP")) ? pair_masked_levels(cfg->skip_levels) : 0u;
    for (int k = 0; k < 3; ++k) {
        f.min_bbox[k] = cfg->min_bbox[k];
        f.bbox_size[k] = cfg->bbox_size[k];
        f.inv_size4[k] = 4.0f / cfg->bbox_size[k];
    }
    hipStream_t st = (hipStream_t)stream;
    note_plan_format(workspace, g.rec8);
    hipLaunchKernelGGL(k_bin_count_rays, dim3(g.W), dim3(1024), (size_t)nbins * 4, st, f, g, w.counts, w.maxbits, overflow_flag(w.recs));
    hipLaunchKernelGGL(k_bin_rowscan, dim3(nbins), dim3(kThreads), 0, st, w.counts, w.totals, g.W);
    hipLaunchKernelGGL(k_bin_starts, dim3(1), dim3(1024), 0, st, w.totals, w.starts, nbins);
    return check_launch("render_scatter_plan");
}

// The plan in two halves around a forward launch that counts (render.hip k_render_fwd_h3<.., true>): attach = carve the
// workspace and point the kernel at its count matrix; finish = the scans.
namespace scanerf {
int scatter_plan_attach(void *workspace, size_t workspace_bytes, int B, int S, int T, int arith, int forward_grid, RenderArgs &a)
{
    BinGeom g;
    SCANERF_REQUIRE(fused_geom(B, S, T, g, arith), "render_forward_plan: shape B=%d S=%d T=%d not supported", B, S, T);
    SCANERF_REQUIRE(g.W == forward_grid && g.rpg == 8, "render_forward_plan: forward grid %d != backward grid %d", forward_grid, g.W);
    SCANERF_REQUIRE(((uintptr_t)workspace & 15) == 0, "render_forward_plan: workspace must be 16-byte aligned");
    BinWorkspace w;
    SCANERF_REQUIRE(bin_workspace_carve(workspace, workspace_bytes, 16 * g.NB, g.W, w),
                    "render_forward_plan: workspace too small (%zu B)", workspace_bytes);
    a.plan_counts = w.counts; a.plan_maxbits = w.maxbits; a.plan_overflow = overflow_flag(w.recs);
    a.plan_NB = g.NB; a.plan_bucket_log = g.bucket_log; a.plan_W = g.W; a.plan_rec8 = g.rec8;
    return 0;
}
int scatter_plan_finish(void *workspace, size_t workspace_bytes, int B, int S, int T, int arith, scanerf_stream_t stream)
{
    BinGeom g;
    BinWorkspace w;
    SCANERF_REQUIRE(fused_geom(B, S, T, g, arith) && bin_workspace_carve(workspace, workspace_bytes, 16 * g.NB, g.W, w),
                    "render_forward_plan: workspace / shape mismatch (B=%d S=%d T=%d, %zu B)", B, S, T, workspace_bytes);
    const int nbins = 16 * g.NB;
    hipStream_t st = (hipStream_t)stream;
    note_plan_format(workspace, g.rec8);
    hipLaunchKernelGGL(k_bin_rowscan, dim3(nbins), dim3(kThreads), 0, st, w.counts, w.totals, g.W);
    hipLaunchKernelGGL(k_bin_starts, dim3(1), dim3(1024), 0, st, w.totals, w.starts, nbins);
    return check_launch("render_forward_plan(scan)");
}
}  // namespace scanerf

// Large tables: partition the coarse records of `workspace` into its fine area (k_bin_split) and rewrite (g, w) to describe the
// fine stream (2^13-entry buckets, 12-byte records).  Returns false -- nothing launched, (g, w) untouched -- where the split does
// not apply: small tables, a workspace without the fine area, no table for the rare window-crossing pairs, or SCANERF_NO_SPLIT=1
// (the windows path stays, for A/B timing and as the fallback).
static bool split_to_fine(int B, int S, BinGeom &g, BinWorkspace &w, void *workspace, size_t workspace_bytes, float *overflow_table,
                          hipStream_t st)
{
    SplitLayout L;
    if (!split_layout(B, S, g, L) || workspace_bytes < L.total_bytes || !overflow_table || tune_set("SCANERF_NO_SPLIT")) return false;
    if (plan_format(workspace) != 0) return false;   // (coarse records of a large table are the 16-byte ones)
    char *base = static_cast<char *>(workspace);
    uint32_t *starts_f = reinterpret_cast<uint32_t *>(base + L.fine_off);
    Rec *recs_f = reinterpret_cast<Rec *>(base + L.fine_recs_off);
    const int nbins_c = 16 * g.NB, wshift = g.bucket_log - 13;
    const uint32_t cap_c = w.capacity < L.budget ? w.capacity : L.budget;
    (void)hipMemsetAsync(overflow_flag(recs_f), 0, 4, st);
    hipLaunchKernelGGL(k_bin_split, dim3(nbins_c), dim3(512), 0, st, w.recs, w.starts, cap_c, recs_f, starts_f, nbins_c, wshift,
                       overflow_table, g.NB, g.bucket_log, g.T);
    g.NB <<= wshift;
    g.bucket_log = 13;
    g.rec8 = 2;
    g.capacity = 0x7ffffff0u;   // (the fine ranges hold exactly the records the split wrote)
    w.recs = recs_f;
    w.starts = starts_f;
    return true;
}

// grad_features [16][T][2] += the records scanerf_render_backward emitted into `workspace`.
SCANERF_API int scanerf_render_scatter_accumulate(float *grad_features, int B, int S, int T, void *workspace,
                                                  size_t workspace_bytes, scanerf_stream_t stream)
{
    if (B == 0) return 0;
    BinGeom g;
    SCANERF_REQUIRE(fused_geom(B, S, T, g), "render_scatter_accumulate: shape B=%d S=%d T=%d not supported", B, S, T);
    SCANERF_REQUIRE(grad_features && workspace, "render_scatter_accumulate: null pointer");
    const int nbins = 16 * g.NB;
    BinWorkspace w;
    SCANERF_REQUIRE(bin_workspace_carve(workspace, workspace_bytes, nbins, g.W, w),
                    "render_scatter_accumulate: workspace too small (%zu B)", workspace_bytes);
    g.capacity = w.capacity = fused_coarse_capacity(w.capacity, B, S, g.bucket_log);
    g.rec8 = -1;  // as the plan recorded it in the workspace
    int nbins_acc = nbins;
    if (split_to_fine(B, S, g, w, workspace, workspace_bytes, grad_features, (hipStream_t)stream)) nbins_acc = 16 * g.NB;
    const size_t lds_bytes = (size_t)(2 << (g.bucket_log < 13 ? g.bucket_log : 13)) * 8;
    // launch shape by the records' format, as the Adam-epilogue entry does (12-byte records: 512 threads x 16 records per lane
    // 1.5 ms against 1.9 ms for the 16-byte records' 256 x 32 at configs[1]; 8-byte ones: 512 x 32)
    const int pf_acc = (nbins_acc != nbins) ? 2 : plan_format(workspace);
    const int variant = tune_int("SCANERF_ACC_VARIANT", pf_acc == 2 ? 8 : (pf_acc == 1 ? 10 : 0));   // (other shapes: experiments build)
#define SCANERF_LAUNCH_ACC(TH, UU)                                                                                  \
    {                                                                                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_accumulate<TH, UU>),               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);             \
        SCANERF_REQUIRE(e == hipSuccess, "render_scatter_accumulate: cannot reserve %zu B of LDS: %s", lds_bytes,    \
                        hipGetErrorString(e));                                                                      \
        hipLaunchKernelGGL((k_bin_accumulate<TH, UU>), dim3(nbins_acc), dim3(TH), lds_bytes, (hipStream_t)stream, w.recs, \
                           w.starts, w.maxbits, g, grad_features, AdamEpilogue{});                                  \
    }
#define SCANERF_LAUNCH_ACC_RUN(TH, UU)                                                                              \
    {                                                                                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_accumulate<TH, UU, true>),         \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);             \
        SCANERF_REQUIRE(e == hipSuccess, "render_scatter_accumulate: cannot reserve %zu B of LDS: %s", lds_bytes,    \
                        hipGetErrorString(e));                                                                      \
        hipLaunchKernelGGL((k_bin_accumulate<TH, UU, true>), dim3(nbins_acc), dim3(TH), lds_bytes, (hipStream_t)stream, \
                           w.recs, w.starts, w.maxbits, g, grad_features, AdamEpilogue{});                          \
    }
    // measured on MI355X (tools/bwd_emit_only.py + bench.py's table_grad_accumulate_adam section, 5.4e8 records = 8.6 GB): record i -> lane i (lane-interleaved) 256x8 3.61 ms,
    // 512x8 3.38, 1024x4 3.31; U consecutive records per lane 1024x4 2.21, 512x8 2.34, 1024x8 2.35, 1024x16 2.01-2.06,
    // 512x16 1.86, 256x16 1.86, 128x16 1.82, 512x32 1.95, 256x32 1.73-1.79 (default), 128x32 1.73, 64x32 1.78.  The
    // interleaved forms were bound by same-address serialisation in the LDS (coarse levels), not by the atomic rate itself
    // (5.8 distinct 64-bit adds per clock per CU: tools/lds_atomic_bench.hip); what is left is mostly the record stream.
#ifdef SCANERF_EXPERIMENTS
    if (variant == 1) SCANERF_LAUNCH_ACC(256, 8)
    else if (variant == 2) SCANERF_LAUNCH_ACC(512, 8)
    else if (variant == 3) SCANERF_LAUNCH_ACC(1024, 4)
    else if (variant == 4) SCANERF_LAUNCH_ACC_RUN(1024, 4)
    else if (variant == 5) SCANERF_LAUNCH_ACC_RUN(512, 8)
    else if (variant == 6) SCANERF_LAUNCH_ACC_RUN(1024, 8)
    else if (variant == 9) SCANERF_LAUNCH_ACC_RUN(256, 16)
    else if (variant == 11) SCANERF_LAUNCH_ACC_RUN(256, 32)
    else if (variant == 12) SCANERF_LAUNCH_ACC_RUN(128, 32)
    else if (variant == 13) SCANERF_LAUNCH_ACC_RUN(64, 32)
    else if (variant == 14) SCANERF_LAUNCH_ACC_RUN(128, 16)
    else if (variant == 7) SCANERF_LAUNCH_ACC_RUN(1024, 16)
    else
#endif
    if (variant == 8) SCANERF_LAUNCH_ACC_RUN(512, 16)
    else if (variant == 10) SCANERF_LAUNCH_ACC_RUN(512, 32)
    else SCANERF_LAUNCH_ACC_RUN(256, 32)
#undef SCANERF_LAUNCH_ACC
#undef SCANERF_LAUNCH_ACC_RUN
    return check_launch("render_scatter_accumulate");
}

// The records of one fused training step applied straight to the table: accumulate + fused sparse Adam in one pass
// (cuda/adam_kernel.cu:24-69 semantics per element: untouched if its gradient is exactly zero; pass the PREVIOUS step count).
// params / exp_avg / exp_avg_sq: [16][T][2] f32.  half_table (may be NULL): f16 / bf16 gather copy of params, refreshed for
// the touched entries.  overflow_grad (may be NULL): the [16][T][2] f32 table given to scanerf_render_backward as
// grad_features (only written if the record workspace overflowed); when the plan's overflow flag is set its entries are added
// to the gradient and re-zeroed, otherwise it is not touched -- it never needs a per-step zero-fill.
static int accumulate_adam(float *params, float *exp_avg, float *exp_avg_sq, void *half_table, int half_dtype,
                           float *overflow_grad, float lr, float beta1, float beta2, float eps, int step, int B, int S, int T,
                           void *workspace, size_t workspace_bytes, int S2, void *workspace2, size_t workspace2_bytes,
                           scanerf_stream_t stream)
{
    if (B == 0) return 0;
    BinGeom g;
    SCANERF_REQUIRE(fused_geom(B, S, T, g), "render_scatter_accumulate_adam: shape B=%d S=%d T=%d not supported", B, S, T);
    SCANERF_REQUIRE(params && exp_avg && exp_avg_sq && workspace, "render_scatter_accumulate_adam: null pointer");
    SCANERF_REQUIRE(!half_table || half_dtype == SCANERF_F16 || half_dtype == SCANERF_BF16,
                    "render_scatter_accumulate_adam: half_dtype=%d", half_dtype);
    const int nbins = 16 * g.NB;
    BinWorkspace w;
    SCANERF_REQUIRE(bin_workspace_carve(workspace, workspace_bytes, nbins, g.W, w),
                    "render_scatter_accumulate_adam: workspace too small (%zu B)", workspace_bytes);
    g.capacity = w.capacity = fused_coarse_capacity(w.capacity, B, S, g.bucket_log);
    g.rec8 = -1;  // as the plan recorded it in the workspace
    AdamEpilogue ad{ nullptr, nullptr, nullptr, 0u, params, exp_avg, exp_avg_sq, half_table, half_dtype, overflow_grad,
                     make_adam_args(lr, beta1, beta2, eps, step), 0 };
    BinGeom g2;
    BinWorkspace w2;
    if (workspace2) {  // the second branch's records: planned on the same B and T (same bins and producer grid), its own S
        SCANERF_REQUIRE(fused_geom(B, S2, T, g2) && g2.NB == g.NB && g2.W == g.W,
                        "render_scatter_accumulate_adam2: second record set B=%d S=%d T=%d does not match the first", B, S2, T);
        SCANERF_REQUIRE(bin_workspace_carve(workspace2, workspace2_bytes, nbins, g2.W, w2),
                        "render_scatter_accumulate_adam2: second workspace too small (%zu B)", workspace2_bytes);
        g2.capacity = w2.capacity = fused_coarse_capacity(w2.capacity, B, S2, g2.bucket_log);
    }
    // large tables: both record sets through the split pass (both or neither: they meet in one image of one geometry)
    int nbins_acc = nbins;
    bool split = false;
    {
        SplitLayout L1, L2;
        const bool can1 = split_layout(B, S, g, L1) && workspace_bytes >= L1.total_bytes && plan_format(workspace) == 0;
        const bool can2 = !workspace2 || (split_layout(B, S2, g2, L2) && workspace2_bytes >= L2.total_bytes && plan_format(workspace2) == 0);
        if (can1 && can2 && overflow_grad && !tune_set("SCANERF_NO_SPLIT")) {
            split = split_to_fine(B, S, g, w, workspace, workspace_bytes, overflow_grad, (hipStream_t)stream);
            if (split && workspace2) split_to_fine(B, S2, g2, w2, workspace2, workspace2_bytes, overflow_grad, (hipStream_t)stream);
            if (split) nbins_acc = 16 * g.NB;
        }
    }
    if (workspace2) { ad.recs2 = w2.recs; ad.starts2 = w2.starts; ad.maxbits2 = w2.maxbits; ad.capacity2 = split ? 0x7ffffff0u : w2.capacity; }
    const size_t lds_bytes = (size_t)(2 << (g.bucket_log < 13 ? g.bucket_log : 13)) * 8;
#define SCANERF_LAUNCH_ACC_ADAM(TH, UU)                                                                                \
    {                                                                                                                 \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_accumulate<TH, UU, true, true>),     \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);               \
        SCANERF_REQUIRE(e == hipSuccess, "render_scatter_accumulate_adam: cannot reserve %zu B of LDS: %s", lds_bytes, \
                        hipGetErrorString(e));                                                                        \
        hipLaunchKernelGGL((k_bin_accumulate<TH, UU, true, true>), dim3(nbins_acc), dim3(TH), lds_bytes, (hipStream_t)stream, \
                           w.recs, w.starts, w.maxbits, g, (float *)nullptr, ad);                                     \
    }
    // measured (configs[1], MI355X), threads x 16-byte loads per lane: 16-byte records 256x32 2.28 ms, 512x32 2.55, 768x32 2.59;
    // 8-byte records 256x32 1.77, 512x16 1.58, 1024x8 1.70, 1024x16 1.55, 768x32 1.51, 512x48 1.53, 512x32 1.47
    // 12-byte records (t16s), threads x records per lane: 256x32 1.72 ms, 512x32 1.74, 768x32 1.67, 1024x16 1.53, 512x8 1.52, 512x16 1.43
    const int pf = split ? 2 : plan_format(workspace);
    const int variant = tune_int("SCANERF_ACC_VARIANT", pf == 1 ? 4 : (pf == 2 ? 1 : 0));   // (other shapes: experiments build)
#ifdef SCANERF_EXPERIMENTS
    if (variant == 2) SCANERF_LAUNCH_ACC_ADAM(1024, 8)
    else if (variant == 3) SCANERF_LAUNCH_ACC_ADAM(256, 16)
    else if (variant == 5) SCANERF_LAUNCH_ACC_ADAM(1024, 16)
    else if (variant == 6) SCANERF_LAUNCH_ACC_ADAM(512, 8)
    else if (variant == 7) SCANERF_LAUNCH_ACC_ADAM(1024, 4)
    else if (variant == 8) SCANERF_LAUNCH_ACC_ADAM(768, 32)
    else if (variant == 9) SCANERF_LAUNCH_ACC_ADAM(512, 48)
    else if (variant == 10) SCANERF_LAUNCH_ACC_ADAM(640, 32)
    else if (variant == 11) SCANERF_LAUNCH_ACC_ADAM(768, 16)
    else
#endif
    if (variant == 1) SCANERF_LAUNCH_ACC_ADAM(512, 16)
    else if (variant == 4) SCANERF_LAUNCH_ACC_ADAM(512, 32)
    else SCANERF_LAUNCH_ACC_ADAM(256, 32)
#undef SCANERF_LAUNCH_ACC_ADAM
    return check_launch("render_scatter_accumulate_adam");
}

SCANERF_API int scanerf_render_scatter_accumulate_adam(float *params, float *exp_avg, float *exp_avg_sq, void *half_table,
                                                       int half_dtype, float *overflow_grad, float lr, float beta1,
                                                       float beta2, float eps, int step, int B, int S, int T,
                                                       void *workspace, size_t workspace_bytes, scanerf_stream_t stream)
{
    return accumulate_adam(params, exp_avg, exp_avg_sq, half_table, half_dtype, overflow_grad, lr, beta1, beta2, eps, step, B, S, T,
                           workspace, workspace_bytes, 0, nullptr, 0, stream);
}

// The same over TWO record sets (a tile's foreground and background branches, tile.py:639-692: each planned and emitted on
// its own workspace over the same B rays and table): both gradients meet in one image and ONE Adam step.
SCANERF_API int scanerf_render_scatter_accumulate_adam2(float *params, float *exp_avg, float *exp_avg_sq, void *half_table,
                                                        int half_dtype, float *overflow_grad, float lr, float beta1,
                                                        float beta2, float eps, int step, int B, int T, int S1,
                                                        void *workspace1, size_t workspace1_bytes, int S2, void *workspace2,
                                                        size_t workspace2_bytes, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(workspace2, "render_scatter_accumulate_adam2: second workspace is null");
    return accumulate_adam(params, exp_avg, exp_avg_sq, half_table, half_dtype, overflow_grad, lr, beta1, beta2, eps, step, B, S1, T,
                           workspace1, workspace1_bytes, S2, workspace2, workspace2_bytes, stream);
}

// ---- test infrastructure (tests/test_gpu_parity.py), not on the product path: the Rec8 codec on its own -------------------
namespace {
__global__ void k_rec8_selftest(const float *gx, const float *gy, const float *tx, const uint32_t *l0, const uint32_t *kk, int n,
                                uint32_t *words, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = kk[i] == 15u ? 0u : (uint32_t)min(__float2int_rn(tx[i] * 8192.0f), 8191);
    const uint2 r = pack_rec8(l0[i], kk[i], t, gx[i], gy[i]);
    words[2 * i] = r.x;
    words[2 * i + 1] = r.y;
    const Rec8Fields f = unpack_rec8(r.x, r.y);
    float *o = out + 8 * (size_t)i;  // l0, l1, then the four contributions (x, y to l0; x, y to l1), E - 25, t
    o[0] = (float)f.l0;
    o[1] = (float)f.l1;
    o[2] = ldexpf((float)(f.mx * (8192 - f.t)), f.e25);
    o[3] = ldexpf((float)(f.my * (8192 - f.t)), f.e25);
    o[4] = ldexpf((float)(f.mx * f.t), f.e25);
    o[5] = ldexpf((float)(f.my * f.t), f.e25);
    o[6] = (float)f.e25;
    o[7] = (float)f.t;
}
}  // namespace

SCANERF_API int scanerf_rec8_selftest(const float *gx, const float *gy, const float *tx, const uint32_t *l0, const uint32_t *k,
                                      int n, uint32_t *words, float *out, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(gx && gy && tx && l0 && k && words && out && n >= 0, "rec8_selftest: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_rec8_selftest, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, gx, gy, tx, l0, k, n, words, out);
    return check_launch("rec8_selftest");
}
