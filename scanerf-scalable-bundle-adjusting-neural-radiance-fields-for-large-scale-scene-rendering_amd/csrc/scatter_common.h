// scatter_common.h -- record format, bin geometry and workspace layout of the atomic-free table-gradient
// scatter, shared by scatter.hip (count / scan / accumulate, stand-alone producer) and render_bwd.hip
// (which emits the records straight from the fused backward kernel).
#pragma once
#include "hashgrid_common.h"

namespace scanerf {

constexpr int kBucketLog = 11;  // 2048 entries (16 KB of fp32 pairs) per bucket
// Fused producer: 2^13 entries per bucket = the largest the accumulate's 64-bit LDS image holds (128 KB).  Every workgroup
// appends to 16 * NB ranges at once; each XCD's L2 (4 MiB) has to hold one partially written 128-B line per range of its
// 32 workgroups to merge the 16-byte records into full-line writes: 2^11-entry buckets = 16 MiB of such lines per XCD,
// 2^13 = 4 MiB.  Measured (two-waves-per-SIMD backward kernel, T = 2^19): plan + backward 5.80 / 5.45 / 5.17 / 5.12 ms at
// 2^11 / 2^12 / 2^13 / 2^14 entries, accumulate 1.75 / 1.78 / 1.83 / 3.9 ms (above 2^13 it works in windows).
constexpr int kFusedBucketLog = 13;

struct BinGeom {
    int N, L, T;
    int bucket_log;   // min(kBucketLog, log2 T)
    int NB;           // buckets per level = T >> bucket_log
    int W;            // producer workgroups
    int per_wg;       // samples per producer workgroup
    int rpg;          // fused producer: rays per workgroup visit (ray = (wg + i*W)*rpg + r): 1 f32 kernel, 4 h3 kernel
    uint32_t capacity;  // records that fit the workspace
    int dbg;
    int rows16;       // stand-alone producer: point-major gradient rows of 16 levels are read once and emitted level by level (host's decision; the kernel follows it)
    int rec8;         // record format: 0 = Rec (16 bytes), 1 = Rec8 (8 bytes), 2 = Rec12 (12 bytes), 3 = Rec12 in 64-byte segments of five (kSegRecs), -1 = read format_word() (accumulate of a fused plan)
};
// Format 3 (round 6, large tables): a (bucket, workgroup) range is a whole number of 64-BYTE SEGMENTS, each holding five Rec12
// records (60 bytes) and one spare word; counts, starts and capacities are in segments.  The stand-alone producer fills a
// segment in its LDS and writes it with ONE aligned 64-byte store (k_bin_scatter_seg); unused record slots of a range's last
// segment are all-zero words, which the accumulate skips (a record whose two gradient words are zero adds nothing).
constexpr int kSegRecs = 5;
// (BinGeom.dbg: timing experiments of a -DSCANERF_EXPERIMENTS build only -- wrong results; always 0 in the product build)

struct Rec {
    uint32_t hdr;
    float tx, gx, gy;
};
// The word right before the records (bin_workspace_head reserves it).  Set (non-zero) when a record did not fit the workspace
// and went to grad_features through atomics instead (emit_pairs / commit_pairs fallback); zeroed by the plan.  Lets the
// Adam-fused accumulate skip the overflow table in the common case.
__host__ __device__ inline uint32_t *overflow_flag(Rec *recs) { return reinterpret_cast<uint32_t *>(recs) - 1; }

// The word before that: the record format the plan chose (0 = Rec, 1 = Rec8); the accumulate of a fused plan reads it, so
// plan, producer and consumer cannot disagree about the stream they share.
__host__ __device__ inline uint32_t *format_word(Rec *recs) { return reinterpret_cast<uint32_t *>(recs) - 2; }
// ... and before that: the levels the plan left out (coarse-to-fine mask: their gradients are exactly zero); the t16 backward
// reads it here and emits no records for them.
__host__ __device__ inline uint32_t *skip_word(Rec *recs) { return reinterpret_cast<uint32_t *>(recs) - 3; }

// One 16-byte record store.  Plain (write-back) stores: measured in the two-waves-per-SIMD backward kernel, non-temporal
// stores of the same records take 2.7x the kernel's time (14.2 vs 5.2 ms) and sc1 (write-through) ones 1.5x: the records of
// a (bin, workgroup) range are written 16 bytes at a time and only the L2 can merge them into full lines.  What helps is
// FEWER ranges being filled at once (kFusedBucketLog).
__device__ __forceinline__ void store_rec(Rec *recs, uint32_t pos, uint32_t hdr, float tx, float gx, float gy)
{
    reinterpret_cast<float4 *>(recs)[pos] = make_float4(__uint_as_float(hdr), tx, gx, gy);
}

// the 4 (y,z) corner pairs of one (point, level): bucket, locals, weights
struct Pairs {
    uint32_t idx0[4], idx1[4];
    uint32_t xm;  // idx0[q] ^ idx1[q] for every q: (x ^ (x+1)) & mask
    float wyz[4], tx;
};

__device__ __forceinline__ void make_pairs(const float p[3], const int32_t *res, uint32_t mask, Pairs &pr)
{
    int b[3];
    float t[3], sc;
    locate_bg(p[0], res[0], b[0], t[0], sc);
    locate_bg(p[1], res[1], b[1], t[1], sc);
    locate_bg(p[2], res[2], b[2], t[2], sc);
    const uint32_t hx0 = (uint32_t)b[0], hx1 = (uint32_t)(b[0] + 1);
    const uint32_t hy[2] = { (uint32_t)b[1] * 2654435761u, (uint32_t)(b[1] + 1) * 2654435761u };
    const uint32_t hz[2] = { (uint32_t)b[2] * 805459861u, (uint32_t)(b[2] + 1) * 805459861u };
    const float wy[2] = { 1 - t[1], t[1] }, wz[2] = { 1 - t[2], t[2] };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int dy = q >> 1, dz = q & 1;
        const uint32_t c = hy[dy] ^ hz[dz];
        pr.idx0[q] = (hx0 ^ c) & mask;
        pr.idx1[q] = (hx1 ^ c) & mask;
        pr.wyz[q] = wy[dy] * wz[dz];
    }
    pr.tx = t[0];
    pr.xm = (hx0 ^ hx1) & mask;
}


// Append the records of one (sample, level) with upstream gradient (gix, giy): one per (y,z) corner pair, two when the
// x-neighbours fall into different buckets.
// cursor_level: this workgroup's LDS cursors of the level's NB bins; grad_level: the level's slice of grad_features, touched
// only when the workspace is too small (slow path, correctness only).
// Shape of the code (it runs inside the backward kernel at one wave per SIMD, where every divergent branch is a bubble):
// idx0 ^ idx1 = (x ^ (x+1)) & mask for all four pairs, so "straddles a bucket boundary" is ONE test per (sample, level);
// the common case is 4 cursor atomics issued back to back and 4 predicated 16-B stores, no branch; straddling lanes and
// workspace overflow share one rarely taken, wave-uniform branch.
template <int DBG = 0>  // timing experiments only (-DSCANERF_BWD_EXPERIMENTS): 1 = no record stores, 2 = no cursor atomics, 3 = neither, 4 / 8 = store shapes
__device__ __forceinline__ void emit_pairs(const Pairs &pr, float gix, float giy, uint32_t *cursor_level, int bucket_log,
                                           uint32_t capacity, Rec *recs, float *grad_level)
{
    const uint32_t lmask = (1u << bucket_log) - 1u;
    const bool straddle = (pr.xm >> bucket_log) != 0u;
    const float a0 = 1.0f - pr.tx;
    auto fallback = [&](uint32_t bkt, uint32_t hdr, float tx, float ax, float ay) {
        *overflow_flag(recs) = 1u;
        float *gs = grad_level + ((size_t)bkt << bucket_log) * 2;
        const uint32_t e0 = hdr & 0xffffu, e1 = hdr >> 16;
        unsafeAtomicAdd(gs + 2 * e0, (1.0f - tx) * ax);
        unsafeAtomicAdd(gs + 2 * e0 + 1, (1.0f - tx) * ay);
        unsafeAtomicAdd(gs + 2 * e1, tx * ax);
        unsafeAtomicAdd(gs + 2 * e1 + 1, tx * ay);
    };
    uint32_t pos[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#ifdef SCANERF_BWD_EXPERIMENTS
        if (DBG == 2 || DBG == 3) { pos[q] = cursor_level[pr.idx0[q] >> bucket_log] + (threadIdx.x & 15); continue; }
#endif
        pos[q] = atomicAdd(&cursor_level[pr.idx0[q] >> bucket_log], 1u);
    }
    const float txr = straddle ? 0.0f : pr.tx;
    bool rare = straddle;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t l0 = pr.idx0[q] & lmask;
        const uint32_t hdr = straddle ? l0 * 0x10001u : (l0 | ((l0 ^ pr.xm) << 16));
        const float gx = pr.wyz[q] * gix, gy = pr.wyz[q] * giy;
        const float ax = straddle ? a0 * gx : gx, ay = straddle ? a0 * gy : gy;
#ifdef SCANERF_BWD_EXPERIMENTS
        if (DBG & 1) {
            if (pos[q] == 0xffffffffu && ax == 1.2345f) reinterpret_cast<float4 *>(recs)[0] = make_float4(__uint_as_float(hdr), txr, ax, ay);
        } else if (DBG == 4) {  // 8-byte stores at 8-byte stride (is the cost per request or per byte?)
            if (pos[q] < capacity) reinterpret_cast<float2 *>(recs)[pos[q]] = make_float2(__uint_as_float(hdr) + txr, ax + ay);
        } else if (DBG == 8) {  // 16-byte stores, every lane group's 16 records contiguous whatever the bins (is it the scatter?)
            reinterpret_cast<float4 *>(recs)[(size_t)blockIdx.x * 65536 + (threadIdx.x & 1023) * 4 + q] = make_float4(__uint_as_float(hdr), txr, ax, ay);
        } else
#endif
        if (pos[q] < capacity) store_rec(recs, pos[q], hdr, txr, ax, ay);
        rare |= pos[q] >= capacity;
    }
    if (DBG == 0 && __builtin_expect(__any(rare), 0)) {
#pragma unroll  // (rolled, pr would be indexed dynamically and live in scratch -- whose stores cost a vmcnt(0) per level)
        for (int q = 0; q < 4; ++q) {
            const uint32_t l0 = pr.idx0[q] & lmask, b0 = pr.idx0[q] >> bucket_log;
            const float gx = pr.wyz[q] * gix, gy = pr.wyz[q] * giy;
            if (pos[q] >= capacity) {
                if (straddle) fallback(b0, l0 * 0x10001u, 0.0f, a0 * gx, a0 * gy);
                else fallback(b0, l0 | ((l0 ^ pr.xm) << 16), pr.tx, gx, gy);
            }
            if (straddle) {  // second record: the x+1 neighbour in its own bucket
                const uint32_t i1 = pr.idx0[q] ^ pr.xm, b1 = i1 >> bucket_log, hdr1 = (i1 & lmask) * 0x10001u;
                const uint32_t p1 = atomicAdd(&cursor_level[b1], 1u);
                if (p1 < capacity) store_rec(recs, p1, hdr1, 0.0f, pr.tx * gx, pr.tx * gy);
                else fallback(b1, hdr1, 0.0f, pr.tx * gx, pr.tx * gy);
            }
        }
    }
}
// emit_pairs in two halves, so that a caller can have the cursor round trips of one (sample, level) in flight while it
// stores the records of the previous one: reserve = the four cursor atomics; commit = records + rare paths.
struct PairSlots {
    uint32_t pos[4];
};
__device__ __forceinline__ void reserve_pairs(const Pairs &pr, uint32_t *cursor_level, int bucket_log, PairSlots &sl)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#if defined(T16_EMIT_DBG) && (T16_EMIT_DBG & 2)   // timing experiments only: no cursor atomics
        sl.pos[q] = cursor_level[pr.idx0[q] >> bucket_log] + (threadIdx.x & 15);
        continue;
#endif
        sl.pos[q] = atomicAdd(&cursor_level[pr.idx0[q] >> bucket_log], 1u);
    }
}
__device__ __forceinline__ void commit_pairs(const Pairs &pr, const PairSlots &sl, float gix, float giy, uint32_t *cursor_level,
                                             int bucket_log, uint32_t capacity, Rec *recs, float *grad_level)
{
    const uint32_t lmask = (1u << bucket_log) - 1u;
    const bool straddle = (pr.xm >> bucket_log) != 0u;
    const float a0 = 1.0f - pr.tx;
    auto fallback = [&](uint32_t bkt, uint32_t hdr, float tx, float ax, float ay) {
        *overflow_flag(recs) = 1u;
        float *gs = grad_level + ((size_t)bkt << bucket_log) * 2;
        const uint32_t e0 = hdr & 0xffffu, e1 = hdr >> 16;
        unsafeAtomicAdd(gs + 2 * e0, (1.0f - tx) * ax);
        unsafeAtomicAdd(gs + 2 * e0 + 1, (1.0f - tx) * ay);
        unsafeAtomicAdd(gs + 2 * e1, tx * ax);
        unsafeAtomicAdd(gs + 2 * e1 + 1, tx * ay);
    };
    const float txr = straddle ? 0.0f : pr.tx;
    bool rare = straddle;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t l0 = pr.idx0[q] & lmask;
        const uint32_t hdr = straddle ? l0 * 0x10001u : (l0 | ((l0 ^ pr.xm) << 16));
        const float gx = pr.wyz[q] * gix, gy = pr.wyz[q] * giy;
        const float ax = straddle ? a0 * gx : gx, ay = straddle ? a0 * gy : gy;
        if (sl.pos[q] < capacity) store_rec(recs, sl.pos[q], hdr, txr, ax, ay);
        rare |= sl.pos[q] >= capacity;
    }
    if (__builtin_expect(__any(rare), 0)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t l0 = pr.idx0[q] & lmask, b0 = pr.idx0[q] >> bucket_log;
            const float gx = pr.wyz[q] * gix, gy = pr.wyz[q] * giy;
            if (sl.pos[q] >= capacity) {
                if (straddle) fallback(b0, l0 * 0x10001u, 0.0f, a0 * gx, a0 * gy);
                else fallback(b0, l0 | ((l0 ^ pr.xm) << 16), pr.tx, gx, gy);
            }
            if (straddle) {
                const uint32_t i1 = pr.idx0[q] ^ pr.xm, b1 = i1 >> bucket_log, hdr1 = (i1 & lmask) * 0x10001u;
                const uint32_t p1 = atomicAdd(&cursor_level[b1], 1u);
                if (p1 < capacity) store_rec(recs, p1, hdr1, 0.0f, pr.tx * gx, pr.tx * gy);
                else fallback(b1, hdr1, 0.0f, pr.tx * gx, pr.tx * gy);
            }
        }
    }
}

// ---- 8-byte records: the stream of the t16 backward (render_bwd_t16.hip) -----------------------------------------------
// That kernel's gradient products run on f16 matrix operands (11-bit significands, ~5e-4 of the largest gradient); carrying
// its records as three f32 spends 8.6 GB per configs[1] step (written, then read) on digits the values do not have.  Rec8
// keeps 12 significant bits + sign per component under the pair's own 8-bit exponent (no launch-wide scale to guess) and
// the x-weight in 13 bits:
//   word0: l0 [12:0] | k [16:13] | t [29:17] | exponent bits 1:0 [31:30]
//   word1: mx [12:0] | my [25:13] (two's complement) | exponent bits 7:2 [31:26]
//   (gx, gy) = (mx, my) * 2^(E - 12), E = exponent - 127 = frexp exponent of max(|gx|, |gy|) * (1 + 2^-13) (clamped to >= -127: smaller
//   values lose bits gradually); entry l0 gets weight (8192 - t) / 8192, entry l1 = l0 ^ ((2 << k) - 1) -- the hash of x + 1
//   differs from that of x in the k + 1 low bits, k = trailing ones of x -- gets t / 8192; k = 15: no second entry (t = 0).
// Rounding: half a unit of the significand (2^-13 .. 2^-12 of the larger component) and 2^-14 in the weight -- measured through
// the fused path: 1.3e-4 relative L2 against 7e-4 for the products' own noise (tests/test_gpu_parity.py); the
// accumulate (k_bin_accumulate) works on the integers, so the sum stays bit-reproducible.  Buckets of at most 2^13 entries.
// (Not representable: inf / NaN gradients -- they come out as finite garbage instead of poisoning the entry as f32 records would;
// the loss of such a step is already non-finite.)
constexpr int kRec8MaxBucketLog = 13;
// record format of a fused plan by the backward's arithmetic: 1 (Rec8) for T16, 2 (Rec12) for T16S, else 0 (Rec, 16 bytes)
inline int fused_rec8(int arith, int bucket_log)
{
    if (tune_set("SCANERF_REC16")) return 0;  // experiments build: the 16-sample-tile kernels on 16-byte records
    if (bucket_log > kRec8MaxBucketLog) return 0;
    return arith == 2 /* SCANERF_ARITH_T16 */ ? 1 : (arith == 3 /* SCANERF_ARITH_T16S */ ? 2 : 0);
}
__device__ __forceinline__ uint2 pack_rec8(uint32_t l0, uint32_t k, uint32_t t, float gx, float gy)
{
    // max < 2^E (0 for g = 0); taken of max * (1 + 2^-13) so that a significand which would round up to 4096 moves to the
    // next exponent instead (the min below never bites)
    const float mabs = fmaxf(fabsf(gx), fabsf(gy));
    int E = __builtin_amdgcn_frexp_expf(fmaf(mabs, 0x1p-13f, mabs));
    E = E < -127 ? -127 : E;   // (frexp: E <= 128)
    const int mx = min(__float2int_rn(__builtin_amdgcn_ldexpf(gx, 12 - E)), 4095);
    const int my = min(__float2int_rn(__builtin_amdgcn_ldexpf(gy, 12 - E)), 4095);
    const uint32_t e = (uint32_t)(E + 127);
    return make_uint2(l0 | (k << 13) | (t << 17) | (e << 30), ((uint32_t)mx & 0x1fffu) | (((uint32_t)my & 0x1fffu) << 13) | ((e >> 2) << 26));
}
// the fields of a Rec8 as the accumulate uses them: entries l0 / l1 (l1 >= 2^13: none), weight t of l1 (8192 - t of l0),
// significands and E - 25 (value = m * weight * 2^(E - 25))
struct Rec8Fields {
    uint32_t l0, l1;
    int t, mx, my, e25;
};
__device__ __forceinline__ Rec8Fields unpack_rec8(uint32_t w0, uint32_t w1)
{
    Rec8Fields f;
    f.l0 = w0 & 0x1fffu;
    f.l1 = f.l0 ^ ((2u << ((w0 >> 13) & 15u)) - 1u);
    f.t = (int)((w0 >> 17) & 0x1fffu);
    f.e25 = (int)((w0 >> 30) | ((w1 >> 26) << 2)) - 127 - 25;
    f.mx = (int)(w1 << 19) >> 19;
    f.my = (int)(w1 << 6) >> 19;
    return f;
}
__device__ __forceinline__ void store_rec8(Rec *recs, uint32_t pos, uint2 r) { reinterpret_cast<uint2 *>(recs)[pos] = r; }

// emit_pairs for the Rec8 stream: same ranges, same cursors, same rare paths (count_pairs is its histogram as well);
// capacity counts 8-byte records here.
template <int DBG = 0>  // timing experiments only (-DSCANERF_BWD_EXPERIMENTS): 5 = records not packed (raw bits)
__device__ __forceinline__ void emit_pairs8(const Pairs &pr, float gix, float giy, uint32_t *cursor_level, int bucket_log,
                                            uint32_t capacity, Rec *recs, float *grad_level)
{
    const uint32_t lmask = (1u << bucket_log) - 1u;
    const bool straddle = (pr.xm >> bucket_log) != 0u;
    const float a0 = 1.0f - pr.tx;
    auto fallback = [&](uint32_t bkt, uint32_t e0, uint32_t e1, float tx, float ax, float ay) {
        *overflow_flag(recs) = 1u;
        float *gs = grad_level + ((size_t)bkt << bucket_log) * 2;
        unsafeAtomicAdd(gs + 2 * e0, (1.0f - tx) * ax);
        unsafeAtomicAdd(gs + 2 * e0 + 1, (1.0f - tx) * ay);
        unsafeAtomicAdd(gs + 2 * e1, tx * ax);
        unsafeAtomicAdd(gs + 2 * e1 + 1, tx * ay);
    };
    uint32_t pos[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pos[q] = atomicAdd(&cursor_level[pr.idx0[q] >> bucket_log], 1u);
    // k = trailing ones of x: xm = 2^(k+1) - 1 (below the bucket size unless the pair straddles)
    const uint32_t k = straddle ? 15u : (uint32_t)(31 - __clz((int)pr.xm));
    const uint32_t t = straddle ? 0u : (uint32_t)min(__float2int_rn(pr.tx * 8192.0f), 8191);
    bool rare = straddle;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float gx = pr.wyz[q] * gix, gy = pr.wyz[q] * giy;
        const float ax = straddle ? a0 * gx : gx, ay = straddle ? a0 * gy : gy;
#ifdef SCANERF_BWD_EXPERIMENTS
        if (DBG == 5) {
            if (pos[q] < capacity) store_rec8(recs, pos[q], make_uint2((pr.idx0[q] & lmask) | (k << 13) | (t << 17), __float_as_uint(ax) ^ __float_as_uint(ay)));
        } else
#endif
        if (pos[q] < capacity) store_rec8(recs, pos[q], pack_rec8(pr.idx0[q] & lmask, k, t, ax, ay));
        rare |= pos[q] >= capacity;
    }
    if (__builtin_expect(__any(rare), 0)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t l0 = pr.idx0[q] & lmask, b0 = pr.idx0[q] >> bucket_log;
            const float gx = pr.wyz[q] * gix, gy = pr.wyz[q] * giy;
            if (pos[q] >= capacity) {
                if (straddle) fallback(b0, l0, l0, 0.0f, a0 * gx, a0 * gy);
                else fallback(b0, l0, l0 ^ pr.xm, pr.tx, gx, gy);
            }
            if (straddle) {  // second record: the x+1 neighbour in its own bucket
                const uint32_t i1 = pr.idx0[q] ^ pr.xm, b1 = i1 >> bucket_log, l1 = i1 & lmask;
                const uint32_t p1 = atomicAdd(&cursor_level[b1], 1u);
                if (p1 < capacity) store_rec8(recs, p1, pack_rec8(l1, 15u, 0u, pr.tx * gx, pr.tx * gy));
                else fallback(b1, l1, l1, 0.0f, pr.tx * gx, pr.tx * gy);
            }
        }
    }
}

// ---- 12-byte records: the stream of the split-gradient 16-sample-tile backward ("t16s") ---------------------------------
// Its gradients are f32-equivalent, so the records keep f32 accuracy -- in 12 instead of 16 bytes (6.4 instead of 8.6 GB per
// configs[1] step, written and read back):
//   word0: l0 [12:0] | k [16:13] | t bits 22:8 [31:17]        (l1 = l0 ^ ((2 << k) - 1), k = 15: no second entry, as Rec8)
//   word1: gx as f32, mantissa rounded to 19 bits; its low 4 bits carry t bits 7:4
//   word2: gy likewise; low 4 bits = t bits 3:0
// t = round(tx * 2^23) (23 bits: weight error 2^-24), gradient components 2^-21 relative.
// Round to nearest on 19 mantissa bits, on the bits (add half a unit, clear the low four).  FINITE inputs only mean what they say:
// +-inf stays +-inf, but a NaN may come out as an infinity, a zero or a finite value (a payload in the low bits is cleared; one
// near 0x7fffff carries into exponent and sign), so a non-finite gradient does not reliably poison its table entry as a 16-byte
// record would -- as with Rec8, the loss of such a step is non-finite already and is what a caller has to test.  Testing for the
// exponent 0xff here would cost ~5 vector instructions per component in the backward kernel's emission (~6 % of its vector work).
// A finite value just below a power of two may round UP to it: the accumulate takes the launch maximum one 19-bit unit larger
// for this format (k_bin_accumulate: `M` for fmt == 2), so the fixed-point conversion's |v * 2^k| < 2^51 holds for rounded values.
__device__ __forceinline__ uint32_t rec12_round(float g) { return (__float_as_uint(g) + 8u) & ~15u; }
__device__ __forceinline__ void store_rec12(Rec *recs, uint32_t pos, uint32_t l0, uint32_t k, uint32_t t, float gx, float gy)
{
    struct __attribute__((aligned(4))) W3 { uint32_t a, b, c; };   // (one global_store_dwordx3; 4-byte aligned)
    *reinterpret_cast<W3 *>(reinterpret_cast<uint32_t *>(recs) + (size_t)pos * 3) =
        W3{ l0 | (k << 13) | ((t >> 8) << 17), rec12_round(gx) | ((t >> 4) & 15u), rec12_round(gy) | (t & 15u) };
}
struct Rec12Fields {
    uint32_t l0, l1;
    float w1, gx, gy;   // weight of l1 (1 - w1 of l0)
};
__device__ __forceinline__ Rec12Fields unpack_rec12(uint32_t w0, uint32_t w1, uint32_t w2)
{
    Rec12Fields f;
    f.l0 = w0 & 0x1fffu;
    f.l1 = f.l0 ^ ((2u << ((w0 >> 13) & 15u)) - 1u);
    const uint32_t t = ((w0 >> 17) << 8) | ((w1 & 15u) << 4) | (w2 & 15u);
    f.w1 = (float)t * 0x1p-23f;
    f.gx = __uint_as_float(w1 & ~15u);
    f.gy = __uint_as_float(w2 & ~15u);
    return f;
}
// emit_pairs for the Rec12 stream: same ranges, same cursors, same rare paths; capacity counts 12-byte records here.  In two
// halves (as reserve_pairs / commit_pairs): the four cursor atomics, then the records -- a caller can have the round trips of
// several (sample, level)s in flight before it stores anything.
__device__ __forceinline__ void commit_pairs12(const Pairs &pr, const PairSlots &sl, float gix, float giy, uint32_t *cursor_level,
                                               int bucket_log, uint32_t capacity, Rec *recs, float *grad_level)
{
    const uint32_t *pos = sl.pos;
    const uint32_t lmask = (1u << bucket_log) - 1u;
    const bool straddle = (pr.xm >> bucket_log) != 0u;
    const float a0 = 1.0f - pr.tx;
    auto fallback = [&](uint32_t bkt, uint32_t e0, uint32_t e1, float tx, float ax, float ay) {
        *overflow_flag(recs) = 1u;
        float *gs = grad_level + ((size_t)bkt << bucket_log) * 2;
        unsafeAtomicAdd(gs + 2 * e0, (1.0f - tx) * ax);
        unsafeAtomicAdd(gs + 2 * e0 + 1, (1.0f - tx) * ay);
        unsafeAtomicAdd(gs + 2 * e1, tx * ax);
        unsafeAtomicAdd(gs + 2 * e1 + 1, tx * ay);
    };
    const uint32_t k = straddle ? 15u : (uint32_t)(31 - __clz((int)pr.xm));
    const uint32_t t = straddle ? 0u : (uint32_t)min(__float2int_rn(pr.tx * 8388608.0f), 8388607);
    bool rare = straddle;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float gx = pr.wyz[q] * gix, gy = pr.wyz[q] * giy;
        const float ax = straddle ? a0 * gx : gx, ay = straddle ? a0 * gy : gy;
#if defined(T16_EMIT_DBG) && (T16_EMIT_DBG & 1)   // timing experiments only: no record stores (the values stay live)
        if (pos[q] == 0xffffffffu && ax == 1.2345f) store_rec12(recs, 0, pr.idx0[q] & lmask, k, t, ax, ay);
#elif defined(T16_EMIT_DBG) && (T16_EMIT_DBG == 8)  // one lane's slot, the others next to it: contiguous 768-B runs at fresh addresses
        store_rec12(recs, (uint32_t)(__builtin_amdgcn_readfirstlane((int)pos[q]) & ~63) + (threadIdx.x & 63u), pr.idx0[q] & lmask, k, t, ax, ay);
#elif defined(T16_EMIT_DBG) && (T16_EMIT_DBG & 4)  // every lane group's records contiguous whatever the bins (is it the scatter?)
        store_rec12(recs, ((blockIdx.x * 1024u + (threadIdx.x & 960u)) * 64u + (pos[q] & 15u) * 256u + q * 64u + (threadIdx.x & 63u)), pr.idx0[q] & lmask, k, t, ax, ay);
#else
        if (pos[q] < capacity) store_rec12(recs, pos[q], pr.idx0[q] & lmask, k, t, ax, ay);
#endif
        rare |= pos[q] >= capacity;
    }
    if (__builtin_expect(__any(rare), 0)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t l0 = pr.idx0[q] & lmask, b0 = pr.idx0[q] >> bucket_log;
            const float gx = pr.wyz[q] * gix, gy = pr.wyz[q] * giy;
            if (pos[q] >= capacity) {
                if (straddle) fallback(b0, l0, l0, 0.0f, a0 * gx, a0 * gy);
                else fallback(b0, l0, l0 ^ pr.xm, pr.tx, gx, gy);
            }
            if (straddle) {  // second record: the x+1 neighbour in its own bucket
                const uint32_t i1 = pr.idx0[q] ^ pr.xm, b1 = i1 >> bucket_log, l1 = i1 & lmask;
                const uint32_t p1 = atomicAdd(&cursor_level[b1], 1u);
                if (p1 < capacity) store_rec12(recs, p1, l1, 15u, 0u, pr.tx * gx, pr.tx * gy);
                else fallback(b1, l1, l1, 0.0f, pr.tx * gx, pr.tx * gy);
            }
        }
    }
}

__device__ __forceinline__ void emit_pairs12(const Pairs &pr, float gix, float giy, uint32_t *cursor_level, int bucket_log,
                                             uint32_t capacity, Rec *recs, float *grad_level)
{
    PairSlots sl;
    reserve_pairs(pr, cursor_level, bucket_log, sl);
    commit_pairs12(pr, sl, gix, giy, cursor_level, bucket_log, capacity, recs, grad_level);
}

// histogram counterpart of emit_pairs (must stay in lock-step with it)
__device__ __forceinline__ void count_pairs(const Pairs &pr, uint32_t *hist_level, int bucket_log)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t b0 = pr.idx0[q] >> bucket_log, b1 = pr.idx1[q] >> bucket_log;
        atomicAdd(&hist_level[b0], 1u);
        if (b1 != b0) atomicAdd(&hist_level[b1], 1u);
    }
}

// workspace carve: [counts nbins*W][totals nbins][starts nbins+1][maxbits][pad to 256 B][records]
struct BinWorkspace {
    uint32_t *counts, *totals, *starts, *maxbits;
    Rec *recs;
    uint32_t capacity;
};
// [counts nbins*W][totals nbins][starts nbins+1][maxbits][... pad ...][skipped levels][format word][overflow flag = the word right before the records]
inline size_t bin_workspace_head(int nbins, int W)
{
    size_t head = ((size_t)nbins * W + 2 * (size_t)nbins + 6) * 4;
    return (head + 255) & ~(size_t)255;
}
// (capacity counts 16-byte records; a Rec8 stream holds twice as many: rec_capacity())
inline bool bin_workspace_carve(void *workspace, size_t bytes, int nbins, int W, BinWorkspace &w)
{
    const size_t head = bin_workspace_head(nbins, W);
    if (bytes < head + sizeof(Rec)) return false;
    w.counts = reinterpret_cast<uint32_t *>(workspace);
    w.totals = w.counts + (size_t)nbins * W;
    w.starts = w.totals + nbins;
    w.maxbits = w.starts + nbins + 1;
    w.recs = reinterpret_cast<Rec *>(reinterpret_cast<char *>(workspace) + head);
    const size_t cap = (bytes - head) / sizeof(Rec);
    w.capacity = cap > 0x7ffffff0u ? 0x7ffffff0u : (uint32_t)cap;
    return true;
}
// Record budget of a fused plan (16-byte records): 4 pairs per (sample, level) + straddle slack.
inline size_t fused_record_budget(int B, int S)
{
    const size_t n = (size_t)B * S * 16;
    return n * 4 + n / 8 + 4096;
}
// Large tables (buckets above 2^13 entries): the workspace scanerf_render_scatter_workspace_bytes sizes carries a second, FINE
// record area behind the coarse records' budget (scatter.hip SplitLayout), and a cached workspace may be larger still.  What the
// producer and the consumers may fill with coarse records ends at the budget: beyond it the overflow-table fallback takes over
// (records written into the fine area would be dropped and overwritten by the split pass).
inline uint32_t fused_coarse_capacity(uint32_t capacity, int B, int S, int bucket_log)
{
    if (bucket_log <= 13) return capacity;
    const size_t budget = fused_record_budget(B, S);
    return capacity < budget ? capacity : (uint32_t)budget;
}
__host__ __device__ inline uint32_t rec_capacity(uint32_t capacity16, int fmt)
{
    if (fmt == 3) return capacity16 / 4u;   // 64-byte segments
    return fmt == 1 ? capacity16 * 2u : (fmt == 2 ? capacity16 + capacity16 / 3u : capacity16);   // 8- / 12- / 16-byte records in the same bytes
}
// bucket size of the FUSED producer (k_render_bwd* emits, scanerf_render_scatter_accumulate consumes): log2 entries
inline int fused_bucket_log(int T);
inline int bin_ilog2(int v)
{
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

inline int fused_bucket_log(int T)
{
    const int lt = bin_ilog2(T);
    int want = kFusedBucketLog;
    want = tune_int("SCANERF_FUSED_BUCKET_LOG", want);  // tuning experiments only
    // at most 256 buckets per level (16 KB of cursors in the producer's LDS), at least `want` entries per bucket
    const int bl = lt - 8 > want ? lt - 8 : want;
    return lt < bl ? lt : bl;
}

// scatter.hip: the fused plan split around a counting forward launch (render.hip)
struct RenderArgs;
int scatter_plan_attach(void *workspace, size_t workspace_bytes, int B, int S, int T, int arith, int forward_grid, RenderArgs &a);
int scatter_plan_finish(void *workspace, size_t workspace_bytes, int B, int S, int T, int arith, void *stream);

}  // namespace scanerf
