// Device-side building blocks of the candidates -> detections stage (kernels:
// yh_nms.hip).
#pragma once
#include "yh_internal.h"

namespace yh {

constexpr int kLaneNmsMax = 16;   // class segments up to this length are walked by a single lane

// Exclusive scan of vals[0..n) in LDS, n <= THREADS * ITEMS, in place.
template <int THREADS, int ITEMS>
__device__ inline void block_exclusive_scan(int *vals, int n, int *wave_sums)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int v[ITEMS];
    int tsum = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = tid * ITEMS + j;
        v[j] = idx < n ? vals[idx] : 0;
        tsum += v[j];
    }
    int incl = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < THREADS / 64; ++w) { const int t = wave_sums[w]; wave_sums[w] = acc; acc += t; }
    }
    __syncthreads();
    int excl = wave_sums[wave] + (incl - tsum);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = tid * ITEMS + j;
        if (idx < n) vals[idx] = excl;
        excl += v[j];
    }
    __syncthreads();
}

// Bitonic sort, ascending, of a[0..cnt) with virtual +inf padding up to n2 (a
// power of two): every compare-exchange moves the smaller key to the lower
// index, so a pair whose upper index is >= cnt is a no-op.
template <int THREADS>
__device__ inline void bitonic_sort(unsigned long long *a, int cnt, int n2)
{
    const int tid = threadIdx.x;
    for (int k = 2, lg = 0; k <= n2; k <<= 1, ++lg) {        // lg = log2(k / 2): every divisor is a power of two
        const int half = k >> 1;
        for (int t = tid; t < (n2 >> 1); t += THREADS) {
            const int blk = t >> lg, off = t & (half - 1);
            const int i = blk * k + off, l = blk * k + (k - 1 - off);
            if (l < cnt) {
                const unsigned long long x = a[i], y = a[l];
                if (x > y) { a[i] = y; a[l] = x; }
            }
        }
        __syncthreads();
        for (int j = half >> 1, lj = lg - 1; j >= 1; j >>= 1, --lj) {
            for (int t = tid; t < (n2 >> 1); t += THREADS) {
                const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1)), l = i + j;
                if (l < cnt) {
                    const unsigned long long x = a[i], y = a[l];
                    if (x > y) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// Same network on (key, value) pairs: the values follow their keys.
template <int THREADS>
__device__ inline void bitonic_sort_kv(unsigned long long *a, float4 *v, int cnt, int n2)
{
    const int tid = threadIdx.x;
    for (int k = 2, lg = 0; k <= n2; k <<= 1, ++lg) {        // lg = log2(k / 2): every divisor is a power of two
        const int half = k >> 1;
        for (int t = tid; t < (n2 >> 1); t += THREADS) {
            const int blk = t >> lg, off = t & (half - 1);
            const int i = blk * k + off, l = blk * k + (k - 1 - off);
            if (l < cnt) {
                const unsigned long long x = a[i], y = a[l];
                if (x > y) { a[i] = y; a[l] = x; const float4 tv = v[i]; v[i] = v[l]; v[l] = tv; }
            }
        }
        __syncthreads();
        for (int j = half >> 1, lj = lg - 1; j >= 1; j >>= 1, --lj) {
            for (int t = tid; t < (n2 >> 1); t += THREADS) {
                const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1)), l = i + j;
                if (l < cnt) {
                    const unsigned long long x = a[i], y = a[l];
                    if (x > y) { a[i] = y; a[l] = x; const float4 tv = v[i]; v[i] = v[l]; v[l] = tv; }
                }
            }
            __syncthreads();
        }
    }
}

// ---- the candidates of one image ------------------------------------------
// They sit in the work units' private slots (ukeys / uboxes; an unused slot holds
// kEmptySlot) and in the image's list (img_count / cand / cand_box, or boxes[row]
// when cand_box is null).  f(key, box) is called for each, in no particular
// order, by whichever thread read it; box is read only when WANT_BOX.  Slot keys
// and boxes are requested together, before a key is looked at, so that one
// memory round trip brings in everything.  list_raw: img_count[b].
template <int THREADS, bool WANT_BOX, typename F>
__device__ inline void cand_for_each(const NmsParams &p, int b, int list_raw, F f)
{
    const int tid = threadIdx.x, nslots = p.upi * kUnitSlots;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long *uk = p.ukeys + (long long)b * nslots;
    const float4 *ub = reinterpret_cast<const float4 *>(p.uboxes) + (long long)b * nslots;
    // U slots per thread are requested back to back (keys and boxes alike, valid or not: an
    // image's slots are a few tens of KiB), so that a small workgroup still pays one memory
    // round trip per U * THREADS slots instead of one per slot.
    constexpr int U = THREADS >= 1024 ? 2 : (THREADS >= 128 ? 2048 / THREADS : 32);
    for (int e0 = 0; e0 < nslots; e0 += THREADS * U) {
        unsigned long long k[U];
        float4 box[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * THREADS + tid;
            const int ec = e < nslots ? e : nslots - 1;
            k[u] = uk[ec];
            box[u] = WANT_BOX ? ub[ec] : zero;
            if (e >= nslots) k[u] = kEmptySlot;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (WANT_BOX) asm volatile("" ::"v"(box[u].x));   // keeps the box loads up there with the keys'
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k[u] != kEmptySlot) f(k[u], box[u]);
    }
    const int nl = list_raw < p.cand_cap ? list_raw : p.cand_cap;
    const unsigned long long *cand = p.cand + (long long)b * p.cand_cap;
    const float4 *cb = p.cand_box ? reinterpret_cast<const float4 *>(p.cand_box) + (long long)b * p.cand_cap : nullptr;

    for (int i = tid; i < nl; i += THREADS) {
        const unsigned long long k = cand[i];
        f(k, WANT_BOX ? (cb ? cb[i] : row_box(p, b, key_row(k))) : zero);
    }
}

__device__ __forceinline__ float lane_bcast(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// NonMaxSuppressionV3's overlap test on corner-normalised boxes with
// precomputed areas: non-positive area never suppresses, the intersection is
// clamped at zero, the comparison is strict: RN(inter / union) > thr.
// The division is the costly part of the test and almost never needed: with
// d = inter - thr * union (one rounding) and the margin m = 2^-20 * union + 2^-100
// (thr lies in [0, 1]), d > m implies RN(inter / union) > thr and d < -m the
// opposite, because the roundings involved stay below 2^-22 relative; only a pair
// inside the margin, or with a union that is not positive, takes the division.
// Same decision in every case.
// The test in two steps, for callers that look at the intersection of a whole wavefront first: with thr in
// [0, 1] an intersection that is not positive (zero after the clamp, or NaN) never suppresses.
__device__ __forceinline__ float intersection(float ymin_i, float xmin_i, float ymax_i, float xmax_i,
                                              float ymin_j, float xmin_j, float ymax_j, float xmax_j)
{
    const float iy = fmaxf(fminf(ymax_i, ymax_j) - fmaxf(ymin_i, ymin_j), 0.0f);
    const float ix = fmaxf(fminf(xmax_i, xmax_j) - fmaxf(xmin_i, xmin_j), 0.0f);
    return iy * ix;
}
__device__ __forceinline__ bool suppresses_at(float inter, float area_i, float area_j, float thr)
{
    if (!(area_i > 0.0f && area_j > 0.0f)) return false;    // a NaN area ends in "false" either way
    const float uni = (area_i + area_j) - inter;
    const float d = __builtin_fmaf(-thr, uni, inter);
    const float m = __builtin_fmaf(0x1p-20f, uni, 0x1p-100f);
    if (uni > 0.0f && uni < 0x1p100f && inter < 0x1p100f) {   // finite, no overflow in d
        if (d > m) return true;
        if (d < -m) return false;
    }
    return inter / uni > thr;
}
__device__ __forceinline__ bool suppresses(float ymin_i, float xmin_i, float ymax_i, float xmax_i, float area_i,
                                           float ymin_j, float xmin_j, float ymax_j, float xmax_j, float area_j,
                                           float thr)
{
    return suppresses_at(intersection(ymin_i, xmin_i, ymax_i, xmax_i, ymin_j, xmin_j, ymax_j, xmax_j), area_i, area_j, thr);
}

// Greedy resolve of a tile in score order: a lane is selected iff no SELECTED earlier lane
// suppresses it, until `room` lanes are selected.  alive: the tile's candidates that survived the
// earlier tiles; mymask: this lane's bits "the earlier lane j would suppress me".  Lanes nobody
// could suppress are selected whatever happens to the others, so only the lanes with a non-empty
// mask are walked (on the scalar unit, in lane order); cutting the selection at `room` afterwards
// is the same as stopping there, since a lane's fate depends on earlier lanes only.
__device__ __forceinline__ unsigned long long resolve_tile(unsigned long long alive, unsigned long long mymask, int room)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long contested = __ballot(mymask != 0ull) & alive;
    unsigned long long keepmask = alive & ~contested;
    const int m_lo = (int)(uint32_t)mymask, m_hi = (int)(uint32_t)(mymask >> 32);
    for (unsigned long long rem = contested; rem != 0ull; rem &= rem - 1ull) {
        const int j = __ffsll((long long)rem) - 1;
        const unsigned long long mj =
            ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(m_hi, j) << 32) |
            (uint32_t)__builtin_amdgcn_readlane(m_lo, j);
        if ((mj & keepmask) == 0ull) keepmask |= 1ull << j;
    }
    // the first `room` selected lanes
    const bool mine = ((keepmask >> lane) & 1ull) && __popcll(keepmask & ((1ull << lane) - 1ull)) < room;
    return __ballot(mine);
}

// Greedy NMS of one sorted segment by ONE wavefront (all 64 lanes call it):
// tf.image.non_max_suppression's walk (candidates in descending score, one is
// selected unless its IoU with an already SELECTED box exceeds the threshold,
// stop at max_boxes), tiled by 64: test the tile against the selected list
// (kbox, LDS), build the 64x64 suppression bit matrix in registers (lane i
// keeps the bits of the earlier lanes that would suppress it), resolve the tile
// serially on the scalar unit with readlane, append the survivors.
// keys[0..cnt): sorted candidate keys (LDS or global); kbox: room for max_boxes
// selected boxes (5 floats each) private to this wavefront.  Returns the number
// selected.
//   IN_LDS = false: the box of row n is box_of(n); the selection goes to
//                   krow / kscore (global, one list per segment);
//   IN_LDS = true : boxes_seg[i] is the box of keys[i] (LDS); the selection goes
//                   to kpos as positions pos_base + i (LDS).
template <bool IN_LDS, typename BoxOf>
__device__ inline int wave_nms(const unsigned long long *keys, int cnt, BoxOf box_of,
                               const float4 *boxes_seg, float (*kbox)[5], float thr, int max_boxes,
                               int32_t *krow, float *kscore, uint16_t *kpos, int pos_base, float4 *kbox_out = nullptr)
{
    const int lane = threadIdx.x & 63;
    int kept = 0;
    for (int t = 0; t < cnt && kept < max_boxes; t += 64) {
        const int i = t + lane;
        const bool valid = i < cnt;
        const unsigned long long key = valid ? keys[i] : 0ull;
        const int n = key_row(key);
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) bx = IN_LDS ? boxes_seg[i] : box_of(n);
        const float ymin = fminf(bx.x, bx.z), xmin = fminf(bx.y, bx.w);
        const float ymax = fmaxf(bx.x, bx.z), xmax = fmaxf(bx.y, bx.w);
        const float area = (ymax - ymin) * (xmax - xmin);
        bool alive = valid;
        // against the boxes selected by earlier tiles
        for (int j = kept - 1; j >= 0; --j) {
            if (__ballot(alive) == 0ull) break;
            const float *kb = kbox[j];
            if (alive && suppresses(ymin, xmin, ymax, xmax, area, kb[0], kb[1], kb[2], kb[3], kb[4], thr))
                alive = false;
        }
        // inside the tile: bit j of mymask = "the earlier lane j would suppress me"
        const unsigned long long alive_mask = __ballot(alive);
        unsigned long long mymask = 0ull;
        for (unsigned long long rem = alive_mask; rem != 0ull; rem &= rem - 1ull) {
            const int j = __ffsll((long long)rem) - 1;
            const float jy0 = lane_bcast(ymin, j), jx0 = lane_bcast(xmin, j);
            const float jy1 = lane_bcast(ymax, j), jx1 = lane_bcast(xmax, j);
            const float ja = lane_bcast(area, j);
            if (alive && lane > j && suppresses(ymin, xmin, ymax, xmax, area, jy0, jx0, jy1, jx1, ja, thr))
                mymask |= 1ull << j;
        }
        const unsigned long long keepmask = resolve_tile(alive_mask, mymask, max_boxes - kept);
        if ((keepmask >> lane) & 1ull) {
            const int pos = kept + __popcll(keepmask & ((1ull << lane) - 1ull));
            kbox[pos][0] = ymin; kbox[pos][1] = xmin; kbox[pos][2] = ymax; kbox[pos][3] = xmax;
            kbox[pos][4] = area;
            if (IN_LDS) {
                kpos[pos] = (uint16_t)(pos_base + i);
            } else {
                krow[pos] = n;
                kscore[pos] = key_score(key);
                if (kbox_out) kbox_out[pos] = bx;
            }
        }
        kept += __popcll(keepmask);
    }
    return kept;
}

// LDS working set of post_image, carved by the caller.
struct PostLds {
    unsigned long long *keys;     // [cap] in arrival order (sorted in place beyond rank_max)
    unsigned long long *sorted;   // [rank_max]
    float4 *ubox;                 // [cap] boxes of keys[]
    float4 *box;                  // [rank_max] boxes of sorted[]
    uint16_t *mask;               // [cap] short segments: bit j = "candidate j of my segment would suppress me"
    uint16_t *kpos;               // [C * max_boxes]
    int *off, *seg, *cnt, *kept;  // [C] each
    int *wave_sums;               // [THREADS / 64]
    int *total;                   // [2]: candidates found, "a segment is longer than kLaneNmsMax"
    float *kbox_all;              // [THREADS / 64][max_boxes][5]
};

// Ranks of KPT keys per thread against all M keys (keys are unique): the list is
// read once per thread, as broadcast LDS reads, several in flight.
template <int THREADS, int KPT>
__device__ __forceinline__ void rank_sort_pass(const PostLds &l, int M, int i0)
{
    const int tid = threadIdx.x;
    unsigned long long key[KPT];
    int rank[KPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int i = i0 + u * THREADS + tid;
        key[u] = i < M ? l.keys[i] : 0ull;
        rank[u] = 0;
    }
    int j = 0;
    for (; j + 4 <= M; j += 4) {
        const unsigned long long k0 = l.keys[j], k1 = l.keys[j + 1], k2 = l.keys[j + 2], k3 = l.keys[j + 3];
#pragma unroll
        for (int u = 0; u < KPT; ++u)
            rank[u] += (k0 < key[u] ? 1 : 0) + (k1 < key[u] ? 1 : 0) + (k2 < key[u] ? 1 : 0) + (k3 < key[u] ? 1 : 0);
    }
    for (; j < M; ++j) {
        const unsigned long long k0 = l.keys[j];
#pragma unroll
        for (int u = 0; u < KPT; ++u) rank[u] += k0 < key[u] ? 1 : 0;
    }
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int i = i0 + u * THREADS + tid;
        if (i < M) {
            l.sorted[rank[u]] = key[u];
            l.box[rank[u]] = l.ubox[i];
        }
    }
}

// Candidates of image b -> its detections, by one workgroup of THREADS threads
// (64 .. 1024), everything in LDS between the candidate slots and the outputs:
// keys and boxes loaded together, keys sorted by the full key (class ascending,
// score descending, row ascending; rank sort up to rank_max keys, bitonic
// beyond) with the boxes following them, per-class greedy NMS, the selections
// concatenated (YOLOv3.py:362-367).  C <= THREADS * SCAN_ITEMS.  Zeroes
// img_count[b] for the next call.
//
// NMS of the short class segments (<= kLaneNmsMax candidates, the usual case at
// detection thresholds) runs in two data-parallel steps: one thread per CANDIDATE
// collects the bits "the earlier candidate j of my segment would suppress me";
// then one thread per CLASS walks its segment in score order over those bits, a
// few integer operations per candidate.  Longer segments take a wavefront each
// (wave_nms).
template <int THREADS, int SCAN_ITEMS>
__device__ inline void post_image(const NmsParams &p, int b, const PostLds &l, int rank_max)
{
    static_assert(kLaneNmsMax <= 16, "suppression bits of a short segment are kept in 16 bits");
    const int tid = threadIdx.x, wave = tid >> 6, C = p.C, mb = p.max_boxes, cap = p.cand_cap;
    const int raw = p.img_count[b];
    for (int c = tid; c < C; c += THREADS) { l.cnt[c] = 0; l.kept[c] = 0; }
    if (tid < 2) l.total[tid] = 0;
    __syncthreads();
    cand_for_each<THREADS, true>(p, b, raw, [&](unsigned long long k, float4 box) {
        const int pos = atomicAdd(&l.total[0], 1);
        if (pos < cap) {
            l.keys[pos] = k;
            l.ubox[pos] = box;
            atomicAdd(&l.cnt[key_class(k)], 1);
        }
    });
    __syncthreads();
    const int found = l.total[0];
    const int M = found < cap ? found : cap;
    if (tid == 0) {
        p.img_count[b] = 0;                                   // leaves the counter clean for the next call
        // the list keeps at most `cap` of the `raw` entries its producers counted
        p.status[b] = image_status(found, raw, cap);
    }
    const unsigned long long *sorted;
    const float4 *sbox;
    if (M <= rank_max) {
        if (M <= THREADS) rank_sort_pass<THREADS, 1>(l, M, 0);
        else if (M <= 2 * THREADS) rank_sort_pass<THREADS, 2>(l, M, 0);
        else if (M <= 4 * THREADS) rank_sort_pass<THREADS, 4>(l, M, 0);
        else
            for (int i0 = 0; i0 < M; i0 += 8 * THREADS) rank_sort_pass<THREADS, 8>(l, M, i0);
        __syncthreads();
        sorted = l.sorted;
        sbox = l.box;
    } else {
        int n2 = 64;
        while (n2 < M) n2 <<= 1;
        bitonic_sort_kv<THREADS>(l.keys, l.ubox, M, n2);
        sorted = l.keys;
        sbox = l.ubox;
    }
    for (int c = tid; c < C; c += THREADS) {
        const int n = l.cnt[c];
        l.off[c] = n;
        if (n > kLaneNmsMax) l.total[1] = 1;
        if (p.out_class_candidates) p.out_class_candidates[b * C + c] = n;
    }
    __syncthreads();
    block_exclusive_scan<THREADS, SCAN_ITEMS>(l.off, C, l.wave_sums);
    for (int c = tid; c < C; c += THREADS) l.seg[c] = l.off[c];
    __syncthreads();

    // short segments, step 1: a thread per candidate
    for (int i = tid; i < M; i += THREADS) {
        const int c = key_class(sorted[i]);
        if (l.cnt[c] > kLaneNmsMax) continue;
        const int off = l.seg[c];
        const float4 bi = sbox[i];
        const float ymin = fminf(bi.x, bi.z), xmin = fminf(bi.y, bi.w);
        const float ymax = fmaxf(bi.x, bi.z), xmax = fmaxf(bi.y, bi.w);
        const float area = (ymax - ymin) * (xmax - xmin);
        uint32_t m = 0;
        for (int j = off; j < i; ++j) {
            const float4 bj = sbox[j];
            const float jy0 = fminf(bj.x, bj.z), jx0 = fminf(bj.y, bj.w);
            const float jy1 = fmaxf(bj.x, bj.z), jx1 = fmaxf(bj.y, bj.w);
            if (suppresses(ymin, xmin, ymax, xmax, area, jy0, jx0, jy1, jx1, (jy1 - jy0) * (jx1 - jx0), p.iou_thr))
                m |= 1u << (j - off);
        }
        l.mask[i] = (uint16_t)m;
    }
    __syncthreads();
    // step 2: a thread per class; a candidate is selected iff no SELECTED earlier one suppresses it
    for (int c = tid; c < C; c += THREADS) {
        const int cnt = l.cnt[c];
        if (cnt == 0 || cnt > kLaneNmsMax) continue;
        const int off = l.seg[c];
        uint16_t *kp = l.kpos + c * mb;
        uint32_t keepmask = 0;
        int kept = 0;
        for (int j = 0; j < cnt && kept < mb; ++j) {
            if ((l.mask[off + j] & keepmask) == 0u) {
                keepmask |= 1u << j;
                kp[kept++] = (uint16_t)(off + j);
            }
        }
        l.kept[c] = kept;
    }
    // longer segments: one wavefront per class
    if (l.total[1]) {
        float (*kbox)[5] = reinterpret_cast<float (*)[5]>(l.kbox_all + (size_t)wave * mb * 5);
        for (int c = wave; c < C; c += THREADS / 64) {
            const int cnt = l.cnt[c];
            if (cnt <= kLaneNmsMax) continue;
            const int off = l.seg[c];
            const int kept = wave_nms<true>(sorted + off, cnt, [](int) { return make_float4(0.f, 0.f, 0.f, 0.f); }, sbox + off, kbox, p.iou_thr, mb,
                                            nullptr, nullptr, l.kpos + c * mb, off);
            if ((tid & 63) == 0) l.kept[c] = kept;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += THREADS) l.off[c] = l.kept[c];
    __syncthreads();
    block_exclusive_scan<THREADS, SCAN_ITEMS>(l.off, C, l.wave_sums);
    const long long obase = (long long)b * C * mb;
    for (int c = tid; c < C; c += THREADS) {                   // a thread per class writes its selections
        const int kept = l.kept[c];
        const long long dst0 = obase + l.off[c];
        for (int j = 0; j < kept; ++j) {
            const int pos = l.kpos[c * mb + j];
            const unsigned long long key = sorted[pos];
            p.out_scores[dst0 + j] = key_score(key);
            reinterpret_cast<float4 *>(p.out_boxes)[dst0 + j] = sbox[pos];
            p.out_class[dst0 + j] = c;
            p.out_index[dst0 + j] = key_row(key);
        }
    }
    if (tid == 0) p.out_count[b] = l.off[C - 1] + l.kept[C - 1];
}

// LDS layout of post_image for a call's bounds (capacity, classes, boxes per class)
// and a workgroup size; every array 16-byte aligned.
struct PostLayout {
    size_t keys, sorted, ubox, box, mask, kpos, off, seg, cnt, kept, wave, total, kbox, bytes;
    int rank_max;
};

__host__ __device__ inline PostLayout post_layout(int cap, int C, int mb, int threads)
{
    PostLayout L;
    L.rank_max = cap < kPostRankMax ? cap : kPostRankMax;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
    L.ubox = take((size_t)cap * 16);
    L.box = take((size_t)L.rank_max * 16);
    L.keys = take((size_t)cap * 8);
    L.sorted = take((size_t)L.rank_max * 8);
    L.kbox = take((size_t)(threads / 64) * mb * 5 * 4);
    L.off = take((size_t)C * 4);
    L.seg = take((size_t)C * 4);
    L.cnt = take((size_t)C * 4);
    L.kept = take((size_t)C * 4);
    L.wave = take((size_t)(threads / 64) * 4);
    L.total = take(8);
    L.mask = take((size_t)cap * 2);
    L.kpos = take((size_t)C * mb * 2);
    L.bytes = o;
    return L;
}

// post_image on a raw LDS block of post_layout(...).bytes bytes.
template <int THREADS>
__device__ inline void post_image_dyn(const NmsParams &p, int b, unsigned char *raw)
{
    const PostLayout L = post_layout(p.cand_cap, p.C, p.max_boxes, THREADS);
    PostLds l;
    l.keys = reinterpret_cast<unsigned long long *>(raw + L.keys);
    l.sorted = reinterpret_cast<unsigned long long *>(raw + L.sorted);
    l.ubox = reinterpret_cast<float4 *>(raw + L.ubox);
    l.box = reinterpret_cast<float4 *>(raw + L.box);
    l.mask = reinterpret_cast<uint16_t *>(raw + L.mask);
    l.kpos = reinterpret_cast<uint16_t *>(raw + L.kpos);
    l.off = reinterpret_cast<int *>(raw + L.off);
    l.seg = reinterpret_cast<int *>(raw + L.seg);
    l.cnt = reinterpret_cast<int *>(raw + L.cnt);
    l.kept = reinterpret_cast<int *>(raw + L.kept);
    l.wave_sums = reinterpret_cast<int *>(raw + L.wave);
    l.total = reinterpret_cast<int *>(raw + L.total);
    l.kbox_all = reinterpret_cast<float *>(raw + L.kbox);
    post_image<THREADS, kPostMaxClasses / THREADS>(p, b, l, L.rank_max);
}

}  // namespace yh
