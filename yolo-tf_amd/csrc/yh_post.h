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
    for (int k = 2; k <= n2; k <<= 1) {
        const int half = k >> 1;
        for (int t = tid; t < (n2 >> 1); t += THREADS) {
            const int blk = t / half, off = t - blk * half;
            const int i = blk * k + off, l = blk * k + (k - 1 - off);
            if (l < cnt) {
                const unsigned long long x = a[i], y = a[l];
                if (x > y) { a[i] = y; a[l] = x; }
            }
        }
        __syncthreads();
        for (int j = half >> 1; j >= 1; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += THREADS) {
                const int i = (t / j) * 2 * j + (t % j), l = i + j;
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
    for (int k = 2; k <= n2; k <<= 1) {
        const int half = k >> 1;
        for (int t = tid; t < (n2 >> 1); t += THREADS) {
            const int blk = t / half, off = t - blk * half;
            const int i = blk * k + off, l = blk * k + (k - 1 - off);
            if (l < cnt) {
                const unsigned long long x = a[i], y = a[l];
                if (x > y) { a[i] = y; a[l] = x; const float4 tv = v[i]; v[i] = v[l]; v[l] = tv; }
            }
        }
        __syncthreads();
        for (int j = half >> 1; j >= 1; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += THREADS) {
                const int i = (t / j) * 2 * j + (t % j), l = i + j;
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
    for (int e = tid; e < nslots; e += THREADS) {
        const unsigned long long k = uk[e];
        float4 box = zero;
        if (WANT_BOX) {
            box = ub[e];
            asm volatile("" ::"v"(box.x));                    // keeps the load next to the key's, ahead of the test
        }
        if (k != kEmptySlot) f(k, box);
    }
    const int nl = list_raw < p.cand_cap ? list_raw : p.cand_cap;
    const unsigned long long *cand = p.cand + (long long)b * p.cand_cap;
    const float4 *cb = p.cand_box ? reinterpret_cast<const float4 *>(p.cand_box) + (long long)b * p.cand_cap : nullptr;
    const float4 *boxes4 = reinterpret_cast<const float4 *>(p.boxes) + (long long)b * p.N;
    for (int i = tid; i < nl; i += THREADS) {
        const unsigned long long k = cand[i];
        f(k, WANT_BOX ? (cb ? cb[i] : boxes4[key_row(k)]) : zero);
    }
}

__device__ __forceinline__ float lane_bcast(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// NonMaxSuppressionV3's overlap test on corner-normalised boxes with
// precomputed areas: non-positive area never suppresses, the intersection is
// clamped at zero, the comparison is strict.
__device__ __forceinline__ bool suppresses(float ymin_i, float xmin_i, float ymax_i, float xmax_i, float area_i,
                                           float ymin_j, float xmin_j, float ymax_j, float xmax_j, float area_j,
                                           float thr)
{
    if (area_i <= 0.0f || area_j <= 0.0f) return false;
    const float iy = fmaxf(fminf(ymax_i, ymax_j) - fmaxf(ymin_i, ymin_j), 0.0f);
    const float ix = fmaxf(fminf(xmax_i, xmax_j) - fmaxf(xmin_i, xmin_j), 0.0f);
    const float inter = iy * ix;
    const float iou = inter / ((area_i + area_j) - inter);
    return iou > thr;
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
//   IN_LDS = false: boxes are gathered from boxes4[row]; the selection goes to
//                   krow / kscore (global, one list per segment);
//   IN_LDS = true : boxes_seg[i] is the box of keys[i] (LDS); the selection goes
//                   to kpos as positions pos_base + i (LDS).
template <bool IN_LDS>
__device__ inline int wave_nms(const unsigned long long *keys, int cnt, const float4 *boxes4,
                               const float4 *boxes_seg, float (*kbox)[5], float thr, int max_boxes,
                               int32_t *krow, float *kscore, uint16_t *kpos, int pos_base)
{
    const int lane = threadIdx.x & 63;
    int kept = 0;
    for (int t = 0; t < cnt && kept < max_boxes; t += 64) {
        const int i = t + lane;
        const bool valid = i < cnt;
        const unsigned long long key = valid ? keys[i] : 0ull;
        const int n = key_row(key);
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) bx = IN_LDS ? boxes_seg[i] : boxes4[n];
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
        // greedy resolve in score order: a lane is selected iff no SELECTED earlier lane suppresses it
        unsigned long long keepmask = 0ull;
        int room = max_boxes - kept;
        const int m_lo = (int)(uint32_t)mymask, m_hi = (int)(uint32_t)(mymask >> 32);
        for (unsigned long long rem = alive_mask; rem != 0ull && room > 0; rem &= rem - 1ull) {
            const int j = __ffsll((long long)rem) - 1;
            const unsigned long long mj =
                ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(m_hi, j) << 32) |
                (uint32_t)__builtin_amdgcn_readlane(m_lo, j);
            if ((mj & keepmask) == 0ull) { keepmask |= 1ull << j; --room; }
        }
        if ((keepmask >> lane) & 1ull) {
            const int pos = kept + __popcll(keepmask & ((1ull << lane) - 1ull));
            kbox[pos][0] = ymin; kbox[pos][1] = xmin; kbox[pos][2] = ymax; kbox[pos][3] = xmax;
            kbox[pos][4] = area;
            if (IN_LDS) {
                kpos[pos] = (uint16_t)(pos_base + i);
            } else {
                krow[pos] = n;
                kscore[pos] = key_score(key);
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
    uint16_t *kpos;               // [C * max_boxes]
    int *off, *seg, *cnt, *kept;  // [C] each
    int *wave_sums;               // [THREADS / 64]
    int *total;                   // [1]
    float *kbox_all;              // [THREADS / 64][max_boxes][5]
};

constexpr int kNmsGroup = 16;     // lanes that share one short class segment (== kLaneNmsMax)

// Candidates of image b -> its detections, by one workgroup of THREADS threads,
// everything in LDS between the candidate slots and the outputs: keys and boxes
// loaded together, keys sorted by the full key (class ascending, score
// descending, row ascending; rank sort up to rank_max keys, bitonic beyond) with
// the boxes following them, the class segments walked by groups of 16 lanes
// (short ones) or wavefronts, the selections concatenated (YOLOv3.py:362-367).
// C <= THREADS * SCAN_ITEMS.  Zeroes img_count[b] for the next call.
template <int THREADS, int SCAN_ITEMS>
__device__ inline void post_image(const NmsParams &p, int b, const PostLds &l, int rank_max)
{
    static_assert(kNmsGroup == kLaneNmsMax, "one lane per candidate of a short segment");
    const int tid = threadIdx.x, wave = tid >> 6, C = p.C, mb = p.max_boxes, cap = p.cand_cap;
    const int raw = p.img_count[b];
    for (int c = tid; c < C; c += THREADS) { l.cnt[c] = 0; l.kept[c] = 0; }
    if (tid == 0) *l.total = 0;
    __syncthreads();
    cand_for_each<THREADS, true>(p, b, raw, [&](unsigned long long k, float4 box) {
        const int pos = atomicAdd(l.total, 1);
        if (pos < cap) {
            l.keys[pos] = k;
            l.ubox[pos] = box;
            atomicAdd(&l.cnt[key_class(k)], 1);
        }
    });
    __syncthreads();
    const int found = *l.total;
    const int M = found < cap ? found : cap;
    if (tid == 0) {
        p.img_count[b] = 0;                                   // leaves the counter clean for the next call
        // the list keeps at most `cap` of the `raw` entries its producers counted
        p.status[b] = (found > cap || raw > cap) ? YH_STATUS_CANDIDATE_OVERFLOW : 0;
    }
    const unsigned long long *sorted;
    const float4 *sbox;
    if (M <= rank_max) {
        // rank sort, four lanes per key: each counts the smaller keys in a quarter of the list
        const int q = tid & 3;
        const int j0 = (M * q) >> 2, j1 = (M * (q + 1)) >> 2;
        for (int i0 = 0; i0 < M; i0 += THREADS / 4) {
            const int i = i0 + (tid >> 2);
            const unsigned long long key = i < M ? l.keys[i] : 0ull;
            int rank = 0;
            for (int j = j0; j < j1; ++j) rank += l.keys[j] < key ? 1 : 0;   // keys are unique
            rank += __shfl_xor(rank, 1, 64);
            rank += __shfl_xor(rank, 2, 64);
            if (i < M && q == 0) {
                l.sorted[rank] = key;
                l.box[rank] = l.ubox[i];
            }
        }
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
        if (p.out_class_candidates) p.out_class_candidates[b * C + c] = n;
    }
    __syncthreads();
    block_exclusive_scan<THREADS, SCAN_ITEMS>(l.off, C, l.wave_sums);
    for (int c = tid; c < C; c += THREADS) l.seg[c] = l.off[c];
    __syncthreads();

    // Short segments (the usual case at detection thresholds): a group of 16 lanes per
    // class, one lane per candidate.  Lane i collects the bits "the earlier candidate j
    // would suppress me"; the greedy walk over those bits is then a few integer steps,
    // done redundantly by every lane of the group.
    {
        const int gl = tid & (kNmsGroup - 1);
        for (int c = tid / kNmsGroup; c < C; c += THREADS / kNmsGroup) {
            const int cnt = l.cnt[c];
            if (cnt == 0 || cnt > kLaneNmsMax) continue;
            const int off = l.seg[c];
            const bool valid = gl < cnt;
            const float4 bi = sbox[off + (valid ? gl : 0)];
            const float ymin = fminf(bi.x, bi.z), xmin = fminf(bi.y, bi.w);
            const float ymax = fmaxf(bi.x, bi.z), xmax = fmaxf(bi.y, bi.w);
            const float area = (ymax - ymin) * (xmax - xmin);
            uint32_t mymask = 0;
            for (int j = 0; j + 1 < cnt; ++j) {
                const float4 bj = sbox[off + j];
                const float jy0 = fminf(bj.x, bj.z), jx0 = fminf(bj.y, bj.w);
                const float jy1 = fmaxf(bj.x, bj.z), jx1 = fmaxf(bj.y, bj.w);
                if (valid && gl > j &&
                    suppresses(ymin, xmin, ymax, xmax, area, jy0, jx0, jy1, jx1, (jy1 - jy0) * (jx1 - jx0), p.iou_thr))
                    mymask |= 1u << j;
            }
            uint32_t keepmask = 0;
            int kept = 0;
            for (int j = 0; j < cnt && kept < mb; ++j) {
                const uint32_t mj = (uint32_t)__shfl((int)mymask, j, kNmsGroup);
                if ((mj & keepmask) == 0u) { keepmask |= 1u << j; ++kept; }
            }
            if ((keepmask >> gl) & 1u) l.kpos[c * mb + __popc(keepmask & ((1u << gl) - 1u))] = (uint16_t)(off + gl);
            if (gl == 0) l.kept[c] = kept;
        }
    }
    // longer segments: one wavefront per class
    float (*kbox)[5] = reinterpret_cast<float (*)[5]>(l.kbox_all + (size_t)wave * mb * 5);
    for (int c = wave; c < C; c += THREADS / 64) {
        const int cnt = l.cnt[c];
        if (cnt <= kLaneNmsMax) continue;
        const int off = l.seg[c];
        const int kept = wave_nms<true>(sorted + off, cnt, nullptr, sbox + off, kbox, p.iou_thr, mb,
                                        nullptr, nullptr, l.kpos + c * mb, off);
        if ((tid & 63) == 0) l.kept[c] = kept;
    }
    __syncthreads();
    for (int c = tid; c < C; c += THREADS) l.off[c] = l.kept[c];
    __syncthreads();
    block_exclusive_scan<THREADS, SCAN_ITEMS>(l.off, C, l.wave_sums);
    const int ocap = C * mb;
    for (int e = tid; e < ocap; e += THREADS) {
        const int c = e / mb, j = e - c * mb;
        if (j < l.kept[c]) {
            const int pos = l.kpos[e];
            const unsigned long long key = sorted[pos];
            const long long dst = (long long)b * ocap + l.off[c] + j;
            p.out_scores[dst] = key_score(key);
            reinterpret_cast<float4 *>(p.out_boxes)[dst] = sbox[pos];
            p.out_class[dst] = c;
            p.out_index[dst] = key_row(key);
        }
    }
    if (tid == 0) p.out_count[b] = l.off[C - 1] + l.kept[C - 1];
}

}  // namespace yh
