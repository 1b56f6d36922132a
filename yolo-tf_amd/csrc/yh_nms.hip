// Candidates -> detections: class binning, per-(image, class) sort by
// (score descending, box row ascending), exact greedy NMS and the gather /
// concat of the outputs.  Replaces YOLOv3.py:355-368 / YOLOv2.py:197-210.
//
// The greedy pass is tf.image.non_max_suppression's (NonMaxSuppressionV3):
// candidates in descending score, one is selected unless its IoU with an
// already SELECTED box exceeds the threshold, stop at max_output_size.  There
// is no cap on how many candidates are walked, so the walk is tiled: one
// wavefront takes 64 sorted candidates at a time, tests them against the
// selected list held in LDS, builds the 64x64 suppression bit matrix in
// registers (lane i keeps the bits of the earlier lanes that would suppress
// it), resolves the tile serially on the scalar unit with readlane, appends
// the survivors and stops as soon as max_boxes are selected.
//
// Two launch shapes:
//  * k_post, one workgroup per image, when an image holds at most kPostCap
//    candidates (the capacity the caller passes): the whole list is sorted in
//    LDS by the full key (class, score, row), the candidates' boxes are
//    gathered into LDS in one go, the wavefronts take the class segments in
//    turn and the same workgroup writes the concatenated outputs.  One launch,
//    no intermediate global traffic.
//  * k_bin -> k_sort_nms -> k_finalize for larger capacities: class histogram
//    and scatter per image, then one workgroup per (image, class) segment.
#include "yh_internal.h"

namespace yh {

constexpr int kBinThreads = 1024;
constexpr int kSegThreads = 256;
constexpr int kSortLds = 2048;       // keys sorted in LDS by k_sort_nms; longer segments sort in global memory
constexpr int kFinThreads = 256;
constexpr int kPostThreads = 1024;

// Exclusive scan of vals[0..n) in LDS, n <= THREADS * ITEMS, in place.
template <int THREADS, int ITEMS>
__device__ void block_exclusive_scan(int *vals, int n, int *wave_sums)
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
__device__ void bitonic_sort(unsigned long long *a, int cnt, int n2)
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

// Greedy NMS of one sorted segment by ONE wavefront (all 64 lanes call it).
// keys[0..cnt): sorted candidate keys (LDS or global); kbox: room for max_boxes
// selected boxes (5 floats each) private to this wavefront.  Returns the number
// selected.
//   IN_LDS = false: boxes are gathered from boxes4[row]; the selection goes to
//                   krow / kscore (global, one list per segment);
//   IN_LDS = true : boxes_seg[i] is the box of keys[i] (LDS); the selection goes
//                   to kpos as positions pos_base + i (LDS).
template <bool IN_LDS>
__device__ int wave_nms(const unsigned long long *keys, int cnt, const float4 *boxes4, const float4 *boxes_seg,
                        float (*kbox)[5], float thr, int max_boxes, int32_t *krow, float *kscore,
                        uint16_t *kpos, int pos_base)
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

// Concatenation of the classes' selections (YOLOv3.py:362-367) for image b.
// s_off[0..C): exclusive scan of the per-class counts kept[0..C) (both LDS or
// global, readable by every thread).
template <int THREADS>
__device__ void write_outputs(const NmsParams &p, int b, const int *s_off, const int32_t *kept)
{
    const int tid = threadIdx.x, C = p.C, mb = p.max_boxes;
    const long long cap = (long long)C * mb;
    const float4 *boxes4 = reinterpret_cast<const float4 *>(p.boxes) + (long long)b * p.N;
    for (long long e = tid; e < cap; e += THREADS) {
        const int c = (int)(e / mb), j = (int)(e - (long long)c * mb);
        if (j < kept[c]) {
            const long long src = ((long long)b * C + c) * mb + j;
            const long long dst = (long long)b * cap + s_off[c] + j;
            const int n = p.kept_row[src];
            p.out_scores[dst] = p.kept_score[src];
            reinterpret_cast<float4 *>(p.out_boxes)[dst] = boxes4[n];
            p.out_class[dst] = c;
            p.out_index[dst] = n;
        }
    }
    if (tid == 0) p.out_count[b] = s_off[C - 1] + kept[C - 1];
}

// ------------------------------------------------------------------ k_post
// One workgroup per image, for capacity <= kPostCap candidates, C <=
// kPostMaxClasses and C * max_boxes <= kPostKeptMax.  Everything between the
// candidate list and the outputs stays in LDS: keys, the candidates' boxes
// (gathered once, all loads in flight together), the selections.
// Dynamic LDS: (kPostThreads / 64) * max_boxes * 5 floats.
__global__ __launch_bounds__(kPostThreads) void k_post(const NmsParams p)
{
    extern __shared__ __align__(16) float s_kbox_all[];
    __shared__ unsigned long long s_keys[kPostCap];
    __shared__ unsigned long long s_sorted[kPostRankMax];
    __shared__ float4 s_box[kPostCap];
    __shared__ uint16_t s_kpos[kPostKeptMax];
    __shared__ int s_off[kPostMaxClasses];      // segment offsets, later output offsets
    __shared__ int s_seg[kPostMaxClasses];      // segment offsets (kept)
    __shared__ int s_cnt[kPostMaxClasses];
    __shared__ int s_kept[kPostMaxClasses];
    __shared__ int s_wave[kPostThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, C = p.C, mb = p.max_boxes;

    const int raw = p.img_count[b];
    const int M = raw < p.cand_cap ? raw : p.cand_cap;
    for (int c = tid; c < C; c += kPostThreads) { s_cnt[c] = 0; s_kept[c] = 0; }
    __syncthreads();
    if (tid == 0) {
        p.img_count[b] = 0;                                   // leaves the counter clean for the next call
        if (raw > p.cand_cap) atomicOr(p.status, YH_STATUS_CANDIDATE_OVERFLOW);
    }
    const unsigned long long *cand = p.cand + (long long)b * p.cand_cap;
    for (int i = tid; i < M; i += kPostThreads) {
        const unsigned long long k = cand[i];
        s_keys[i] = k;
        atomicAdd(&s_cnt[key_class(k)], 1);
    }
    __syncthreads();
    // sort by the full key: class ascending, score descending, row ascending
    const unsigned long long *sorted;
    if (M <= kPostRankMax) {
        for (int i = tid; i < M; i += kPostThreads) {
            const unsigned long long key = s_keys[i];
            int rank = 0;
            for (int j = 0; j < M; ++j) rank += s_keys[j] < key ? 1 : 0;   // broadcast reads; keys are unique
            s_sorted[rank] = key;
        }
        __syncthreads();
        sorted = s_sorted;
    } else {
        int n2 = kPostRankMax;
        while (n2 < M) n2 <<= 1;
        bitonic_sort<kPostThreads>(s_keys, M, n2);
        sorted = s_keys;
    }
    // every candidate's box, one gather with all loads in flight
    const float4 *boxes4 = reinterpret_cast<const float4 *>(p.boxes) + (long long)b * p.N;
    for (int i = tid; i < M; i += kPostThreads) s_box[i] = boxes4[key_row(sorted[i])];
    for (int c = tid; c < C; c += kPostThreads) {
        const int n = s_cnt[c];
        s_off[c] = n;
        if (p.out_class_candidates) p.out_class_candidates[b * C + c] = n;
    }
    __syncthreads();
    block_exclusive_scan<kPostThreads, kPostMaxClasses / kPostThreads>(s_off, C, s_wave);
    for (int c = tid; c < C; c += kPostThreads) s_seg[c] = s_off[c];
    __syncthreads();

    // the wavefronts take the non-empty class segments in turn
    float (*kbox)[5] = reinterpret_cast<float (*)[5]>(s_kbox_all + (size_t)wave * mb * 5);
    for (int c = wave; c < C; c += kPostThreads / 64) {
        const int cnt = s_cnt[c];
        if (cnt == 0) continue;
        const int off = s_seg[c];
        const int kept = wave_nms<true>(sorted + off, cnt, nullptr, s_box + off, kbox, p.iou_thr, mb,
                                        nullptr, nullptr, s_kpos + c * mb, off);
        if ((tid & 63) == 0) s_kept[c] = kept;
    }
    __syncthreads();
    for (int c = tid; c < C; c += kPostThreads) s_off[c] = s_kept[c];
    __syncthreads();
    block_exclusive_scan<kPostThreads, kPostMaxClasses / kPostThreads>(s_off, C, s_wave);
    // concatenation of the classes' selections (YOLOv3.py:362-367)
    const int cap = C * mb;
    for (int e = tid; e < cap; e += kPostThreads) {
        const int c = e / mb, j = e - c * mb;
        if (j < s_kept[c]) {
            const int pos = s_kpos[e];
            const unsigned long long key = sorted[pos];
            const long long dst = (long long)b * cap + s_off[c] + j;
            p.out_scores[dst] = key_score(key);
            reinterpret_cast<float4 *>(p.out_boxes)[dst] = s_box[pos];
            p.out_class[dst] = c;
            p.out_index[dst] = key_row(key);
        }
    }
    if (tid == 0) p.out_count[b] = s_off[C - 1] + s_kept[C - 1];
}

// ------------------------------------------------ general path: bin, sort+nms, finalize
// One workgroup per image: class histogram, exclusive scan, scatter into class
// segments.  The order inside a segment is arbitrary; the sort that follows
// makes the result independent of it because keys are unique.
__global__ __launch_bounds__(kBinThreads) void k_bin(const NmsParams p)
{
    __shared__ int s_hist[YH_MAX_CLASSES];
    __shared__ int s_wave[kBinThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x, C = p.C;
    const int raw = p.img_count[b];
    const int M = raw < p.cand_cap ? raw : p.cand_cap;
    for (int c = tid; c < C; c += kBinThreads) s_hist[c] = 0;
    __syncthreads();
    if (tid == 0) {
        p.img_count[b] = 0;                                   // leaves the counter clean for the next call
        if (raw > p.cand_cap) atomicOr(p.status, YH_STATUS_CANDIDATE_OVERFLOW);
    }
    const unsigned long long *keys = p.cand + (long long)b * p.cand_cap;
    for (int i = tid; i < M; i += kBinThreads) atomicAdd(&s_hist[key_class(keys[i])], 1);
    __syncthreads();
    for (int c = tid; c < C; c += kBinThreads) {
        const int n = s_hist[c];
        p.seg_cnt[b * C + c] = n;
        if (p.out_class_candidates) p.out_class_candidates[b * C + c] = n;
    }
    __syncthreads();
    block_exclusive_scan<kBinThreads, YH_MAX_CLASSES / kBinThreads>(s_hist, C, s_wave);
    for (int c = tid; c < C; c += kBinThreads) p.seg_off[b * C + c] = s_hist[c];
    __syncthreads();
    unsigned long long *out = p.seg_keys + (long long)b * p.cand_cap;
    for (int i = tid; i < M; i += kBinThreads) {
        const unsigned long long k = keys[i];
        const int pos = atomicAdd(&s_hist[key_class(k)], 1);
        out[pos] = k;
    }
}

// One workgroup per (image, class) segment.
__global__ __launch_bounds__(kSegThreads) void k_sort_nms(const NmsParams p)
{
    __shared__ unsigned long long s_keys[kSortLds];
    __shared__ float s_kbox[YH_MAX_NMS_BOXES][5];
    const int seg = blockIdx.x, tid = threadIdx.x;
    const int b = seg / p.C;
    const int cnt = p.seg_cnt[seg];
    if (cnt == 0) {
        if (tid == 0) p.kept_cnt[seg] = 0;
        return;
    }
    unsigned long long *keys = p.seg_keys + (long long)b * p.cand_cap + p.seg_off[seg];

    if (cnt <= 64) {
        // rank sort inside one wavefront; keys are unique
        if (tid < 64) {
            const unsigned long long key = tid < cnt ? keys[tid] : ~0ull;
            const int lo = (int)(uint32_t)key, hi = (int)(uint32_t)(key >> 32);
            int rank = 0;
            for (int j = 0; j < cnt; ++j) {
                const unsigned long long kj =
                    ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(hi, j) << 32) |
                    (uint32_t)__builtin_amdgcn_readlane(lo, j);
                rank += kj < key ? 1 : 0;
            }
            if (tid < cnt) s_keys[rank] = key;
        }
        __syncthreads();
        if (tid < cnt) keys[tid] = s_keys[tid];
    } else if (cnt <= kSortLds) {
        int n2 = 128;
        while (n2 < cnt) n2 <<= 1;
        for (int i = tid; i < cnt; i += kSegThreads) s_keys[i] = keys[i];
        __syncthreads();
        bitonic_sort<kSegThreads>(s_keys, cnt, n2);
        for (int i = tid; i < cnt; i += kSegThreads) keys[i] = s_keys[i];
    } else {
        int n2 = kSortLds;
        while (n2 < cnt) n2 <<= 1;
        __syncthreads();
        bitonic_sort<kSegThreads>(keys, cnt, n2);
    }
    __syncthreads();          // sorted keys in global memory, visible to wave 0
    if (tid >= 64) return;
    const int kept = wave_nms<false>(keys, cnt, reinterpret_cast<const float4 *>(p.boxes) + (long long)b * p.N,
                                     nullptr, s_kbox, p.iou_thr, p.max_boxes,
                                     p.kept_row + (long long)seg * p.max_boxes,
                                     p.kept_score + (long long)seg * p.max_boxes, nullptr, 0);
    if (tid == 0) p.kept_cnt[seg] = kept;
}

// One workgroup per image: concatenate the classes' selections.
__global__ __launch_bounds__(kFinThreads) void k_finalize(const NmsParams p)
{
    __shared__ int s_off[YH_MAX_CLASSES];
    __shared__ int s_wave[kFinThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x, C = p.C;
    for (int c = tid; c < C; c += kFinThreads) s_off[c] = p.kept_cnt[b * C + c];
    __syncthreads();
    block_exclusive_scan<kFinThreads, YH_MAX_CLASSES / kFinThreads>(s_off, C, s_wave);
    write_outputs<kFinThreads>(p, b, s_off, p.kept_cnt + (long long)b * C);
}

hipError_t launch_nms_pipeline(const NmsParams &p, hipStream_t stm)
{
    if (p.B == 0) return hipSuccess;
    if (p.cand_cap <= kPostCap && p.C <= kPostMaxClasses && p.max_boxes <= kPostMaxBoxes &&
        p.C * p.max_boxes <= kPostKeptMax) {
        const size_t lds = (size_t)(kPostThreads / 64) * p.max_boxes * 5 * sizeof(float);
        hipLaunchKernelGGL(k_post, dim3(p.B), dim3(kPostThreads), lds, stm, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_bin, dim3(p.B), dim3(kBinThreads), 0, stm, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sort_nms, dim3((uint32_t)p.B * (uint32_t)p.C), dim3(kSegThreads), 0, stm, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_finalize, dim3(p.B), dim3(kFinThreads), 0, stm, p);
    return hipGetLastError();
}

}  // namespace yh
