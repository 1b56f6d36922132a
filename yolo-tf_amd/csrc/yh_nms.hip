// Candidates -> detections as stand-alone launches: class binning,
// per-(image, class) sort by (score descending, box row ascending), exact
// greedy NMS and the gather / concat of the outputs.  Replaces
// YOLOv3.py:355-368 / YOLOv2.py:197-210.  The building blocks (wave_nms,
// post_image) are in yh_post.h.
//
// Launch shapes, chosen from the candidate capacity the caller passes:
//  * k_post, one workgroup per image, for capacity <= kPostCap: one launch, no
//    intermediate global traffic;
//  * k_bin -> k_sort_nms -> k_finalize for larger capacities: class histogram
//    and scatter per image, then one workgroup per (image, class) segment.
#include "yh_post.h"

namespace yh {

constexpr int kBinThreads = 1024;
constexpr int kSegThreads = 256;
constexpr int kSortLds = 2048;       // keys sorted in LDS by k_sort_nms; longer segments sort in global memory
constexpr int kFinThreads = 256;

// ------------------------------------------------------------------ k_post
// All LDS is dynamic and sized by the call's bounds (capacity, classes, boxes per
// class), not by the kernel's limits: with a small capacity a workgroup fits into
// a compute unit next to the streaming kernel's workgroups, so that on a second
// stream it overlaps the next batch's streaming pass instead of waiting behind it.
__global__ __launch_bounds__(kPostThreads) void k_post(const NmsParams p)
{
    extern __shared__ __align__(16) unsigned char s_raw[];
    post_image_dyn<kPostThreads>(p, blockIdx.x, s_raw);
}

// ------------------------------------------------ general path: bin, sort+nms, finalize
// One workgroup per image: class histogram, exclusive scan, scatter into class
// segments.  The order inside a segment is arbitrary; the sort that follows
// makes the result independent of it because keys are unique.  When more
// candidates were found than the capacity holds, the segments are cut off at the
// capacity in class order and the image is flagged.
__global__ __launch_bounds__(kBinThreads) void k_bin(const NmsParams p)
{
    __shared__ int s_cur[YH_MAX_CLASSES];    // histogram, then each segment's write cursor
    __shared__ int s_end[YH_MAX_CLASSES];    // count, then each segment's end
    __shared__ int s_wave[kBinThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x, C = p.C, cap = p.cand_cap;
    const int raw = p.img_count[b];
    for (int c = tid; c < C; c += kBinThreads) s_cur[c] = 0;
    __syncthreads();
    cand_for_each<kBinThreads, false>(p, b, raw, [&](unsigned long long k, float4) {
        atomicAdd(&s_cur[key_class(k)], 1);
    });
    __syncthreads();
    for (int c = tid; c < C; c += kBinThreads) s_end[c] = s_cur[c];
    __syncthreads();
    block_exclusive_scan<kBinThreads, YH_MAX_CLASSES / kBinThreads>(s_cur, C, s_wave);
    const int found = s_cur[C - 1] + s_end[C - 1];
    __syncthreads();
    for (int c = tid; c < C; c += kBinThreads) {
        const int o = s_cur[c] < cap ? s_cur[c] : cap;
        const int n = s_end[c] < cap - o ? s_end[c] : cap - o;
        p.seg_off[b * C + c] = o;
        p.seg_cnt[b * C + c] = n;
        if (p.out_class_candidates) p.out_class_candidates[b * C + c] = n;
        s_cur[c] = o;
        s_end[c] = o + n;
    }
    if (tid == 0) {
        p.img_count[b] = 0;                                   // leaves the counter clean for the next call
        p.status[b] = image_status(found, raw, cap);
    }
    __syncthreads();
    unsigned long long *out = p.seg_keys + (long long)b * cap;
    float4 *fill = p.boxes_fill ? reinterpret_cast<float4 *>(p.boxes_fill) + (long long)b * p.N : nullptr;
    if (fill) {
        // the later launches find a candidate's box under its row
        cand_for_each<kBinThreads, true>(p, b, raw, [&](unsigned long long k, float4 box) {
            const int c = key_class(k);
            const int pos = atomicAdd(&s_cur[c], 1);
            if (pos < s_end[c]) {
                out[pos] = k;
                fill[key_row(k)] = box;
            }
        });
    } else {
        cand_for_each<kBinThreads, false>(p, b, raw, [&](unsigned long long k, float4) {
            const int c = key_class(k);
            const int pos = atomicAdd(&s_cur[c], 1);
            if (pos < s_end[c]) out[pos] = k;
        });
    }
}

// The same binning for long candidate lists, spread over kBinSlice-sized slices of an image's
// list (one workgroup each) in two launches: k_bin_count adds the slices' class histograms
// into class_hist; k_bin_scatter scans the image's histogram (every slice's workgroup does,
// redundantly: C values), reserves its share of every class segment with one global atomic
// per class it holds, and scatters its slice.  One workgroup per image walked the whole list
// twice with every thread contending for the C counters in LDS.  For candidates that live in
// the list only (no private slots, no slot-indexed boxes).  class_hist, class_cursor and
// img_count are zero between calls: k_sort_nms clears them.
constexpr int kBinSlice = 2048;
constexpr int kBinSliceThreads = 256;

__device__ __forceinline__ void slice_range(const NmsParams &p, int b, int sub, int &begin, int &end, int &raw)
{
    raw = p.img_count[b];
    const int nl = raw < p.cand_cap ? raw : p.cand_cap;
    begin = sub * kBinSlice;
    end = begin + kBinSlice < nl ? begin + kBinSlice : nl;
}

// The class tables of these kernels are dynamic LDS sized by the call's class count (launch_nms_pipeline):
// a table for the largest class count the ABI admits would hold a compute unit to three such workgroups.
__global__ __launch_bounds__(kBinSliceThreads) void k_bin_count(const NmsParams p, int slices)
{
    extern __shared__ int s_h[];             // [C]
    const int b = blockIdx.x / slices, sub = blockIdx.x - b * slices, tid = threadIdx.x, C = p.C;
    int begin, end, raw;
    slice_range(p, b, sub, begin, end, raw);
    if (begin >= end) return;
    for (int c = tid; c < C; c += kBinSliceThreads) s_h[c] = 0;
    __syncthreads();
    const unsigned long long *cand = p.cand + (long long)b * p.cand_cap;
    for (int i = begin + tid; i < end; i += kBinSliceThreads) atomicAdd(&s_h[key_class(cand[i])], 1);
    __syncthreads();
    for (int c = tid; c < C; c += kBinSliceThreads)
        if (s_h[c]) atomicAdd(&p.class_hist[b * C + c], s_h[c]);
}

__global__ __launch_bounds__(kBinSliceThreads) void k_bin_scatter(const NmsParams p, int slices)
{
    extern __shared__ int s_tab[];           // [3][C]
    __shared__ int s_wave[kBinSliceThreads / 64];
    const int b = blockIdx.x / slices, sub = blockIdx.x - b * slices, tid = threadIdx.x, C = p.C, cap = p.cand_cap;
    int *s_cur = s_tab;                      // histogram, then each segment's write cursor for this slice
    int *s_end = s_tab + C;                  // count, then each segment's end
    int *s_mine = s_tab + 2 * C;             // this slice's histogram
    int begin, end, raw;
    slice_range(p, b, sub, begin, end, raw);
    if (sub > 0 && begin >= end) return;
    for (int c = tid; c < C; c += kBinSliceThreads) { s_cur[c] = p.class_hist[b * C + c]; s_mine[c] = 0; }
    __syncthreads();
    for (int c = tid; c < C; c += kBinSliceThreads) s_end[c] = s_cur[c];
    __syncthreads();
    block_exclusive_scan<kBinSliceThreads, YH_MAX_CLASSES / kBinSliceThreads>(s_cur, C, s_wave);
    const int found = s_cur[C - 1] + s_end[C - 1];
    __syncthreads();
    for (int c = tid; c < C; c += kBinSliceThreads) {
        const int o = s_cur[c] < cap ? s_cur[c] : cap;
        const int n = s_end[c] < cap - o ? s_end[c] : cap - o;
        if (sub == 0) {
            p.seg_off[b * C + c] = o;
            p.seg_cnt[b * C + c] = n;
            if (p.out_class_candidates) p.out_class_candidates[b * C + c] = n;
        }
        s_cur[c] = o;
        s_end[c] = o + n;
    }
    if (sub == 0 && tid == 0) p.status[b] = image_status(found, raw, cap);
    __syncthreads();
    const unsigned long long *cand = p.cand + (long long)b * cap;
    for (int i = begin + tid; i < end; i += kBinSliceThreads) atomicAdd(&s_mine[key_class(cand[i])], 1);
    __syncthreads();
    for (int c = tid; c < C; c += kBinSliceThreads)
        if (s_mine[c]) s_cur[c] += atomicAdd(&p.class_cursor[b * C + c], s_mine[c]);
    __syncthreads();
    unsigned long long *out = p.seg_keys + (long long)b * cap;
    for (int i = begin + tid; i < end; i += kBinSliceThreads) {
        const unsigned long long k = cand[i];
        const int c = key_class(k);
        const int pos = atomicAdd(&s_cur[c], 1);
        if (pos < s_end[c]) out[pos] = k;
    }
}

// k_bin_scatter for class counts that leave room in LDS for the slice's keys: the slice (eight keys per thread, held
// in registers) is brought into class order in LDS first (a counting sort: histogram, scan, placement), and written out in that order, so
// that the lanes of a store go to a few runs of consecutive keys instead of 64 segments (8-byte stores spread
// over 64 lines each were what the plain scatter's time went into).  Needs class_hist complete (hist_ready, or
// k_bin_count ahead of it), like k_bin_scatter.
constexpr int kSortedSlice = 4096;
constexpr int kSortedThreads = 512;
constexpr int kSortedMaxClasses = 1024;          // five class tables of this size beside 32 KiB of keys

__global__ __launch_bounds__(kSortedThreads) void k_bin_scatter_sorted(const NmsParams p, int slices)
{
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ int s_wave[kSortedThreads / 64];
    unsigned long long *s_out = reinterpret_cast<unsigned long long *>(s_dyn);
    int *s_tab = reinterpret_cast<int *>(s_out + kSortedSlice);
    const int b = blockIdx.x / slices, sub = blockIdx.x - b * slices, tid = threadIdx.x, C = p.C, cap = p.cand_cap;
    int *s_cur = s_tab;              // the image's histogram, then each segment's start
    int *s_end = s_tab + C;          // count, then each segment's end
    int *s_mine = s_tab + 2 * C;     // this slice's histogram
    int *s_loc = s_tab + 3 * C;      // where a class starts in the sorted slice, then its write cursor there
    int *s_gpos = s_tab + 4 * C;     // global position of a class's keys minus their position in the sorted slice
    const int raw = p.img_count[b];
    const int nl = raw < cap ? raw : cap;
    const int begin = sub * kSortedSlice;
    const int end = begin + kSortedSlice < nl ? begin + kSortedSlice : nl;
    if (sub > 0 && begin >= end) return;
    const int n = end > begin ? end - begin : 0;
    for (int c = tid; c < C; c += kSortedThreads) { s_cur[c] = p.class_hist[b * C + c]; s_mine[c] = 0; }
    __syncthreads();
    for (int c = tid; c < C; c += kSortedThreads) s_end[c] = s_cur[c];
    const unsigned long long *cand = p.cand + (long long)b * cap + begin;
    constexpr int kPer = kSortedSlice / kSortedThreads;
    unsigned long long key[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = tid + j * kSortedThreads;
        key[j] = i < n ? cand[i] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if (tid + j * kSortedThreads < n) atomicAdd(&s_mine[key_class(key[j])], 1);
    __syncthreads();
    block_exclusive_scan<kSortedThreads, kSortedMaxClasses / kSortedThreads>(s_cur, C, s_wave);
    const int found = s_cur[C - 1] + s_end[C - 1];
    __syncthreads();
    for (int c = tid; c < C; c += kSortedThreads) {
        const int o = s_cur[c] < cap ? s_cur[c] : cap;
        const int m = s_end[c] < cap - o ? s_end[c] : cap - o;
        if (sub == 0) {
            p.seg_off[b * C + c] = o;
            p.seg_cnt[b * C + c] = m;
            if (p.out_class_candidates) p.out_class_candidates[b * C + c] = m;
        }
        s_cur[c] = o;
        s_end[c] = o + m;
        s_loc[c] = s_mine[c];
    }
    if (sub == 0 && tid == 0) p.status[b] = image_status(found, raw, cap);
    __syncthreads();
    block_exclusive_scan<kSortedThreads, kSortedMaxClasses / kSortedThreads>(s_loc, C, s_wave);
    __syncthreads();
    for (int c = tid; c < C; c += kSortedThreads) {
        const int mine = s_mine[c];
        const int at = mine ? s_cur[c] + atomicAdd(&p.class_cursor[b * C + c], mine) : 0;   // this slice's share of the segment
        s_gpos[c] = at - s_loc[c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if (tid + j * kSortedThreads < n) s_out[atomicAdd(&s_loc[key_class(key[j])], 1)] = key[j];
    __syncthreads();
    unsigned long long *out = p.seg_keys + (long long)b * cap;
    for (int j = tid; j < n; j += kSortedThreads) {
        const unsigned long long k = s_out[j];
        const int c = key_class(k);
        const int pos = s_gpos[c] + j;
        if (pos < s_end[c]) out[pos] = k;
    }
}

// ---- k_sort_nms: one workgroup per (image, class) segment ------------------------------
// The greedy walk rarely looks past the first few hundred candidates of a long segment, so a
// segment that fits into LDS is not sorted whole: a radix selection on the score bits of the
// keys (two 8-bit levels at most) picks a prefix of the score order that holds at least
// `target` candidates, only that prefix is sorted, and the walk runs on it.  If the walk uses
// the prefix up before max_boxes boxes are selected, the whole segment is sorted and walked
// again: the result is that of the full sort in every case.
constexpr int kSelMax = 1024;        // longest prefix sorted on its own

// The prefix a call aims at, and the room its selection needs (a power of two: the prefix is sorted padded to
// one; 0: the bounds do not allow a prefix, whole segments are sorted).  The room is dynamic LDS behind the
// selected boxes, sized per call: 4 KiB at nms_max_boxes <= 204, so that more of these workgroups fit beside
// another stream's kernels than with room for kSelMax keys in every launch.
__host__ __device__ inline int prefix_target(int max_boxes) { return 2 * max_boxes > 240 ? 2 * max_boxes : 240; }
__host__ __device__ inline int prefix_room(int max_boxes)
{
    const int target = prefix_target(max_boxes), most = target + (target >> 2);
    if (most > kSelMax) return 0;
    int room = 128;
    while (room < most) room <<= 1;
    return room;
}
__host__ __device__ inline size_t kept_boxes_bytes(int max_boxes) { return ((size_t)max_boxes * 5 * sizeof(float) + 15) & ~(size_t)15; }

// Bins of s_hist -> the first bin at which the running count reaches `target`; wave 0 only.
// Returns the bin, its inclusive running count in *cum; `before` is added to the counts.  s_excl, when
// given, receives every bin's exclusive running count (without `before`).
__device__ __forceinline__ int first_bin_reaching(const int *s_hist, int before, int target, int *cum, int *s_excl = nullptr)
{
    const int lane = threadIdx.x & 63;
    int a[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = s_hist[4 * lane + k]; sum += a[k]; }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    int c = before + incl - sum, found = -1, at = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (s_excl) s_excl[4 * lane + k] = c - before;
        c += a[k];
        if (found < 0 && c >= target) { found = 4 * lane + k; at = c; }
    }
    const unsigned long long m = __ballot(found >= 0);
    if (m == 0ull) { *cum = before + __builtin_amdgcn_readlane(incl, 63); return 255; }
    const int src = __ffsll((long long)m) - 1;
    *cum = __builtin_amdgcn_readlane(at, src);
    return __builtin_amdgcn_readlane(found, src);
}

// The tables select_prefix leaves for sort_selected.  The selection is grouped by the top byte of the keys'
// score bits, and the keys of the boundary bin d1 (in practice nearly all of them: the top byte is sign and
// exponent, and the best scores of a segment share theirs) by the next byte.  end1 / end2: one past a group's
// last key in the selection; count1 / count2: its size.
struct PrefixBins {
    int end1[kSegThreads], count1[kSegThreads];
    int end2[kSegThreads], count2[kSegThreads];
    int d1;
};

// Block-wide.  s_keys[0..cnt) -> s_sel[0..n): the n smallest keys for some n in [target, room], group after
// group in key order, any order inside a group (bins has the layout); returns n, or 0 when no such prefix
// exists at the two levels' resolution.
__device__ inline int select_prefix(const unsigned long long *s_keys, int cnt, int target, int room,
                                    unsigned long long *s_sel, int *s_ctl, PrefixBins &bins)
{
    const int tid = threadIdx.x;
    constexpr int kShift1 = 44, kShift2 = 36;              // the top two bytes of the 32 score bits at [51:20]
    bins.count1[tid] = 0;                                   // kSegThreads == 256 bins
    bins.count2[tid] = 0;
    __syncthreads();
    for (int i = tid; i < cnt; i += kSegThreads) atomicAdd(&bins.count1[(int)(s_keys[i] >> kShift1) & 0xFF], 1);
    __syncthreads();
    if (tid < 64) {
        int cum;
        const int d1 = first_bin_reaching(bins.count1, 0, target, &cum, bins.end1);    // end1: the groups' starts for now
        if (tid == 0) { s_ctl[0] = d1; s_ctl[1] = cum; s_ctl[2] = cum - bins.count1[d1]; }
    }
    __syncthreads();
    const int d1 = s_ctl[0], cum1 = s_ctl[1], below = s_ctl[2];
    __syncthreads();
    for (int i = tid; i < cnt; i += kSegThreads) {          // the boundary bin by the next byte
        const unsigned long long k = s_keys[i];
        if (((int)(k >> kShift1) & 0xFF) == d1) atomicAdd(&bins.count2[(int)(k >> kShift2) & 0xFF], 1);
    }
    __syncthreads();
    // whole when it does not overshoot (a target nobody reaches: the last bin, every key of the bin)
    if (tid < 64) {
        int cum;
        const int d = first_bin_reaching(bins.count2, below, cum1 > target + (target >> 2) ? target : 0x7fffffff, &cum, bins.end2);
        if (tid == 0) { s_ctl[0] = d; s_ctl[1] = cum; }
    }
    __syncthreads();
    const int d2 = s_ctl[0], n = s_ctl[1];
    if (n > room) return 0;                                 // s_sel holds `room` keys
    if (tid == 0) bins.d1 = d1;
    bins.end2[tid] += below;                                // starts inside the selection
    __syncthreads();
    for (int i = tid; i < cnt; i += kSegThreads) {
        const unsigned long long k = s_keys[i];
        const int b1 = (int)(k >> kShift1) & 0xFF, b2 = (int)(k >> kShift2) & 0xFF;
        if (b1 < d1) s_sel[atomicAdd(&bins.end1[b1], 1)] = k;
        else if (b1 == d1 && b2 <= d2) s_sel[atomicAdd(&bins.end2[b2], 1)] = k;
    }
    __syncthreads();
    return n;
}

// Sorts the selection select_prefix left (keys are unique): a key's place is its group's start plus the number
// of smaller keys in its group, a handful of comparisons where the ranking of the whole selection against
// itself takes n per key.  Every thread holds its keys (n <= KPT * kSegThreads) and their places before any is
// written.
template <int KPT>
__device__ __forceinline__ void sort_selected(unsigned long long *s_sel, int n, const PrefixBins &bins)
{
    const int tid = threadIdx.x;
    const int d1 = bins.d1;
    unsigned long long key[KPT];
    int place[KPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int i = u * kSegThreads + tid;
        key[u] = i < n ? s_sel[i] : 0ull;
        place[u] = 0;
        if (i < n) {
            const int b1 = (int)(key[u] >> 44) & 0xFF, b2 = (int)(key[u] >> 36) & 0xFF;
            const int end = b1 < d1 ? bins.end1[b1] : bins.end2[b2];
            const int start = end - (b1 < d1 ? bins.count1[b1] : bins.count2[b2]);
            int r = start;
            for (int j = start; j < end; ++j) r += s_sel[j] < key[u] ? 1 : 0;
            place[u] = r;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < KPT; ++u)
        if (u * kSegThreads + tid < n) s_sel[place[u]] = key[u];
    __syncthreads();
}

// Sorts keys[0..n) in place by rank (keys are unique): a thread holds KPT keys in registers, reads the whole
// list once as broadcast LDS reads (two keys per read) and counts the smaller ones; one barrier, then every
// key goes to its rank.  For the few hundred keys of a selected prefix this is a third of the bitonic
// network's time, which has a barrier per stage (36 stages at 256 keys).  n <= KPT * kSegThreads.
template <int KPT>
__device__ __forceinline__ void rank_sort_inplace(unsigned long long *keys, int n)
{
    const int tid = threadIdx.x;
    unsigned long long key[KPT];
    int rank[KPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int i = u * kSegThreads + tid;
        key[u] = i < n ? keys[i] : 0ull;
        rank[u] = 0;
    }
    int j = 0;
    for (; j + 4 <= n; j += 4) {
        const unsigned long long k0 = keys[j], k1 = keys[j + 1], k2 = keys[j + 2], k3 = keys[j + 3];
#pragma unroll
        for (int u = 0; u < KPT; ++u)
            rank[u] += (k0 < key[u] ? 1 : 0) + (k1 < key[u] ? 1 : 0) + (k2 < key[u] ? 1 : 0) + (k3 < key[u] ? 1 : 0);
    }
    for (; j < n; ++j) {
        const unsigned long long k0 = keys[j];
#pragma unroll
        for (int u = 0; u < KPT; ++u) rank[u] += k0 < key[u] ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < KPT; ++u)
        if (u * kSegThreads + tid < n) keys[rank[u]] = key[u];
    __syncthreads();
}

__device__ __forceinline__ void sort_prefix(unsigned long long *keys, int n)
{
    if (n <= kSegThreads) rank_sort_inplace<1>(keys, n);
    else if (n <= 2 * kSegThreads) rank_sort_inplace<2>(keys, n);
    else if (n <= 4 * kSegThreads) rank_sort_inplace<4>(keys, n);
    else {
        int n2 = 1024;
        while (n2 < n) n2 <<= 1;
        bitonic_sort<kSegThreads>(keys, n, n2);
    }
}

// wave_nms's walk by the four wavefronts of a workgroup (all kSegThreads threads call it; the
// result is the same in every thread).  Every wavefront holds the tile's 64 candidates in its
// registers; the tests against the boxes selected so far and the 64 x 64 tests inside the tile
// are split four ways (a quarter of the selected list, sixteen columns of the bit matrix per
// wavefront) and meet in LDS; wavefront 0 resolves the tile on the scalar unit and appends the
// survivors.  Same decisions in the same order as wave_nms.
struct BlockNmsLds {
    unsigned long long dead[kSegThreads / 64];
    uint16_t m16[64 * (kSegThreads / 64)];
    int kept;
};

template <typename BoxOf>
__device__ inline int block_nms(const unsigned long long *keys, int cnt, BoxOf box_of, float (*kbox)[5],
                                float thr, int max_boxes, int32_t *krow, float *kscore, float4 *kbox_out, BlockNmsLds &l,
                                const float4 *cbox = nullptr)
{
    static_assert(kSegThreads == 256, "sixteen columns of the bit matrix per wavefront");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int kept = 0;
    for (int t = 0; t < cnt && kept < max_boxes; t += 64) {
        const int i = t + lane;
        const bool valid = i < cnt;
        const unsigned long long key = valid ? keys[i] : 0ull;
        const int n = key_row(key);
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) bx = cbox ? cbox[i] : box_of(n);
        const float ymin = fminf(bx.x, bx.z), xmin = fminf(bx.y, bx.w);
        const float ymax = fmaxf(bx.x, bx.z), xmax = fmaxf(bx.y, bx.w);
        const float area = (ymax - ymin) * (xmax - xmin);
        // against the boxes selected by earlier tiles: this wavefront's share of the list
        bool alive = valid;
        for (int j = kept - 1 - wave; j >= 0; j -= kSegThreads / 64) {
            if (__ballot(alive) == 0ull) break;
            const float *kb = kbox[j];
            // the intersections of the whole wavefront first: most selected boxes touch none of the tile's candidates
            const float inter = intersection(ymin, xmin, ymax, xmax, kb[0], kb[1], kb[2], kb[3]);
            if (__ballot(alive && inter > 0.0f) == 0ull) continue;
            if (alive && suppresses_at(inter, area, kb[4], thr)) alive = false;
        }
        const unsigned long long dead = __ballot(valid && !alive);
        if (lane == 0) l.dead[wave] = dead;
        __syncthreads();
        unsigned long long alive_mask = __ballot(valid);
#pragma unroll
        for (int w = 0; w < kSegThreads / 64; ++w) alive_mask &= ~l.dead[w];
        alive = (alive_mask >> lane) & 1ull;
        // inside the tile: bit j of the mask = "the earlier lane j would suppress me"; columns 16 * wave ..
        uint32_t m = 0u;
        if (cbox) {
            // the columns' boxes as broadcast reads of the LDS copy, eight asked for before the first is
            // used: eight independent tests (column after column out of a lane's registers, every test waits
            // for the scalar unit to hand the column's five values over)
            constexpr int kCols = 8;
#pragma unroll 1
            for (int c0 = 0; c0 < 16; c0 += kCols) {
                if (((alive_mask >> (16 * wave + c0)) & ((1ull << kCols) - 1ull)) == 0ull) continue;   // nobody left in these columns
                float4 cb[kCols];
#pragma unroll
                for (int c = 0; c < kCols; ++c) {
                    const int j = t + 16 * wave + c0 + c;
                    cb[c] = j < cnt ? cbox[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int c = 0; c < kCols; ++c) {
                    const int j = 16 * wave + c0 + c;
                    const float jy0 = fminf(cb[c].x, cb[c].z), jx0 = fminf(cb[c].y, cb[c].w);
                    const float jy1 = fmaxf(cb[c].x, cb[c].z), jx1 = fmaxf(cb[c].y, cb[c].w);
                    // the intersections of the whole wavefront first: most columns touch no later candidate
                    const float inter = intersection(ymin, xmin, ymax, xmax, jy0, jx0, jy1, jx1);
                    const bool mine = alive && lane > j && ((alive_mask >> j) & 1ull);
                    if (__ballot(mine && inter > 0.0f) == 0ull) continue;
                    if (mine && suppresses_at(inter, area, (jy1 - jy0) * (jx1 - jx0), thr)) m |= 1u << (c0 + c);
                }
            }
        } else
        for (unsigned long long rem = alive_mask & (0xFFFFull << (16 * wave)); rem != 0ull; rem &= rem - 1ull) {
            const int j = __ffsll((long long)rem) - 1;
            const float jy0 = lane_bcast(ymin, j), jx0 = lane_bcast(xmin, j);
            const float jy1 = lane_bcast(ymax, j), jx1 = lane_bcast(xmax, j);
            const float ja = lane_bcast(area, j);
            if (alive && lane > j && suppresses(ymin, xmin, ymax, xmax, area, jy0, jx0, jy1, jx1, ja, thr))
                m |= 1u << (j - 16 * wave);
        }
        l.m16[lane * (kSegThreads / 64) + wave] = (uint16_t)m;
        __syncthreads();
        if (wave == 0) {
            const unsigned long long mymask = *reinterpret_cast<const unsigned long long *>(&l.m16[lane * (kSegThreads / 64)]);
            const unsigned long long keepmask = resolve_tile(alive_mask, mymask, max_boxes - kept);
            if ((keepmask >> lane) & 1ull) {
                const int pos = kept + __popcll(keepmask & ((1ull << lane) - 1ull));
                kbox[pos][0] = ymin; kbox[pos][1] = xmin; kbox[pos][2] = ymax; kbox[pos][3] = xmax;
                kbox[pos][4] = area;
                krow[pos] = n;
                kscore[pos] = key_score(key);
                kbox_out[pos] = bx;
            }
            if (lane == 0) l.kept = kept + __popcll(keepmask);
        }
        __syncthreads();
        kept = l.kept;
    }
    return kept;
}

#ifdef YH_ROWS_TRACE
// measurement build: thread 0 of a segment's workgroup stamps the phases of k_sort_nms (tools/trace_sort.py)
constexpr int kSortTraceSegs = 1 << 14, kSortTraceWords = 8;
__device__ unsigned long long g_sort_trace[kSortTraceSegs * kSortTraceWords];
#define YH_ST(i) do { if (threadIdx.x == 0 && blockIdx.x < (unsigned)kSortTraceSegs) g_sort_trace[(size_t)blockIdx.x * kSortTraceWords + (i)] = wall_clock64(); } while (0)
extern "C" int yh_trace_sort_read(void *dst, size_t bytes, int clear)
{
    const size_t have = sizeof(unsigned long long) * kSortTraceSegs * kSortTraceWords;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_sort_trace));
    if (e != hipSuccess) return (int)e;
    if (clear) return (int)hipMemset(ptr, 0, have);
    return (int)hipMemcpy(dst, ptr, bytes < have ? bytes : have, hipMemcpyDeviceToHost);
}
#else
#define YH_ST(i) do { } while (0)
#endif

__global__ __launch_bounds__(kSegThreads) void k_sort_nms(const NmsParams p)
{
    __shared__ unsigned long long s_keys[kSortLds];
    extern __shared__ __align__(16) unsigned char s_dyn[];    // the selected boxes (max_boxes x 5 floats), then the prefix's keys
    float (*s_kbox)[5] = reinterpret_cast<float (*)[5]>(s_dyn);
    unsigned long long *s_sel = reinterpret_cast<unsigned long long *>(s_dyn + kept_boxes_bytes(p.max_boxes));
    const int sel_room = prefix_room(p.max_boxes);
    __shared__ int s_ctl[4];
    __shared__ PrefixBins s_bins;
    __shared__ BlockNmsLds s_walk;
    const int seg = blockIdx.x, tid = threadIdx.x;
    const int b = seg / p.C;
    YH_ST(0);
    const int cnt = p.seg_cnt[seg];
    if (tid == 0) {                           // the binning's counters, clean for the next call
        p.class_hist[seg] = 0;
        p.class_cursor[seg] = 0;
        if (seg == b * p.C) p.img_count[b] = 0;
    }
    if (cnt == 0) {
        if (tid == 0) p.kept_cnt[seg] = 0;
        return;
    }
    unsigned long long *keys = p.seg_keys + (long long)b * p.cand_cap + p.seg_off[seg];
    auto box_of = [&](int n) { return row_box(p, b, n); };
    float4 *seg_boxes = reinterpret_cast<float4 *>(p.kept_box) + (long long)seg * p.max_boxes;
    auto walk = [&](const unsigned long long *sorted, int n) {          // wave 0
        return wave_nms<false>(sorted, n, box_of, nullptr, s_kbox, p.iou_thr, p.max_boxes,
                               p.kept_row + (long long)seg * p.max_boxes,
                               p.kept_score + (long long)seg * p.max_boxes, nullptr, 0, seg_boxes);
    };
    auto block_walk = [&](const unsigned long long *sorted, int n) {    // the whole workgroup
        return block_nms(sorted, n, box_of, s_kbox, p.iou_thr, p.max_boxes,
                         p.kept_row + (long long)seg * p.max_boxes,
                         p.kept_score + (long long)seg * p.max_boxes, seg_boxes, s_walk);
    };

    if (cnt <= 64) {
        // rank sort inside one wavefront; keys are unique
        if (tid >= 64) return;
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int kept = walk(s_keys, cnt);
        if (tid == 0) p.kept_cnt[seg] = kept;
        return;
    }
    if (cnt <= kSortLds) {
        for (int i = tid; i < cnt; i += kSegThreads) s_keys[i] = keys[i];
        __syncthreads();
        YH_ST(1);
        // a prefix of about 240 keys sorts as 256 (the selection overshoots its target by a few keys)
        const int target = prefix_target(p.max_boxes);
        if (cnt >= 2 * target && sel_room > 0) {
            const int n = select_prefix(s_keys, cnt, target, sel_room, s_sel, s_ctl, s_bins);
            YH_ST(2);
            if (n > 0) {
                if (n <= kSegThreads) sort_selected<1>(s_sel, n, s_bins);
                else if (n <= 2 * kSegThreads) sort_selected<2>(s_sel, n, s_bins);
                else sort_selected<kSelMax / kSegThreads>(s_sel, n, s_bins);
                YH_ST(3);
                // the selection's boxes, decoded once by the whole workgroup into the LDS the unsorted keys leave
                // free (the walk's wavefronts all look at every tile; a gather per tile and wavefront was a memory
                // round trip each)
                float4 *s_cbox = reinterpret_cast<float4 *>(s_keys);
                static_assert(kSelMax * sizeof(float4) <= kSortLds * sizeof(unsigned long long), "the boxes of a prefix fit where the keys were");
                for (int i = tid; i < n; i += kSegThreads) s_cbox[i] = box_of(key_row(s_sel[i]));
                __syncthreads();
                YH_ST(4);
                const int kept = block_nms(s_sel, n, box_of, s_kbox, p.iou_thr, p.max_boxes,
                                           p.kept_row + (long long)seg * p.max_boxes,
                                           p.kept_score + (long long)seg * p.max_boxes, seg_boxes, s_walk, s_cbox);
                YH_ST(5);
#ifdef YH_ROWS_TRACE
                if (tid == 0 && blockIdx.x < (unsigned)kSortTraceSegs) {
                    g_sort_trace[(size_t)blockIdx.x * kSortTraceWords + 6] = ((unsigned long long)cnt << 40) | ((unsigned long long)n << 20) | (unsigned long long)kept;
                    g_sort_trace[(size_t)blockIdx.x * kSortTraceWords + 7] = 1ull | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8) |
                        ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) << 40);
                }
#endif
                if (kept >= p.max_boxes || n == cnt) {
                    if (tid == 0) p.kept_cnt[seg] = kept;
                    return;
                }
                __syncthreads();
                for (int i = tid; i < cnt; i += kSegThreads) s_keys[i] = keys[i];      // the boxes took the keys' place
                __syncthreads();
            }
        }
        sort_prefix(s_keys, cnt);
        const int kept = block_walk(s_keys, cnt);
        if (tid == 0) p.kept_cnt[seg] = kept;
        return;
    }
    int n2 = kSortLds;
    while (n2 < cnt) n2 <<= 1;
    __syncthreads();
    bitonic_sort<kSegThreads>(keys, cnt, n2);
    __syncthreads();          // sorted keys in global memory, visible to the whole workgroup
    const int kept = block_walk(keys, cnt);
    if (tid == 0) p.kept_cnt[seg] = kept;
}

// Concatenates the classes' selections of an image (YOLOv3.py:362-367).  `parts` workgroups
// per image share the C * max_boxes slots (each scans the image's C counts itself): the walk
// is a chain of dependent loads per slot, so it wants many lanes, not many iterations.
constexpr int kFinSlots = 4 * kFinThreads;   // slots per workgroup

__global__ __launch_bounds__(kFinThreads) void k_finalize(const NmsParams p, int parts)
{
    extern __shared__ int s_off[];           // [C]
    __shared__ int s_wave[kFinThreads / 64];
    const int b = blockIdx.x / parts, part = blockIdx.x - b * parts, tid = threadIdx.x, C = p.C, mb = p.max_boxes;
    const int32_t *kept = p.kept_cnt + (long long)b * C;
    for (int c = tid; c < C; c += kFinThreads) s_off[c] = kept[c];
    __syncthreads();
    block_exclusive_scan<kFinThreads, YH_MAX_CLASSES / kFinThreads>(s_off, C, s_wave);
    const long long cap = (long long)C * mb;
    const long long e_end = (long long)(part + 1) * kFinSlots < cap ? (long long)(part + 1) * kFinSlots : cap;
    for (long long e = (long long)part * kFinSlots + tid; e < e_end; e += kFinThreads) {
        const int c = (int)(e / mb), j = (int)(e - (long long)c * mb);
        if (j < kept[c]) {
            const long long src = ((long long)b * C + c) * mb + j;
            const long long dst = (long long)b * cap + s_off[c] + j;
            p.out_scores[dst] = p.kept_score[src];
            reinterpret_cast<float4 *>(p.out_boxes)[dst] = reinterpret_cast<const float4 *>(p.kept_box)[src];
            p.out_class[dst] = c;
            p.out_index[dst] = p.kept_row[src];
        }
    }
    if (part == 0 && tid == 0) p.out_count[b] = s_off[C - 1] + kept[C - 1];
}

bool select_single_launch(int cand_cap, int num_classes, int max_boxes)
{
    return cand_cap <= kPostCap && num_classes <= kPostMaxClasses && max_boxes <= kPostMaxBoxes &&
           num_classes * max_boxes <= kPostKeptMax;
}

bool select_bins_by_slices(int cand_cap, int num_classes, int max_boxes)
{
    return !select_single_launch(cand_cap, num_classes, max_boxes) && cand_cap > kBinSlice;
}

static bool single_launch_shape(const NmsParams &p) { return select_single_launch(p.cand_cap, p.C, p.max_boxes); }

bool select_fits_stream(const NmsParams &p)
{
    return single_launch_shape(p) && post_layout(p.cand_cap, p.C, p.max_boxes, 64).bytes <= kFuseLdsMax;
}

bool select_fits_rows(const NmsParams &p)
{
    return single_launch_shape(p) && post_layout(p.cand_cap, p.C, p.max_boxes, kRowsThreads).bytes <= kRowsLdsMax;
}

hipError_t launch_nms_pipeline(const NmsParams &p, hipStream_t stm)
{
    if (p.B == 0) return hipSuccess;
    if (single_launch_shape(p)) {
        const size_t lds = post_layout(p.cand_cap, p.C, p.max_boxes, kPostThreads).bytes;
        const hipError_t le = raise_lds_limit(reinterpret_cast<const void *>(k_post), lds);
        if (le != hipSuccess) return le;
        hipLaunchKernelGGL(k_post, dim3(p.B), dim3(kPostThreads), lds, stm, p);
        return hipGetLastError();
    }
    hipError_t e;
    if (p.upi == 0 && !p.boxes_fill && !p.cand_box && select_bins_by_slices(p.cand_cap, p.C, p.max_boxes)) {
        const int slices = (p.cand_cap + kBinSlice - 1) / kBinSlice;
        const dim3 grid((uint32_t)p.B * (uint32_t)slices);
        if (!p.hist_ready) {
            hipLaunchKernelGGL(k_bin_count, grid, dim3(kBinSliceThreads), (size_t)p.C * sizeof(int), stm, p, slices);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        if (p.C <= kSortedMaxClasses) {
            const int wide = (p.cand_cap + kSortedSlice - 1) / kSortedSlice;
            const size_t lds = (size_t)kSortedSlice * sizeof(unsigned long long) + (size_t)p.C * 5 * sizeof(int);
            if ((e = raise_lds_limit(reinterpret_cast<const void *>(k_bin_scatter_sorted), lds)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_bin_scatter_sorted, dim3((uint32_t)p.B * (uint32_t)wide), dim3(kSortedThreads), lds, stm, p, wide);
        } else {
            hipLaunchKernelGGL(k_bin_scatter, grid, dim3(kBinSliceThreads), (size_t)p.C * 3 * sizeof(int), stm, p, slices);
        }
    } else {
        hipLaunchKernelGGL(k_bin, dim3(p.B), dim3(kBinThreads), 0, stm, p);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sort_nms, dim3((uint32_t)p.B * (uint32_t)p.C), dim3(kSegThreads),
                       kept_boxes_bytes(p.max_boxes) + (size_t)prefix_room(p.max_boxes) * sizeof(unsigned long long), stm, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int parts = (int)(((long long)p.C * p.max_boxes + kFinSlots - 1) / kFinSlots);
    hipLaunchKernelGGL(k_finalize, dim3((uint32_t)p.B * (uint32_t)parts), dim3(kFinThreads), (size_t)p.C * sizeof(int), stm, p, parts);
    return hipGetLastError();
}

}  // namespace yh
