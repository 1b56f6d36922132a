// Internal declarations shared by the kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "yolohead.h"

namespace yh {

// ---- candidate key -------------------------------------------------------
// key64 = class(12) | ~orderable(score)(32) | box row(20): ascending key order
// is class ascending, score descending, box row ascending.
constexpr int kRowBits = 20;
constexpr int kScoreShift = kRowBits;
constexpr int kClassShift = 52;

__host__ __device__ inline uint32_t orderable_bits(uint32_t u)
{
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending with the float order
}
__host__ __device__ inline uint32_t orderable_bits_inv(uint32_t o)
{
    return (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
}
__device__ inline unsigned long long make_key(int cls, float score, int row)
{
    const uint32_t part = ~orderable_bits(__float_as_uint(score));
    return ((unsigned long long)(uint32_t)cls << kClassShift) |
           ((unsigned long long)part << kScoreShift) | (unsigned long long)(uint32_t)row;
}
__device__ inline int key_class(unsigned long long k) { return (int)(k >> kClassShift); }
__device__ inline int key_row(unsigned long long k) { return (int)(k & ((1u << kRowBits) - 1u)); }
__device__ inline float key_score(unsigned long long k)
{
    const uint32_t part = (uint32_t)(k >> kScoreShift);
    return __uint_as_float(orderable_bits_inv(~part));
}

// ---- arithmetic shared by every kernel (fp32, no contraction) -----------
// Both the decode (YOLOv3.py:338-345: sigmoid of the class, objectness and centre logits, exp of the
// size logits) and the loss (sigmoid cross-entropy and its derivative) use the hardware's exp2, log2
// and reciprocal, one ulp each.  exp(x) = exp2(x * log2(e)): a plain product would carry its rounding
// (and that of the constant) into the exponent, a relative error that grows as 8e-8 |x| and passes
// 5e-6 at the largest finite results.  The product is therefore taken in two parts: p = RN(x * hi),
// d = the exact remainder x * hi - p plus x * lo (hi + lo = log2(e) to 2^-50), and
// exp2(p + d) = exp2(p) * (1 + d ln 2) with |d| < 2^-17 while the result is finite: three ulp whatever |x| is.
// Beyond that d means nothing (for |x| of 1e30 it is huge, for an infinite x it is inf - inf), so it is held
// inside +-2^-10: an overflowing exp2(p) stays +inf, an underflowing one 0, exp(+-inf) is +inf / 0 as the
// reference's, and a NaN logit stays NaN through exp2(p).
// Every decode kernel uses these two functions, so the staged and the fused paths agree bit for bit.
__device__ __forceinline__ float decode_exp(float x)
{
    const float hi = 1.44269502162933349609375f;        // RN(log2 e)
    const float lo = 1.92596299112661747e-8f;             // log2 e - hi
    const float p = x * hi;
    const float d = __builtin_amdgcn_fmed3f(__builtin_fmaf(x, lo, __builtin_fmaf(x, hi, -p)), -0x1p-10f, 0x1p-10f);   // a NaN d: one of the bounds
    return __builtin_amdgcn_exp2f(p) * __builtin_fmaf(d, 0.693147180559945309417f, 1.0f);
}
__device__ __forceinline__ float decode_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + decode_exp(-x)); }
__device__ __forceinline__ float sigmoidf(float x) { return decode_sigmoid(x); }
// tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log(1 + exp(-|x|)).  With t = exp(-|x|)
// in (0, 1]: below 1/16 the series of log1p (its sixth term is under 2e-8 of the sum); from 1/16 on
// log(u) * t / (u - 1) with u = 1 + t rounded, which takes the rounding of the sum back out (u - 1
// is exact).  A few ulp in either branch, against the library's 147 instructions some 25.
__device__ __forceinline__ float sigmoid_ce(float x, float z)
{
    const float t = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896340736f);
    const float u = 1.0f + t, d = u - 1.0f;
    const float wide = (__builtin_amdgcn_logf(u) * 0.693147180559945309417f) * (t * __builtin_amdgcn_rcpf(d));
    float poly = __builtin_fmaf(t, 0.2f, -0.25f);
    poly = __builtin_fmaf(t, poly, 0.333333333333333333f);
    poly = __builtin_fmaf(t, poly, -0.5f);
    poly = __builtin_fmaf(t, poly, 1.0f);
    const float l = t < 0.0625f ? t * poly : wide;
    return (fmaxf(x, 0.0f) - x * z) + l;
}

// ---- streaming pass (decode / threshold / compaction) --------------------
// Candidates leave the streaming pass in two places.  The wave-private variant
// gives every work unit (one wavefront's run of rows of one image) kUnitSlots
// private slots of (key, box), written with plain stores and no atomic (unused
// slots get kEmptySlot); whatever exceeds them, and everything the workgroup-tile variant
// finds, goes to the image's list, whose range is reserved with a global atomic.
constexpr int kUnitSlots = 16;
constexpr int kMaxUnitsPerImage = 1024;
constexpr unsigned long long kEmptySlot = ~0ull;   // no candidate: its score field would be a NaN

struct StreamScale {
    const float *src;       // head [B][rows_img][K] or conf [B][rows_img][K]
    uint32_t rows_img;      // rows of one image at this scale
    uint32_t span;          // rows_img * K floats
    uint32_t grid_w;
    uint32_t magic_w;       // floor(2^32 / grid_w): quotient estimate, at most one too small
    uint32_t magic_p;       // floor(2^32 / P)
    uint32_t row_base;      // first box row of this scale inside an image
    uint32_t blk_begin;     // first workgroup of this scale in the launch
    uint32_t unit_base;     // wave-private variant: first work unit of this scale inside an image
    uint32_t chunks_per_img;
    uint32_t tiles_per_chunk;   // tiles one workgroup walks
    float mult;
    float prior_hw[YH_MAX_PRIORS][2];
};

struct StreamParams {
    int32_t S, B, C, P, K, N;
    uint32_t magic_k;       // floor(2^32 / K) + 1: row of a flat index by multiply-high (K >= 2)
    int32_t scan_col_step;  // (floats a workgroup scans per step: 4 * kTileThreads) mod K
    int32_t tile_rows;      // rows per LDS tile: floor((tile floats - 3) / K)
    int32_t wave_ni;        // 0: workgroup tiles of kTileFloats; 4 or 8: wave-private tiles of wave_ni KiB
    int32_t wave_nbuf;      // wave-private variant: tile buffers per wavefront
    int32_t row_gather;     // 1: objectness-first variant (one work unit per wavefront, tiles_per_chunk steps of 64 rows)
    uint32_t units;         // objectness-first variant: workgroups per image (kRowsThreads / 64 units each)
    int32_t dense_min;      // objectness-first variant: a unit of whose sampled rows this many in 64 survive is read whole (> 64: never)
    int32_t dense_probe;    // ... 0: no unit is read whole; 1: a unit samples sixteen of its rows first
    int32_t dense_col_step; // ... (floats a wavefront covers per 16-byte load: 256) mod K
    int32_t dense_row_step; // ... and 256 / K
    int32_t upi;            // wave-private variant: work units per image (all scales), <= kMaxUnitsPerImage
    uint32_t cuts;          // workgroup-tile variant: runs of tiles the pass is cut into (a workgroup's worth each)
    uint32_t grid_blocks;   // workgroup-tile variant: workgroups launched; fewer than `cuts`: each walks cuts
                            // blockIdx.x, blockIdx.x + gridDim.x, ...
    float thr;              // score threshold
    float prefilter;        // logits below this cannot reach thr
    int32_t cand_cap;
    unsigned long long *cand;   // [B][cand_cap]
    float *cand_box;            // [B][cand_cap][4] boxes of the list entries (wave-private variant)
    int32_t *img_count;         // [B], zero on entry; the kernel that consumes it zeroes it again
    int32_t *class_hist;         // [B][C], or null: the workgroup-tile kernel adds the classes of the keys it lists (the multi-launch
                                 // select stage then starts from finished counts)
    unsigned long long *ukeys;  // [B][upi][kUnitSlots]
    float *uboxes;              // [B][upi][kUnitSlots][4]
    float *boxes;               // [B][N][4] (workgroup-tile variant and full decode)
    float *conf;                // [B][N][C] (full decode only, may be null)
    StreamScale sc[YH_MAX_SCALES];
};

// Box of row `row` (YOLOv3.py:340-348): yx = cell + 0.5 + sigmoid(t_yx),
// hw = prior + exp(t_hw), corners scaled by the per-scale multiplier.
template <typename Scale>
__device__ __forceinline__ float4 decode_box(int P, Scale &sc, float ty, float tx, float th, float tw, uint32_t row)
{
    // row = (gy * grid_w + gx) * P + pi, taken apart without integer division: the product with
    // floor(2^32 / d) gives the quotient or one less, one comparison settles it
    uint32_t cell = __umulhi(row, sc.magic_p);
    uint32_t pi = row - cell * (uint32_t)P;
    if (pi >= (uint32_t)P) { pi -= (uint32_t)P; ++cell; }
    uint32_t gy = __umulhi(cell, sc.magic_w);
    uint32_t gx = cell - gy * sc.grid_w;
    if (gx >= sc.grid_w) { gx -= sc.grid_w; ++gy; }
    // The prior of a per-lane index: a uniform walk over the table (scalar reads of the parameter block
    // and a select each) instead of an indexed read, which would be a vector load from memory with a
    // full round trip in the middle of the decode.
    float ph = sc.prior_hw[0][0], pw = sc.prior_hw[0][1];
    for (int k = 1; k < P; ++k) {
        const float kh = sc.prior_hw[k][0], kw = sc.prior_hw[k][1];
        ph = pi == (uint32_t)k ? kh : ph;
        pw = pi == (uint32_t)k ? kw : pw;
    }
    const float by = ((float)gy + 0.5f) + decode_sigmoid(ty);
    const float bx = ((float)gx + 0.5f) + decode_sigmoid(tx);
    const float bh = ph + decode_exp(th);
    const float bw = pw + decode_exp(tw);
    const float m = sc.mult;
    float4 o;
    o.x = (by - bh / 2.0f) * m;
    o.y = (bx - bw / 2.0f) * m;
    o.z = (by + bh / 2.0f) * m;
    o.w = (bx + bw / 2.0f) * m;
    return o;
}

// Raises a kernel's dynamic-LDS limit on the current device when `bytes` needs it (cached).
hipError_t raise_lds_limit(const void *kernel, size_t bytes);
constexpr size_t kLdsPerWorkgroupMax = 160 * 1024;   // LDS of one gfx950 compute unit

hipError_t launch_probe_read(const void *src, size_t bytes, hipStream_t st);

// Workgroup-tile variant of the streaming pass: two wavefronts on a 16 KiB LDS tile.  On the dense workload (128 images of
// 100x100x10 boxes, 80 classes, one candidate per box) four wavefronts on 24 KiB are 1 % faster for the pass on its own
// (924 against 935 us), but the pass shares the chip with the select stage's kernels of the batches before it on other
// streams, and the smaller workgroup leaves them LDS: 1.219 against 1.286 ms per step (12 / 20 KiB tiles 1.26 / 1.25 ms;
// one wavefront on 8 KiB 1.40 ms).
constexpr int kTileThreads = 128;
constexpr int kTileFloats = 4096;

enum StreamMode { kModeDetect = 0, kModeFull = 1, kModeConf = 2 };

// fused: a candidates -> detections stage (of the previous batch) to run inside the same launch, one
// extra single-wavefront workgroup per image; only with the wave-private variant and select_fits_stream()
hipError_t launch_stream(const StreamParams &p, StreamMode mode, uint32_t blocks, hipStream_t st,
                         const struct NmsParams *fused = nullptr);

// Status word of an image (yolohead.h): the overflow bit and, with it, how many candidates the
// image had (stored and dropped), in units of 256, saturating.
__device__ __forceinline__ int32_t image_status(int stored, int raw, int cap)
{
    // `stored` counts what this stage read: slot candidates plus min(raw, cap) list entries; `raw` is the
    // list's counter, which kept counting past the capacity
    if (!(stored > cap || raw > cap)) return 0;
    const long long total = (long long)stored + (raw > cap ? (long long)raw - cap : 0);
    long long units = (total + 255) >> 8;
    units = units > 0x7fffff ? 0x7fffff : units;
    return (int32_t)(YH_STATUS_CANDIDATE_OVERFLOW | (units << YH_STATUS_COUNT_SHIFT));
}

// ---- candidates -> detections --------------------------------------------
struct NmsParams {
    int32_t B, C, N;
    int32_t cand_cap;
    int32_t max_boxes;
    float iou_thr;
    int32_t upi;                       // work units per image with private slots; 0: the list only
    const unsigned long long *ukeys;   // [B][upi][kUnitSlots]
    const float *uboxes;               // [B][upi][kUnitSlots][4]
    const unsigned long long *cand;    // [B][cand_cap] unordered
    const float *cand_box;             // [B][cand_cap][4], or null: a candidate's box is boxes[row]
    unsigned long long *seg_keys;      // [B][cand_cap] binned by class, then sorted
    int32_t *img_count;                // [B]; zeroed again once consumed
    int32_t *seg_off;                  // [B][C]
    int32_t *seg_cnt;                  // [B][C]
    int32_t *class_hist;               // [B][C] sliced binning: class counts; zero between calls
    int32_t *class_cursor;             // [B][C] sliced binning: reserved entries per class; zero between calls
    int32_t *kept_cnt;                 // [B][C]
    int32_t *kept_row;                 // [B][C][max_boxes]
    float *kept_score;                 // [B][C][max_boxes]
    float *kept_box;                   // [B][C][max_boxes][4]: the selected boxes as decoded (what k_finalize copies out)
    int32_t hist_ready;                // class_hist holds the list's class counts already (the streaming kernel counted)
    const float *boxes;                // [B][N][4] indexed by box row; null: a box is decoded from the head tensors (src)
    int32_t src_P, src_K, src_C, src_S;   // with boxes == null: the head's geometry, as in StreamParams
    StreamScale src[YH_MAX_SCALES];
    float *boxes_fill;                 // same array when the multi-launch path has to fill it from the
                                       // slot-indexed boxes, else null
    int32_t *out_count;                // [B]
    float *out_scores;                 // [B][C*max_boxes]
    float *out_boxes;                  // [B][C*max_boxes][4]
    int32_t *out_class;
    int32_t *out_index;
    int32_t *out_class_candidates;     // [B][C] or null
    int32_t *status;
};

// The box of box row n of image b for the stages behind the streaming pass: from the caller's or the workspace's
// array by row, or decoded from the head tensors where the streaming pass left no such array (YOLOv3.py:340-348;
// the same function and operands as every other decode of the library: the same bits).
__device__ __forceinline__ float4 row_box(const NmsParams &p, int b, int n)
{
    if (p.boxes) return reinterpret_cast<const float4 *>(p.boxes)[(long long)b * p.N + n];
    int s = 0;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.src_S && (uint32_t)n >= p.src[i].row_base) s = i;
    // the scale of a per-lane row: uniform reads of the parameter block and selects
    const float *src = p.src[0].src;
    uint32_t rows_img = p.src[0].rows_img, row_base = 0u;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.src_S && s == i) { src = p.src[i].src; rows_img = p.src[i].rows_img; row_base = p.src[i].row_base; }
    const uint32_t row = (uint32_t)n - row_base;
    const float *rp = src + ((long long)b * rows_img + row) * p.src_K + p.src_C;
    const float ty = rp[0], tx = rp[1], th = rp[2], tw = rp[3];
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < YH_MAX_SCALES; ++i)
        if (i < p.src_S && s == i) o = decode_box(p.src_P, p.src[i], ty, tx, th, tw, row);
    return o;
}

// single-launch shape of the candidates -> detections stage (k_post)
constexpr int kPostCap = 2048;          // candidates per image
constexpr int kPostRankMax = 1024;      // up to here: rank sort; above: bitonic
constexpr int kPostMaxClasses = 1024;
constexpr int kPostMaxBoxes = 128;
constexpr int kPostKeptMax = 4096;      // num_classes * nms_max_boxes
constexpr int kPostThreads = 1024;      // workgroup of the stand-alone kernel
constexpr size_t kFuseLdsMax = 32 * 1024;   // largest LDS block of the stage when it rides in the streaming launch

hipError_t launch_nms_pipeline(const NmsParams &p, hipStream_t st);
bool select_single_launch(int cand_cap, int num_classes, int max_boxes);   // k_post takes the whole stage
bool select_bins_by_slices(int cand_cap, int num_classes, int max_boxes);  // the multi-launch stage bins a list-only image slice by slice,
                                                                            // from class counts the workgroup-tile kernel can leave it
bool select_fits_stream(const NmsParams &p);   // the stage can ride in a streaming launch
bool select_fits_rows(const NmsParams &p);     // ... in a launch of the objectness-first variant
constexpr int kRowsThreads = 256;           // objectness-first variant: four wavefronts per workgroup
constexpr size_t kRowsLdsMax = 48 * 1024;   // three such workgroups per compute unit

// ---- loss ------------------------------------------------------------------
struct LossScale {
    const float *head;
    int32_t H, W;
    uint32_t rows_img;
    uint32_t row_base;
    uint32_t blk_begin;     // first negative-branch block of this scale inside an image
    float gt_rate;
    float *grad;            // backward: d/d(head), same shape as head
    uint32_t magic_p, magic_w;   // row / P and cell / W by multiply-high; 0: divide (divisor 1, or rows too many for the magic)
    float prior_hw[YH_MAX_PRIORS][2];
};

// What the forward pass's positive branch leaves for the gradient launch that follows it: the
// assignment and targets of one ground-truth row.
struct PosRecord {
    long long at;        // first float of the assigned row inside its head tensor; -1: no term
    int32_t s, label;    // scale, class
    float fy, fx, lh, lw;
};

struct LossParams {
    int32_t S, B, C, P, K, N, pad;
    int32_t blocks_per_img;       // negative-branch blocks per image (all scales)
    float w_coord, w_noobj, w_obj, w_class;
    float ignore_thr;
    int32_t normalize;
    const float *gt;              // [B][pad][5]
    float *neg_partial;           // [B][blocks_per_img]
    float *pos_partial;           // [B][pos_blocks][3]
    int32_t pos_blocks;
    int32_t *gcount;              // [B] ground-truth rows per image
    float *loss_per_image;        // [B]
    float *loss_parts;            // [B][4] or null
    float *iou_max;               // [B][N] or null
    float *loss_sum;              // [2]: sum, mean over the call's images
    double *reduce_pair;          // [2] at the start of the workspace: fp64 sum of the per-image losses, image count
    uint8_t *noobj_mask;          // [B][N] or null: 1 where the box enters the no-object sum
    const float *grad_w;          // backward: [B] upstream gradient of every per-image loss
    PosRecord *pos_rec;           // [B][pos_blocks][kLossPosGtPerBlock], written when grad_w is set; null in the backward-only launches
    float *pair_tab;              // [B][S][pair_cap][12]: ground-truth corners and areas in grid units, two rows side by side:
                                  //   written and read back by the negative-branch workgroups of the main launch
    int32_t pair_cap;             // pairs per (image, scale) in pair_tab: a multiple of 8 (whole 128-byte lines), >= ceil(pad / 2)
    uint32_t *occ_tab;            // [B][S][occ_cap]: cell-occupancy bitmaps written by yh_loss_prepare
    int32_t *prep_flags;          // [B][S]: 1 when every operand of the table is within the fast path's bounds
    int32_t occ_cap;              // words per bitmap in occ_tab: a multiple of 32 (whole 128-byte lines)
    int32_t prepared;             // the launch starts from what yh_loss_prepare left in the workspace
    int32_t ticket_slot;          // the launch's cell of the library's ticket array; -1: a launch of its own finalizes
    int32_t cu_slots;             // compute units of the device (0: keep the launch order of the negative-branch blocks)
    LossScale sc[YH_MAX_SCALES];
};

hipError_t launch_loss(const LossParams &p, hipStream_t st);
hipError_t launch_loss_prepare(const LossParams &p, hipStream_t st);
size_t loss_lds_bytes(int pad, int occ_words);      // dynamic LDS of the loss kernel: raw GT rows + occupancy bitmap
hipError_t launch_loss_backward(const LossParams &p, hipStream_t st);

constexpr int kLossNegThreads = 256;
constexpr int kLossBoxes = 2;            // box rows per lane of the negative branch
constexpr int kLossNegRows = kLossNegThreads * kLossBoxes;   // box rows per workgroup
constexpr int kLossMaxPad = 2048;       // ground-truth rows (with padding) of an image: what a workgroup's tables in LDS hold
constexpr int kLossPosGtPerBlock = 16;  // ground-truth rows per workgroup of the positive branch

}  // namespace yh
