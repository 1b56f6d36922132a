// Training loss of the detection head: ground-truth normalisation, best-prior
// and scale assignment with the coordinate / class / object sums (positive
// branch), and the IoU ignore mask with the no-object sum (negative branch).
// Replaces YOLOv3.py:117-310 / YOLOv2.py:105-173.
//
// The reference materialises [A][G][P][2] broadcast tiles for the ignore mask;
// here a lane owns one (cell, prior) box, the image's ground-truth corners sit
// in LDS and are read as broadcasts, and the only HBM traffic of the negative
// branch is the objectness logit of every row.  IoU uses + - * / min max only,
// in the reference's operation order, compiled without contraction, so the
// mask decisions are the same bits a CPU evaluation produces.
#include "yh_internal.h"

namespace yh {

constexpr int kPosThreads = 64 * kLossPosGtPerBlock;
constexpr int kPairStride = 12;              // floats per pair of ground-truth rows in LDS (16-byte aligned)
constexpr float kLossFastBound = 0x1p40f;
typedef float f2 __attribute__((ext_vector_type(2)));

// fminf / fmaxf as the bare instructions (IEEE mode: a NaN operand yields the other one, as
// fminf / fmaxf do).  The compiler's own expansion first canonicalises operands it loaded from
// memory, one extra instruction per operand, in the innermost loop of the negative branch.
__device__ __forceinline__ float vmin(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// _get_normlized_gn's slice (YOLOv3.py:437): the first index of the minimum of
// column 0 of gt [pad][5].  Block-wide; result broadcast through *out.
__device__ int block_gt_count(const float *gt, int pad, float *s_val, int *s_idx, int nthreads)
{
    const int tid = threadIdx.x;
    float bv = INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < pad; j += nthreads) {
        const float v = gt[j * 5];
        if (v < bv) { bv = v; bi = j; }          // strict: keeps the first index per thread
    }
    // wave reduce on (value, index), lower index wins ties
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ov = __shfl_xor(bv, d, 64);
        const int oi = __shfl_xor(bi, d, 64);
        if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { s_val[tid >> 6] = bv; s_idx[tid >> 6] = bi; }
    __syncthreads();
    float fv = s_val[0];
    int fi = s_idx[0];
    for (int w = 1; w < nthreads / 64; ++w) {
        const float ov = s_val[w];
        const int oi = s_idx[w];
        if (ov < fv || (ov == fv && oi < fi)) { fv = ov; fi = oi; }
    }
    __syncthreads();
    return (pad > 0 && fi != 0x7fffffff) ? fi : 0;
}

// The image's ground-truth block [pad][5] staged in LDS with coalesced loads (one memory
// round trip for everything the workgroup needs from it), and _get_normlized_gn's row count
// (block_gt_count's result) taken from there.
__device__ int stage_gt(const float *gt, int pad, float *s_gt, float *s_val, int *s_idx, int nthreads)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < pad * 5; i += nthreads) s_gt[i] = gt[i];
    __syncthreads();
    float bv = INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < pad; j += nthreads) {
        const float v = s_gt[j * 5];
        if (v < bv) { bv = v; bi = j; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ov = __shfl_xor(bv, d, 64);
        const int oi = __shfl_xor(bi, d, 64);
        if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { s_val[tid >> 6] = bv; s_idx[tid >> 6] = bi; }
    __syncthreads();
    float fv = s_val[0];
    int fi = s_idx[0];
    for (int w = 1; w < nthreads / 64; ++w) {
        const float ov = s_val[w];
        const int oi = s_idx[w];
        if (ov < fv || (ov == fv && oi < fi)) { fv = ov; fi = oi; }
    }
    __syncthreads();
    return (pad > 0 && fi != 0x7fffffff) ? fi : 0;
}

// ---------------------------------------------------------------- negative
// B * blocks_per_img workgroups; one owns 256 consecutive box rows of one scale
// of one image.  s_raw: 5 * pad floats (GT corners + area at this scale)
// followed by the cell-occupancy bitmap of the scale.
//
// EXACT_MAX = false (the caller does not ask for the IoU maxima): only the
// predicate "some IoU exceeds the threshold" is needed, and it is decided
// without the division wherever that is safe: RN(inter / union) > thr follows from
// sign(union) * inter > thr * |union| by a margin far wider than the roundings
// involved, and its negation likewise; a box with a pair inside the margin, or
// with a zero or non-finite union, is walked again with the division.  Same
// decisions, less arithmetic.
//
// GRAD = true: the workgroup also writes its rows of the gradient tensor (what
// k_loss_grad_fill writes: zeros, and the no-object term in the objectness column).  The
// zeros are 16-byte stores spread over the IoU loop, one every other step, so the memory
// system drains them while the loop keeps the vector units busy (the loop is compute-bound,
// the fill is write-bound); lanes that skip the loop catch up after it; the objectness
// column follows once the mask is known, behind the barrier that orders it after the zeros.
template <bool EXACT_MAX, bool GRAD>
__device__ __forceinline__ void loss_neg_block(const LossParams &p, uint32_t block, unsigned char *s_raw,
                                               float *s_val, int *s_idx)
{
    // [ceil(pad/2)][12]: two ground-truth rows side by side, y1 y1' x1 x1' y2 y2' x2 x2' area area' - -
    float *s_pair = reinterpret_cast<float *>(s_raw);
    float *s_gt = s_pair + (size_t)((p.pad + 1) >> 1) * kPairStride;    // [pad][5]: y1 x1 y2 x2 area
    uint32_t *s_occ = reinterpret_cast<uint32_t *>(s_gt + (size_t)p.pad * 5);

    const int tid = threadIdx.x;
    const int b = block / p.blocks_per_img;
    const uint32_t r = block - (uint32_t)b * p.blocks_per_img;
    int s = 0;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.S && r >= p.sc[i].blk_begin) s = i;
    const LossScale &sc = p.sc[s];
    const uint32_t row = (r - sc.blk_begin) * kLossNegThreads + tid;
    const int C = p.C, K = p.K, P = p.P, H = sc.H, W = sc.W;
    const float *gt = p.gt + (long long)b * p.pad * 5;

    const bool in_range = row < sc.rows_img;

    const int occ_words = (H * W + 31) >> 5;
    for (int i = tid; i < occ_words; i += kLossNegThreads) s_occ[i] = 0u;
    const int G = stage_gt(gt, p.pad, s_gt, s_val, s_idx, kLossNegThreads);   // its barriers cover s_occ too
    const float rate = sc.gt_rate;
    int in_bounds = 1;                                          // every corner and area below 2^40 in magnitude
    for (int g = tid; g < G; g += kLossNegThreads) {           // row g is rewritten in place by its only reader
        const float gy = s_gt[g * 5 + 0] / rate, gx = s_gt[g * 5 + 1] / rate;
        const float gh = s_gt[g * 5 + 2] / rate, gw = s_gt[g * 5 + 3] / rate;
        const float y1 = gy - gh / 2.0f, x1 = gx - gw / 2.0f;
        const float y2 = gy + gh / 2.0f, x2 = gx + gw / 2.0f;
        s_gt[g * 5 + 0] = y1; s_gt[g * 5 + 1] = x1; s_gt[g * 5 + 2] = y2; s_gt[g * 5 + 3] = x2;
        const float area = (y2 - y1) * (x2 - x1);
        s_gt[g * 5 + 4] = area;
        if (!EXACT_MAX) {
            const float v[5] = {y1, x1, y2, x2, area};
            float *slot = s_pair + (g >> 1) * kPairStride + (g & 1);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                slot[2 * k] = v[k];
                if (g == G - 1 && (g & 1) == 0) slot[2 * k + 1] = v[k];    // odd count: the last row twice
                in_bounds &= fabsf(v[k]) < kLossFastBound ? 1 : 0;
            }
        }
        const int cy = (int)floorf(gy), cx = (int)floorf(gx);
        if (cy >= 0 && cy < H && cx >= 0 && cx < W) {
            const int cell = cy * W + cx;
            atomicOr(&s_occ[cell >> 5], 1u << (cell & 31));
        }
    }
    const bool gt_in_bounds = __syncthreads_and(in_bounds) != 0;

    // The objectness logit is asked for here, after the ground-truth round trip (memory loads
    // return in issue order: an earlier request would hold that round trip back), and is not
    // needed before the IoU loop below has run: the loop hides its latency.
    float obj = 0.0f;
    if (in_range) obj = sc.head[((long long)b * sc.rows_img + row) * K + C + 4];

    // the workgroup's run of the gradient tensor: `lead` single floats up to 16-byte alignment,
    // nb 16-byte blocks, then fewer than four single floats
    const uint32_t row0 = (r - sc.blk_begin) * kLossNegThreads;
    float *gdst = nullptr;
    float4 *gdst4 = nullptr;
    uint32_t gtotal = 0, lead = 0, nb = 0, filled = 0;
    float gscale = 0.0f;
    if (GRAD) {
        const uint32_t nrows = sc.rows_img - row0 < (uint32_t)kLossNegThreads ? sc.rows_img - row0 : (uint32_t)kLossNegThreads;
        gdst = sc.grad + ((long long)b * sc.rows_img + row0) * K;
        gtotal = nrows * (uint32_t)K;
        lead = (4u - (uint32_t)((reinterpret_cast<uintptr_t>(gdst) >> 2) & 3u)) & 3u;
        lead = lead < gtotal ? lead : gtotal;
        nb = (gtotal - lead) >> 2;
        gdst4 = reinterpret_cast<float4 *>(gdst + lead);
        gscale = p.w_noobj * p.grad_w[b];
        if (p.normalize) gscale = gscale / (float)G;
    }
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);

    float contrib = 0.0f;
    bool counted_row = false;
    if (in_range) {
        const uint32_t cell = row / (uint32_t)P;
        const uint32_t pi = row - cell * (uint32_t)P;
        const uint32_t cy = cell / (uint32_t)W, cx = cell - cy * (uint32_t)W;
        const bool occupied = (s_occ[cell >> 5] >> (cell & 31)) & 1u;
        float m = __builtin_nanf("");
        bool counted = false;                 // the box enters the no-object sum
        if (!occupied) {
            // YOLOv3.py:246-257: L = a_yx - a_hw/2 and U = a_yx + a_hw/2 are then used
            // as (centre, size): corners L - U/2 and L + U/2.
            const float ay = (float)cy + 0.5f, ax = (float)cx + 0.5f;
            float ah = 0.0f, aw = 0.0f;                   // uniform walk: scalar reads of the parameter block
            for (int q = 0; q < P; ++q)
                if (pi == (uint32_t)q) { ah = sc.prior_hw[q][0]; aw = sc.prior_hw[q][1]; }
            const float ly = ay - ah / 2.0f, lx = ax - aw / 2.0f;
            const float uy = ay + ah / 2.0f, ux = ax + aw / 2.0f;
            const float by1 = ly - uy / 2.0f, bx1 = lx - ux / 2.0f;
            const float by2 = ly + uy / 2.0f, bx2 = lx + ux / 2.0f;
            const float aarea = (by2 - by1) * (bx2 - bx1);
            if (EXACT_MAX) {
                m = -INFINITY;
                bool any_nan = false;
                for (int g = 0; g < G; ++g) {
                    const float gy1 = s_gt[g * 5 + 0], gx1 = s_gt[g * 5 + 1];
                    const float gy2 = s_gt[g * 5 + 2], gx2 = s_gt[g * 5 + 3];
                    const float garea = s_gt[g * 5 + 4];
                    // intersection NOT clamped at zero (YOLOv3.py:286)
                    const float inter = (vmin(gy2, by2) - vmax(gy1, by1)) * (vmin(gx2, bx2) - vmax(gx1, bx1));
                    const float iou = inter / ((aarea + garea) - inter);
                    if (iou != iou) any_nan = true;
                    else if (iou > m) m = iou;
                }
                if (any_nan) m = __builtin_nanf("");
                counted = m <= p.ignore_thr;
            } else {
                // Division-free pass, two ground-truth rows per step in packed arithmetic.  With
                // d = sign(uni) * inter - thr * |uni| (one rounding) and the margin
                // m = 2^-20 * max(1, thr) * |uni| + 2^-100:  d > m implies RN(inter / uni) > thr and
                // d < -m implies the opposite (the roundings involved stay below 2^-22 relative; the
                // constant keeps tiny unions out).  e = d - m and h = max(d - m, -d - m) are folded
                // into a running maximum and minimum: some e > 0 means "masked", every h > 0 means
                // every pair was decided.  That bookkeeping needs NaN-free values, which the bounds
                // on the operands guarantee (no product can overflow); otherwise, and for a pair
                // inside the margin, the exact walk below decides.
                const float thr = p.ignore_thr;
                const bool fast = gt_in_bounds && thr >= 0.0f && thr < 0x1p20f &&
                                  fabsf(ah) < kLossFastBound && fabsf(aw) < kLossFastBound;
                const float mfac = 0x1p-20f * fmaxf(1.0f, thr);
                float e_max = -INFINITY, h_min = fast ? INFINITY : -INFINITY;
                const int pairs = fast ? (G + 1) >> 1 : 0;
                const f2 nthr = {-thr, -thr}, mf = {mfac, mfac}, tiny = {0x1p-100f, 0x1p-100f};
                const f2 aa = {aarea, aarea};
                for (int j = 0; j < pairs; ++j) {
                    if (GRAD && !(j & 1)) {
                        const uint32_t i = filled * (uint32_t)kLossNegThreads + (uint32_t)tid;
                        if (i < nb) gdst4[i] = zero4;
                        ++filled;
                    }
                    const float4 lo = *reinterpret_cast<const float4 *>(s_pair + j * kPairStride);      // y1 y1' x1 x1'
                    const float4 hi = *reinterpret_cast<const float4 *>(s_pair + j * kPairStride + 4);  // y2 y2' x2 x2'
                    const float2 ar = *reinterpret_cast<const float2 *>(s_pair + j * kPairStride + 8);
                    const f2 dy = f2{vmin(hi.x, by2), vmin(hi.y, by2)} - f2{vmax(lo.x, by1), vmax(lo.y, by1)};
                    const f2 dx = f2{vmin(hi.z, bx2), vmin(hi.w, bx2)} - f2{vmax(lo.z, bx1), vmax(lo.w, bx1)};
                    const f2 inter = dy * dx;
                    const f2 uni = (aa + f2{ar.x, ar.y}) - inter;
                    const f2 au = {fabsf(uni.x), fabsf(uni.y)};
                    const f2 si = {__uint_as_float(__float_as_uint(inter.x) ^ (__float_as_uint(uni.x) & 0x80000000u)),
                                   __uint_as_float(__float_as_uint(inter.y) ^ (__float_as_uint(uni.y) & 0x80000000u))};
                    const f2 d = __builtin_elementwise_fma(nthr, au, si);
                    const f2 m = __builtin_elementwise_fma(mf, au, tiny);
                    const f2 e = d - m, f = -d - m;
                    e_max = fmaxf(e_max, fmaxf(e.x, e.y));
                    h_min = fminf(h_min, fminf(fmaxf(e.x, f.x), fmaxf(e.y, f.y)));
                }
                int masked = e_max > 0.0f ? 1 : 0;
                const bool recheck = !(h_min > 0.0f);
                if (recheck) {                    // rare: the exact walk decides
                    masked = 0;
                    for (int g = 0; g < G; ++g) {
                        const float gy1 = s_gt[g * 5 + 0], gx1 = s_gt[g * 5 + 1];
                        const float gy2 = s_gt[g * 5 + 2], gx2 = s_gt[g * 5 + 3];
                        const float garea = s_gt[g * 5 + 4];
                        const float inter = (vmin(gy2, by2) - vmax(gy1, by1)) * (vmin(gx2, bx2) - vmax(gx1, bx1));
                        const float iou = inter / ((aarea + garea) - inter);
                        masked |= (iou != iou) | (iou > thr);
                    }
                }
                counted = !masked;
            }
        }
        if (counted) contrib = sigmoid_ce(obj, 0.0f);
        counted_row = counted;
        if (EXACT_MAX) p.iou_max[(long long)b * p.N + sc.row_base + row] = m;
        if (p.noobj_mask) p.noobj_mask[(long long)b * p.N + sc.row_base + row] = counted ? 1 : 0;
    }

    // block sum, fixed order
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) contrib += __shfl_xor(contrib, d, 64);
    if ((tid & 63) == 0) s_val[tid >> 6] = contrib;
    if (GRAD) {                                                 // the zeros the loop above did not get to
        for (uint32_t i = filled * (uint32_t)kLossNegThreads + (uint32_t)tid; i < nb; i += kLossNegThreads) gdst4[i] = zero4;
        if ((uint32_t)tid < lead) gdst[tid] = 0.0f;
        const uint32_t e = lead + 4u * nb + (uint32_t)tid;
        if (e < gtotal) gdst[e] = 0.0f;
    }
    __syncthreads();
    if (GRAD && counted_row) gdst[(row - row0) * (uint32_t)K + (uint32_t)(C + 4)] = sigmoidf(obj) * gscale;
    if (tid == 0) {
        float acc = s_val[0];
        for (int w = 1; w < kLossNegThreads / 64; ++w) acc += s_val[w];
        p.neg_partial[(long long)b * p.blocks_per_img + r] = acc;
    }
}

// Target assignment of one ground-truth row (yc, xc, h, w, class in pixels): best
// prior per scale by the reference's IoU (YOLOv3.py:159-177: intersection not
// clamped, anchor area = prod(hw); argmax = first maximum), then the scale
// (YOLOv3.py:185-189: scale s < S-1 takes the GT when it beats every other scale
// strictly; everything else goes to the last scale).  All wave-uniform.
struct GtAssign {
    int s, p;            // scale and prior
    uint32_t row;        // box row inside the scale: (cy * W + cx) * P + p
    int label;
    float gy, gx, gh, gw;   // the GT in grid units of scale s
    float prior_h, prior_w; // the assigned prior
    bool inside;         // the centre's cell lies inside the grid
};

__device__ __forceinline__ GtAssign assign_gt(const LossParams &p, const float *gtrow)
{
    const int S = p.S, P = p.P;
    const float y = gtrow[0], x = gtrow[1], h = gtrow[2], w = gtrow[3];
    GtAssign a;
    a.label = (int)gtrow[4];                                  // tf.cast(float -> int32) truncates
    float best_iou[YH_MAX_SCALES];
    int best_p[YH_MAX_SCALES];
#pragma unroll
    for (int s = 0; s < YH_MAX_SCALES; ++s) {
        best_iou[s] = 0.0f; best_p[s] = 0;
        if (s < S) {
            const float rate = p.sc[s].gt_rate;
            const float gy = y / rate, gx = x / rate, gh = h / rate, gw = w / rate;
            const float ay = floorf(gy) + 0.5f, ax = floorf(gx) + 0.5f;
            const float gy1 = gy - gh / 2.0f, gx1 = gx - gw / 2.0f;
            const float gy2 = gy + gh / 2.0f, gx2 = gx + gw / 2.0f;
            const float garea = (gy2 - gy1) * (gx2 - gx1);
            for (int q = 0; q < P; ++q) {
                const float ah = p.sc[s].prior_hw[q][0], aw = p.sc[s].prior_hw[q][1];
                const float ay1 = ay - ah / 2.0f, ax1 = ax - aw / 2.0f;
                const float ay2 = ay + ah / 2.0f, ax2 = ax + aw / 2.0f;
                const float inter = (fminf(gy2, ay2) - fmaxf(gy1, ay1)) * (fminf(gx2, ax2) - fmaxf(gx1, ax1));
                const float iou = inter / ((ah * aw + garea) - inter);
                if (q == 0 || iou > best_iou[s]) { best_iou[s] = iou; best_p[s] = q; }
            }
        }
    }
    int ssel = S - 1;
    bool found = false;
#pragma unroll
    for (int s = 0; s < YH_MAX_SCALES; ++s) {
        if (s < S - 1 && !found) {
            bool wins = true;
#pragma unroll
            for (int t = 0; t < YH_MAX_SCALES; ++t)
                if (t < S && t != s && !(best_iou[s] > best_iou[t])) wins = false;
            if (wins) { ssel = s; found = true; }
        }
    }
    int psel = 0;
#pragma unroll
    for (int s = 0; s < YH_MAX_SCALES; ++s) if (s == ssel) psel = best_p[s];
    // wave-uniform by construction; saying so keeps the parameter-block reads below scalar
    ssel = __builtin_amdgcn_readfirstlane(ssel);
    psel = __builtin_amdgcn_readfirstlane(psel);
    const LossScale &sc = p.sc[ssel];
    const float rate = sc.gt_rate;
    a.s = ssel; a.p = psel;
    a.gy = y / rate; a.gx = x / rate; a.gh = h / rate; a.gw = w / rate;
    const int cy = (int)floorf(a.gy), cx = (int)floorf(a.gx);
    a.inside = cy >= 0 && cy < sc.H && cx >= 0 && cx < sc.W;
    a.row = a.inside ? (uint32_t)((cy * sc.W + cx) * P + psel) : 0u;
    a.prior_h = sc.prior_hw[psel][0]; a.prior_w = sc.prior_hw[psel][1];
    return a;
}

// ---------------------------------------------------------------- positive
// B * pos_blocks workgroups; one wavefront per ground-truth row.  Every lane
// evaluates the (cheap) assignment redundantly, then the lanes split the K
// channels of the assigned head row.
__device__ __forceinline__ void loss_pos_block(const LossParams &p, uint32_t block, float *s_val, int *s_idx,
                                               float (*s_part)[3])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (int)(block / (uint32_t)p.pos_blocks);
    const int blk = (int)block - b * p.pos_blocks;
    const float *gt = p.gt + (long long)b * p.pad * 5;
    const int G = block_gt_count(gt, p.pad, s_val, s_idx, kPosThreads);
    if (blk == 0 && tid == 0) p.gcount[b] = G;

    const int g = blk * kLossPosGtPerBlock + wave;
    float coord = 0.0f, cls = 0.0f, obj = 0.0f;
    if (g < G) {
        const int C = p.C, K = p.K;
        const GtAssign a = assign_gt(p, gt + g * 5);
        if (a.inside) {
            const LossScale &sc = p.sc[a.s];
            const float *rowp = sc.head + (long long)b * sc.rows_img * K + (long long)a.row * K;
            for (int k = lane; k < K; k += 64) {
                const float v = rowp[k];
                if (k < C) {
                    cls += sigmoid_ce(v, k == a.label ? 1.0f : 0.0f);
                } else if (k == C) {
                    coord += sigmoid_ce(v, a.gy - floorf(a.gy));
                } else if (k == C + 1) {
                    coord += sigmoid_ce(v, a.gx - floorf(a.gx));
                } else if (k == C + 2) {
                    const float d = v - logf(a.gh / a.prior_h);
                    coord += 0.5f * (d * d);
                } else if (k == C + 3) {
                    const float d = v - logf(a.gw / a.prior_w);
                    coord += 0.5f * (d * d);
                } else {
                    obj += sigmoid_ce(v, 1.0f);
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        coord += __shfl_xor(coord, d, 64);
        cls += __shfl_xor(cls, d, 64);
        obj += __shfl_xor(obj, d, 64);
    }
    if (lane == 0) { s_part[wave][0] = coord; s_part[wave][1] = cls; s_part[wave][2] = obj; }
    __syncthreads();
    if (tid < 3) {
        float acc = s_part[0][tid];
        for (int w2 = 1; w2 < kPosThreads / 64; ++w2) acc += s_part[w2][tid];
        p.pos_partial[((long long)b * p.pos_blocks + blk) * 3 + tid] = acc;
    }
}

// One launch for both branches: the last B * pos_blocks workgroups take the
// positive branch (short, latency-bound: they fill in behind the negative ones).
static_assert(kPosThreads == kLossNegThreads, "both branches share one launch");
template <bool EXACT_MAX, bool GRAD>
__global__ __launch_bounds__(kLossNegThreads) void k_loss_main(const LossParams p)
{
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ float s_val[kLossNegThreads / 64];
    __shared__ int s_idx[kLossNegThreads / 64];
    __shared__ float s_part[kLossNegThreads / 64][3];
    const uint32_t nneg = (uint32_t)p.B * (uint32_t)p.blocks_per_img;
    if (blockIdx.x < nneg) loss_neg_block<EXACT_MAX, GRAD>(p, blockIdx.x, s_raw, s_val, s_idx);
    else loss_pos_block(p, blockIdx.x - nneg, s_val, s_idx, s_part);
}

// ---------------------------------------------------------------- finalize
// One workgroup: per-image combination (YOLOv3.py:308-310 / YOLOv2.py:172) and
// the batch sum.  Sixteen lanes per image (64 images at a time with 1024 threads): they
// fetch the image's partial sums together and add them up in double in a fixed tree.  The
// batch sum is defined over kFinalizeClasses residue classes of the image index (class g
// adds the losses of images g, g + 64, ... in that order, then the classes are added in
// order), whatever the workgroup's size: a smaller workgroup keeps several classes per lane
// group.  The result depends neither on scheduling nor on the workgroup's size.
constexpr int kFinalizeThreads = 1024;
constexpr int kFinalizeLanes = 16;
constexpr int kFinalizeClasses = 64;

__device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int d = kFinalizeLanes / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

template <int THREADS>
__device__ __forceinline__ void loss_finalize(const LossParams &p, double *s_sum)
{
    constexpr int kFinalizeGroups = THREADS / kFinalizeLanes;
    constexpr int kPerGroup = kFinalizeClasses / kFinalizeGroups;   // residue classes a lane group carries
    static_assert(kFinalizeGroups * kPerGroup == kFinalizeClasses, "the lane groups split the residue classes evenly");
    const int tid = threadIdx.x, gl = tid & (kFinalizeLanes - 1), grp = tid / kFinalizeLanes;
    double local[kPerGroup];                                   // lane 0 of each group: its images' losses
#pragma unroll
    for (int k = 0; k < kPerGroup; ++k) local[k] = 0.0;
    for (int b = grp, turn = 0; b < p.B; b += kFinalizeGroups, ++turn) {
        double coord = 0.0, cls = 0.0, obj = 0.0, noobj = 0.0;
        const float gf = p.normalize ? (float)p.gcount[b] : 1.0f;
        const float *pp = p.pos_partial + (long long)b * p.pos_blocks * 3;
        for (int i = gl; i < p.pos_blocks; i += kFinalizeLanes) { coord += pp[i * 3]; cls += pp[i * 3 + 1]; obj += pp[i * 3 + 2]; }
        const float *np_ = p.neg_partial + (long long)b * p.blocks_per_img;
        for (int i = gl; i < p.blocks_per_img; i += kFinalizeLanes) noobj += np_[i];
        coord = group_sum(coord); cls = group_sum(cls); obj = group_sum(obj); noobj = group_sum(noobj);
        const float fc = (float)coord, fk = (float)cls, fo = (float)obj, fn = (float)noobj;
        float pos = (p.w_coord * fc + p.w_class * fk) + p.w_obj * fo;
        float neg = p.w_noobj * fn;
        if (p.normalize) { pos = pos / gf; neg = neg / gf; }
        const float li = pos + neg;
        if (gl == 0) {
            p.loss_per_image[b] = li;
            if (p.loss_parts) {
                p.loss_parts[b * 4 + 0] = fc; p.loss_parts[b * 4 + 1] = fk;
                p.loss_parts[b * 4 + 2] = fo; p.loss_parts[b * 4 + 3] = fn;
            }
#pragma unroll
            for (int k = 0; k < kPerGroup; ++k)               // image b belongs to class b mod 64 = grp + 16 * (turn mod kPerGroup)
                if (turn % kPerGroup == k) local[k] += (double)li;
        }
    }
    if (gl == 0) {
#pragma unroll
        for (int k = 0; k < kPerGroup; ++k) s_sum[grp + k * kFinalizeGroups] = local[k];
    }
    __syncthreads();
    if (tid == 0) {
        double acc = 0.0;
        for (int g = 0; g < kFinalizeClasses; ++g) acc += s_sum[g];
        if (p.loss_sum) {
            p.loss_sum[0] = (float)acc;
            p.loss_sum[1] = (float)acc / (float)p.B;          // reduce_mean over this call's images, in fp32
        }
        p.reduce_pair[0] = acc;                                // what a data-parallel caller all-reduces in place
        p.reduce_pair[1] = (double)p.B;
    }
}

__global__ __launch_bounds__(kFinalizeThreads) void k_loss_finalize(const LossParams p)
{
    __shared__ double s_sum[kFinalizeClasses];
    loss_finalize<kFinalizeThreads>(p, s_sum);
}

// ---------------------------------------------------------------- backward
// d(sum_b grad_w[b] * loss_per_image[b]) / d(head), closed form:
//   no-object term   w_noobj * sigmoid(obj)                          on the boxes of noobj_mask
//   assigned rows    w_class * (sigmoid(x_k) - [k == label]), w_coord * (sigmoid(t_yx) - frac),
//                    w_coord * (t_hw - log(g_hw / prior_hw)), w_obj * (sigmoid(obj) - 1)
// each divided by the image's GT count when normalize is set.
//
// k_loss_grad_fill writes the whole gradient tensor (zeros and the objectness column of the
// no-object boxes): a workgroup owns 256 consecutive rows, the row owners compute the column,
// then all threads store the rows' K floats as one contiguous, coalesced run.  Bound: HBM write.
__global__ __launch_bounds__(kLossNegThreads) void k_loss_grad_fill(const LossParams p)
{
    __shared__ float s_g[kLossNegThreads];
    __shared__ float s_val[kLossNegThreads / 64];
    __shared__ int s_idx[kLossNegThreads / 64];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / p.blocks_per_img;
    const uint32_t r = blockIdx.x - (uint32_t)b * p.blocks_per_img;
    int s = 0;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.S && r >= p.sc[i].blk_begin) s = i;
    const LossScale &sc = p.sc[s];
    const uint32_t row0 = (r - sc.blk_begin) * kLossNegThreads;
    const uint32_t row = row0 + tid;
    const int C = p.C, K = p.K;
    float g = 0.0f;
    const bool in_range = row < sc.rows_img;
    float obj = 0.0f;
    int counted = 0;
    if (in_range) {
        obj = sc.head[((long long)b * sc.rows_img + row) * K + C + 4];
        counted = p.noobj_mask[(long long)b * p.N + sc.row_base + row];
    }
    const int G = block_gt_count(p.gt + (long long)b * p.pad * 5, p.pad, s_val, s_idx, kLossNegThreads);
    float scale = p.w_noobj * p.grad_w[b];
    if (p.normalize) scale = scale / (float)G;
    if (counted) g = sigmoidf(obj) * scale;
    s_g[tid] = g;
    __syncthreads();
    const uint32_t nrows = sc.rows_img - row0 < (uint32_t)kLossNegThreads ? sc.rows_img - row0 : (uint32_t)kLossNegThreads;
    float *dst = sc.grad + ((long long)b * sc.rows_img + row0) * K;
    const uint32_t total = nrows * (uint32_t)K;
    const uint32_t objk = (uint32_t)(C + 4);
    auto value = [&](uint32_t rr, uint32_t k) { return k == objk ? s_g[rr] : 0.0f; };
    // 16-byte stores over the aligned body of the run, single floats at its ends
    uint32_t head = (4u - (uint32_t)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3u)) & 3u;
    head = head < total ? head : total;
    if ((uint32_t)tid < head) dst[tid] = value(0u, (uint32_t)tid);        // head < 4 <= K: still row 0
    const uint32_t nb = (total - head) >> 2;
    float4 *dst4 = reinterpret_cast<float4 *>(dst + head);
    for (uint32_t i = tid; i < nb; i += kLossNegThreads) {
        const uint32_t e0 = head + 4u * i;
        uint32_t rr = e0 / (uint32_t)K, k = e0 - rr * (uint32_t)K;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = value(rr, k);
            if (++k == (uint32_t)K) { k = 0; ++rr; }
        }
        dst4[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
    const uint32_t e = head + 4u * nb + (uint32_t)tid;
    if (e < total) {
        const uint32_t rr = e / (uint32_t)K;
        dst[e] = value(rr, e - rr * (uint32_t)K);
    }
}

// Adds the assigned rows' terms (atomically: several GT rows may share a box).  Runs after
// k_loss_grad_fill on the same stream.  One wavefront per GT row.
__device__ __forceinline__ void loss_grad_pos(const LossParams &p, uint32_t block, float *s_val, int *s_idx)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (int)(block / (uint32_t)p.pos_blocks);
    const int blk = (int)block - b * p.pos_blocks;
    const float *gt = p.gt + (long long)b * p.pad * 5;
    const int G = block_gt_count(gt, p.pad, s_val, s_idx, kPosThreads);
    const int g = blk * kLossPosGtPerBlock + wave;
    if (g >= G) return;
    const int C = p.C, K = p.K;
    const GtAssign a = assign_gt(p, gt + g * 5);
    if (!a.inside) return;
    const LossScale &sc = p.sc[a.s];
    float w = p.grad_w[b];
    if (p.normalize) w = w / (float)G;
    const long long at = (long long)b * sc.rows_img * K + (long long)a.row * K;
    const float *rowp = sc.head + at;
    float *grow = sc.grad + at;
    for (int k = lane; k < K; k += 64) {
        const float v = rowp[k];
        float d;
        if (k < C) d = p.w_class * (sigmoidf(v) - (k == a.label ? 1.0f : 0.0f));
        else if (k == C) d = p.w_coord * (sigmoidf(v) - (a.gy - floorf(a.gy)));
        else if (k == C + 1) d = p.w_coord * (sigmoidf(v) - (a.gx - floorf(a.gx)));
        else if (k == C + 2) d = p.w_coord * (v - logf(a.gh / a.prior_h));
        else if (k == C + 3) d = p.w_coord * (v - logf(a.gw / a.prior_w));
        else d = p.w_obj * (sigmoidf(v) - 1.0f);
        atomicAdd(&grow[k], d * w);
    }
}

__global__ __launch_bounds__(kPosThreads) void k_loss_grad_pos(const LossParams p)
{
    __shared__ float s_val[kPosThreads / 64];
    __shared__ int s_idx[kPosThreads / 64];
    loss_grad_pos(p, blockIdx.x, s_val, s_idx);
}

// After the fused forward + gradient-fill pass: the assigned rows' terms, and in one more
// workgroup of the same launch the finalize step (the two are independent of each other).
__global__ __launch_bounds__(kPosThreads) void k_loss_grad_pos_finalize(const LossParams p)
{
    __shared__ float s_val[kPosThreads / 64];
    __shared__ int s_idx[kPosThreads / 64];
    __shared__ double s_sum[kFinalizeClasses];
    if (blockIdx.x + 1u == gridDim.x) loss_finalize<kPosThreads>(p, s_sum);
    else loss_grad_pos(p, blockIdx.x, s_val, s_idx);
}

hipError_t launch_loss_backward(const LossParams &p, hipStream_t stm)
{
    if (p.B == 0) return hipSuccess;
    hipLaunchKernelGGL(k_loss_grad_fill, dim3((uint32_t)p.B * (uint32_t)p.blocks_per_img), dim3(kLossNegThreads), 0, stm, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_loss_grad_pos, dim3((uint32_t)p.B * (uint32_t)p.pos_blocks), dim3(kPosThreads), 0, stm, p);
    return hipGetLastError();
}

size_t loss_lds_bytes(int pad, int occ_words)
{
    return (size_t)((pad + 1) / 2) * kPairStride * sizeof(float) + (size_t)pad * 5 * sizeof(float) +
           (size_t)occ_words * sizeof(uint32_t);
}

hipError_t launch_loss(const LossParams &p, hipStream_t stm)
{
    if (p.B == 0) return hipSuccess;
    int occ_words = 0;
    for (int s = 0; s < p.S; ++s) {
        const int w = (p.sc[s].H * p.sc[s].W + 31) >> 5;
        if (w > occ_words) occ_words = w;
    }
    const size_t lds = loss_lds_bytes(p.pad, occ_words);
    const dim3 grid((uint32_t)p.B * (uint32_t)(p.pos_blocks + p.blocks_per_img));
    const bool grad = p.grad_w != nullptr;                     // forward and gradient in one pass over the rows
    void (*kernel)(const LossParams) = p.iou_max ? (grad ? k_loss_main<true, true> : k_loss_main<true, false>)
                                                 : (grad ? k_loss_main<false, true> : k_loss_main<false, false>);
    hipError_t e = raise_lds_limit(reinterpret_cast<const void *>(kernel), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, grid, dim3(kLossNegThreads), lds, stm, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (grad)
        hipLaunchKernelGGL(k_loss_grad_pos_finalize, dim3((uint32_t)p.B * (uint32_t)p.pos_blocks + 1u), dim3(kPosThreads), 0, stm, p);
    else
        hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(kFinalizeThreads), 0, stm, p);
    return hipGetLastError();
}

}  // namespace yh
