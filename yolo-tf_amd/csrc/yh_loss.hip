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

#ifdef YH_ROWS_TRACE
// Measurement build only (tools/trace_loss.py): per-workgroup timestamps of the negative branch.
constexpr int kLossTraceGroups = 1 << 14, kLossTraceWords = 8;
__device__ unsigned long long g_loss_trace[kLossTraceGroups * kLossTraceWords];
#define YH_LT(var) do { var = wall_clock64(); } while (0)
#else
#define YH_LT(var) do { } while (0)
#endif

constexpr int kPosThreads = kLossNegThreads;     // the positive branch shares the negative branch's launch
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
// B * blocks_per_img workgroups; one owns kLossNegRows consecutive box rows of one scale
// of one image, kLossBoxes per lane.  s_raw: 5 * pad floats (GT corners + area at this scale)
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

    unsigned long long lt0 = 0, lt1 = 0, lt2 = 0, lt3 = 0, lt4 = 0, lt5 = 0;
    (void)lt0; (void)lt1; (void)lt2; (void)lt3; (void)lt4; (void)lt5;
    YH_LT(lt0);
    const int tid = threadIdx.x;
    const int b = block / p.blocks_per_img;
    const uint32_t r = block - (uint32_t)b * p.blocks_per_img;
    int s = 0;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.S && r >= p.sc[i].blk_begin) s = i;
    const LossScale &sc = p.sc[s];
    const int C = p.C, K = p.K, P = p.P, H = sc.H, W = sc.W;
    const float *gt = p.gt + (long long)b * p.pad * 5;

    const int occ_words = (H * W + 31) >> 5;
    for (int i = tid; i < occ_words; i += kLossNegThreads) s_occ[i] = 0u;
    const int G = stage_gt(gt, p.pad, s_gt, s_val, s_idx, kLossNegThreads);   // its barriers cover s_occ too
    YH_LT(lt1);
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
    YH_LT(lt2);

    // The objectness logits are asked for here, after the ground-truth round trip (memory loads
    // return in issue order: an earlier request would hold that round trip back), and are not
    // needed before the IoU loop below has run: the loop hides their latency.
    // A lane owns kLossBoxes box rows of the workgroup's run, kLossNegThreads apart: every
    // ground-truth pair read from LDS serves all of them.
    const uint32_t row0 = (r - sc.blk_begin) * kLossNegRows;
    uint32_t row[kLossBoxes];
    bool in_range[kLossBoxes];
    float obj[kLossBoxes];
#pragma unroll
    for (int i = 0; i < kLossBoxes; ++i) {
        row[i] = row0 + (uint32_t)(i * kLossNegThreads + tid);
        in_range[i] = row[i] < sc.rows_img;
        obj[i] = 0.0f;
        if (in_range[i]) obj[i] = sc.head[((long long)b * sc.rows_img + row[i]) * K + C + 4];
    }

    // the workgroup's run of the gradient tensor: `lead` single floats up to 16-byte alignment,
    // nb 16-byte blocks, then fewer than four single floats
    float *gdst = nullptr;
    float4 *gdst4 = nullptr;
    uint32_t gtotal = 0, lead = 0, nb = 0, filled = 0;
    float gscale = 0.0f;
    if (GRAD) {
        const uint32_t nrows = sc.rows_img - row0 < (uint32_t)kLossNegRows ? sc.rows_img - row0 : (uint32_t)kLossNegRows;
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
    auto fill_some = [&]() {                                    // kLossBoxes 16-byte stores of zeros per lane
#pragma unroll
        for (int i = 0; i < kLossBoxes; ++i) {
            const uint32_t at = filled * (uint32_t)kLossNegThreads + (uint32_t)tid;
            if (at < nb) gdst4[at] = zero4;
            ++filled;
        }
    };

    // each box: the reference's no-object box (YOLOv3.py:246-257: L = a_yx - a_hw/2 and
    // U = a_yx + a_hw/2 are then used as (centre, size): corners L - U/2 and L + U/2)
    float by1[kLossBoxes], bx1[kLossBoxes], by2[kLossBoxes], bx2[kLossBoxes], aarea[kLossBoxes];
    bool live[kLossBoxes];                                      // in range and in a cell without a GT centre
    bool dims_ok = true;
    bool any_live = false;
#pragma unroll
    for (int i = 0; i < kLossBoxes; ++i) {
        by1[i] = bx1[i] = by2[i] = bx2[i] = aarea[i] = 0.0f;
        live[i] = false;
        if (in_range[i]) {
            const uint32_t cell = row[i] / (uint32_t)P;
            const uint32_t pi = row[i] - cell * (uint32_t)P;
            const uint32_t cy = cell / (uint32_t)W, cx = cell - cy * (uint32_t)W;
            live[i] = !((s_occ[cell >> 5] >> (cell & 31)) & 1u);
            const float ay = (float)cy + 0.5f, ax = (float)cx + 0.5f;
            float ah = 0.0f, aw = 0.0f;                       // uniform walk: scalar reads of the parameter block
            for (int q = 0; q < P; ++q)
                if (pi == (uint32_t)q) { ah = sc.prior_hw[q][0]; aw = sc.prior_hw[q][1]; }
            const float ly = ay - ah / 2.0f, lx = ax - aw / 2.0f;
            const float uy = ay + ah / 2.0f, ux = ax + aw / 2.0f;
            by1[i] = ly - uy / 2.0f; bx1[i] = lx - ux / 2.0f;
            by2[i] = ly + uy / 2.0f; bx2[i] = lx + ux / 2.0f;
            aarea[i] = (by2[i] - by1[i]) * (bx2[i] - bx1[i]);
            dims_ok = dims_ok && fabsf(ah) < kLossFastBound && fabsf(aw) < kLossFastBound;
            any_live = any_live || live[i];
        }
    }

    float m[kLossBoxes];
    bool counted[kLossBoxes];                                   // the box enters the no-object sum
#pragma unroll
    for (int i = 0; i < kLossBoxes; ++i) { m[i] = __builtin_nanf(""); counted[i] = false; }
    // the exact walk of one box: IoU maximum (NaN if any IoU is NaN), intersection NOT clamped at zero (YOLOv3.py:286)
    auto exact_max = [&](int i) {
        float mx = -INFINITY;
        bool any_nan = false;
        for (int g = 0; g < G; ++g) {
            const float gy1 = s_gt[g * 5 + 0], gx1 = s_gt[g * 5 + 1];
            const float gy2 = s_gt[g * 5 + 2], gx2 = s_gt[g * 5 + 3];
            const float garea = s_gt[g * 5 + 4];
            const float inter = (vmin(gy2, by2[i]) - vmax(gy1, by1[i])) * (vmin(gx2, bx2[i]) - vmax(gx1, bx1[i]));
            const float iou = inter / ((aarea[i] + garea) - inter);
            if (iou != iou) any_nan = true;
            else if (iou > mx) mx = iou;
        }
        return any_nan ? __builtin_nanf("") : mx;
    };
    if (EXACT_MAX) {
#pragma unroll
        for (int i = 0; i < kLossBoxes; ++i) {
            if (live[i]) {
                m[i] = exact_max(i);
                counted[i] = m[i] <= p.ignore_thr;
            }
        }
    } else if (any_live) {
        // Division-free pass, two ground-truth rows per step in packed arithmetic.  With the
        // margin eps = 2^-20 * max(1, thr):
        //   a = inter - (thr + eps) * uni,  e = uni * a:   e > 0  implies  RN(inter / uni) > thr,
        //   b = inter - (thr - eps) * uni,  f = -uni * b:  f > 0  implies  RN(inter / uni) < thr
        // for either sign of the union (a product with uni keeps or flips the inequality exactly as
        // the division by uni does; one rounding in a and b, none that could change the sign of e
        // or f short of an underflow to zero, and a zero decides nothing).  e is folded into a
        // running maximum (some e > 0: masked), max(e, f) into a running minimum (every pair
        // decided).  That bookkeeping needs NaN-free values, which the bounds on the operands
        // guarantee (no product can overflow); otherwise, and for a pair inside the margin or
        // with a zero union, the exact walk decides.
        const float thr = p.ignore_thr;
        const bool fast_all = gt_in_bounds && thr >= 0.0f && thr < 0x1p20f;   // the same for the whole workgroup
        const float eps = 0x1p-20f * fmaxf(1.0f, thr);
        float e_max[kLossBoxes], h_min[kLossBoxes];
#pragma unroll
        for (int i = 0; i < kLossBoxes; ++i) { e_max[i] = -INFINITY; h_min[i] = (fast_all && dims_ok) ? INFINITY : -INFINITY; }
        // a lane whose own box is out of bounds runs along (the trip count stays uniform) and is walked exactly afterwards:
        // its h_min starts below zero and v_min / v_max never let a NaN in
        const int pairs = fast_all ? (G + 1) >> 1 : 0;
        const f2 c_hi = {-(thr + eps), -(thr + eps)}, c_lo = {-(thr - eps), -(thr - eps)};
        auto pair_step = [&](int j) {
            const float4 lo = *reinterpret_cast<const float4 *>(s_pair + j * kPairStride);      // y1 y1' x1 x1'
            const float4 hi = *reinterpret_cast<const float4 *>(s_pair + j * kPairStride + 4);  // y2 y2' x2 x2'
            const float2 ar = *reinterpret_cast<const float2 *>(s_pair + j * kPairStride + 8);
#pragma unroll
            for (int i = 0; i < kLossBoxes; ++i) {
                const f2 dy = f2{vmin(hi.x, by2[i]), vmin(hi.y, by2[i])} - f2{vmax(lo.x, by1[i]), vmax(lo.y, by1[i])};
                const f2 dx = f2{vmin(hi.z, bx2[i]), vmin(hi.w, bx2[i])} - f2{vmax(lo.z, bx1[i]), vmax(lo.w, bx1[i])};
                const f2 inter = dy * dx;
                const f2 uni = (f2{aarea[i], aarea[i]} + f2{ar.x, ar.y}) - inter;
                const f2 a = __builtin_elementwise_fma(c_hi, uni, inter);
                const f2 b2 = __builtin_elementwise_fma(c_lo, uni, inter);
                const f2 e = uni * a, f = -uni * b2;
                e_max[i] = fmaxf(e_max[i], fmaxf(e.x, e.y));
                h_min[i] = fminf(h_min[i], fminf(fmaxf(e.x, f.x), fmaxf(e.y, f.y)));
            }
        };
        int j = 0;
        for (; j + 1 < pairs; j += 2) {                        // two steps per turn: the second step's LDS reads travel
            if (GRAD) fill_some();                             //  while the first step computes
            pair_step(j);
            pair_step(j + 1);
        }
        if (j < pairs) pair_step(j);
#pragma unroll
        for (int i = 0; i < kLossBoxes; ++i) {
            if (live[i]) {
                bool masked = e_max[i] > 0.0f;
                if (!(h_min[i] > 0.0f)) {                      // rare: the exact walk decides
                    const float mx = exact_max(i);
                    masked = (mx != mx) || mx > thr;
                }
                counted[i] = !masked;
            }
        }
    }
    YH_LT(lt3);
    float contrib = 0.0f;
#pragma unroll
    for (int i = 0; i < kLossBoxes; ++i) {
        if (in_range[i]) {
            if (counted[i]) contrib += sigmoid_ce(obj[i], 0.0f);
            if (EXACT_MAX) p.iou_max[(long long)b * p.N + sc.row_base + row[i]] = m[i];
            if (p.noobj_mask) p.noobj_mask[(long long)b * p.N + sc.row_base + row[i]] = counted[i] ? 1 : 0;
        }
    }

    YH_LT(lt4);
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
    if (GRAD) {
#pragma unroll
        for (int i = 0; i < kLossBoxes; ++i)
            if (counted[i]) gdst[(row[i] - row0) * (uint32_t)K + (uint32_t)(C + 4)] = sigmoidf(obj[i]) * gscale;
    }
    if (tid == 0) {
        float acc = s_val[0];
        for (int w = 1; w < kLossNegThreads / 64; ++w) acc += s_val[w];
        p.neg_partial[(long long)b * p.blocks_per_img + r] = acc;
    }
#ifdef YH_ROWS_TRACE
    YH_LT(lt5);
    if ((tid & 63) == 0 && block < (uint32_t)kLossTraceGroups && (tid >> 6) < 1) {
        unsigned long long *tr = g_loss_trace + (size_t)block * kLossTraceWords;
        tr[0] = lt0; tr[1] = lt1; tr[2] = lt2; tr[3] = lt3; tr[4] = lt4; tr[5] = lt5; tr[6] = (unsigned long long)s; tr[7] = 1;
    }
#endif
}

#ifdef YH_ROWS_TRACE
extern "C" int yh_trace_loss_read(void *dst, size_t bytes, int clear)
{
    const size_t have = sizeof(unsigned long long) * kLossTraceGroups * kLossTraceWords;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_loss_trace));
    if (e != hipSuccess) return (int)e;
    if (clear) return (int)hipMemset(ptr, 0, have);
    return (int)hipMemcpy(dst, ptr, bytes < have ? bytes : have, hipMemcpyDeviceToHost);
}
#endif

// Target assignment of one ground-truth row (yc, xc, h, w, class in pixels): best
// prior per scale by the reference's IoU (YOLOv3.py:159-177: intersection not
// clamped, anchor area = prod(hw); argmax = first maximum), then the scale
// (YOLOv3.py:185-189: scale s < S-1 takes the GT when it beats every other scale
// strictly; everything else goes to the last scale).  Every lane may hold a row of its
// own: the parameter block is only read at uniform indices (an indexed read by a per-lane
// scale or prior would be a vector load from memory with a round trip of its own), and
// what belongs to the chosen scale and prior is picked with selects.
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
    float best_iou[YH_MAX_SCALES], sgy[YH_MAX_SCALES], sgx[YH_MAX_SCALES], sgh[YH_MAX_SCALES], sgw[YH_MAX_SCALES];
    float sph[YH_MAX_SCALES], spw[YH_MAX_SCALES];
    int best_p[YH_MAX_SCALES], srow[YH_MAX_SCALES];
    bool sin[YH_MAX_SCALES];
#pragma unroll
    for (int s = 0; s < YH_MAX_SCALES; ++s) {
        best_iou[s] = 0.0f; best_p[s] = 0; sgy[s] = sgx[s] = sgh[s] = sgw[s] = 0.0f; sph[s] = spw[s] = 1.0f;
        srow[s] = 0; sin[s] = false;
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
                if (q == 0 || iou > best_iou[s]) { best_iou[s] = iou; best_p[s] = q; sph[s] = ah; spw[s] = aw; }
            }
            sgy[s] = gy; sgx[s] = gx; sgh[s] = gh; sgw[s] = gw;
            const int cy = (int)floorf(gy), cx = (int)floorf(gx);
            sin[s] = cy >= 0 && cy < p.sc[s].H && cx >= 0 && cx < p.sc[s].W;
            srow[s] = (cy * p.sc[s].W + cx) * P + best_p[s];
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
    a.s = ssel; a.p = 0; a.row = 0u; a.inside = false;
    a.gy = a.gx = a.gh = a.gw = 0.0f; a.prior_h = a.prior_w = 1.0f;
#pragma unroll
    for (int s = 0; s < YH_MAX_SCALES; ++s) {
        if (s == ssel) {
            a.p = best_p[s];
            a.gy = sgy[s]; a.gx = sgx[s]; a.gh = sgh[s]; a.gw = sgw[s];
            a.prior_h = sph[s]; a.prior_w = spw[s];
            a.inside = sin[s];
            a.row = sin[s] ? (uint32_t)srow[s] : 0u;
        }
    }
    return a;
}

// What the positive branch keeps per ground-truth row between its two phases.
struct PosLds {
    const float *rowp[kLossPosGtPerBlock];     // the assigned head row; null: no term (centre outside the grid, or no row)
    float *growp[kLossPosGtPerBlock];          // the same row of the gradient tensor (backward)
    float fy[kLossPosGtPerBlock], fx[kLossPosGtPerBlock];   // cross-entropy labels of t_y, t_x: the centre's offset inside its cell
    float lh[kLossPosGtPerBlock], lw[kLossPosGtPerBlock];   // regression targets of t_h, t_w: log(size / prior)
    int label[kLossPosGtPerBlock];
};

// Phase one: a lane per ground-truth row of this workgroup's share (rows blk * kLossPosGtPerBlock ...),
// assignment and targets into LDS.  Returns the number of rows in the share.  Ends with a barrier.
__device__ __forceinline__ int pos_assign(const LossParams &p, int b, int blk, int G, PosLds &l, bool want_grad)
{
    const int tid = threadIdx.x;
    const int g0 = blk * kLossPosGtPerBlock;
    int n = G - g0;
    n = n < 0 ? 0 : (n > kLossPosGtPerBlock ? kLossPosGtPerBlock : n);
    if (tid < kLossPosGtPerBlock) {
        const float *rowp = nullptr;
        float *growp = nullptr;
        float fy = 0.0f, fx = 0.0f, lh = 0.0f, lw = 0.0f;
        int label = -1;
        if (tid < n) {
            const GtAssign a = assign_gt(p, p.gt + ((long long)b * p.pad + g0 + tid) * 5);
            if (a.inside) {
                // the head of the chosen scale: uniform reads of the parameter block, selects
                const float *base = nullptr;
                float *gbase = nullptr;
                long long rows_img = 0;
#pragma unroll
                for (int s = 0; s < YH_MAX_SCALES; ++s)
                    if (s < p.S && s == a.s) { base = p.sc[s].head; gbase = p.sc[s].grad; rows_img = p.sc[s].rows_img; }
                const long long at = ((long long)b * rows_img + a.row) * p.K;
                rowp = base + at;
                growp = want_grad ? gbase + at : nullptr;
                fy = a.gy - floorf(a.gy); fx = a.gx - floorf(a.gx);
                lh = logf(a.gh / a.prior_h); lw = logf(a.gw / a.prior_w);
                label = a.label;
            }
        }
        l.rowp[tid] = rowp; l.growp[tid] = growp;
        l.fy[tid] = fy; l.fx[tid] = fx; l.lh[tid] = lh; l.lw[tid] = lw; l.label[tid] = label;
    }
    __syncthreads();
    return n;
}

// ---------------------------------------------------------------- positive
// B * pos_blocks workgroups; one takes kLossPosGtPerBlock ground-truth rows of one image.
// Phase one (pos_assign): a lane per row.  Phase two: the rows' K channels as one flat list
// of (row, channel) items spread over the workgroup's lanes, so that the cross-entropy code
// runs on full wavefronts (a wavefront per row ran it twice on 85 of 128 lanes, after every
// lane had repeated the row's assignment).
__device__ __forceinline__ void loss_pos_block(const LossParams &p, uint32_t block, float *s_val, int *s_idx,
                                               float (*s_part)[3], PosLds &l)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (int)(block / (uint32_t)p.pos_blocks);
    const int blk = (int)block - b * p.pos_blocks;
    const float *gt = p.gt + (long long)b * p.pad * 5;
    const int G = block_gt_count(gt, p.pad, s_val, s_idx, kPosThreads);
    if (blk == 0 && tid == 0) p.gcount[b] = G;
    const int n = pos_assign(p, b, blk, G, l, false);
    const int C = p.C, K = p.K;
    const uint32_t magic = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)K) + 1u;   // item -> row by multiply-high (items < 2^17)
    float coord = 0.0f, cls = 0.0f, obj = 0.0f;
    for (int it = tid; it < n * K; it += kPosThreads) {
        const int gl = (int)__umulhi((uint32_t)it, magic), k = it - gl * K;
        const float *rowp = l.rowp[gl];
        if (!rowp) continue;
        const float v = rowp[k];
        if (k < C) {
            cls += sigmoid_ce(v, k == l.label[gl] ? 1.0f : 0.0f);
        } else if (k == C) {
            coord += sigmoid_ce(v, l.fy[gl]);
        } else if (k == C + 1) {
            coord += sigmoid_ce(v, l.fx[gl]);
        } else if (k == C + 2) {
            const float d = v - l.lh[gl];
            coord += 0.5f * (d * d);
        } else if (k == C + 3) {
            const float d = v - l.lw[gl];
            coord += 0.5f * (d * d);
        } else {
            obj += sigmoid_ce(v, 1.0f);
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

// One launch for both branches: the first B * pos_blocks workgroups take the positive
// branch (few, each a chain of dependent memory round trips: started first, they run
// beside the compute-bound negative branch instead of after it).
static_assert(kPosThreads == kLossNegThreads, "both branches share one launch");
template <bool EXACT_MAX, bool GRAD>
__global__ __launch_bounds__(kLossNegThreads) void k_loss_main(const LossParams p)
{
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ float s_val[kLossNegThreads / 64];
    __shared__ int s_idx[kLossNegThreads / 64];
    __shared__ float s_part[kLossNegThreads / 64][3];
    __shared__ PosLds s_pos;
    const uint32_t npos = (uint32_t)p.B * (uint32_t)p.pos_blocks;
    if (blockIdx.x < npos) loss_pos_block(p, blockIdx.x, s_val, s_idx, s_part, s_pos);
    else loss_neg_block<EXACT_MAX, GRAD>(p, blockIdx.x - npos, s_raw, s_val, s_idx);
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
    __shared__ float s_g[kLossNegRows];
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
    const uint32_t row0 = (r - sc.blk_begin) * kLossNegRows;
    const int C = p.C, K = p.K;
    float obj[kLossBoxes];
    int counted[kLossBoxes];
#pragma unroll
    for (int i = 0; i < kLossBoxes; ++i) {
        const uint32_t row = row0 + (uint32_t)(i * kLossNegThreads + tid);
        obj[i] = 0.0f; counted[i] = 0;
        if (row < sc.rows_img) {
            obj[i] = sc.head[((long long)b * sc.rows_img + row) * K + C + 4];
            counted[i] = p.noobj_mask[(long long)b * p.N + sc.row_base + row];
        }
    }
    const int G = block_gt_count(p.gt + (long long)b * p.pad * 5, p.pad, s_val, s_idx, kLossNegThreads);
    float scale = p.w_noobj * p.grad_w[b];
    if (p.normalize) scale = scale / (float)G;
#pragma unroll
    for (int i = 0; i < kLossBoxes; ++i) s_g[i * kLossNegThreads + tid] = counted[i] ? sigmoidf(obj[i]) * scale : 0.0f;
    __syncthreads();
    const uint32_t nrows = sc.rows_img - row0 < (uint32_t)kLossNegRows ? sc.rows_img - row0 : (uint32_t)kLossNegRows;
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
// k_loss_grad_fill on the same stream.  Same two phases as the forward's positive branch.
__device__ __forceinline__ void loss_grad_pos(const LossParams &p, uint32_t block, float *s_val, int *s_idx, PosLds &l)
{
    const int tid = threadIdx.x;
    const int b = (int)(block / (uint32_t)p.pos_blocks);
    const int blk = (int)block - b * p.pos_blocks;
    const float *gt = p.gt + (long long)b * p.pad * 5;
    const int G = block_gt_count(gt, p.pad, s_val, s_idx, kPosThreads);
    const int n = pos_assign(p, b, blk, G, l, true);
    const int C = p.C, K = p.K;
    const uint32_t magic = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)K) + 1u;
    float w = p.grad_w[b];
    if (p.normalize) w = w / (float)G;
    for (int it = tid; it < n * K; it += kPosThreads) {
        const int gl = (int)__umulhi((uint32_t)it, magic), k = it - gl * K;
        const float *rowp = l.rowp[gl];
        if (!rowp) continue;
        const float v = rowp[k];
        float d;
        if (k < C) d = p.w_class * (sigmoidf(v) - (k == l.label[gl] ? 1.0f : 0.0f));
        else if (k == C) d = p.w_coord * (sigmoidf(v) - l.fy[gl]);
        else if (k == C + 1) d = p.w_coord * (sigmoidf(v) - l.fx[gl]);
        else if (k == C + 2) d = p.w_coord * (v - l.lh[gl]);
        else if (k == C + 3) d = p.w_coord * (v - l.lw[gl]);
        else d = p.w_obj * (sigmoidf(v) - 1.0f);
        atomicAdd(&l.growp[gl][k], d * w);
    }
}

__global__ __launch_bounds__(kPosThreads) void k_loss_grad_pos(const LossParams p)
{
    __shared__ float s_val[kPosThreads / 64];
    __shared__ int s_idx[kPosThreads / 64];
    __shared__ PosLds s_pos;
    loss_grad_pos(p, blockIdx.x, s_val, s_idx, s_pos);
}

// After the fused forward + gradient-fill pass: the assigned rows' terms, and in one more
// workgroup of the same launch the finalize step (the two are independent of each other).
__global__ __launch_bounds__(kPosThreads) void k_loss_grad_pos_finalize(const LossParams p)
{
    __shared__ float s_val[kPosThreads / 64];
    __shared__ int s_idx[kPosThreads / 64];
    __shared__ double s_sum[kFinalizeClasses];
    __shared__ PosLds s_pos;
    if (blockIdx.x + 1u == gridDim.x) loss_finalize<kPosThreads>(p, s_sum);
    else loss_grad_pos(p, blockIdx.x, s_val, s_idx, s_pos);
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
