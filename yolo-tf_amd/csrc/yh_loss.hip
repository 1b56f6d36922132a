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
// Measurement build only (tools/trace_loss.py): per-wavefront timestamps of the negative branch.
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

// Partial sums travel between workgroups of one launch (the last one to finish combines them)
// without fences: a store and a load that go to device-coherent memory themselves (the L2 of
// another XCD is not looked up or left with a stale line), and the writer waits for its stores
// to be acknowledged before it takes its ticket.  A release / acquire fence pair would do the
// same by writing back and invalidating the whole L2 in every workgroup, which costs more than
// the loss kernel itself.
template <typename T>
__device__ __forceinline__ void publish(T *ptr, T v) { __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T observe(const T *ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stores_acknowledged() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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

// Tickets of the launches whose last workgroup finalizes: library-owned, so that they are known to be
// zero (a caller's workspace may hold anything); the finalizing workgroup puts its ticket back to zero.
constexpr int kLossTickets = 4096;
__device__ unsigned int g_loss_tickets[kLossTickets];

// ---------------------------------------------------------------- negative
// B * blocks_per_img workgroups; one owns kLossNegRows consecutive box rows of one scale
// of one image, kLossBoxes per lane.  Dynamic LDS: the image's raw ground-truth rows
// [pad][5] and the cell-occupancy bitmap of the scale.
//
// The workgroup turns the ground-truth rows into corners and areas in the scale's grid units
// (YOLOv3.py:246-257 operand order), two rows side by side, padded with copies of the last row
// up to a multiple of four rows (a maximum over the rows does not notice a repeated row), and
// writes that table to global memory: every workgroup of (image, scale) writes the same bits to
// the same table, so no workgroup waits for another, and once its own stores are acknowledged
// the table it reads back is complete.  It reads it back through the scalar data path: the
// table is the same for every lane, so its values are operands in scalar registers and the loop
// issues no LDS or vector-memory reads at all (read from LDS, the same values cost as much
// return bandwidth as 64 different ones, and that bandwidth, not the arithmetic, then set the
// loop's pace).
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
// zeros are 16-byte stores spread over the IoU loop, so the memory system drains them while
// the loop keeps the vector units busy (the loop is compute-bound, the fill is write-bound);
// lanes that skip the loop catch up after it; the objectness column follows once the mask is
// known, behind the barrier that orders it after the zeros.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const f4v const_f4_t;      // read through the scalar data path
typedef __attribute__((address_space(4))) const float const_float_t;
typedef __attribute__((address_space(4))) const int32_t const_int_t;

typedef float f8v __attribute__((ext_vector_type(8)));

// Four ground-truth rows (two pairs) of the prepared table in scalar registers.
struct Turn { f8v lohi0; f2 ar0; f8v lohi1; f2 ar1; };

// Starts the four scalar loads of a turn without waiting for them.  Every register they write is one
// the loop reads later (no padding is loaded): a register with no reader would be free for the
// compiler to reuse while its load is still in flight.
__device__ __forceinline__ void turn_issue(Turn &t, const const_float_t *src)
{
    asm volatile("s_load_dwordx8 %0, %4, 0x0\n\t"
                 "s_load_dwordx2 %1, %4, 0x20\n\t"
                 "s_load_dwordx8 %2, %4, 0x30\n\t"
                 "s_load_dwordx2 %3, %4, 0x50"
                 : "=&s"(t.lohi0), "=&s"(t.ar0), "=&s"(t.lohi1), "=&s"(t.ar1) : "s"(src));
}
// Waits for every scalar load in flight; the values of t may be used after it and not before.
__device__ __forceinline__ void turn_wait(Turn &t)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t.lohi0), "+s"(t.ar0), "+s"(t.lohi1), "+s"(t.ar1));
}

// A value that is the same in every lane, kept in a scalar register.
__device__ __forceinline__ float uniform_value(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// v_min_f32 / v_max_f32 with the first operand in a scalar register
__device__ __forceinline__ float vmin_s(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax_s(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b));
    return r;
}

// The ground-truth tables of one (image, scale), built by a whole workgroup: the raw rows staged in s_gt, the
// corners-and-areas table written to p.pair_tab (YOLOv3.py:246-257 operand order, two rows side by side, padded
// with copies of the last row up to a multiple of four rows: a maximum over the rows does not notice a repeated
// row), the cell-occupancy bitmap in s_occ (YOLOv3.py:123-131).  Returns the row count; *in_bounds: this
// thread's rows have every corner and area below 2^40 in magnitude (the caller combines the threads' flags
// behind a barrier, which also completes s_occ).
__device__ __forceinline__ int gt_tables(const LossParams &p, int b, int s, float *s_gt, uint32_t *s_occ, float *s_val, int *s_idx,
                                         int *in_bounds_out)
{
    const int tid = threadIdx.x;
    const LossScale &sc = p.sc[s];
    const int W = sc.W, H = sc.H;
    const float *gt = p.gt + (long long)b * p.pad * 5;
    const int occ_words = (H * W + 31) >> 5;
    for (int i = tid; i < occ_words; i += kLossNegThreads) s_occ[i] = 0u;
    const int G = stage_gt(gt, p.pad, s_gt, s_val, s_idx, kLossNegThreads);   // its barriers cover s_occ too
    const size_t bs = (size_t)b * p.S + s;
    float *tab_w = p.pair_tab + bs * (size_t)p.pair_cap * kPairStride;
    const float rate = sc.gt_rate;
    const int slots = ((G + 3) >> 2) << 2;
    int in_bounds = 1;
    for (int g = tid; g < slots; g += kLossNegThreads) {
        const float *row = s_gt + (g < G ? g : G - 1) * 5;
        const float gy = row[0] / rate, gx = row[1] / rate;
        const float gh = row[2] / rate, gw = row[3] / rate;
        const float y1 = gy - gh / 2.0f, x1 = gx - gw / 2.0f;
        const float y2 = gy + gh / 2.0f, x2 = gx + gw / 2.0f;
        const float area = (y2 - y1) * (x2 - x1);
        const float v[5] = {y1, x1, y2, x2, area};
        float *slot = tab_w + (g >> 1) * kPairStride + (g & 1);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            slot[2 * k] = v[k];
            in_bounds &= fabsf(v[k]) < kLossFastBound ? 1 : 0;
        }
        const int cy = (int)floorf(gy), cx = (int)floorf(gx);
        if (g < G && cy >= 0 && cy < H && cx >= 0 && cx < W) {
            const int cell = cy * W + cx;
            atomicOr(&s_occ[cell >> 5], 1u << (cell & 31));
        }
    }
    *in_bounds_out = in_bounds;
    return G;
}

template <bool EXACT_MAX, bool GRAD, bool PREPARED>
__device__ __forceinline__ void loss_neg_block(const LossParams &p, uint32_t block, unsigned char *s_raw, float *s_val, int *s_idx)
{
    unsigned long long lt0 = 0, lt1 = 0, lt2 = 0, lt3 = 0, lt4 = 0, lt5 = 0;
    (void)lt0; (void)lt1; (void)lt2; (void)lt3; (void)lt4; (void)lt5;
    unsigned slow_turns = 0;                                    // measurement build: turns that took the exact path
    (void)slow_turns;
    YH_LT(lt0);
    const int tid = threadIdx.x;
    const int b = block / p.blocks_per_img;
    const uint32_t r = block - (uint32_t)b * p.blocks_per_img;
    int s = 0;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.S && r >= p.sc[i].blk_begin) s = i;
    const LossScale &sc = p.sc[s];
    const int C = p.C, K = p.K, P = p.P, W = sc.W;
    float *s_gt = reinterpret_cast<float *>(s_raw);                         // [pad][5]: the raw rows
    uint32_t *s_occ = reinterpret_cast<uint32_t *>(s_gt + (size_t)p.pad * 5);

    int G;
    bool gt_in_bounds;
    uint32_t occ_word[kLossBoxes], occ_bit[kLossBoxes];         // PREPARED: the occupancy words of the lane's cells, in flight
#pragma unroll
    for (int i = 0; i < kLossBoxes; ++i) { occ_word[i] = 0u; occ_bit[i] = 0u; }
    const size_t bs = (size_t)b * p.S + s;
    const const_float_t *tab = (const const_float_t *)(p.pair_tab + bs * (size_t)p.pair_cap * kPairStride);
    if (PREPARED) {
        // yh_loss_prepare left the row count, the table, the bitmap and the bounds flag of (image, scale) in the
        // workspace: two scalar loads instead of the staging above, and the lane's occupancy words are asked
        // for further down and looked at after the loop
        G = ((const const_int_t *)p.gcount)[b];
        gt_in_bounds = ((const const_int_t *)p.prep_flags)[bs] != 0;
    } else {
        int in_bounds;
        G = gt_tables(p, b, s, s_gt, s_occ, s_val, s_idx, &in_bounds);
        stores_acknowledged();                                  // the table is in the L2 the scalar loads below read from
        gt_in_bounds = __syncthreads_and(in_bounds) != 0;
    }
    YH_LT(lt1);

    // A lane owns kLossBoxes box rows of the workgroup's run, kLossNegThreads apart.  The
    // objectness logits are not needed before the IoU loop below has run, and the loop hides
    // their latency.  They are asked for halfway through the loop rather than here: the
    // workgroups of a compute unit start within a microsecond or two of each other, and a table
    // store of a late one queued behind the objectness loads of an early one in the unit's
    // memory pipeline (some 40 MB for the whole launch) waited up to 10 us for its acknowledgement.
    const uint32_t row0 = (r - sc.blk_begin) * kLossNegRows;
    uint32_t row[kLossBoxes];
    bool in_range[kLossBoxes];
    float obj[kLossBoxes];
#pragma unroll
    for (int i = 0; i < kLossBoxes; ++i) {
        row[i] = row0 + (uint32_t)(i * kLossNegThreads + tid);
        in_range[i] = row[i] < sc.rows_img;
        obj[i] = 0.0f;
    }
    bool obj_asked = false;
    auto ask_obj = [&]() {
#pragma unroll
        for (int i = 0; i < kLossBoxes; ++i)
            if (in_range[i]) obj[i] = sc.head[((long long)b * sc.rows_img + row[i]) * K + C + 4];
        obj_asked = true;
    };

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
            const uint32_t cell = sc.magic_p ? __umulhi(row[i], sc.magic_p) : row[i] / (uint32_t)P;
            const uint32_t pi = row[i] - cell * (uint32_t)P;
            const uint32_t cy = sc.magic_w ? __umulhi(cell, sc.magic_w) : cell / (uint32_t)W;
            const uint32_t cx = cell - cy * (uint32_t)W;
            if (PREPARED) {
                occ_word[i] = p.occ_tab[bs * (size_t)p.occ_cap + (cell >> 5)];
                occ_bit[i] = cell & 31u;
                live[i] = EXACT_MAX ? !((occ_word[i] >> occ_bit[i]) & 1u) : true;   // otherwise settled behind the loop (resolve_live)
            } else {
                live[i] = !((s_occ[cell >> 5] >> (cell & 31)) & 1u);
            }
            const float ay = (float)cy + 0.5f, ax = (float)cx + 0.5f;
            // uniform walk over the priors: scalar reads of the parameter block and selects.  (Written as
            // plain selects of loads, the compiler turns the walk into one load at a per-lane address: a
            // vector load, whose wait also waits for the objectness loads issued above.)
            float ah = 0.0f, aw = 0.0f;
            for (int q = 0; q < P; ++q) {
                const float ph = uniform_value(sc.prior_hw[q][0]), pw = uniform_value(sc.prior_hw[q][1]);
                if (pi == (uint32_t)q) { ah = ph; aw = pw; }
            }
            const float ly = ay - ah / 2.0f, lx = ax - aw / 2.0f;
            const float uy = ay + ah / 2.0f, ux = ax + aw / 2.0f;
            by1[i] = ly - uy / 2.0f; bx1[i] = lx - ux / 2.0f;
            by2[i] = ly + uy / 2.0f; bx2[i] = lx + ux / 2.0f;
            aarea[i] = (by2[i] - by1[i]) * (bx2[i] - bx1[i]);
            dims_ok = dims_ok && fabsf(ah) < kLossFastBound && fabsf(aw) < kLossFastBound;
            any_live = any_live || live[i];
        }
    }
    YH_LT(lt2);

    float m[kLossBoxes];
    bool counted[kLossBoxes];                                   // the box enters the no-object sum
#pragma unroll
    for (int i = 0; i < kLossBoxes; ++i) { m[i] = __builtin_nanf(""); counted[i] = false; }
    // the exact walk of one box: IoU maximum (NaN if any IoU is NaN), intersection NOT clamped at zero (YOLOv3.py:286)
    // (the table's copies of the last row change neither the maximum nor the NaN flag)
    auto exact_max = [&](int i) {
        float mx = -INFINITY;
        bool any_nan = false;
        auto one = [&](float gy1, float gx1, float gy2, float gx2, float garea) {
            const float inter = (vmin_s(gy2, by2[i]) - vmax_s(gy1, by1[i])) * (vmin_s(gx2, bx2[i]) - vmax_s(gx1, bx1[i]));
            const float iou = inter / ((aarea[i] + garea) - inter);
            if (iou != iou) any_nan = true;
            else if (iou > mx) mx = iou;
        };
        auto four = [&](const Turn &t) {
            one(t.lohi0.s0, t.lohi0.s2, t.lohi0.s4, t.lohi0.s6, t.ar0.x);
            one(t.lohi0.s1, t.lohi0.s3, t.lohi0.s5, t.lohi0.s7, t.ar0.y);
            one(t.lohi1.s0, t.lohi1.s2, t.lohi1.s4, t.lohi1.s6, t.ar1.x);
            one(t.lohi1.s1, t.lohi1.s3, t.lohi1.s5, t.lohi1.s7, t.ar1.y);
        };
        const int xturns = (G + 3) >> 2, xlast = xturns - 1;
        Turn xa, xb;
        if (xturns > 0) { turn_issue(xa, tab); turn_wait(xa); }
        for (int t = 0; t < xturns; t += 2) {
            turn_issue(xb, tab + (t + 1 < xlast ? t + 1 : xlast) * (2 * kPairStride));
            four(xa);
            turn_wait(xb);
            turn_issue(xa, tab + (t + 2 < xlast ? t + 2 : xlast) * (2 * kPairStride));
            if (t + 1 < xturns) four(xb);
            turn_wait(xa);
        }
        return any_nan ? __builtin_nanf("") : mx;
    };
    // PREPARED: the occupancy words were asked for above; whether a box lies in a cell with a ground-truth centre is
    // only needed where the box is counted, behind the loop, so nobody waits for them before it
    auto resolve_live = [&]() {
        if (PREPARED && !EXACT_MAX) {
#pragma unroll
            for (int i = 0; i < kLossBoxes; ++i) live[i] = in_range[i] && !((occ_word[i] >> occ_bit[i]) & 1u);
        }
    };
    if (EXACT_MAX) {
#pragma unroll
        for (int i = 0; i < kLossBoxes; ++i) {
            if (live[i]) {
                m[i] = exact_max(i);
                counted[i] = m[i] <= p.ignore_thr;
            }
        }
    } else if (__any(any_live)) {
        // Division-free pass, two ground-truth rows per step in packed arithmetic, two steps (four
        // rows) per turn.  With the margin eps = 2^-20 * max(1, thr):
        //   a = inter - (thr + eps) * uni,  e = uni * a:   e > 0  implies  RN(inter / uni) > thr,
        //   b = inter - (thr - eps) * uni,  f = -uni * b:  f > 0  implies  RN(inter / uni) < thr
        // for either sign of the union (a product with uni keeps or flips the inequality exactly as
        // the division by uni does; one rounding in a and b, none that could change the sign of e
        // or f short of an underflow to zero, and a zero decides nothing).  e is folded into a
        // running maximum (some e > 0: masked).  A turn in which some lane has a row with neither
        // e > 0 nor f > 0 (inside the margin, or a zero union) settles its four rows with the
        // division for those lanes there and then: rare, and no second walk afterwards.  The
        // bookkeeping needs NaN-free values, which the bounds on the operands guarantee (no product
        // can overflow); a wavefront with an operand out of bounds takes the exact walk.
        const float thr = p.ignore_thr;
        const bool fast = gt_in_bounds && thr >= 0.0f && thr < 0x1p20f && __all(dims_ok);   // the same for the whole wavefront
        if (fast) {
            const float eps = 0x1p-20f * fmaxf(1.0f, thr);
            float e_max[kLossBoxes], h_turn[kLossBoxes];
            bool masked_exact[kLossBoxes];
#pragma unroll
            for (int i = 0; i < kLossBoxes; ++i) { e_max[i] = -INFINITY; h_turn[i] = INFINITY; masked_exact[i] = false; }
            const int turns = (G + 3) >> 2;
            const f2 c_hi = {-(thr + eps), -(thr + eps)}, c_lo = {-(thr - eps), -(thr - eps)};
            auto pair_step = [&](const f8v lh, const f2 ar, const bool first) {     // y1 y1' x1 x1' y2 y2' x2 x2' / area area'
#pragma unroll
                for (int i = 0; i < kLossBoxes; ++i) {
                    const f2 dy = f2{vmin_s(lh.s4, by2[i]), vmin_s(lh.s5, by2[i])} - f2{vmax_s(lh.s0, by1[i]), vmax_s(lh.s1, by1[i])};
                    const f2 dx = f2{vmin_s(lh.s6, bx2[i]), vmin_s(lh.s7, bx2[i])} - f2{vmax_s(lh.s2, bx1[i]), vmax_s(lh.s3, bx1[i])};
                    const f2 inter = dy * dx;
                    const f2 uni = (f2{aarea[i], aarea[i]} + ar) - inter;
                    const f2 a = __builtin_elementwise_fma(c_hi, uni, inter);
                    const f2 b2 = __builtin_elementwise_fma(c_lo, uni, inter);
                    const f2 e = uni * a, f = -uni * b2;
                    e_max[i] = fmaxf(e_max[i], fmaxf(e.x, e.y));
                    const float hx = fmaxf(e.x, f.x), hy = fmaxf(e.y, f.y);
                    h_turn[i] = first ? fminf(hx, hy) : fminf(h_turn[i], fminf(hx, hy));
                }
            };
            auto one_exact = [&](int i, float gy1, float gx1, float gy2, float gx2, float garea) {
                const float inter = (vmin_s(gy2, by2[i]) - vmax_s(gy1, by1[i])) * (vmin_s(gx2, bx2[i]) - vmax_s(gx1, bx1[i]));
                const float iou = inter / ((aarea[i] + garea) - inter);
                return (iou != iou) || iou > thr;
            };
            auto turn_steps = [&](const Turn &t) {
                pair_step(t.lohi0, t.ar0, true);
                pair_step(t.lohi1, t.ar1, false);
                bool undecided = false;
#pragma unroll
                for (int i = 0; i < kLossBoxes; ++i) undecided = undecided || !(h_turn[i] > 0.0f);
                if (__any(undecided)) {
#ifdef YH_ROWS_TRACE
                    ++slow_turns;
#endif
#pragma unroll
                    for (int i = 0; i < kLossBoxes; ++i) {
                        if (!(h_turn[i] > 0.0f)) {
                            const bool m0 = one_exact(i, t.lohi0.s0, t.lohi0.s2, t.lohi0.s4, t.lohi0.s6, t.ar0.x);
                            const bool m1 = one_exact(i, t.lohi0.s1, t.lohi0.s3, t.lohi0.s5, t.lohi0.s7, t.ar0.y);
                            const bool m2 = one_exact(i, t.lohi1.s0, t.lohi1.s2, t.lohi1.s4, t.lohi1.s6, t.ar1.x);
                            const bool m3 = one_exact(i, t.lohi1.s1, t.lohi1.s3, t.lohi1.s5, t.lohi1.s7, t.ar1.y);
                            masked_exact[i] = masked_exact[i] || m0 || m1 || m2 || m3;
                        }
                    }
                }
            };
            // Two register sets take turns: the scalar loads of the next turn's rows are in flight while this
            // turn computes.  Issue and wait are written out (the compiler, left to itself, issues a turn's
            // loads at the top of that turn and waits for them at once); between the two the set being
            // loaded is not touched.
            Turn ta, tb;
            const int last = turns - 1;
            const int obj_turn = (turns >> 1) & ~1;           // halfway through the loop (measured at 25 turns: 2, 4: 29.5 us per call, 8: 28.7, 12: 27.2, 16: 28.0, after the loop: 28.5)
            if (turns > 0) { turn_issue(ta, tab); turn_wait(ta); }
            for (int t = 0; t < turns; t += 2) {
                turn_issue(tb, tab + (t + 1 < last ? t + 1 : last) * (2 * kPairStride));
                if (t == obj_turn) ask_obj();
                if (GRAD) fill_some();
                turn_steps(ta);
                turn_wait(tb);
                turn_issue(ta, tab + (t + 2 < last ? t + 2 : last) * (2 * kPairStride));
                if (t + 1 < turns) {
                    if (GRAD) fill_some();
                    turn_steps(tb);
                }
                turn_wait(ta);
            }
            resolve_live();
#pragma unroll
            for (int i = 0; i < kLossBoxes; ++i) counted[i] = live[i] && !(e_max[i] > 0.0f || masked_exact[i]);
        } else {
            resolve_live();
#pragma unroll
            for (int i = 0; i < kLossBoxes; ++i) {
                if (live[i]) {
                    const float mx = exact_max(i);
                    counted[i] = !((mx != mx) || mx > thr);
                }
            }
        }
    }
    if (!obj_asked) ask_obj();
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
        publish(&p.neg_partial[(long long)b * p.blocks_per_img + r], acc);
    }
#ifdef YH_ROWS_TRACE
    YH_LT(lt5);
    if ((tid & 63) == 0 && block * (kLossNegThreads / 64) + (tid >> 6) < (uint32_t)kLossTraceGroups) {
        unsigned long long *tr = g_loss_trace + ((size_t)block * (kLossNegThreads / 64) + (tid >> 6)) * kLossTraceWords;
        tr[0] = lt0; tr[1] = lt1; tr[2] = lt2; tr[3] = lt3; tr[4] = lt4; tr[5] = lt5; tr[6] = (unsigned long long)s | ((unsigned long long)slow_turns << 8);
        // where it ran: HW_ID (wave, SIMD, CU, shader array, shader engine) and the XCD
        tr[7] = 1ull | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8) |
                ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) << 40);
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
        long long rec_at = -1;
        int rec_s = 0;
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
                rec_at = at; rec_s = a.s;
                fy = a.gy - floorf(a.gy); fx = a.gx - floorf(a.gx);
                lh = logf(a.gh / a.prior_h); lw = logf(a.gw / a.prior_w);
                label = a.label;
            }
        }
        l.rowp[tid] = rowp; l.growp[tid] = growp;
        l.fy[tid] = fy; l.fx[tid] = fx; l.lh[tid] = lh; l.lw[tid] = lw; l.label[tid] = label;
        if (p.pos_rec) {                                        // the gradient launch that follows starts from here
            PosRecord rec;
            rec.at = rec_at; rec.s = rec_s; rec.label = label;
            rec.fy = fy; rec.fx = fx; rec.lh = lh; rec.lw = lw;
            p.pos_rec[((long long)b * p.pos_blocks + blk) * kLossPosGtPerBlock + tid] = rec;
        }
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
template <bool PREPARED>
__device__ __forceinline__ void loss_pos_block(const LossParams &p, uint32_t block, float *s_val, int *s_idx,
                                               float (*s_part)[3], PosLds &l)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (int)(block / (uint32_t)p.pos_blocks);
    const int blk = (int)block - b * p.pos_blocks;
    const float *gt = p.gt + (long long)b * p.pad * 5;
    int G;
    if (PREPARED) {
        G = ((const const_int_t *)p.gcount)[b];                // yh_loss_prepare counted the rows
    } else {
        G = block_gt_count(gt, p.pad, s_val, s_idx, kPosThreads);
        if (blk == 0 && tid == 0) publish(&p.gcount[b], G);
    }
    const int n = pos_assign(p, b, blk, G, l, false);
    const int C = p.C, K = p.K;
    const uint32_t magic = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)K) + 1u;   // item -> row by multiply-high (items < 2^17)
    float coord = 0.0f, cls = 0.0f, obj = 0.0f;
    for (int it = tid; it < n * K; it += kPosThreads) {
        const int gl = (int)__umulhi((uint32_t)it, magic), k = it - gl * K;
        const float *rowp = l.rowp[gl];
        if (!rowp) continue;
        const float v = rowp[k];
        // one cross-entropy and one squared error per item, the term picked by selects: the branches of
        // a per-channel if-chain would each run the cross-entropy code for the whole wavefront
        const float z = k < C ? (k == l.label[gl] ? 1.0f : 0.0f) : (k == C ? l.fy[gl] : (k == C + 1 ? l.fx[gl] : 1.0f));
        const float ce = sigmoid_ce(v, z);
        const float dv = v - (k == C + 2 ? l.lh[gl] : l.lw[gl]);
        const float term = (k == C + 2 || k == C + 3) ? 0.5f * (dv * dv) : ce;
        if (k < C) cls += term;
        else if (k < C + 4) coord += term;
        else obj += term;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        coord += __shfl_xor(coord, d, 64);
        cls += __shfl_xor(cls, d, 64);
        obj += __shfl_xor(obj, d, 64);
    }
    if (lane == 0) { s_part[wave][0] = coord; s_part[wave][1] = cls; s_part[wave][2] = obj; }
    __syncthreads();
    if (tid == 0) {                                            // the thread that takes the workgroup's ticket writes all of it
        for (int k = 0; k < 3; ++k) {
            float acc = s_part[0][k];
            for (int w2 = 1; w2 < kPosThreads / 64; ++w2) acc += s_part[w2][k];
            publish(&p.pos_partial[((long long)b * p.pos_blocks + blk) * 3 + k], acc);
        }
    }
}

// ---------------------------------------------------------------- finalize
// One workgroup: per-image combination (YOLOv3.py:308-310 / YOLOv2.py:172) and
// the batch sum.  Sixteen lanes per image (16 images at a time with 256 threads): they
// fetch the image's partial sums together and add them up in double in a fixed tree.  The
// batch sum is defined over kFinalizeClasses residue classes of the image index (class g
// adds the losses of images g, g + 64, ... in that order, then the classes are added in
// order), whatever the workgroup's size: a smaller workgroup keeps several classes per lane
// group.  The result depends neither on scheduling nor on the workgroup's size.
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
        const float gf = p.normalize ? (float)observe(&p.gcount[b]) : 1.0f;
        const float *pp = p.pos_partial + (long long)b * p.pos_blocks * 3;
        for (int i = gl; i < p.pos_blocks; i += kFinalizeLanes) { coord += observe(&pp[i * 3]); cls += observe(&pp[i * 3 + 1]); obj += observe(&pp[i * 3 + 2]); }
        const float *np_ = p.neg_partial + (long long)b * p.blocks_per_img;
        for (int i = gl; i < p.blocks_per_img; i += kFinalizeLanes) noobj += observe(&np_[i]);
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

// While a launch fits the device at once, workgroup q lands on compute unit q mod n (round-robin
// over the XCDs and their CUs): the workgroups sharing a CU are q, q + n, q + 2n, ...  This maps
// the launch ids q0 .. q1-1 of the negative branch onto blocks so that those get consecutive
// blocks (same image, same scale): one ground-truth table per CU in its scalar data cache instead
// of five or six tables taking turns in it.  A bijection for any n; n = 0 keeps the launch order.
__device__ __forceinline__ uint32_t cu_major_block(uint32_t q, uint32_t q0, uint32_t q1, uint32_t n)
{
    if (n == 0u) return q - q0;
    auto below = [&](uint32_t x, uint32_t c) { const uint32_t m = x % n; return (x / n) * c + (m < c ? m : c); };   // ids in [0, x) with a residue below c
    auto same = [&](uint32_t x, uint32_t c) { return x / n + (x % n > c ? 1u : 0u); };                              // ids in [0, x) with residue c
    const uint32_t c = q % n;
    return (below(q1, c) - below(q0, c)) + (same(q, c) - same(q0, c));
}

// One launch for both branches: the first B * pos_blocks workgroups take the positive
// branch (few, each a chain of dependent memory round trips: started first, they run
// beside the compute-bound negative branch instead of after it).
static_assert(kPosThreads == kLossNegThreads, "both branches share one launch");
template <bool EXACT_MAX, bool GRAD, bool PREPARED>
__global__ __launch_bounds__(kLossNegThreads) void k_loss_main(const LossParams p)
{
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ float s_val[kLossNegThreads / 64];
    __shared__ int s_idx[kLossNegThreads / 64];
    __shared__ float s_part[kLossNegThreads / 64][3];
    __shared__ PosLds s_pos;
    __shared__ double s_sum[kFinalizeClasses];
    __shared__ int s_last;
    const uint32_t npos = (uint32_t)p.B * (uint32_t)p.pos_blocks;
#ifdef YH_ROWS_TRACE
    const unsigned long long pt0 = wall_clock64();
#endif
    if (blockIdx.x < npos) loss_pos_block<PREPARED>(p, blockIdx.x, s_val, s_idx, s_part, s_pos);
    else loss_neg_block<EXACT_MAX, GRAD, PREPARED>(p, cu_major_block(blockIdx.x, npos, gridDim.x, (uint32_t)p.cu_slots), s_raw, s_val, s_idx);
#ifdef YH_ROWS_TRACE
    if (blockIdx.x < npos && threadIdx.x == 0) {                // positive branch: start, end, place, marker 2
        const size_t at = (size_t)(gridDim.x - npos) * (kLossNegThreads / 64) + blockIdx.x;
        if (at < (size_t)kLossTraceGroups) {
            unsigned long long *tr = g_loss_trace + at * kLossTraceWords;
            tr[0] = pt0; tr[5] = wall_clock64(); tr[6] = 255;
            tr[7] = 1ull | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8) |
                    ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) << 40);
        }
    }
#endif
    if (!GRAD && p.ticket_slot >= 0) {
        // The workgroup that finishes last combines the partial sums (with GRAD the launch that
        // follows does): every workgroup publishes its sums and takes a ticket; the holder of the
        // last ticket reads them all.  No workgroup waits for another, and the combination's order
        // does not depend on who is last.
        if (threadIdx.x == 0) {                                // thread 0 wrote the workgroup's sums
            stores_acknowledged();
            s_last = atomicAdd(&g_loss_tickets[p.ticket_slot], 1u) + 1u == gridDim.x ? 1 : 0;
        }
        __syncthreads();
        if (s_last) {
            loss_finalize<kLossNegThreads>(p, s_sum);
            if (threadIdx.x == 0) publish(&g_loss_tickets[p.ticket_slot], 0u);
        }
    }
}

// yh_loss_prepare: everything the loss needs from the ground truth alone, once per (image, scale) instead of once
// per workgroup of the main launch: the row count, the corners-and-areas table, the cell-occupancy bitmap and the
// "every operand within bounds" flag.  A training loop issues it as soon as a batch's ground truth is on the
// device, ahead of the convolutions; the main launch that follows then starts its IoU loop after two scalar loads.
__global__ __launch_bounds__(kLossNegThreads) void k_loss_prepare(const LossParams p)
{
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ float s_val[kLossNegThreads / 64];
    __shared__ int s_idx[kLossNegThreads / 64];
    const int tid = threadIdx.x;
    const int b = (int)(blockIdx.x / (uint32_t)p.S), s = (int)(blockIdx.x - (uint32_t)b * (uint32_t)p.S);
    float *s_gt = reinterpret_cast<float *>(s_raw);
    uint32_t *s_occ = reinterpret_cast<uint32_t *>(s_gt + (size_t)p.pad * 5);
    int in_bounds;
    const int G = gt_tables(p, b, s, s_gt, s_occ, s_val, s_idx, &in_bounds);
    const bool ok = __syncthreads_and(in_bounds) != 0;         // the barrier completes s_occ
    const size_t bs = (size_t)b * p.S + s;
    const int occ_words = (p.sc[s].H * p.sc[s].W + 31) >> 5;
    for (int i = tid; i < p.occ_cap; i += kLossNegThreads) p.occ_tab[bs * (size_t)p.occ_cap + i] = i < occ_words ? s_occ[i] : 0u;
    if (tid == 0) {
        p.prep_flags[bs] = ok ? 1 : 0;
        if (s == 0) p.gcount[b] = G;
    }
}

__global__ __launch_bounds__(kLossNegThreads) void k_loss_finalize(const LossParams p)
{
    __shared__ double s_sum[kFinalizeClasses];
    loss_finalize<kLossNegThreads>(p, s_sum);
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

// What the gradient launch keeps per image to find the ground-truth rows that share a box.
constexpr unsigned long long kNoBox = ~0ull;
struct PosShare {
    unsigned long long key[kLossMaxPad];       // (scale, first float of the assigned row) of every row of the image; kNoBox: no term
    uint32_t flag[kLossPosGtPerBlock];         // of the workgroup's own rows: bit 0 shares its box, bit 1 is not the first of them
    uint32_t later[kLossPosGtPerBlock][kLossMaxPad / 32];   // of the workgroup's own rows: the later rows of the image with the same box, a bit each
    uint32_t first[kLossPosGtPerBlock];        // the first of them
    uint32_t nlater[kLossPosGtPerBlock];       // their number
};

__device__ __forceinline__ unsigned long long box_key(int s, long long at)
{
    return ((unsigned long long)s << 56) | (unsigned long long)at;
}

// Adds the assigned rows' terms.  Runs after the gradient fill on the same stream.  Same two phases
// as the forward's positive branch.  Several ground-truth rows may be assigned one box (same cell,
// same prior): their terms are added up in the order of the rows by the lanes of the FIRST of them
// and stored once, so the result does not depend on which workgroup comes first (atomic additions
// of three or more terms would).  A box is assigned to one row in all but a few cases: those take
// one addition per channel as before.
__device__ __forceinline__ void loss_grad_pos(const LossParams &p, uint32_t block, float *s_val, int *s_idx, PosLds &l, PosShare &sh)
{
    const int tid = threadIdx.x;
    const int b = (int)(block / (uint32_t)p.pos_blocks);
    const int blk = (int)block - b * p.pos_blocks;
    const int g0 = blk * kLossPosGtPerBlock;
    const float *gt = p.gt + (long long)b * p.pad * 5;
    int G, n;       // G: rows of the image that may carry a key (the count behind k_loss_grad_fill, every record slot behind a forward pass)
    int Gn;         // the image's row count (the normalisation's divisor)
    if (tid < kLossPosGtPerBlock) { sh.flag[tid] = 0u; sh.first[tid] = 0x7fffffffu; sh.nlater[tid] = 0u; }
    for (int i = tid; i < kLossPosGtPerBlock * (kLossMaxPad / 32); i += kPosThreads) (&sh.later[0][0])[i] = 0u;
    const PosRecord *recs = nullptr;
    if (p.pos_rec) {
        // Behind a forward pass of the same call: the records it left, one per slot of the image's positive-branch
        // workgroups (at = -1 where the slot has no row or the row no term).  Everything this phase needs is asked
        // for at once: one memory round trip.
        G = p.pos_blocks * kLossPosGtPerBlock;
        recs = p.pos_rec + (long long)b * G;
        Gn = p.gcount[b];
        for (int g = tid; g < G; g += kPosThreads) {
            const PosRecord rec = recs[g];
            sh.key[g] = rec.at >= 0 ? box_key(rec.s, rec.at) : kNoBox;
            const int gl = g - g0;
            if (gl >= 0 && gl < kLossPosGtPerBlock) {
                const float *rowp = nullptr;
                float *growp = nullptr;
                if (rec.at >= 0) {
#pragma unroll
                    for (int s = 0; s < YH_MAX_SCALES; ++s)
                        if (s < p.S && s == rec.s) { rowp = p.sc[s].head + rec.at; growp = p.sc[s].grad + rec.at; }
                }
                l.rowp[gl] = rowp; l.growp[gl] = growp;
                l.fy[gl] = rec.fy; l.fx[gl] = rec.fx; l.lh[gl] = rec.lh; l.lw[gl] = rec.lw; l.label[gl] = rec.label;
            }
        }
        __syncthreads();
        n = kLossPosGtPerBlock;
    } else {
        G = Gn = block_gt_count(gt, p.pad, s_val, s_idx, kPosThreads);
        for (int g = tid; g < G; g += kPosThreads) {
            const GtAssign a = assign_gt(p, gt + (long long)g * 5);
            long long rows_img = 0;
#pragma unroll
            for (int s = 0; s < YH_MAX_SCALES; ++s)
                if (s < p.S && s == a.s) rows_img = p.sc[s].rows_img;
            sh.key[g] = a.inside ? box_key(a.s, ((long long)b * rows_img + a.row) * p.K) : kNoBox;
        }
        n = pos_assign(p, b, blk, G, l, true);
    }
    // (both branches end with a barrier: the keys and the cleared flags are visible)
    for (int g = tid; g < G; g += kPosThreads) {
        const unsigned long long key = sh.key[g];
        if (key == kNoBox) continue;
        for (int gl = 0; gl < kLossPosGtPerBlock; ++gl) {
            const int mine = g0 + gl;
            if (mine < G && mine != g && sh.key[mine] == key) {
                atomicOr(&sh.flag[gl], g < mine ? 3u : 1u);
                if (g > mine) { atomicOr(&sh.later[gl][g >> 5], 1u << (g & 31)); atomicMin(&sh.first[gl], (uint32_t)g); atomicAdd(&sh.nlater[gl], 1u); }
            }
        }
    }
    __syncthreads();
    const int C = p.C, K = p.K;
    const uint32_t magic = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)K) + 1u;
    float w = p.grad_w[b];
    if (p.normalize) w = w / (float)Gn;
    auto term = [&](float v, int k, int label, float fy, float fx, float lh, float lw) {
        const float z = k < C ? (k == label ? 1.0f : 0.0f) : (k == C ? fy : (k == C + 1 ? fx : 1.0f));
        const float tgt = k == C + 2 ? lh : lw;
        const float diff = (k == C + 2 || k == C + 3) ? v - tgt : sigmoidf(v) - z;
        const float d = (k < C ? p.w_class : (k < C + 4 ? p.w_coord : p.w_obj)) * diff;
        return d * w;
    };
    // Behind the fused forward pass the word added to is known to be zero: that pass counts no box of a cell that
    // holds a ground-truth centre, and an assigned box lies in such a cell; nothing is read back then.  Behind
    // k_loss_grad_fill the caller's mask decides what the word holds.
    const bool after_forward = p.pos_rec != nullptr;
    // kGradBatch items per lane at a time: the head values, the gradient words where they are read and, for the
    // first row of a box shared by two, the record of the other row are asked for together (one memory round trip
    // per batch; item after item, every item's loads wait behind the store of the item before, which the compiler
    // may not move them across).  A row that is not the first of its box has nothing to do.
    constexpr int kGradBatch = 6;
    bool leftovers = false;                                     // a box shared by three or more rows, or shared and no records
    for (int it0 = tid; it0 < n * K; it0 += kPosThreads * kGradBatch) {
        int gls[kGradBatch], ks[kGradBatch];
        bool on[kGradBatch], two[kGradBatch];
        float v[kGradBatch], old[kGradBatch];
        PosRecord nrec[kGradBatch];
#pragma unroll
        for (int j = 0; j < kGradBatch; ++j) {
            const int it = it0 + j * kPosThreads;
            on[j] = it < n * K;
            gls[j] = on[j] ? (int)__umulhi((uint32_t)it, magic) : 0;
            ks[j] = it - gls[j] * K;
            const uint32_t flag = sh.flag[gls[j]];
            on[j] = on[j] && l.rowp[gls[j]] != nullptr && !(flag & 2u);
            two[j] = on[j] && (flag & 1u);
            if (two[j] && !(recs && sh.nlater[gls[j]] == 1u)) { on[j] = two[j] = false; leftovers = true; }
            v[j] = on[j] ? l.rowp[gls[j]][ks[j]] : 0.0f;
            old[j] = (on[j] && !after_forward) ? l.growp[gls[j]][ks[j]] : 0.0f;
            nrec[j] = PosRecord{-1, 0, 0, 0.0f, 0.0f, 0.0f, 0.0f};
            if (two[j]) nrec[j] = recs[sh.first[gls[j]]];
        }
#pragma unroll
        for (int j = 0; j < kGradBatch; ++j) {
            if (!on[j]) continue;
            const int gl = gls[j], k = ks[j];
            float sum = old[j] + term(v[j], k, l.label[gl], l.fy[gl], l.fx[gl], l.lh[gl], l.lw[gl]);
            if (two[j]) sum += term(v[j], k, nrec[j].label, nrec[j].fy, nrec[j].fx, nrec[j].lh, nrec[j].lw);
            l.growp[gl][k] = sum;
        }
    }
    // What is left: the later rows of the box in their order (the set bits of later[], lowest first), with the targets
    // of the rows' records, or computed again as pos_assign computes them.
    if (!__syncthreads_or(leftovers)) return;
    for (int it = tid; it < n * K; it += kPosThreads) {
        const int gl = (int)__umulhi((uint32_t)it, magic), k = it - gl * K;
        const float *rowp = l.rowp[gl];
        if (!rowp || sh.flag[gl] != 1u || (recs && sh.nlater[gl] == 1u)) continue;
        const float v = rowp[k];
        float sum = (after_forward ? 0.0f : l.growp[gl][k]) + term(v, k, l.label[gl], l.fy[gl], l.fx[gl], l.lh[gl], l.lw[gl]);
        for (int w = (g0 + gl) >> 5; w * 32 < G; ++w) {
            uint32_t bits = sh.later[gl][w];                    // only rows after this one were entered
            while (bits != 0u) {
                const int g = w * 32 + __builtin_ctz(bits);
                bits &= bits - 1u;
                if (recs) {
                    const PosRecord rec = recs[g];
                    sum += term(v, k, rec.label, rec.fy, rec.fx, rec.lh, rec.lw);
                } else {
                    const GtAssign a = assign_gt(p, gt + (long long)g * 5);
                    sum += term(v, k, a.label, a.gy - floorf(a.gy), a.gx - floorf(a.gx), logf(a.gh / a.prior_h), logf(a.gw / a.prior_w));
                }
            }
        }
        l.growp[gl][k] = sum;
    }
}

__global__ __launch_bounds__(kPosThreads) void k_loss_grad_pos(const LossParams p)
{
    __shared__ float s_val[kPosThreads / 64];
    __shared__ int s_idx[kPosThreads / 64];
    __shared__ PosLds s_pos;
    __shared__ PosShare s_share;
    loss_grad_pos(p, blockIdx.x, s_val, s_idx, s_pos, s_share);
}

// After the fused forward + gradient-fill pass: the assigned rows' terms, and in one more
// workgroup of the same launch the finalize step (the two are independent of each other).
__global__ __launch_bounds__(kPosThreads) void k_loss_grad_pos_finalize(const LossParams p)
{
    __shared__ float s_val[kPosThreads / 64];
    __shared__ int s_idx[kPosThreads / 64];
    __shared__ double s_sum[kFinalizeClasses];
    __shared__ PosLds s_pos;
    __shared__ PosShare s_share;
    if (blockIdx.x + 1u == gridDim.x) loss_finalize<kPosThreads>(p, s_sum);
    else loss_grad_pos(p, blockIdx.x, s_val, s_idx, s_pos, s_share);
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

hipError_t launch_loss_prepare(const LossParams &p, hipStream_t stm)
{
    if (p.B == 0) return hipSuccess;
    int occ_words = 0;
    for (int s = 0; s < p.S; ++s) {
        const int w = (p.sc[s].H * p.sc[s].W + 31) >> 5;
        if (w > occ_words) occ_words = w;
    }
    const size_t lds = loss_lds_bytes(p.pad, occ_words);
    const hipError_t e = raise_lds_limit(reinterpret_cast<const void *>(k_loss_prepare), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_loss_prepare, dim3((uint32_t)p.B * (uint32_t)p.S), dim3(kLossNegThreads), lds, stm, p);
    return hipGetLastError();
}

size_t loss_lds_bytes(int pad, int occ_words)
{
    return (size_t)pad * 5 * sizeof(float) + (size_t)occ_words * sizeof(uint32_t);
}

hipError_t launch_loss(const LossParams &p, hipStream_t stm)
{
    if (p.B == 0) return hipSuccess;
    int occ_words = 0;
    for (int s = 0; s < p.S; ++s) {
        const int w = (p.sc[s].H * p.sc[s].W + 31) >> 5;
        if (w > occ_words) occ_words = w;
    }
    const size_t lds = p.prepared ? 0 : loss_lds_bytes(p.pad, occ_words);   // a prepared launch stages nothing
    const dim3 grid((uint32_t)p.B * (uint32_t)(p.pos_blocks + p.blocks_per_img));
    const bool grad = p.grad_w != nullptr;                     // forward and gradient in one pass over the rows
    void (*kernel)(const LossParams);
    if (p.prepared)
        kernel = p.iou_max ? (grad ? k_loss_main<true, true, true> : k_loss_main<true, false, true>)
                           : (grad ? k_loss_main<false, true, true> : k_loss_main<false, false, true>);
    else
        kernel = p.iou_max ? (grad ? k_loss_main<true, true, false> : k_loss_main<true, false, false>)
                           : (grad ? k_loss_main<false, true, false> : k_loss_main<false, false, false>);
    hipError_t e = raise_lds_limit(reinterpret_cast<const void *>(kernel), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, grid, dim3(kLossNegThreads), lds, stm, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (grad)
        hipLaunchKernelGGL(k_loss_grad_pos_finalize, dim3((uint32_t)p.B * (uint32_t)p.pos_blocks + 1u), dim3(kPosThreads), 0, stm, p);
    else if (p.ticket_slot < 0)
        hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(kLossNegThreads), 0, stm, p);
    return hipGetLastError();
}

}  // namespace yh
