// Streaming pass over the head tensor: decode, score filter and candidate
// compaction in one read of HBM.  Replaces YOLOv3.py:338-357 / YOLOv2.py:183-199.
//
// The NHWC head of one scale is a row-major matrix [B * rows_img][K], K = C + 5.
// A workgroup owns a run of 16-byte slots of one image and every lane reads
// whole float4 slots of the flat stream (lane i -> base + 16 * i), so the HBM
// access is fully coalesced whatever K is.  The loads of one loop trip are
// unconditional (indices are clamped, results masked), so eight of them are in
// flight per lane before the first is consumed.
//
// Detect mode never materialises the [N][C] confidence matrix.  A class logit x
// can only reach conf = sigmoid(x) * sigmoid(obj) >= thr if sigmoid(x) >= thr,
// so phase 1 is one compare per element against a conservatively lowered
// logit(thr); the channel of an element is tracked incrementally (no division).
// Survivors (about one per box) are staged in LDS as flat indices.  Phase 2
// gives each staged element to a lane: load the row's objectness, evaluate the
// exact confidence, compare with the threshold, decode and store the box,
// stage the candidate key; one global atomic per workgroup then reserves the
// range in the image's candidate list.
#include "yh_internal.h"

namespace yh {

constexpr int kThreads = 256;
constexpr int kUnroll = 8;
constexpr int kPreStage = 4096;  // phase-1 survivors staged per workgroup (flat indices)
constexpr int kStage = 1024;     // candidate keys staged per workgroup

static_assert(kThreads * kUnroll == (int)kStreamSlotGranule, "slot granule");

struct Stage {
    unsigned long long keys[kStage];
    uint32_t pre[kPreStage];
    int pre_count;
    int count;
    int base;
};

__device__ __forceinline__ void push_candidate(Stage &st, const StreamParams &p, int b,
                                               unsigned long long key)
{
    const int slot = atomicAdd(&st.count, 1);
    if (slot < kStage) {
        st.keys[slot] = key;
    } else {
        const int g = atomicAdd(&p.img_count[b], 1);
        if (g < p.cand_cap) p.cand[(long long)b * p.cand_cap + g] = key;
    }
}

// Box of row `row` (YOLOv3.py:340-348): yx = cell + 0.5 + sigmoid(t_yx),
// hw = prior + exp(t_hw), corners scaled by the per-scale multiplier.
__device__ __forceinline__ float4 decode_box(int P, const StreamScale &sc, const float *t, uint32_t row)
{
    const uint32_t cell = row / (uint32_t)P;
    const uint32_t pi = row - cell * (uint32_t)P;
    const uint32_t gy = cell / sc.grid_w;
    const uint32_t gx = cell - gy * sc.grid_w;
    const float by = ((float)gy + 0.5f) + sigmoidf(t[0]);
    const float bx = ((float)gx + 0.5f) + sigmoidf(t[1]);
    const float bh = sc.prior_hw[pi][0] + expf(t[2]);
    const float bw = sc.prior_hw[pi][1] + expf(t[3]);
    const float m = sc.mult;
    float4 o;
    o.x = (by - bh / 2.0f) * m;
    o.y = (bx - bw / 2.0f) * m;
    o.z = (by + bh / 2.0f) * m;
    o.w = (bx + bw / 2.0f) * m;
    return o;
}

// Exact evaluation of one element (flat index l inside the image's span).
template <int MODE>
__device__ __forceinline__ void evaluate(Stage &st, const StreamParams &p, const StreamScale &sc,
                                         const float *img, int b, uint32_t l)
{
    const uint32_t K = (uint32_t)p.K, C = (uint32_t)p.C;
    const uint32_t row = __umulhi(l, p.magic_k);          // l / K, exact for l * K < 2^32
    const uint32_t k = l - row * K;
    if (k >= C) return;
    const float x = img[l];
    const int n = (int)(sc.row_base + row);
    if (MODE == kModeConf) {                               // the stream is the confidence matrix itself
        if (x >= p.thr) push_candidate(st, p, b, make_key((int)k, x, n));
        return;
    }
    const float *rowp = img + (size_t)row * K;
    const float cf = sigmoidf(x) * sigmoidf(rowp[C + 4]);
    if (cf >= p.thr) {
        reinterpret_cast<float4 *>(p.boxes)[(long long)b * p.N + n] = decode_box(p.P, sc, rowp + C, row);
        push_candidate(st, p, b, make_key((int)k, cf, n));
    }
}

// Phase-1 survivor: remember its flat index.  Overflow is detected after the
// loop (pre_count > kPreStage) and handled by re-walking the chunk.
__device__ __forceinline__ void stage_survivor(Stage &st, uint32_t l)
{
    const int slot = atomicAdd(&st.pre_count, 1);
    if (slot < kPreStage) st.pre[slot] = l;
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void k_stream(const StreamParams p)
{
    __shared__ Stage st;
    const int tid = threadIdx.x;
    if (tid == 0) { st.count = 0; st.pre_count = 0; }
    __syncthreads();

    int s = 0;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.S && blockIdx.x >= p.sc[i].blk_begin) s = i;
    const StreamScale &sc = p.sc[s];
    const uint32_t r = blockIdx.x - sc.blk_begin;
    const int b = (int)(r / sc.chunks_per_img);
    const uint32_t chunk = r - (uint32_t)b * sc.chunks_per_img;

    const int K = p.K, C = p.C;
    const uint32_t span = sc.span;
    const long long s0 = (long long)b * span;              // first float of this image at this scale
    const float *img = sc.src + s0;
    // slots that lie completely inside the image: [q_lo_img, q_hi_img)
    const long long q_lo_img = (s0 + 3) >> 2;
    const long long q_hi_img = (s0 + span) >> 2;
    long long q_lo = q_lo_img + (long long)chunk * p.chunk_slots;
    long long q_hi = q_lo + p.chunk_slots;
    if (q_hi > q_hi_img) q_hi = q_hi_img;
    const float pf = (MODE == kModeConf) ? p.thr : p.prefilter;

    if (q_lo < q_hi) {
        const float4 *src4 = reinterpret_cast<const float4 *>(sc.src) + q_lo;
        const int nslots = (int)(q_hi - q_lo);              // <= chunk_slots
        // channel of this lane's first element, then advanced by 1024 floats per load
        const uint32_t l_first = (uint32_t)(q_lo * 4 - s0) + 4u * (uint32_t)tid;
        int k = (int)(l_first - __umulhi(l_first, p.magic_k) * (uint32_t)K);
        const int dk = (4 * kThreads) % K;
        for (int it = 0; it < nslots; it += kThreads * kUnroll) {
            float4 v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                int i = it + u * kThreads + tid;
                i = i < nslots ? i : nslots - 1;            // clamped: the load is unconditional
                v[u] = src4[i];
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int i = it + u * kThreads + tid;
                const float xs[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                if (i < nslots && fmaxf(fmaxf(xs[0], xs[1]), fmaxf(xs[2], xs[3])) >= pf) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int kk = k + j;
                        kk = kk >= K ? kk - K : kk;
                        if (kk < C && xs[j] >= pf)
                            stage_survivor(st, l_first + 4u * (uint32_t)(it + u * kThreads) + j);
                    }
                }
                k += dk;
                k = k >= K ? k - K : k;
            }
        }
    }
    // the (up to 3 + 3) elements of the image that share a slot with a neighbour
    // image are evaluated directly
    if (chunk == 0) {
        const uint32_t head_n = (uint32_t)(q_lo_img * 4 - s0);
        const uint32_t tail_n = (uint32_t)((s0 + span) - q_hi_img * 4);
        uint32_t l = 0xffffffffu;
        if (q_lo_img <= q_hi_img) {
            if ((uint32_t)tid < head_n) l = tid;
            else if ((uint32_t)tid >= 32u && (uint32_t)tid - 32u < tail_n) l = span - tail_n + ((uint32_t)tid - 32u);
        } else if ((uint32_t)tid < span) {                   // image smaller than one slot
            l = tid;
        }
        if (l != 0xffffffffu && img[l] >= pf) evaluate<MODE>(st, p, sc, img, b, l);
    }

    // phase 2: exact evaluation of the staged survivors, one per lane
    __syncthreads();
    const int npre = st.pre_count;
    if (npre <= kPreStage) {
        for (int i = tid; i < npre; i += kThreads) evaluate<MODE>(st, p, sc, img, b, st.pre[i]);
    } else if (q_lo < q_hi) {
        // more survivors than the staging area holds (very low thresholds):
        // walk the chunk again, element by element, from the cache
        const uint32_t l0 = (uint32_t)(q_lo * 4 - s0);
        const uint32_t nel = 4u * (uint32_t)(q_hi - q_lo);
        for (uint32_t i = tid; i < nel; i += kThreads)
            if (img[l0 + i] >= pf) evaluate<MODE>(st, p, sc, img, b, l0 + i);
    }

    __syncthreads();
    const int cnt = st.count < kStage ? st.count : kStage;
    if (cnt > 0) {
        if (tid == 0) st.base = atomicAdd(&p.img_count[b], cnt);
        __syncthreads();
        const int base = st.base;
        for (int i = tid; i < cnt; i += kThreads) {
            const int g = base + i;
            if (g < p.cand_cap) p.cand[(long long)b * p.cand_cap + g] = st.keys[i];
        }
    }
}

// Full decode (yh_decode): boxes [B][N][4] and, optionally, conf [B][N][C].
// grid = (ceil(span / 256), B) per scale: one element of the head per lane,
// coalesced loads and stores; the row's objectness comes from the cache.
__global__ __launch_bounds__(kThreads) void k_decode_full(const StreamParams p, int s)
{
    const StreamScale &sc = p.sc[s];
    const int b = blockIdx.y;
    const uint32_t l = blockIdx.x * kThreads + threadIdx.x;
    if (l >= sc.span) return;
    const uint32_t K = (uint32_t)p.K, C = (uint32_t)p.C;
    const float *img = sc.src + (long long)b * sc.span;
    const uint32_t row = __umulhi(l, p.magic_k);
    const uint32_t k = l - row * K;
    const long long n = (long long)b * p.N + sc.row_base + row;
    if (k < C) {
        if (p.conf) p.conf[n * C + k] = sigmoidf(img[l]) * sigmoidf(img[(size_t)row * K + C + 4]);
    } else if (k == C) {
        reinterpret_cast<float4 *>(p.boxes)[n] = decode_box(p.P, sc, img + l, row);
    }
}

hipError_t launch_stream(const StreamParams &p, StreamMode mode, uint32_t blocks, hipStream_t stm)
{
    if (mode == kModeFull) {
        for (int s = 0; s < p.S; ++s) {
            const dim3 grid((p.sc[s].span + kThreads - 1) / kThreads, (uint32_t)p.B);
            hipLaunchKernelGGL(k_decode_full, grid, dim3(kThreads), 0, stm, p, s);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    if (blocks == 0) return hipSuccess;
    const dim3 grid(blocks), block(kThreads);
    if (mode == kModeDetect) hipLaunchKernelGGL(k_stream<kModeDetect>, grid, block, 0, stm, p);
    else hipLaunchKernelGGL(k_stream<kModeConf>, grid, block, 0, stm, p);
    return hipGetLastError();
}

}  // namespace yh
