// Streaming pass over the head tensor: decode, score filter and candidate
// compaction in one read of HBM.  Replaces YOLOv3.py:338-357 / YOLOv2.py:183-199.
//
// The NHWC head of one scale is a row-major matrix [B * rows_img][K], K = C + 5,
// one row per box.  A workgroup walks tiles of whole rows of one image.  A tile
// (up to 32 KiB) goes from HBM straight into LDS with LDS-DMA loads
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR destination,
// no VALU work), issued back to back so that the whole tile is in flight at
// once; the 16-byte phase of the tile start is absorbed by an aligned window
// plus a dword shift inside LDS.
//
// The tile is then consumed row-wise from LDS (row pitch K dwords; lanes on
// consecutive rows or consecutive channels hit distinct banks):
//   A. one lane per row reads the objectness logit; since
//      conf = sigmoid(x) * sigmoid(obj) <= min(sigmoid(x), sigmoid(obj)),
//      a row whose objectness is below a conservatively lowered logit(thr)
//      cannot yield a candidate and is dropped with a single compare;
//   B. the surviving rows' class logits are spread over all lanes; those that
//      pass the same prefilter get the exact sigmoid product and threshold, and
//      real candidates decode their box from the tile and stage a key.
// The [N][C] confidence matrix is never materialised and nothing is read
// twice from memory.  One global atomic per workgroup reserves the range in
// the image's candidate list; the launch is oversubscribed (a few thousand
// short workgroups) so that this latency-bound tail overlaps other workgroups'
// streaming.
#include "yh_internal.h"

namespace yh {

constexpr int kThreads = 256;
constexpr int kStage = 512;      // candidate keys staged per workgroup

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

struct Stage {
    float tile[kTileFloats];
    unsigned long long keys[kStage];
    uint16_t rows[kTileFloats / 6 + 1];   // rows that passed step A (K >= 6)
    int nrows;
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
__device__ __forceinline__ float4 decode_box(int P, const StreamScale &sc, float ty, float tx, float th,
                                             float tw, uint32_t row)
{
    const uint32_t cell = row / (uint32_t)P;
    const uint32_t pi = row - cell * (uint32_t)P;
    const uint32_t gy = cell / sc.grid_w;
    const uint32_t gx = cell - gy * sc.grid_w;
    const float by = ((float)gy + 0.5f) + sigmoidf(ty);
    const float bx = ((float)gx + 0.5f) + sigmoidf(tx);
    const float bh = sc.prior_hw[pi][0] + expf(th);
    const float bw = sc.prior_hw[pi][1] + expf(tw);
    const float m = sc.mult;
    float4 o;
    o.x = (by - bh / 2.0f) * m;
    o.y = (bx - bw / 2.0f) * m;
    o.z = (by + bh / 2.0f) * m;
    o.w = (bx + bw / 2.0f) * m;
    return o;
}

// Appends the staged candidate keys to the image's list: one global atomic per
// call reserves the range.  Called by every thread of the workgroup.
__device__ __forceinline__ void flush_candidates(Stage &st, const StreamParams &p, int b)
{
    const int tid = threadIdx.x;
    const int cnt = st.count < kStage ? st.count : kStage;   // same value in every thread (read after a barrier)
    if (cnt > 0) {
        if (tid == 0) st.base = atomicAdd(&p.img_count[b], cnt);
        __syncthreads();
        const int base = st.base;
        for (int i = tid; i < cnt; i += kThreads) {
            const int g = base + i;
            if (g < p.cand_cap) p.cand[(long long)b * p.cand_cap + g] = st.keys[i];
        }
        __syncthreads();
        if (tid == 0) st.count = 0;
        __syncthreads();
    }
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void k_stream(const StreamParams p)
{
    __shared__ __attribute__((aligned(16))) Stage st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { st.count = 0; st.nrows = 0; }

    int s = 0;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.S && blockIdx.x >= p.sc[i].blk_begin) s = i;
    const StreamScale &sc = p.sc[s];
    const uint32_t r = blockIdx.x - sc.blk_begin;
    const int b = (int)(r / sc.chunks_per_img);
    const uint32_t chunk = r - (uint32_t)b * sc.chunks_per_img;

    const int K = p.K, C = p.C, R = p.tile_rows;
    const long long s0 = (long long)b * sc.span;            // first float of this image at this scale
    const long long total = (long long)p.B * sc.span;       // floats in the tensor
    const long long q_last = (total >> 2) - 1;              // last slot that lies completely inside the tensor
    const float4 *src4 = reinterpret_cast<const float4 *>(sc.src);
    const float pf = (MODE == kModeConf) ? p.thr : p.prefilter;

    const uint32_t row_begin = chunk * sc.tiles_per_chunk * (uint32_t)R;
    uint32_t row_end = row_begin + sc.tiles_per_chunk * (uint32_t)R;
    if (row_end > sc.rows_img) row_end = sc.rows_img;

    for (uint32_t row_t = row_begin; row_t < row_end; row_t += (uint32_t)R) {
        const int rows_t = (int)((row_end - row_t) < (uint32_t)R ? (row_end - row_t) : (uint32_t)R);
        const long long f0 = s0 + (long long)row_t * K;      // first float of the tile
        const long long q0 = f0 >> 2;                         // aligned window start (slots)
        const int sh = (int)(f0 & 3);                         // dword shift of the tile inside the window
        const int wslots = (sh + rows_t * K + 3) >> 2;       // window slots, <= kTileFloats / 4
        // ---- HBM -> LDS, whole tile in flight
        __syncthreads();                                      // previous tile fully consumed
        long long nd = q_last - q0 + 1;                       // slots of the window that may be read as 16 bytes
        const int ndma = nd < wslots ? (int)(nd < 0 ? 0 : nd) : wslots;
#pragma unroll
        for (int j = 0; j < kTileFloats / 4 / kThreads; ++j) {
            const int slot0 = (j * (kThreads / 64) + wave) * 64;
            if (slot0 < ndma) {
                int slot = slot0 + lane;
                slot = slot < ndma ? slot : ndma - 1;        // lanes past the end re-read the last slot
                __builtin_amdgcn_global_load_lds((gptr_t *)(src4 + q0 + slot), (lptr_t *)(st.tile + slot0 * 4), 16, 0, 0);
            }
        }
        if (ndma < wslots) {                                  // the tensor's last, partial slot: 1..3 floats
            const int left = (int)(total - (q0 + ndma) * 4);
            if (tid < left) st.tile[ndma * 4 + tid] = sc.src[(q0 + ndma) * 4 + tid];
        }
        __syncthreads();                                      // waits for the DMA (vmcnt) and publishes the tile
        const float *tile = st.tile + sh;

        // ---- A: rows whose objectness can still reach the threshold
        int nrows;
        if (MODE == kModeDetect) {
            for (int rr = tid; rr < rows_t; rr += kThreads) {
                if (tile[rr * K + C + 4] >= pf) st.rows[atomicAdd(&st.nrows, 1)] = (uint16_t)rr;
            }
            __syncthreads();
            nrows = st.nrows;
        } else {
            nrows = rows_t;
        }
        // ---- B: class logits of those rows, spread over all lanes
        const int items = nrows * C;
        for (int item = tid; item < items; item += kThreads) {
            const int ri = item / C, c = item - ri * C;
            const int rr = (MODE == kModeDetect) ? (int)st.rows[ri] : ri;
            const float *rp = tile + rr * K;
            const float x = rp[c];
            if (!(x >= pf)) continue;
            const uint32_t row = row_t + (uint32_t)rr;
            const int n = (int)(sc.row_base + row);
            if (MODE == kModeConf) {                          // the stream is the confidence matrix itself
                push_candidate(st, p, b, make_key(c, x, n));
            } else {
                const float cf = sigmoidf(x) * sigmoidf(rp[C + 4]);
                if (cf >= p.thr) {
                    reinterpret_cast<float4 *>(p.boxes)[(long long)b * p.N + n] =
                        decode_box(p.P, sc, rp[C], rp[C + 1], rp[C + 2], rp[C + 3], row);
                    push_candidate(st, p, b, make_key(c, cf, n));
                }
            }
        }
        __syncthreads();
        if (tid == 0) st.nrows = 0;
        if (st.count >= kStage / 2) flush_candidates(st, p, b);
    }
    __syncthreads();
    flush_candidates(st, p, b);
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
        reinterpret_cast<float4 *>(p.boxes)[n] = decode_box(p.P, sc, img[l], img[l + 1], img[l + 2], img[l + 3], row);
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
