// Streaming pass over the head tensor: decode, score filter and candidate
// compaction in one read of HBM.  Replaces YOLOv3.py:338-357 / YOLOv2.py:183-199.
//
// The NHWC head of one scale is a row-major matrix [B * rows_img][K], K = C + 5,
// one row per box.  A workgroup walks tiles of whole rows of one image.  A tile
// (up to 16 KiB) goes from HBM straight into LDS with LDS-DMA loads
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR destination,
// no VALU work), issued back to back so that the whole tile is in flight at
// once; the 16-byte phase of the tile start is absorbed by an aligned window
// plus a dword shift inside LDS.
//
// The tile is then consumed from LDS in two steps:
//   A. a flat scan, 16 bytes per lane and step, tests every float against a
//      conservatively lowered logit(thr) (conf = sigmoid(x) * sigmoid(obj) <=
//      min(sigmoid(x), sigmoid(obj)), so a class logit below it cannot yield a
//      candidate) and a cheap "is a class column" rule, and leaves one entry per
//      lane with a hit in an LDS list;
//   B. a dense pass over the entries gives them the exact column check, the
//      exact sigmoid product and threshold, and real candidates decode their
//      box from the tile and stage a key.
// The [N][C] confidence matrix is never materialised and nothing is read
// twice from memory.  One global atomic per flush of the staged keys reserves
// the range in the image's candidate list; the launch is oversubscribed (a few
// thousand short workgroups) so that this latency-bound tail overlaps other
// workgroups' streaming.
//
// Three kernels share this file: k_stream (workgroup tiles, described above:
// dense candidate sets, yh_nms), k_stream_wave (wave-private ring of tiles,
// reads the whole head without barriers) and k_detect_rows (objectness-first:
// does not read the whole head; the default of yh_detect).
#include <cstring>
#include <mutex>
#include <type_traits>
#include "yh_post.h"

namespace yh {

constexpr int kThreads = 256;
constexpr int kStage = 512;      // candidate keys staged per workgroup
constexpr int kHitCap = 512;     // class logits that passed the prefilter, collected before they are worked on

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

static_assert(kTileFloats / 4 <= 4096, "a scan entry keeps the slot number in 12 bits");
struct Stage {
    float tile[kTileFloats];
    unsigned long long keys[kStage];
    uint16_t hits[kHitCap];               // scan entries: 16-byte slot of the window (12 bits) | a bit per float that passed
    int nhits;
    int count;
    int base;
};

// Classes of the keys this workgroup has put on the image's list (dynamic LDS, [C], when the launch counts them).
extern __shared__ int s_class_hist[];

__device__ __forceinline__ void push_candidate(Stage &st, const StreamParams &p, int b,
                                               unsigned long long key)
{
    const int slot = atomicAdd(&st.count, 1);
    if (slot < kStage) {
        st.keys[slot] = key;
    } else {
        const int g = atomicAdd(&p.img_count[b], 1);
        if ((uint32_t)g < (uint32_t)p.cand_cap) {
            p.cand[(long long)b * p.cand_cap + g] = key;
            if (p.class_hist) atomicAdd(&s_class_hist[key_class(key)], 1);
        }
    }
}

// Appends the staged candidate keys to the image's list: one global atomic per
// call reserves the range.  Called by every thread of the workgroup with the
// value st.count had after the last barrier (the caller reads it before it puts
// the next tile's LDS-DMA loads in flight: an LDS read behind them would wait for
// them, and the atomic below is meant to travel together with them).
__device__ __forceinline__ void flush_candidates(Stage &st, const StreamParams &p, int b, int staged)
{
    const int tid = threadIdx.x;
    const int cnt = staged < kStage ? staged : kStage;
    if (cnt > 0) {
        if (tid == 0) st.base = atomicAdd(&p.img_count[b], cnt);
        __syncthreads();
        const int base = st.base;
        for (int i = tid; i < cnt; i += kTileThreads) {
            const int g = base + i;
            if ((uint32_t)g < (uint32_t)p.cand_cap) {
                const unsigned long long key = st.keys[i];
                p.cand[(long long)b * p.cand_cap + g] = key;
                if (p.class_hist) atomicAdd(&s_class_hist[key_class(key)], 1);
            }
        }
        __syncthreads();
        if (tid == 0) st.count = 0;
        __syncthreads();
    }
}

template <int MODE>
__global__ __launch_bounds__(kTileThreads) void k_stream(const StreamParams p)
{
    __shared__ __attribute__((aligned(16))) Stage st;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);   // the wavefront's number, known to be uniform
    if (tid == 0) { st.count = 0; st.nhits = 0; }
    if (p.class_hist)
        for (int c = tid; c < p.C; c += kTileThreads) s_class_hist[c] = 0;

    // A cut is one workgroup's worth of tiles.  Launched with one workgroup per cut the hardware hands them
    // out; launched with fewer (a fixed number per compute unit, so that LDS and wavefront slots stay free
    // for another stream's kernels) every workgroup walks its cuts in turn.
    for (uint32_t cut = blockIdx.x; cut < p.cuts; cut += gridDim.x) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.S && cut >= p.sc[i].blk_begin) s = i;
    const StreamScale &sc = p.sc[s];
    const uint32_t r = cut - sc.blk_begin;
    const int b = (int)(r / sc.chunks_per_img);
    const uint32_t chunk = r - (uint32_t)b * sc.chunks_per_img;

    const int K = p.K, C = p.C, R = p.tile_rows;
    const long long s0 = (long long)b * sc.span;            // first float of this image at this scale
    const long long total = (long long)p.B * sc.span;       // floats in the tensor
    const long long q_last = (total >> 2) - 1;              // last slot that lies completely inside the tensor
    const float4 *src4 = reinterpret_cast<const float4 *>(sc.src);
    const float pf = (MODE == kModeConf) ? p.thr : p.prefilter;

    // The workgroup's tiles are chunk, chunk + chunks_per_img, chunk + 2 * chunks_per_img, ... of the image:
    // the workgroups that are resident together (consecutive launch ids) then read one contiguous,
    // advancing window of memory, whatever the number of tiles each takes.  (With a run of consecutive
    // tiles per workgroup the kernel's time swung by 8 % with the run's length: 955 us at 48 tiles,
    // 900 at 6, 976 at 7, 949 at 8, 906 at 9 on the dense workload: 128 images of 100x100x10 boxes, 80 classes.)
    const uint32_t tile_stride = sc.chunks_per_img * (uint32_t)R;
    const uint32_t row_begin = chunk * (uint32_t)R;
    const uint32_t row_end = sc.rows_img;

    // HBM -> LDS, the whole tile in flight at once
    auto issue = [&](uint32_t row_t) {
        const int rows_t = (int)((row_end - row_t) < (uint32_t)R ? (row_end - row_t) : (uint32_t)R);
        const long long f0 = s0 + (long long)row_t * K;      // first float of the tile
        const long long q0 = f0 >> 2;                         // aligned window start (slots)
        const int sh = (int)(f0 & 3);                         // dword shift of the tile inside the window
        const int wslots = (sh + rows_t * K + 3) >> 2;       // window slots, <= kTileFloats / 4
        long long nd = q_last - q0 + 1;                       // slots of the window that may be read as 16 bytes
        const int ndma = nd < wslots ? (int)(nd < 0 ? 0 : nd) : wslots;
        // a wavefront's 64 slots per load: the group's first slot is a scalar (the branches below are
        // scalar branches), the lane's address one 64-bit add away from a pointer computed once per tile
        const float4 *lane_src = src4 + q0 + lane;
#pragma unroll
        for (int j = 0; j < kTileFloats / 4 / kTileThreads; ++j) {
            const int slot0 = (j * (kTileThreads / 64) + wave_u) * 64;
            if (slot0 + 64 <= ndma) {
                __builtin_amdgcn_global_load_lds((gptr_t *)(lane_src + slot0), (lptr_t *)(st.tile + slot0 * 4), 16, 0, 0);
            } else if (slot0 < ndma) {
                // The window's last group: the lanes past the end load nothing.  (A lane that re-read the last
                // slot instead would also write the LDS slots behind it, and slot ndma is where another
                // wavefront's plain store puts the tensor's partial slot below: the later of the two would win.)
                if (slot0 + lane < ndma)
                    __builtin_amdgcn_global_load_lds((gptr_t *)(lane_src + slot0), (lptr_t *)(st.tile + slot0 * 4), 16, 0, 0);
            }
        }
        if (ndma < wslots) {                                  // the tensor's last, partial slot: 1..3 floats
            const int left = (int)(total - (q0 + ndma) * 4);
            if (tid < left) st.tile[ndma * 4 + tid] = sc.src[(q0 + ndma) * 4 + tid];
        }
    };
    if (row_begin < row_end) issue(row_begin);
    for (uint32_t row_t = row_begin; row_t < row_end; row_t += tile_stride) {
        const int rows_t = (int)((row_end - row_t) < (uint32_t)R ? (row_end - row_t) : (uint32_t)R);
        const long long f0 = s0 + (long long)row_t * K;
        const int sh = (int)(f0 & 3);
        const int wslots = (sh + rows_t * K + 3) >> 2;
        __syncthreads();                                      // waits for the DMA (vmcnt) and publishes the tile
        const float *tile = st.tile + sh;

        // ---- scan: every float of the window against the prefilter, 16 bytes per lane and step.
        // The scan is conservative and cheap: a float passes when it reaches the prefilter and its
        // column (float index mod K; the column of a lane's first float is kept up to date by
        // adding (4 * kTileThreads) mod K per step) is not one of the K - C columns behind the class
        // logits.  A lane with a hit leaves ONE entry (slot number and a bit per float) in the
        // list; the lanes then take the entries densely and the exact step sorts out what the
        // scan let through (floats outside the tile, columns that wrapped into the next row).
        // (Doing the exact test where an item is found leaves one or two lanes of a wavefront
        // working while the others wait: with dense candidates that made the kernel
        // compute-bound at a few percent lane utilisation.)  An entry that finds the list full
        // is worked on where it was found.
        const int valid = rows_t * K;
        auto exact = [&](int fi) {
            if ((uint32_t)fi >= (uint32_t)valid) return;      // the window's few floats around the tile
            // fi < 2^13; K = 1 (a one-class confidence matrix) has no 32-bit reciprocal
            const int rr = K == 1 ? fi : (int)__umulhi((uint32_t)fi, p.magic_k), c = fi - rr * K;
            if (c >= C) return;
            const float *rp = tile + rr * K;
            const float x = rp[c];
            const uint32_t row = row_t + (uint32_t)rr;
            const int n = (int)(sc.row_base + row);
            if (MODE == kModeConf) {                          // the stream is the confidence matrix itself
                if (x >= pf) push_candidate(st, p, b, make_key(c, x, n));
            } else {
                // the candidate's box is decoded by whoever looks at it later (row_box): the greedy walk sees a
                // few hundred of a class's candidates, the outputs hold max_boxes of them, and a box by row for
                // every candidate was a scattered 16-byte store and a hundred instructions each in this pass
                const float cf = decode_sigmoid(x) * decode_sigmoid(rp[C + 4]);
                if (cf >= p.thr) push_candidate(st, p, b, make_key(c, cf, n));
            }
        };
        auto exact_bits = [&](int w, uint32_t bits) {        // the set floats of slot w
            while (bits) {
                const int k = __ffs((int)bits) - 1;
                bits &= bits - 1u;
                exact(4 * w + k - sh);
            }
        };
        const uint32_t behind = (uint32_t)(K - C);            // columns C .. K-1 hold no class logit
        const int fi_first = 4 * tid - sh;                    // -3 .. -1: before the tile
        int col0 = fi_first >= 0 ? (K == 1 ? 0 : fi_first - (int)__umulhi((uint32_t)fi_first, p.magic_k) * K) : fi_first + K;
        for (int w = tid; w < wslots; w += kTileThreads) {
            const float4 v = reinterpret_cast<const float4 *>(st.tile)[w];
            const uint32_t t0 = (uint32_t)(col0 - C);
            uint32_t hit = (v.x >= pf && t0 >= behind) ? 1u : 0u;
            hit |= (v.y >= pf && t0 + 1u >= behind) ? 2u : 0u;
            hit |= (v.z >= pf && t0 + 2u >= behind) ? 4u : 0u;
            hit |= (v.w >= pf && t0 + 3u >= behind) ? 8u : 0u;
            if (hit) {
                const int slot = atomicAdd(&st.nhits, 1);
                if (slot < kHitCap) st.hits[slot] = (uint16_t)((uint32_t)w | (hit << 12));
                else exact_bits(w, hit);
            }
            col0 += p.scan_col_step;
            col0 = col0 >= K ? col0 - K : col0;
        }
        __syncthreads();
        const int nhits = st.nhits < kHitCap ? st.nhits : kHitCap;
        for (int h = tid; h < nhits; h += kTileThreads) {
            const uint32_t e = st.hits[h];
            exact_bits((int)(e & 0xFFFu), e >> 12);
        }
        __syncthreads();                                      // the tile is consumed
        // The next tile's loads go out before the staged keys are flushed: the flush waits for a
        // global atomic, and that wait and the loads' flight time then overlap.
        const int staged = st.count;
        if (tid == 0) st.nhits = 0;
        if (row_t + tile_stride < row_end) issue(row_t + tile_stride);
        // late: a flush costs a round trip to the image's counter with the whole workgroup waiting; what does
        // not fit the stage any more goes to the list key by key (push_candidate), slower but as correct
        if (staged >= kStage - 128) flush_candidates(st, p, b, staged);
    }
    __syncthreads();
    flush_candidates(st, p, b, st.count);
    if (p.class_hist) {
        // the cut's share of the image's class counts (a cut lies in one image): what k_bin_count would count again
        // from the list
        __syncthreads();
        for (int c = tid; c < C; c += kTileThreads) {
            const int v = s_class_hist[c];
            if (v) { atomicAdd(&p.class_hist[(long long)b * C + c], v); s_class_hist[c] = 0; }
        }
        __syncthreads();
    }
    }
}

// ------------------------------------------------------------ wave-private variant
// Same pass with no workgroup-level synchronisation at all: a workgroup is one
// wavefront that owns a ring of NBUF LDS tile buffers (NI KiB each, whole rows)
// and a run of tiles of one image.  The LDS-DMA loads of the next NBUF-1 tiles
// stay in flight while the wavefront looks at the current one; it waits with a
// counted s_waitcnt instead of a barrier.  Vector-memory operations of a
// wavefront complete in issue order, and every tile is exactly NI loads, so
// "at most NI * (tiles issued later) operations outstanding" proves that the
// current tile has landed, whatever stores were issued in between.
//
// Every LDS read of a tile goes through inline assembly: before an LDS read it
// can see, the compiler waits for all outstanding LDS-DMA loads, the later
// tiles included, which would serialise the ring.
//
// Per tile: wait, one ds_read of the rows' objectness logits, one ballot, issue
// the tile NBUF ahead.  Rows that pass the objectness prefilter (a few percent)
// spread their class logits over the 64 lanes; candidates (key and decoded box)
// are compacted by ballot into a wave-private stage.  At the end of its work
// unit the wavefront stores them into the unit's kUnitSlots private slots and
// kEmptySlot into the rest: plain stores, no atomic, nothing a later wavefront
// has to wait for (one global atomic per unit on the 64 per-image counters cost
// 5-14 us of a 46 us launch).  Only the 17th and later candidates of a unit
// reserve a range of the image's list with a global atomic.
//
// This is the variant yh_detect uses unless a row is wider than a tile or the
// caller's capacity announces dense candidates; the workgroup-tile kernel above
// stays for those and for yh_nms.
constexpr int kWaveStage = 128;   // staged keys per wave; flushed at >= 64, one step adds <= 64

typedef __attribute__((address_space(3))) float lds_float_t;

__device__ __forceinline__ float lds_peek(uint32_t addr)
{
    float v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// five consecutive floats (t_y, t_x, t_h, t_w, objectness)
__device__ __forceinline__ void lds_peek5(uint32_t addr, float &a, float &b, float &c, float &d, float &e)
{
    asm volatile("ds_read_b32 %0, %5\n\tds_read_b32 %1, %5 offset:4\n\tds_read_b32 %2, %5 offset:8\n\t"
                 "ds_read_b32 %3, %5 offset:12\n\tds_read_b32 %4, %5 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e)
                 : "v"(addr)
                 : "memory");
}

// A staged candidate into LDS.  Also assembly: a ds_write the compiler can see is
// ordered behind the outstanding LDS-DMA loads just like a read.
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_poke_candidate(uint32_t key_addr, uint32_t box_addr, unsigned long long key, float4 box)
{
    v4f_t b;
    b.x = box.x; b.y = box.y; b.z = box.z; b.w = box.w;
    asm volatile("ds_write_b64 %0, %2\n\tds_write_b128 %1, %3" ::"v"(key_addr), "v"(box_addr), "v"(key), "v"(b) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct WaveCtx {
    const StreamScale *sc;
    const float4 *src4;
    long long s0, total, q_last;
    uint32_t row_end;
    int K, R;
};

// Issues the loads of the tile that starts at row `row_t` into `tb`.  Returns true
// when the tile is covered by exactly NI LDS-DMA instructions and nothing else.
template <int NI>
__device__ __forceinline__ bool wave_issue_tile(const WaveCtx &w, uint32_t row_t, float *tb, int lane)
{
    const int rows_t = (int)((w.row_end - row_t) < (uint32_t)w.R ? (w.row_end - row_t) : (uint32_t)w.R);
    const long long f0 = w.s0 + (long long)row_t * w.K;
    const long long q0 = f0 >> 2;
    const int sh = (int)(f0 & 3);
    const int wslots = (sh + rows_t * w.K + 3) >> 2;          // <= NI * 64
    const long long nd = w.q_last - q0 + 1;
    const int ndma = nd < wslots ? (int)(nd < 0 ? 0 : nd) : wslots;
    if (ndma > 0) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            int slot = j * 64 + lane;
            slot = slot < ndma ? slot : ndma - 1;             // lanes past the end re-read the last slot
            __builtin_amdgcn_global_load_lds((gptr_t *)(w.src4 + q0 + slot), (lptr_t *)(tb + j * 256), 16, 0, 0);
        }
    }
    if (ndma < wslots) {                                      // the tensor's last, partial slot: 1..3 floats
        const int left = (int)(w.total - (q0 + ndma) * 4);
        if (lane < left) tb[ndma * 4 + lane] = w.sc->src[(q0 + ndma) * 4 + lane];
    }
    return ndma == wslots;
}

// Writes the staged candidates out.  `emitted` counts the unit's candidates written by
// earlier calls: the first kUnitSlots of a unit go to its private slots, the rest to the
// image's list (one global atomic reserves the range).
struct WaveOut {
    unsigned long long *ukeys;   // this unit's slots
    float4 *uboxes;
    unsigned long long *cand;    // this image's list
    float4 *cand_box;
    int32_t *img_count;          // this image's list counter
    int cand_cap;
};

__device__ __forceinline__ void wave_flush(const WaveOut &o, const unsigned long long *keys, const float4 *boxes,
                                           int &count, int &emitted, int lane)
{
    int n_unit = kUnitSlots - emitted;
    n_unit = n_unit < 0 ? 0 : (n_unit < count ? n_unit : count);
    const int n_list = count - n_unit;
    int base = 0;
    if (n_list > 0) {
        if (lane == 0) base = atomicAdd(o.img_count, n_list);
        base = __builtin_amdgcn_readfirstlane(base);
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < count; i += 64) {
        if (i < n_unit) {
            o.ukeys[emitted + i] = keys[i];
            o.uboxes[emitted + i] = boxes[i];
        } else {
            const int g = base + (i - n_unit);
            if ((uint32_t)g < (uint32_t)o.cand_cap) {
                o.cand[g] = keys[i];
                o.cand_box[g] = boxes[i];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    emitted += count;
    count = 0;
}

// LDS of a streaming workgroup: NBUF tiles, then the staged boxes and keys.
constexpr size_t wave_lds_bytes(int ni, int nbuf) { return (size_t)nbuf * ni * 1024 + (size_t)kWaveStage * (16 + 8); }

// The first q.B workgroups of the launch do not stream: each runs the candidates ->
// detections stage of one image of the PREVIOUS batch (q; its candidates were left in
// another workspace by the launch before this one on the same stream), so that the
// latency-bound stage hides behind this batch's streaming instead of following it.
template <int NI, int NBUF>
__global__ __launch_bounds__(64) void k_stream_wave(const StreamParams p, const NmsParams q)
{
    constexpr int kBuf = NI * 256;                            // floats per tile buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    if (blockIdx.x < (uint32_t)q.B) {
        post_image_dyn<64>(q, (int)blockIdx.x, s_dyn);
        return;
    }
    float (*s_tile)[kBuf] = reinterpret_cast<float (*)[kBuf]>(s_dyn);
    float4 *s_box = reinterpret_cast<float4 *>(s_dyn + (size_t)NBUF * kBuf * 4);
    unsigned long long *s_keys = reinterpret_cast<unsigned long long *>(s_dyn + (size_t)NBUF * kBuf * 4 + (size_t)kWaveStage * 16);
    const int lane = threadIdx.x;
    const uint32_t unit = blockIdx.x - (uint32_t)q.B;

    int s = 0;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.S && unit >= p.sc[i].blk_begin) s = i;
    const StreamScale &sc = p.sc[s];
    const uint32_t r = unit - sc.blk_begin;
    const int b = (int)(r / sc.chunks_per_img);
    const uint32_t chunk = r - (uint32_t)b * sc.chunks_per_img;
    const long long u = (long long)b * p.upi + sc.unit_base + chunk;   // this unit's index in the batch

    const int K = p.K, C = p.C, R = p.tile_rows;
    WaveCtx w;
    w.sc = &sc;
    w.src4 = reinterpret_cast<const float4 *>(sc.src);
    w.s0 = (long long)b * sc.span;
    w.total = (long long)p.B * sc.span;
    w.q_last = (w.total >> 2) - 1;
    w.K = K; w.R = R;
    const uint32_t row_begin = chunk * sc.tiles_per_chunk * (uint32_t)R;
    uint32_t row_end = row_begin + sc.tiles_per_chunk * (uint32_t)R;
    if (row_end > sc.rows_img) row_end = sc.rows_img;
    const int ntiles = row_begin < row_end ? (int)((row_end - row_begin + (uint32_t)R - 1) / (uint32_t)R) : 0;
    w.row_end = row_end;
    const float pf = p.prefilter;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    WaveOut o;
    o.ukeys = p.ukeys + u * kUnitSlots;
    o.uboxes = reinterpret_cast<float4 *>(p.uboxes) + u * kUnitSlots;
    o.cand = p.cand + (long long)b * p.cand_cap;
    o.cand_box = reinterpret_cast<float4 *>(p.cand_box) + (long long)b * p.cand_cap;
    o.img_count = p.img_count + b;
    o.cand_cap = p.cand_cap;

    // `irregular`: a tile in flight is not exactly NI loads (the tensor's last
    // slots), so only a full wait proves anything from here on.
    bool irregular = false;
#pragma unroll
    for (int i = 0; i < NBUF; ++i)
        if (i < ntiles) irregular |= !wave_issue_tile<NI>(w, row_begin + (uint32_t)i * (uint32_t)R, s_tile[i], lane);
    int count = 0;                                            // staged candidates, wave-uniform
    int emitted = 0;                                          // candidates of this unit already written out
    int buf = 0;

    for (int t = 0; t < ntiles; ++t) {
        const uint32_t row_t = row_begin + (uint32_t)t * (uint32_t)R;
        const int rows_t = (int)((row_end - row_t) < (uint32_t)R ? (row_end - row_t) : (uint32_t)R);
        float *tb = s_tile[buf];
        const int sh = (int)((w.s0 + (long long)row_t * K) & 3);
        const int later = ntiles - 1 - t;                     // tiles issued after this one: min(later, NBUF - 1)
        if (irregular || later == 0) wait_vmcnt<0>();
        else if (NBUF == 2 || later == 1) wait_vmcnt<NI>();
        else if (NBUF == 3 || later == 2) wait_vmcnt<2 * NI>();
        else wait_vmcnt<3 * NI>();
        const uint32_t row0_off = (uint32_t)(uintptr_t)(lds_float_t *)tb + (uint32_t)sh * 4u;
        for (int r0 = 0; r0 < rows_t; r0 += 64) {
            const int rr = r0 + lane;
            const int rc = rr < rows_t ? rr : rows_t - 1;
            const float obj_l = lds_peek(row0_off + (uint32_t)(rc * K + C + 4) * 4u);
            unsigned long long mask = __ballot(rr < rows_t && obj_l >= pf);
            while (mask) {                                    // wave-uniform: one surviving row per turn
                const int bit = __builtin_ctzll(mask);
                mask &= mask - 1;
                const int rit = r0 + bit;
                const uint32_t rp_off = row0_off + (uint32_t)(rit * K) * 4u;
                float ty, tx, th, tw, obj;
                lds_peek5(rp_off + (uint32_t)C * 4u, ty, tx, th, tw, obj);
                const float so = decode_sigmoid(obj);
                const uint32_t row = row_t + (uint32_t)rit;
                const int n = (int)(sc.row_base + row);
                bool have_box = false;
                float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int c0 = 0; c0 < C; c0 += 64) {
                    const int c = c0 + lane;
                    const float x = lds_peek(rp_off + (uint32_t)(c < C ? c : C - 1) * 4u);
                    bool hit = false;
                    float cf = 0.0f;
                    if (c < C && x >= pf) {
                        cf = decode_sigmoid(x) * so;
                        hit = cf >= p.thr;
                    }
                    const unsigned long long hm = __ballot(hit);
                    if (hm == 0) continue;
                    if (!have_box) {                          // the row's box, once, in every lane
                        box = decode_box(p.P, sc, ty, tx, th, tw, row);
                        have_box = true;
                    }
                    if (hit) {
                        const int pos = count + __builtin_popcountll(hm & lanes_below);
                        lds_poke_candidate((uint32_t)(uintptr_t)(lds_float_t *)(void *)(s_keys + pos),
                                           (uint32_t)(uintptr_t)(lds_float_t *)(void *)(s_box + pos),
                                           make_key(c, cf, n), box);
                    }
                    count += __builtin_popcountll(hm);
                    if (count >= 64) wave_flush(o, s_keys, s_box, count, emitted, lane);
                }
            }
        }
        if (t + NBUF < ntiles)
            irregular |= !wave_issue_tile<NI>(w, row_t + (uint32_t)NBUF * (uint32_t)R, tb, lane);
        buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
    if (count > 0) wave_flush(o, s_keys, s_box, count, emitted, lane);
    if (lane >= emitted && lane < kUnitSlots) o.ukeys[lane] = kEmptySlot;        // the unit's unused slots
}

// ------------------------------------------------------------ objectness-first variant
// conf = sigmoid(x) * sigmoid(obj) >= thr needs sigmoid(obj) >= thr, so a row's class logits
// matter only if its objectness passes the (conservatively lowered) logit(thr): with the
// few-percent survival rate of a detector's head, most of the tensor need not be read at all.
//
// A wavefront owns one work unit: `steps` steps of 64 rows of one image and scale, and walks
// them on its own (a workgroup is four such wavefronts on consecutive units of one image and
// has no barrier).  Per step:
//   1. a lane reads the objectness logit of one row (one float per K: the memory system
//      fetches a 128-byte line per row instead of the row); the next step's logits are
//      requested before this step's are looked at;
//   2. the rows that pass are appended to the wavefront's ring of survivors in LDS;
//   3. whenever kRowsBatch survivors are queued (and at the end of the unit) the queued rows are
//      worked on in batches of kRowsBatch: sixteen lanes per row, four rows per turn, the loads of all
//      kRowsTurns turns in flight at once (a lane holds kRowsX class logits of its row per turn); a
//      turn only compares its logits with the prefilter and queues those that pass; one pass per
//      batch, a lane per queued logit, does the exact sigmoid product and threshold test, decodes the
//      box and stages (key, box) per wavefront.
// How much is in flight is what matters: with every row of the launch requested at once (one
// row per lane and everything resident) a survivor's gather queues behind the whole objectness
// column, so the launch becomes column phase, then gather phase, then stores, each waiting for
// the slowest of the one before; with a few hundred rows in flight per wavefront and the
// wavefronts at different steps the gathers travel between the column reads.
// A wavefront writes what it staged into the private slots of its own unit, the 17th and
// later candidates into the image's list, exactly as k_stream_wave does, and the previous
// batch's candidates -> detections stage rides in the first q.B workgroups in the same way.
#ifdef YH_ROWS_TRACE
// Measurement build only (tools/trace_rows.py): per-wavefront timestamps of k_detect_rows.
constexpr int kTraceWaves = 1 << 16, kTraceWords = 8;
__device__ unsigned long long g_rows_trace[kTraceWaves * kTraceWords];
#define YH_TRACE_STAMP(var) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); var = wall_clock64(); } while (0)
#define YH_TRACE_NOW(var) do { var = wall_clock64(); } while (0)
#else
#define YH_TRACE_STAMP(var) do { } while (0)
#define YH_TRACE_NOW(var) do { } while (0)
#endif

constexpr int kRowsWaves = kRowsThreads / 64;
constexpr int kRowsTurns = 4;                 // turns of four rows whose loads are in flight together (six or eight: no faster)
constexpr int kRowsBatch = 4 * kRowsTurns;    // survivors worked on together
// Steps whose objectness logits are requested together (a round).  One: a wavefront has this step's and the
// next step's logits in flight, 128 lines.  (64 images of 10647 boxes, four steps per unit, per launch alone /
// per step on three streams: 4 -> 22.1 / 16.6 us, 3 -> 21.4 / 16.6, 2 -> 21.2 / 16.5, 1 -> 20.8 / 16.4; 22743
// boxes per image, nine steps per unit: 4 -> 43.8 / 34.9, 2 -> 41.7 / 34.8, 1 -> 42.1 / 34.8.)
constexpr int kRowsAhead = 1;
constexpr int kRowsList = 128;                // ring of queued survivors per wavefront (a step adds <= 64 to < kRowsBatch)
constexpr int kRowsX = 6;                     // class logits per lane and row held in registers: 16 * 6 = 96 classes
constexpr int kRowsPend = 64;                 // class logits past the prefilter, collected before the exact test: one per lane
constexpr int kRowsStage = 64;                // candidates (key, box) staged per wavefront before they are written out
// per wavefront: staged boxes and keys, the survivors' ring (row, objectness), the pending class logits
// (16 bytes each) and the box channels of the batch's sixteen rows
// ... and the chunk buffer of a unit that is read whole: 4 KiB, four 16-byte loads per lane in flight
constexpr uint32_t kRowsChunkSlots = 256;
constexpr int kRowsDepth = 1;                 // chunks of a unit read whole that are in flight (in registers) ahead of the one in LDS
constexpr size_t kRowsLdsWave = (size_t)kRowsStage * (16 + 8) + (size_t)kRowsList * 8 + (size_t)kRowsPend * 16 + (size_t)kRowsBatch * 16 +
                                (size_t)kRowsChunkSlots * 16;
constexpr size_t kRowsLds = (size_t)kRowsWaves * kRowsLdsWave;
static_assert(kRowsLds <= kRowsLdsMax, "the select stage riding in the launch is planned against kRowsLdsMax");
static_assert((kRowsList & (kRowsList - 1)) == 0 && kRowsList >= 64 + kRowsBatch, "ring indices are masked");
static_assert(kRowsLdsWave % 16 == 0, "every wavefront's block starts 16-byte aligned");

// ---- a unit read whole -----------------------------------------------------------------------------------
// A step most of whose rows survive the objectness prefilter is better read whole: its 64 rows are one contiguous
// span of 64 * K floats, which 16-byte loads at consecutive addresses move at streaming rate, where the row-by-row
// walk asks for the same lines in 4-byte pieces, sixteen lanes per row, after the objectness column has already
// fetched (and the cache has already dropped) a line of every row: a line per row fetched twice, 1.38 times the
// tensor at 85 floats per row.  The choice is made once per unit (k_detect_rows, from sixteen sampled rows), and a
// unit that is read whole runs here, in a function of its own that is NOT inlined: inlined, in whatever arrangement,
// it changed the register allocation of the row-by-row walk (scalar registers spilled inside its loops; in one build
// the next step's objectness logits, which are meant to stay in flight across a batch's gathers, landed in a
// register the gathers' address arithmetic reused, so every gather waited for all loads before it: twice the time
// at 50 % surviving rows).  The call costs the kernel a stack of 8 bytes per lane (the callee saves registers it
// need not, the wavefront ends after it), which is 0.3 us per launch for the scratch set-up.
// Its arguments arrive in vector registers and are uniform.
//
// A span starts at any 4-byte phase, so the loads cover the aligned window around it.  It travels in chunks of
// 4 KiB, four 16-byte loads per lane: the next two chunks are on their way into registers while a chunk is looked
// at in the wavefront's LDS buffer, and each is copied there when its turn comes (a unit is a chain of dependent round
// trips, 87 KB at 85 floats per row and four steps: with one chunk ahead a unit read whole among units walked row by
// row, which finish in half the time, was what the launch waited for) (loads into registers are waited for one by one
// where they are used; LDS-DMA loads would make the compiler wait for all of them in front of every LDS read it
// sees).  A row's objectness is its last float.  Every float of the chunk is compared with the prefilter, 16 bytes
// per lane and turn; one that passes and is a class logit looks its row's objectness up in the chunk and, if that
// passes too, joins the pending list; a full list (and the end of a step) gets the exact sigmoid product and
// threshold, one logit per lane, and the box channels of those that pass are read back from memory (rare, and the
// row's lines were fetched a moment ago).  A logit of the one row that straddles the chunk's end is queued with its
// objectness to be read back from memory as well.
typedef __attribute__((address_space(4))) const StreamParams kparams_t;    // the kernel's argument block
typedef __attribute__((address_space(4))) const StreamScale kscale_t;
// batch-slot field of a pending logit whose row's objectness is read back from memory before the exact test
constexpr uint32_t kRowsObjKnown = 0u, kRowsObjFromMemory = 1u;

__device__ __attribute__((noinline)) void rows_whole_unit(kparams_t *pv, int bv, uint32_t uiv, uint32_t ldsv)
{
    const unsigned long long pa = reinterpret_cast<unsigned long long>(pv);
    kparams_t &p = *reinterpret_cast<kparams_t *>(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) |
                                                  (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)pa));
    const int b = __builtin_amdgcn_readfirstlane(bv);
    const uint32_t ui = (uint32_t)__builtin_amdgcn_readfirstlane((int)uiv);
    unsigned char *at = (unsigned char *)(lds_float_t *)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)ldsv);
    const int lane = (int)(threadIdx.x & 63u);
    float4 *s_box = reinterpret_cast<float4 *>(at);
    at += (size_t)kRowsStage * 16;
    uint4 *s_pend = reinterpret_cast<uint4 *>(at);             // class logit, objectness logit, row, class | kRowsObj* << 16
    at += (size_t)kRowsPend * 16 + (size_t)kRowsBatch * 16;
    unsigned long long *s_keys = reinterpret_cast<unsigned long long *>(at);
    at += (size_t)kRowsStage * 8 + (size_t)kRowsList * 8;
    float *s_chunk = reinterpret_cast<float *>(at);            // the chunk that is looked at

    int s = 0;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.S && ui >= p.sc[i].unit_base) s = i;
    kscale_t &sc = p.sc[s];
    const uint32_t chunk = ui - sc.unit_base;
    const int K = p.K, C = p.C;
    const uint32_t steps = p.sc[0].tiles_per_chunk;
    const float *img = sc.src + (long long)b * sc.span;
    const uint32_t row_begin = chunk * steps * 64u;
    uint32_t row_end = row_begin + steps * 64u;               // the caller found rows in this unit
    if (row_end > sc.rows_img) row_end = sc.rows_img;
    const float pf = p.prefilter, thr = p.thr;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const long long u = (long long)b * p.upi + ui;
    WaveOut o;
    o.ukeys = p.ukeys + u * kUnitSlots;
    o.uboxes = reinterpret_cast<float4 *>(p.uboxes) + u * kUnitSlots;
    o.cand = p.cand + (long long)b * p.cand_cap;
    o.cand_box = reinterpret_cast<float4 *>(p.cand_box) + (long long)b * p.cand_cap;
    o.img_count = p.img_count + b;
    o.cand_cap = p.cand_cap;
    int count = 0, emitted = 0, npend = 0;                    // as in k_detect_rows, wave-uniform
#ifdef YH_ROWS_TRACE
    unsigned long long wt0 = 0, wt1 = 0, wta = 0, wtb = 0, wait_sum = 0, scan_sum = 0, nchunks = 0;
    YH_TRACE_NOW(wt0);
#endif

    auto settle = [&]() {
        __builtin_amdgcn_wave_barrier();
        const bool live = lane < npend;
        uint4 e = make_uint4(0u, 0u, 0u, 0u);
        if (live) e = s_pend[lane];
        if ((e.w >> 16) == kRowsObjFromMemory) e.y = __float_as_uint(img[(size_t)e.z * K + C + 4]);
        const float cf = decode_sigmoid(__uint_as_float(e.x)) * decode_sigmoid(__uint_as_float(e.y));
        const bool hit = live && cf >= thr;
        const unsigned long long hm = __ballot(hit);
        npend = 0;
        if (hm == 0ull) return;
        const int add = __builtin_popcountll(hm);
        if (count + add > kRowsStage) wave_flush(o, s_keys, s_box, count, emitted, lane);
        if (hit) {
            const float *rp = img + (size_t)e.z * K + C;
            const int pos = count + __builtin_popcountll(hm & lanes_below);
            s_keys[pos] = make_key((int)(e.w & 0xffffu), cf, (int)(sc.row_base + e.z));
            s_box[pos] = decode_box(p.P, sc, rp[0], rp[1], rp[2], rp[3], e.z);
        }
        count += add;
    };
    auto pend = [&](bool hit, int c, float x, float ob, uint32_t row, uint32_t how) {
        const unsigned long long hm = __ballot(hit);
        if (hm == 0ull) return;
        const int add = __builtin_popcountll(hm);
        if (npend + add > kRowsPend) settle();
        if (hit)
            s_pend[npend + __builtin_popcountll(hm & lanes_below)] =
                make_uint4(__float_as_uint(x), __float_as_uint(ob), row, (uint32_t)c | (how << 16));
        npend += add;
    };

    // 32-bit indices relative to the image's 16-byte slot: the image starts at any 4-byte phase of it
    const uint32_t img_phase = (uint32_t)(reinterpret_cast<uintptr_t>(img) >> 2) & 3u;
    const float4 *img4 = reinterpret_cast<const float4 *>(img - img_phase);
    // floats from the image's slot to the end of the tensor, saturated (only a window that reaches the tensor's
    // end looks at it, and such a window starts less than 2^32 floats before it)
    const unsigned long long img_floats64 = (unsigned long long)(p.B - b) * sc.span + img_phase;
    const uint32_t img_floats = img_floats64 > 0xffffffffull ? 0xffffffffu : (uint32_t)img_floats64;
    const uint32_t behind = (uint32_t)(K - C);                // columns C .. K-1 hold no class logit
    const float kNaN = __builtin_nanf("");

    for (uint32_t row0 = row_begin; row0 < row_end; row0 += 64u) {
        const uint32_t nrows = row_end - row0 < 64u ? row_end - row0 : 64u;
        const uint32_t nfl = nrows * (uint32_t)K;             // floats of the span
        const uint32_t a0 = img_phase + row0 * (uint32_t)K;   // the span's first float
        const int sh = (int)(a0 & 3u);
        const uint32_t wslots = ((uint32_t)sh + nfl + 3u) >> 2;
        const float4 *src4 = img4 + (a0 >> 2);                // first 16-byte slot of the window
        const int tail_valid = (int)(nfl + (uint32_t)sh) - 4 * (int)(wslots - 1u);   // floats of the last slot inside the span: 1..4
        int fi0 = 4 * lane - sh;                              // the lane's first float, relative to the span; -3 .. -1: before it
        int rr0 = fi0 >= 0 ? (int)__umulhi((uint32_t)fi0, p.magic_k) : -1;   // its row of the step and its column
        int col0 = fi0 >= 0 ? fi0 - rr0 * K : fi0 + K;
        const uint32_t left_f = img_floats - (a0 & ~3u);      // floats from the window's start to the tensor's end
        const uint32_t full = (left_f >> 2) < wslots ? (left_f >> 2) : wslots;   // slots to read as 16 bytes
        // the next kRowsDepth chunks travel in registers while this one is looked at in LDS
        v4f_t nx[kRowsDepth][kRowsChunkSlots / 64];
        auto request = [&](v4f_t (&dst)[kRowsChunkSlots / 64], uint32_t w0) {
#pragma unroll
            for (int j = 0; j < (int)kRowsChunkSlots / 64; ++j) {
                const uint32_t w = w0 + (uint32_t)j * 64u + (uint32_t)lane;
                dst[j] = (v4f_t){0.0f, 0.0f, 0.0f, 0.0f};
                if (w < full) dst[j] = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(src4 + w));
            }
        };
#pragma unroll
        for (int d = 0; d < kRowsDepth; ++d)
            if ((uint32_t)d * kRowsChunkSlots < wslots) request(nx[d], (uint32_t)d * kRowsChunkSlots);
        for (uint32_t wd = 0; wd < wslots; wd += kRowsDepth * kRowsChunkSlots) {
#pragma unroll
        for (int d = 0; d < kRowsDepth; ++d) {
            const uint32_t w0 = wd + (uint32_t)d * kRowsChunkSlots;
            if (w0 >= wslots) break;                          // wave-uniform
            const uint32_t cs = wslots - w0 < kRowsChunkSlots ? wslots - w0 : kRowsChunkSlots;   // slots of this chunk
            __builtin_amdgcn_wave_barrier();
#ifdef YH_ROWS_TRACE
            YH_TRACE_NOW(wta);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            YH_TRACE_NOW(wtb);
            wait_sum += wtb - wta;
            ++nchunks;
#endif
#pragma unroll
            for (int j = 0; j < (int)kRowsChunkSlots / 64; ++j)
                reinterpret_cast<v4f_t *>(s_chunk)[j * 64 + lane] = nx[d][j];
            if (w0 + kRowsDepth * kRowsChunkSlots < wslots) request(nx[d], w0 + kRowsDepth * kRowsChunkSlots);
            if (full < w0 + cs) {                             // the tensor's last, partial slot: 1..3 floats
                if ((uint32_t)lane < (left_f & 3u))
                    s_chunk[(full - w0) * 4u + (uint32_t)lane] = reinterpret_cast<const float *>(src4 + full)[lane];
            }
            __builtin_amdgcn_wave_barrier();
            const int cbase = 4 * (int)w0 - sh;               // index, relative to the span, of the buffer's first float
            int r_hi = (int)__umulhi((uint32_t)(cbase + 4 * (int)cs), p.magic_k);   // rows that end before the chunk does
            r_hi = r_hi < (int)nrows ? r_hi : (int)nrows;
            const int nu = (int)((cs + 63u) >> 6);
#ifdef YH_ROWS_TRACE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            YH_TRACE_NOW(wta);
#endif
            // First the whole chunk at once: is there a float anywhere in it that reaches the prefilter in a column
            // that may hold a class logit?  Four independent 16-byte reads and comparison chains per lane (a
            // wavefront on its own SIMD advances at the latency of dependent instructions, not at their issue
            // rate: one slot per turn took 420 cycles a turn).  Stale floats behind a short chunk and the floats
            // around the span can only make this answer yes.  Usually it is no, and the chunk is done.
            {
                float mx = kNaN;
                int cc = col0;
#pragma unroll
                for (int t = 0; t < (int)kRowsChunkSlots / 64; ++t) {
                    const float4 x = reinterpret_cast<const float4 *>(s_chunk)[t * 64 + lane];
                    const uint32_t t0 = (uint32_t)(cc - C);
                    const float m0 = t0 >= behind ? x.x : kNaN, m1 = t0 + 1u >= behind ? x.y : kNaN;
                    const float m2 = t0 + 2u >= behind ? x.z : kNaN, m3 = t0 + 3u >= behind ? x.w : kNaN;
                    mx = fmaxf(mx, fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)));
                    cc += p.dense_col_step;
                    cc = cc >= K ? cc - K : cc;
                }
                if (__ballot(mx >= pf) == 0ull) {
                    for (int t = 0; t < nu; ++t) {            // the lane's place after the chunk
                        rr0 += p.dense_row_step;
                        col0 += p.dense_col_step;
                        if (col0 >= K) { col0 -= K; ++rr0; }
                    }
#ifdef YH_ROWS_TRACE
                    YH_TRACE_NOW(wtb);
                    scan_sum += wtb - wta;
#endif
                    continue;
                }
            }
#pragma unroll 1
            for (int t = 0; t < nu; ++t) {
                const uint32_t wl = (uint32_t)t * 64u + (uint32_t)lane;     // the lane's slot of the chunk
                float4 x = reinterpret_cast<const float4 *>(s_chunk)[wl];
                // Conservative and cheap, because it runs for every 16 bytes of the tensor: a float counts when its
                // column (that of the lane's first float plus its place) is not one of the K - C columns behind the
                // class logits (a column that wrapped into the next row passes too; the exact row and column follow
                // for the few floats that get further), anything else is replaced by NaN (which no comparison passes, whatever the prefilter, and
                // which the maximum ignores); when no lane's largest
                // float reaches the prefilter, which is the usual case, the turn ends here.
                const uint32_t t0 = (uint32_t)(col0 - C);
                x.x = t0 >= behind ? x.x : kNaN;
                x.y = t0 + 1u >= behind ? x.y : kNaN;
                x.z = t0 + 2u >= behind ? x.z : kNaN;
                x.w = t0 + 3u >= behind ? x.w : kNaN;
                if ((w0 == 0u && t == 0) || (t == nu - 1 && w0 + cs == wslots)) {   // wave-uniform: the window's first and last slots
                    uint32_t ok = wl < cs ? 0xfu : 0u;        // floats of the window outside the span do not count
                    ok = w0 + wl == wslots - 1u ? (1u << tail_valid) - 1u : ok;
                    ok = w0 + wl == 0u ? ok & (0xfu << sh) : ok;
                    x.x = (ok & 1u) ? x.x : kNaN;
                    x.y = (ok & 2u) ? x.y : kNaN;
                    x.z = (ok & 4u) ? x.z : kNaN;
                    x.w = (ok & 8u) ? x.w : kNaN;
                }
                uint32_t bits = 0u;
                if (__ballot(fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w)) >= pf) != 0ull)
                    bits = (x.x >= pf ? 1u : 0u) | (x.y >= pf ? 2u : 0u) | (x.z >= pf ? 4u : 0u) | (x.w >= pf ? 8u : 0u);
                while (__ballot(bits != 0u) != 0ull) {
                    const int i = bits != 0u ? __builtin_ctz(bits) : 0;
                    const float xi = i == 0 ? x.x : i == 1 ? x.y : i == 2 ? x.z : x.w;
                    int rr = rr0, c = col0 + i;
                    if (c >= K) { c -= K; ++rr; }
                    bool hit = bits != 0u && c < C;
                    float obr = 0.0f;
                    uint32_t how = kRowsObjKnown;
                    if (hit) {
                        if (rr < r_hi) {
                            obr = s_chunk[rr * K + K - 1 - cbase];
                            hit = obr >= pf;
                        } else {
                            how = kRowsObjFromMemory;
                        }
                    }
                    pend(hit, c, xi, obr, row0 + (uint32_t)rr, how);
                    bits &= bits - 1u;
                }
                rr0 += p.dense_row_step;
                col0 += p.dense_col_step;
                if (col0 >= K) { col0 -= K; ++rr0; }
            }
#ifdef YH_ROWS_TRACE
            YH_TRACE_NOW(wtb);
            scan_sum += wtb - wta;
#endif
        }
        }
        if (npend > 0) settle();
    }
    if (count > 0) wave_flush(o, s_keys, s_box, count, emitted, lane);
    if (lane >= emitted && lane < kUnitSlots) o.ukeys[lane] = kEmptySlot;        // the unit's unused slots
#ifdef YH_ROWS_TRACE
    YH_TRACE_STAMP(wt1);
    if (lane == 0 && u < (long long)kTraceWaves) {            // measurement build: this unit's record, by unit number
        unsigned long long *tr = g_rows_trace + u * kTraceWords;
        tr[0] = wt0; tr[1] = wt1; tr[2] = wait_sum; tr[3] = scan_sum; tr[4] = nchunks; tr[5] = 0; tr[6] = 0; tr[7] = ~0ull;
    }
#endif
}

__global__ __launch_bounds__(kRowsThreads) void k_detect_rows(const StreamParams p, const NmsParams q)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    if (blockIdx.x < (uint32_t)q.B) {
        post_image_dyn<kRowsThreads>(q, (int)blockIdx.x, s_dyn);
        return;
    }
    // the wavefront's number as a scalar: the unit, its scale and every address derived from them then live in
    // scalar registers, and the scale's parameters are scalar loads of the kernel arguments (a per-lane
    // number makes them a vector load from memory, one more round trip before the first objectness read)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char *at = s_dyn + (size_t)wave * kRowsLdsWave;
    float4 *s_box = reinterpret_cast<float4 *>(at);
    at += (size_t)kRowsStage * 16;
    uint4 *s_pend = reinterpret_cast<uint4 *>(at);             // class logit, objectness logit, row, class | batch slot << 16
    at += (size_t)kRowsPend * 16;
    float4 *s_tb = reinterpret_cast<float4 *>(at);             // t_y, t_x, t_h, t_w of the batch's rows, by batch slot
    at += (size_t)kRowsBatch * 16;
    unsigned long long *s_keys = reinterpret_cast<unsigned long long *>(at);
    at += (size_t)kRowsStage * 8;
    uint32_t *s_lrow = reinterpret_cast<uint32_t *>(at);       // row inside the scale
    float *s_lobj = reinterpret_cast<float *>(at + (size_t)kRowsList * 4);

    const uint32_t g = blockIdx.x - (uint32_t)q.B;
    const int b = (int)(g / p.units);                         // p.units: workgroups per image
    const uint32_t ui = (g - (uint32_t)b * p.units) * kRowsWaves + (uint32_t)wave;   // unit inside the image (< p.upi)
    int s = 0;
#pragma unroll
    for (int i = 1; i < YH_MAX_SCALES; ++i)
        if (i < p.S && ui >= p.sc[i].unit_base) s = i;
    const StreamScale &sc = p.sc[s];
    const uint32_t chunk = ui - sc.unit_base;
    const bool has_rows = chunk < sc.chunks_per_img;          // the image's last workgroup may hold spare units
    const int K = p.K, C = p.C;
    const uint32_t steps = p.sc[0].tiles_per_chunk;           // the same for every scale
    const float *img = sc.src + (long long)b * sc.span;
    const uint32_t row_begin = chunk * steps * 64u;
    uint32_t row_end = has_rows ? row_begin + steps * 64u : row_begin;
    if (row_end > sc.rows_img) row_end = sc.rows_img;
    const uint32_t nsteps = row_begin < row_end ? (row_end - row_begin + 63u) / 64u : 0u;
    const float pf = p.prefilter, thr = p.thr;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const long long u = (long long)b * p.upi + ui;
    WaveOut o;
    o.ukeys = p.ukeys + u * kUnitSlots;
    o.uboxes = reinterpret_cast<float4 *>(p.uboxes) + u * kUnitSlots;
    o.cand = p.cand + (long long)b * p.cand_cap;
    o.cand_box = reinterpret_cast<float4 *>(p.cand_box) + (long long)b * p.cand_cap;
    o.img_count = p.img_count + b;
    o.cand_cap = p.cand_cap;
    int count = 0;                                            // staged candidates, wave-uniform
    int emitted = 0;                                          // candidates of this unit already written out
    int npend = 0;                                            // pending class logits, wave-uniform
    int head = 0, tail = 0;                                   // the survivors' ring: entries [head, tail), indices masked
    unsigned long long trB = 0, trC = 0;                      // measurement build: the last batch's data arrived / worked on
    (void)trB; (void)trC;
    const int grp = lane >> 4, j = lane & 15;                 // sixteen lanes per row

    // rows outside the unit read as NaN: no comparison with the prefilter (-inf for thr <= 0) holds for them
    const float kNoRow = __builtin_nanf("");
    auto objectness = [&](uint32_t row) { return row < row_end ? img[(size_t)row * K + C + 4] : kNoRow; };

    // The exact test of the pending class logits, one per lane: sigmoid product against the threshold, and
    // for those that pass the box of their row and the key, staged for the write-out.  A wavefront that
    // found a few logits past the prefilter in a batch of rows pays this once, not once per row.
    auto settle = [&]() {
        __builtin_amdgcn_wave_barrier();
        const bool live = lane < npend;
        uint4 e = make_uint4(0u, 0u, 0u, 0u);
        if (live) e = s_pend[lane];
        const float cf = decode_sigmoid(__uint_as_float(e.x)) * decode_sigmoid(__uint_as_float(e.y));
        const bool hit = live && cf >= thr;
        const unsigned long long hm = __ballot(hit);
        npend = 0;
        if (hm == 0ull) return;
        const int add = __builtin_popcountll(hm);
        if (count + add > kRowsStage) wave_flush(o, s_keys, s_box, count, emitted, lane);
        if (hit) {
            const float4 tb = s_tb[e.w >> 16];
            const int pos = count + __builtin_popcountll(hm & lanes_below);
            s_keys[pos] = make_key((int)(e.w & 0xffffu), cf, (int)(sc.row_base + e.z));
            s_box[pos] = decode_box(p.P, sc, tb.x, tb.y, tb.z, tb.w, e.z);
        }
        count += add;
    };
    // Queues the lanes' class logits that passed the prefilter (hit: this lane has one, x of class c, in
    // the row of batch slot `slot` with objectness logit ob).
    auto pend = [&](bool hit, int c, float x, float ob, uint32_t row, int slot) {
        const unsigned long long hm = __ballot(hit);
        if (hm == 0ull) return;
        const int add = __builtin_popcountll(hm);
        if (npend + add > kRowsPend) settle();
        if (hit)
            s_pend[npend + __builtin_popcountll(hm & lanes_below)] =
                make_uint4(__float_as_uint(x), __float_as_uint(ob), row, (uint32_t)c | ((uint32_t)slot << 16));
        npend += add;
    };

    // Works on the first n (1 .. kRowsBatch) queued survivors.  Every row read here was queued by this
    // wavefront from a row < row_end of its own image and scale.
    auto work = [&](int n) {
        float xv[kRowsTurns][kRowsX], tb[kRowsTurns], ob[kRowsTurns];
        uint32_t rw[kRowsTurns];
        bool val[kRowsTurns];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < kRowsTurns; ++t) {
            val[t] = 4 * t + grp < n;
            rw[t] = 0u; ob[t] = 0.0f; tb[t] = 0.0f;
#pragma unroll
            for (int i = 0; i < kRowsX; ++i) xv[t][i] = 0.0f;
            if (4 * t < n) {                                  // wave-uniform: the turn holds at least one row
                const int idx = (head + (val[t] ? 4 * t + grp : 0)) & (kRowsList - 1);   // spare groups re-read entry `head`
                rw[t] = s_lrow[idx];
                ob[t] = s_lobj[idx];
                const float *rp = img + (size_t)rw[t] * K;
#pragma unroll
                for (int i = 0; i < kRowsX; ++i) {
                    const int c = j + 16 * i;
                    if (c < C) xv[t][i] = rp[c];
                }
                tb[t] = rp[C + (j & 3)];                       // t_y, t_x, t_h, t_w in lanes 0-3 of the group
            }
        }
        YH_TRACE_STAMP(trB);
#pragma unroll
        for (int t = 0; t < kRowsTurns; ++t) {
            if (4 * t >= n) break;
            const int slot = 4 * t + grp;
            if (val[t] && j < 4) reinterpret_cast<float *>(s_tb + slot)[j] = tb[t];
            // a bit per class held in registers whose logit can still reach the threshold
            uint32_t bits = 0u;
#pragma unroll
            for (int i = 0; i < kRowsX; ++i)
                if (val[t] && j + 16 * i < C && xv[t][i] >= pf) bits |= 1u << i;
            while (__ballot(bits != 0u) != 0ull) {            // a lane's logits one at a time: mostly one turn of this loop
                const bool hit = bits != 0u;
                const int i = hit ? __builtin_ctz(bits) : 0;
                float x = xv[t][0];
#pragma unroll
                for (int k = 1; k < kRowsX; ++k) x = i == k ? xv[t][k] : x;
                pend(hit, j + 16 * i, x, ob[t], rw[t], slot);
                bits &= bits - 1u;
            }
            if (16 * kRowsX < C) {                            // more classes than the registers hold
                const float *rp = img + (size_t)rw[t] * K;
                for (int c0 = 16 * kRowsX; c0 < C; c0 += 16) {
                    const int c = c0 + j;
                    const float x = val[t] && c < C ? rp[c] : kNoRow;
                    pend(x >= pf, c, x, ob[t], rw[t], slot);
                }
            }
        }
        if (npend > 0) settle();                              // s_tb belongs to this batch
        head += n;
        YH_TRACE_NOW(trC);
    };

    // The unit's steps, in one of two ways, chosen once per unit (a unit that mixed them would keep the state of
    // both alive across each other's code, and the row-by-row walk, the common case, would pay for it in registers).
    // The unit first looks at the objectness of sixteen of its rows, spread evenly over its steps: sixteen lines.
    // When p.dense_min in 64 of them or more pass, every step of the unit is read whole, with no objectness column
    // requested at all (the column read of a step that is then read whole is wasted traffic: a line per row fetched
    // twice, 1.38 times the tensor at 85 floats per row, because the lines do not survive in the cache until the
    // rows are read).
    // Otherwise (what a detector's head looks like) the walk is row by row: the objectness logits of step t + 1 are
    // requested before those of step t are looked at; rows that pass are queued, and as soon as a whole batch is
    // queued it is worked on (with one step requested ahead little stands between a batch's gathers and the memory
    // system; waiting until the ring could not take another step was 2 % slower from 16 % surviving rows up and no
    // faster below).  One more pass of the loop, after the last step, works the ring empty: written this way the
    // batch code exists once in the kernel.
    static_assert(kRowsAhead == 1, "one step's objectness logits are requested ahead");
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0, tr4 = 0;
    (void)tr0; (void)tr1; (void)tr2; (void)tr3; (void)tr4;
    int trace_left = 0;
    (void)trace_left;
    YH_TRACE_NOW(tr0);
    bool dense = false;                                       // wave-uniform
    if (nsteps > 0u && p.dense_min <= 64) {
        if (p.dense_probe == 1) {
            const uint32_t prow = row_begin + (uint32_t)lane * (nsteps * 4u);   // every (4 * steps)-th row: 16 lanes span the unit
            const float po = lane < 16 ? objectness(prow) : kNoRow;
            const int have = __builtin_popcountll(__ballot(lane < 16 && prow < row_end));
            const int pass = __builtin_popcountll(__ballot(po >= pf));
            dense = pass * 64 >= p.dense_min * have;
        }
    }
    if (dense) {
        // out of line, with nothing of this function alive across the call: the walk below keeps its registers
        rows_whole_unit((kparams_t *)__builtin_amdgcn_kernarg_segment_ptr(), b, ui,
                        (uint32_t)(uintptr_t)(lds_float_t *)(void *)(s_dyn + (size_t)wave * kRowsLdsWave));
        return;
    } else {
        float obj_next = nsteps > 0u ? objectness(row_begin + (uint32_t)lane) : kNoRow;
        for (uint32_t t = 0;; ++t) {
            const bool last = t >= nsteps;                    // wave-uniform
            if (!last) {
                const uint32_t row0 = row_begin + t * 64u;
                const float ob = obj_next;
                obj_next = t + 1u < nsteps ? objectness(row0 + 64u + (uint32_t)lane) : kNoRow;
                const unsigned long long mask = __ballot(ob >= pf);
#ifdef YH_ROWS_TRACE
                if (t == 0) { asm volatile("" ::"s"(mask)); YH_TRACE_NOW(tr1); }
#endif
                if (mask) {
                    if ((mask >> lane) & 1ull) {
                        const int e = (tail + __builtin_popcountll(mask & lanes_below)) & (kRowsList - 1);
                        s_lrow[e] = row0 + (uint32_t)lane;
                        s_lobj[e] = ob;
                    }
                    tail += __builtin_popcountll(mask);
                }
            } else {
                YH_TRACE_STAMP(tr2);
                trace_left = tail - head;
            }
            const int keep = last ? 0 : kRowsBatch - 1;
            while (tail - head > keep) work(tail - head < kRowsBatch ? tail - head : kRowsBatch);
            if (last) break;
        }
    }
    YH_TRACE_STAMP(tr3);
    if (count > 0) wave_flush(o, s_keys, s_box, count, emitted, lane);
    if (lane >= emitted && lane < kUnitSlots) o.ukeys[lane] = kEmptySlot;        // the unit's unused slots
#ifdef YH_ROWS_TRACE
    YH_TRACE_STAMP(tr4);
    const unsigned long long wid = (unsigned long long)g * kRowsWaves + wave;
    if (lane == 0 && wid < (unsigned long long)kTraceWaves) {
        unsigned long long *tr = g_rows_trace + wid * kTraceWords;
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        tr[0] = tr0; tr[1] = tr1; tr[2] = tr2; tr[3] = tr3; tr[4] = tr4; tr[5] = hwid;
        tr[6] = ((unsigned long long)(uint32_t)tail << 32) | (uint32_t)trace_left; tr[7] = trB; tr[5] = trC;
    }
#endif
}

#ifdef YH_ROWS_TRACE
extern "C" int yh_trace_clear(void)
{
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_rows_trace));
    if (e == hipSuccess) e = hipMemset(ptr, 0, sizeof(unsigned long long) * kTraceWaves * kTraceWords);
    return (int)e;
}

extern "C" int yh_trace_read(void *dst, size_t bytes)
{
    const size_t have = sizeof(unsigned long long) * kTraceWaves * kTraceWords;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rows_trace), bytes < have ? bytes : have, 0, hipMemcpyDeviceToHost);
}
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) and size: the
// largest value asked for so far is remembered, a launch within it costs no runtime call.
hipError_t raise_lds_limit(const void *kernel, size_t bytes)
{
    constexpr int kSlots = 32, kDevices = 64;
    static std::mutex mu;
    static const void *fn[kSlots];
    static size_t limit[kSlots][kDevices];
    if (bytes <= 48 * 1024) return hipSuccess;                // the default limit of dynamic LDS
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    int at = -1;
    for (int i = 0; i < kSlots && at < 0; ++i)
        if (fn[i] == kernel || fn[i] == nullptr) at = i;
    if (at < 0) return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    fn[at] = kernel;
    if (limit[at][dev] >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) limit[at][dev] = bytes;
    return e;
}

static hipError_t launch_rows(const StreamParams &p, uint32_t blocks, hipStream_t stm, const NmsParams &q)
{
    size_t lds = kRowsLds;
    if (q.B > 0) {
        const size_t need = post_layout(q.cand_cap, q.C, q.max_boxes, kRowsThreads).bytes;
        lds = need > lds ? need : lds;
    }
    const hipError_t le = raise_lds_limit(reinterpret_cast<const void *>(k_detect_rows), lds);
    if (le != hipSuccess) return le;
    hipLaunchKernelGGL(k_detect_rows, dim3(blocks + (uint32_t)q.B), dim3(kRowsThreads), lds, stm, p, q);
    return hipGetLastError();
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
        if (p.conf) p.conf[n * C + k] = decode_sigmoid(img[l]) * decode_sigmoid(img[(size_t)row * K + C + 4]);
    } else if (k == C) {
        reinterpret_cast<float4 *>(p.boxes)[n] = decode_box(p.P, sc, img[l], img[l + 1], img[l + 2], img[l + 3], row);
    }
}

template <int NI, int NBUF>
static hipError_t launch_wave(const StreamParams &p, uint32_t blocks, hipStream_t stm, const NmsParams &q)
{
    size_t lds = wave_lds_bytes(NI, NBUF);
    if (q.B > 0) {
        const size_t need = post_layout(q.cand_cap, q.C, q.max_boxes, 64).bytes;
        lds = need > lds ? need : lds;
    }
    const hipError_t le = raise_lds_limit(reinterpret_cast<const void *>(k_stream_wave<NI, NBUF>), lds);
    if (le != hipSuccess) return le;
    hipLaunchKernelGGL((k_stream_wave<NI, NBUF>), dim3(blocks + (uint32_t)q.B), dim3(64), lds, stm, p, q);
    return hipGetLastError();
}

// Plain streaming read of a buffer (yh_probe_read): what the memory system delivers to a
// kernel that only reads, the yardstick next to the nominal HBM peak.  Grid-stride, eight
// 16-byte loads per lane in flight.
__global__ __launch_bounds__(kThreads) void k_probe_read(const float4 *src, long long nslots, float *sink)
{
    float acc = 0.f;
    const long long stride = (long long)gridDim.x * kThreads * 8;
    for (long long base = (long long)blockIdx.x * kThreads * 8; base < nslots; base += stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            long long i = base + u * kThreads + threadIdx.x;
            i = i < nslots ? i : nslots - 1;
            v[u] = src[i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w));
    }
    if (acc == 12345.678f && sink) sink[0] = acc;     // keeps the loads alive; never true for finite data in practice
}

hipError_t launch_probe_read(const void *src, size_t bytes, hipStream_t stm)
{
    const long long nslots = (long long)(bytes / 16);
    if (nslots == 0) return hipSuccess;
    hipLaunchKernelGGL(k_probe_read, dim3(1024), dim3(kThreads), 0, stm, static_cast<const float4 *>(src), nslots,
                       static_cast<float *>(nullptr));
    return hipGetLastError();
}

hipError_t launch_stream(const StreamParams &p, StreamMode mode, uint32_t blocks, hipStream_t stm, const NmsParams *fused)
{
    if (fused && !(mode == kModeDetect && (p.wave_ni || p.row_gather))) return hipErrorInvalidValue;
    if (mode == kModeFull) {
        for (int s = 0; s < p.S; ++s) {
            const dim3 grid((p.sc[s].span + kThreads - 1) / kThreads, (uint32_t)p.B);
            hipLaunchKernelGGL(k_decode_full, grid, dim3(kThreads), 0, stm, p, s);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    if (mode == kModeDetect && p.row_gather) {
        NmsParams none;
        std::memset(&none, 0, sizeof(none));
        const NmsParams &q = fused ? *fused : none;
        if (blocks == 0 && q.B == 0) return hipSuccess;
        return launch_rows(p, blocks, stm, q);
    }
    if (mode == kModeDetect && p.wave_ni) {
        NmsParams none;
        std::memset(&none, 0, sizeof(none));
        const NmsParams &q = fused ? *fused : none;
        if (blocks == 0 && q.B == 0) return hipSuccess;
        switch (p.wave_ni * 10 + p.wave_nbuf) {
        case 42: return launch_wave<4, 2>(p, blocks, stm, q);
        case 43: return launch_wave<4, 3>(p, blocks, stm, q);
        case 44: return launch_wave<4, 4>(p, blocks, stm, q);
        case 82: return launch_wave<8, 2>(p, blocks, stm, q);
        case 83: return launch_wave<8, 3>(p, blocks, stm, q);
        default: return hipErrorInvalidValue;
        }
    }
    if (blocks == 0) return hipSuccess;
    if (p.cuts != blocks || p.grid_blocks == 0 || p.grid_blocks > blocks) return hipErrorInvalidValue;
    const dim3 grid(p.grid_blocks), block(kTileThreads);
    const size_t hist_lds = p.class_hist ? (size_t)p.C * sizeof(int) : 0;
    if (mode == kModeDetect) hipLaunchKernelGGL(k_stream<kModeDetect>, grid, block, hist_lds, stm, p);
    else hipLaunchKernelGGL(k_stream<kModeConf>, grid, block, hist_lds, stm, p);
    return hipGetLastError();
}

}  // namespace yh
