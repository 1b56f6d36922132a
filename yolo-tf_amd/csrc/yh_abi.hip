// extern "C" boundary of libyolohead: argument checks, workspace carving and
// kernel sequencing.  Declarations and the reference regions each entry point
// replaces: include/yolohead.h.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include "yh_internal.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    return fail(YH_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int check_config(const yh_head_config *cfg)
{
    if (!cfg) return fail(YH_ERR_INVALID_ARG, "cfg is null");
    if (cfg->num_scales < 1 || cfg->num_scales > YH_MAX_SCALES)
        return fail(YH_ERR_LIMIT, "num_scales %d outside [1, %d]", cfg->num_scales, YH_MAX_SCALES);
    if (cfg->num_priors < 1 || cfg->num_priors > YH_MAX_PRIORS)
        return fail(YH_ERR_LIMIT, "num_priors %d outside [1, %d]", cfg->num_priors, YH_MAX_PRIORS);
    if (cfg->num_classes < 1 || cfg->num_classes > YH_MAX_CLASSES)
        return fail(YH_ERR_LIMIT, "num_classes %d outside [1, %d]", cfg->num_classes, YH_MAX_CLASSES);
    int64_t n = 0;
    const int64_t K = cfg->num_classes + 5;
    for (int s = 0; s < cfg->num_scales; ++s) {
        if (cfg->grid_h[s] < 1 || cfg->grid_w[s] < 1)
            return fail(YH_ERR_INVALID_ARG, "grid %d is %d x %d", s, cfg->grid_h[s], cfg->grid_w[s]);
        const int64_t rows = (int64_t)cfg->grid_h[s] * cfg->grid_w[s] * cfg->num_priors;
        // the flat-index -> (row, channel) split uses a 32-bit multiply-high
        if ((rows * K + K) * K >= (int64_t)1 << 32)
            return fail(YH_ERR_LIMIT, "scale %d: rows * (C+5)^2 = %lld exceeds 2^32", s, (long long)(rows * K * K));
        n += rows;
    }
    if (n > YH_MAX_BOXES_PER_IMAGE)
        return fail(YH_ERR_LIMIT, "%lld boxes per image exceed %d", (long long)n, YH_MAX_BOXES_PER_IMAGE);
    return YH_OK;
}

int64_t boxes_per_image(const yh_head_config *cfg)
{
    int64_t n = 0;
    for (int s = 0; s < cfg->num_scales; ++s)
        n += (int64_t)cfg->grid_h[s] * cfg->grid_w[s] * cfg->num_priors;
    return n;
}

int check_heads(const yh_head_config *cfg, const float *const *heads)
{
    if (!heads) return fail(YH_ERR_INVALID_ARG, "heads is null");
    for (int s = 0; s < cfg->num_scales; ++s) {
        if (!heads[s]) return fail(YH_ERR_INVALID_ARG, "heads[%d] is null", s);
        if (reinterpret_cast<uintptr_t>(heads[s]) & 15u)
            return fail(YH_ERR_ALIGN, "heads[%d] is not 16-byte aligned", s);
    }
    return YH_OK;
}

// Lower bound of the class logits that can still pass conf >= thr: conf <=
// sigmoid(x) in fp32, so x must reach logit(thr) up to the error of the device
// sigmoid; the margin below is orders of magnitude wider than that error.
float prefilter_logit(float thr)
{
    if (!(thr > 0.0f)) return -std::numeric_limits<float>::infinity();   // thr <= 0 (or NaN): every finite logit goes to the exact test
    if (thr > 1.0f) return std::numeric_limits<float>::infinity();
    double t = (double)thr;
    const double tmax = 1.0 - std::ldexp(1.0, -25);
    if (t > tmax) t = tmax;
    const double l = std::log(t / (1.0 - t));
    return (float)(l - 1e-3 * (1.0 + std::fabs(l)));
}

struct DetectWs {
    unsigned long long *cand, *seg_keys, *ukeys;
    float *boxes, *cand_box, *uboxes;
    int32_t *img_count, *seg_off, *seg_cnt, *class_hist, *class_cursor, *kept_cnt, *kept_row;
    float *kept_score, *kept_box;
    size_t zero_begin, zero_bytes;
    size_t bytes;
};

// row_boxes: the workspace holds the decoded boxes by box row [B][N][4] (the workgroup-tile
// streaming kernel writes them, the multi-launch select stage reads them); cand_boxes: it holds
// the list entries' boxes [B][cap][4]; upi: work units per image with private slots.
// [zero_begin, zero_begin + zero_bytes): the counters a call expects to find zero (a zero-filled
// workspace has them so, and every call leaves them so).
DetectWs carve_detect(void *base, int64_t B, int64_t N, int64_t C, int64_t cap, int64_t mb, bool row_boxes,
                      bool cand_boxes, int64_t upi)
{
    DetectWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    // the base may be null (a size query): offsets are added to its integer value, not to the pointer
    const uintptr_t p = reinterpret_cast<uintptr_t>(base);
    const size_t o_cand = take((size_t)B * cap * 8), o_seg = take((size_t)B * cap * 8);
    const size_t o_box = row_boxes ? take((size_t)B * N * 16) : 0;
    const size_t o_cbox = cand_boxes ? take((size_t)B * cap * 16) : 0;
    const size_t o_uk = take((size_t)B * upi * yh::kUnitSlots * 8);
    const size_t o_ub = take((size_t)B * upi * yh::kUnitSlots * 16);
    w.cand_box = cand_boxes ? reinterpret_cast<float *>(p + o_cbox) : nullptr;
    w.ukeys = reinterpret_cast<unsigned long long *>(p + o_uk);
    w.uboxes = reinterpret_cast<float *>(p + o_ub);
    const size_t o_off = take((size_t)B * C * 4), o_cnt = take((size_t)B * C * 4);
    const size_t o_img = take((size_t)B * 4);
    const size_t o_hist = take((size_t)B * C * 4), o_cur = take((size_t)B * C * 4);
    const size_t o_kc = take((size_t)B * C * 4), o_kr = take((size_t)B * C * mb * 4), o_ks = take((size_t)B * C * mb * 4);
    // the selected boxes: only the multi-launch select stage passes them from one launch to the next
    const bool multi = !yh::select_single_launch((int)cap, (int)C, (int)mb);
    const size_t o_kb = multi ? take((size_t)B * C * mb * 16) : 0;
    w.zero_begin = o_img;
    w.zero_bytes = o_kc - o_img;                 // img_count, class_hist, class_cursor
    w.cand = reinterpret_cast<unsigned long long *>(p + o_cand);
    w.seg_keys = reinterpret_cast<unsigned long long *>(p + o_seg);
    w.boxes = row_boxes ? reinterpret_cast<float *>(p + o_box) : nullptr;
    w.img_count = reinterpret_cast<int32_t *>(p + o_img);
    w.seg_off = reinterpret_cast<int32_t *>(p + o_off);
    w.seg_cnt = reinterpret_cast<int32_t *>(p + o_cnt);
    w.class_hist = reinterpret_cast<int32_t *>(p + o_hist);
    w.class_cursor = reinterpret_cast<int32_t *>(p + o_cur);
    w.kept_cnt = reinterpret_cast<int32_t *>(p + o_kc);
    w.kept_row = reinterpret_cast<int32_t *>(p + o_kr);
    w.kept_score = reinterpret_cast<float *>(p + o_ks);
    w.kept_box = multi ? reinterpret_cast<float *>(p + o_kb) : nullptr;
    w.bytes = off;
    return w;
}

// yh_detect's workspace for a streaming plan.  The boxes-by-row array, the largest part by far, is only needed
// between a slot-writing kernel and the multi-launch select stage (which looks boxes up by row: k_bin copies
// the slots' boxes there); the workgroup-tile kernel leaves no boxes at all, the stages behind it decode the
// box of a row they look at from the head tensors.
DetectWs carve_for_plan(void *base, const yh::StreamParams &p, int64_t B, int64_t cap, int64_t mb)
{
    const bool slots = p.wave_ni || p.row_gather;
    const bool row_boxes = slots && !yh::select_single_launch((int)cap, p.C, (int)mb);
    return carve_detect(base, B, p.N, p.C, cap, mb, row_boxes, slots, p.upi);
}

// The workgroup-tile streaming kernel leaves no boxes: the select stage decodes the boxes it looks at from the
// head tensors.  When the two stages are separate calls (yh_detect_candidates / yh_detect_select, or a batch's
// select stage riding in the next call of yh_detect_pipelined) the select stage has no head argument of its
// own: the library remembers, per workspace, the head tensors of the last streaming call on it.  A small
// table; when it is full the entry whose streaming call is the longest ago goes (a workspace's record is
// lost only once kWorkspaceHeads OTHER workspaces have had a streaming call since its own).
struct WorkspaceHeads { const void *ws; const float *head[YH_MAX_SCALES]; unsigned long long stamp; };
constexpr int kWorkspaceHeads = 64;
std::mutex g_ws_heads_mu;
WorkspaceHeads g_ws_heads[kWorkspaceHeads];
unsigned long long g_ws_heads_clock = 0;

void remember_heads(const void *ws, const float *const *heads, int S)
{
    std::lock_guard<std::mutex> lock(g_ws_heads_mu);
    int at = -1, oldest = 0;
    for (int i = 0; i < kWorkspaceHeads && at < 0; ++i) {
        if (g_ws_heads[i].ws == ws) at = i;
        if (g_ws_heads[i].stamp < g_ws_heads[oldest].stamp) oldest = i;      // never used: stamp 0
    }
    if (at < 0) at = oldest;
    g_ws_heads[at].ws = ws;
    g_ws_heads[at].stamp = ++g_ws_heads_clock;
    for (int s = 0; s < YH_MAX_SCALES; ++s) g_ws_heads[at].head[s] = s < S ? heads[s] : nullptr;
}

bool recall_heads(const void *ws, const float **heads)
{
    std::lock_guard<std::mutex> lock(g_ws_heads_mu);
    for (int i = 0; i < kWorkspaceHeads; ++i)
        if (g_ws_heads[i].ws == ws) {
            for (int s = 0; s < YH_MAX_SCALES; ++s) heads[s] = g_ws_heads[i].head[s];
            return true;
        }
    return false;
}

int check_nms_args(int32_t batch, float iou_thr, int32_t max_boxes, int32_t max_cand, const void *ws,
                   const void *oc, const void *os, const void *ob, const void *ocl, const void *oi, const void *status)
{
    if (batch < 0) return fail(YH_ERR_INVALID_ARG, "batch %d is negative", batch);
    // tf.image.non_max_suppression requires iou_threshold in [0, 1]
    if (!(iou_thr >= 0.0f && iou_thr <= 1.0f)) return fail(YH_ERR_INVALID_ARG, "iou_threshold %g outside [0, 1]", iou_thr);
    if (max_boxes < 1 || max_boxes > YH_MAX_NMS_BOXES)
        return fail(YH_ERR_LIMIT, "nms_max_boxes %d outside [1, %d]", max_boxes, YH_MAX_NMS_BOXES);
    if (max_cand < 1) return fail(YH_ERR_INVALID_ARG, "max_candidates %d < 1", max_cand);
    if (batch > 0 && (!ws || !oc || !os || !ob || !ocl || !oi || !status))
        return fail(YH_ERR_INVALID_ARG, "null workspace, output or status pointer");
    return YH_OK;
}

// Fills the per-scale block decomposition of the streaming pass.  A workgroup
// walks tiles of whole rows (<= 16 KiB each) and ends with a latency-bound tail
// (the atomic that reserves its candidate range), so the launch is
// oversubscribed: about kStreamTargetBlocks workgroups, a few per resident slot.
constexpr uint64_t kStreamTargetBlocks = 3600;
// Few tiles per workgroup: what a workgroup stages then leaves in one flush at its end (a flush is a
// round trip to the image's candidate counter with the whole workgroup waiting), and the launch ends
// without a long tail.  128 images of 100x100x10 boxes, 80 classes, one candidate per box, 16 KiB tiles,
// pass on its own / step with the select stages of other batches beside it: 3 tiles per workgroup
// 1108 / 1257 us, 4: 973 / 1247, 5: 921 / 1251, 6: 916 / 1184, 7: 923 / 1224, 8: 940 / 1243, 10: 981 / 1229,
// 12: 946 / 1236, 16: 957 / 1259.
constexpr uint64_t kStreamTilesPerBlock = 6;
// The confidence matrix of yh_nms (no decode, rows of C floats, its launches on one stream): 8 tiles per workgroup
// 786 us for 128 images of 1e5 x 80 confidences, 6 tiles 810 us.
constexpr uint64_t kStreamTilesPerBlockConf = 8;

constexpr uint64_t kStreamTargetUnits = 8192;    // wave-private variant: work units (one per wavefront)
// Objectness-first variant: workgroups (four wavefronts each) a launch may hold and still be resident at once with
// room to spare, three on each of 256 compute units.  (Measured at 64 images of 10647 boxes with the select stage
// riding along: 512-thread workgroups, 512 of them: 23.8 us per launch alone, 17.2 us per step on three streams;
// 256-thread workgroups: 768 -> 22.1 / 16.6 us, 1024 or 1280 -> 23.1 / 16.5 us; 128-thread workgroups 25.1 / 16.4 us,
// the select stage's workgroup then being the launch's slowest.)
constexpr uint64_t kRowsResidentGroups = 768;

// Launch-plan knobs (yh_set_tuning): the one piece of process-wide state, plain words read with
// relaxed atomics; 0 = the built-in default.
constexpr int kTuneFields = 9;
std::atomic<int32_t> g_tuning[kTuneFields];
enum { kTuneDetectKernel, kTuneRowSteps, kTuneStreamNi, kTuneStreamNbuf, kTuneStreamUnits, kTuneStreamBlocks, kTuneStreamResident,
       kTuneRowsDenseMin, kTuneRowsProbe };

long knob(int which, long fallback)
{
    const long v = g_tuning[which].load(std::memory_order_relaxed);
    return v > 0 ? v : fallback;
}

int compute_units();

uint32_t plan_stream(yh::StreamParams &p, int B, yh::StreamMode mode, int32_t max_candidates = 0)
{
    // The detection pass is objectness-first (or, by the knob, wave-private tiles of 4 or 8 KiB when a
    // row fits one).  A capacity beyond the single-launch select stage announces a dense candidate set
    // (low thresholds): there the workgroup-tile variant, which reads the whole head and appends a few
    // hundred candidates per atomic, is the faster one.
    p.wave_ni = 0;
    p.row_gather = 0;
    p.units = 0;
    if (mode == yh::kModeDetect && max_candidates <= yh::kPostCap) {
        // tuning knob: 1 = objectness-first variant (default), 2 = streaming variants
        if (knob(kTuneDetectKernel, 1) == 1) {
            p.row_gather = 1;
        } else {
            const long ni = knob(kTuneStreamNi, 8);         // tuning knob: 4, 8, anything else = workgroup tiles
            if ((ni == 4 || ni == 8) && (256 * ni - 3) / p.K >= 1) p.wave_ni = (int32_t)ni;
            else if (ni == 4 && (256 * 8 - 3) / p.K >= 1) p.wave_ni = 8;
        }
    }
    if (p.row_gather) {
        // a work unit = `steps` steps of 64 rows (one per lane) of one image and scale; a workgroup
        // takes four consecutive units of one image (the last workgroup of an image may hold spare
        // ones); an image's units are indexed by the candidates -> detections stage, so their
        // number is bounded
        const uint32_t waves = (uint32_t)(yh::kRowsThreads / 64);
        auto units_per_image = [&](uint64_t steps) {
            uint64_t upi = 0;
            for (int s = 0; s < p.S; ++s) upi += ((uint64_t)p.sc[s].rows_img + 64 * steps - 1) / (64 * steps);
            return (upi + waves - 1) / waves * waves;
        };
        // Default: the fewest steps per unit (down to two) with which the whole launch is resident at once,
        // three workgroups per compute unit, counting the select stage's workgroup per image that rides in
        // a pipelined launch (it holds its slot for about 12 us): a unit of up to four steps asks for all
        // its objectness logits at the start, and no workgroup waits for a slot before it can ask.
        uint64_t steps = (uint64_t)knob(kTuneRowSteps, 0);
        if (steps == 0) {
            steps = 4;
            for (uint64_t cand = 2; cand <= 16; ++cand)
                if ((uint64_t)B * (units_per_image(cand) / waves + 1) <= kRowsResidentGroups) { steps = cand; break; }
        }
        while (units_per_image(steps) > (uint64_t)yh::kMaxUnitsPerImage) ++steps;
        p.tile_rows = 64;
        uint32_t upi = 0;
        for (int s = 0; s < p.S; ++s) {
            const uint64_t chunks = ((uint64_t)p.sc[s].rows_img + 64 * steps - 1) / (64 * steps);
            p.sc[s].tiles_per_chunk = (uint32_t)steps;
            p.sc[s].chunks_per_img = (uint32_t)chunks;
            p.sc[s].blk_begin = 0;
            p.sc[s].unit_base = upi;
            upi += (uint32_t)chunks;
        }
        // A unit is read whole (its steps as contiguous spans of 64 rows) instead of row by row once so many of its
        // rows survive that the gathers would fetch as many cache lines as the spans hold: lines of a span
        // 64 * 4K / 128 = 2K; row by row the objectness column costs min(64, 2K) lines per step and a surviving row
        // 4K / 128 more (its objectness line does not survive in the cache until the row is read).  K = 85: 40 of 64
        // rows.  The kernel decides per unit from sixteen sampled rows (10 of 16 at K = 85), and a launch lasts as
        // long as its slowest unit: a unit read whole takes 32-36 us at K = 85 and four steps however many rows
        // survive, a unit walked row by row 20 us plus 35 us times the share of surviving rows.  With this threshold
        // a sparse launch (16 % surviving rows: 26 us) sees no unit read whole by chance (2e-5 per unit), and from
        // 30 % on, where chance picks a few per launch, they end with the others.  (64 images of 10647 boxes, no
        // candidates: row by row 20.1 / 25.5 / 39.5 / 50.7 / 54.8 us at 2 / 16 / 50 / 84 / 100 % surviving rows;
        // with whole units 20.8 / 26.6 / 40.7 / 43.1 / 41.1 us, 235 MB of traffic at 100 % instead of 319 MB.)
        // Rows no longer than a cache line: the column read fetches every line of the tensor anyway, a unit is read
        // whole from 16 surviving rows in 64 on.
        {
            const double span_lines = 2.0 * p.K, col_lines = std::min(64.0, span_lines);
            const double per_row = std::max(0.25, p.K / 32.0);
            long dm = std::lround((span_lines - col_lines) / per_row);
            dm = std::min<long>(64, std::max<long>(16, dm));
            p.dense_min = (int32_t)knob(kTuneRowsDenseMin, dm);
            p.dense_probe = knob(kTuneRowsProbe, 1) == 1 ? 1 : 0;   // tuning knob: 1 = sample (default), 2 = never
            p.dense_col_step = 256 % p.K;
            p.dense_row_step = 256 / p.K;
        }
        p.units = (upi + waves - 1) / waves;                     // workgroups per image
        p.upi = (int32_t)(p.units * waves);
        return p.units * (uint32_t)B;
    }
    if (p.wave_ni) {
        const long nb = knob(kTuneStreamNbuf, 3);
        p.wave_nbuf = (nb >= 2 && nb <= (p.wave_ni == 4 ? 4 : 3)) ? (int32_t)nb : 3;
        p.tile_rows = (256 * p.wave_ni - 3) / p.K;
        uint64_t total_tiles = 0;
        for (int s = 0; s < p.S; ++s)
            total_tiles += ((uint64_t)p.sc[s].rows_img + p.tile_rows - 1) / p.tile_rows * (uint64_t)B;
        const uint64_t want = (uint64_t)knob(kTuneStreamUnits, (long)kStreamTargetUnits);
        uint64_t tpc = (total_tiles + want / 2) / want;          // tiles per wavefront, nearest
        if (tpc < 1) tpc = 1;
        // an image's units are indexed by the candidates -> detections stage: bounded
        const uint64_t tiles_one_img = total_tiles / (uint64_t)B;
        const uint64_t tpc_min = (tiles_one_img + (yh::kMaxUnitsPerImage - p.S) - 1) / (yh::kMaxUnitsPerImage - p.S);
        if (tpc < tpc_min) tpc = tpc_min;
        uint32_t units = 0, upi = 0;
        for (int s = 0; s < p.S; ++s) {
            const uint64_t tiles_img = ((uint64_t)p.sc[s].rows_img + p.tile_rows - 1) / p.tile_rows;
            const uint64_t chunks = (tiles_img + tpc - 1) / tpc;
            p.sc[s].tiles_per_chunk = (uint32_t)((tiles_img + chunks - 1) / chunks);   // even split
            p.sc[s].chunks_per_img = (uint32_t)chunks;
            p.sc[s].blk_begin = units;
            p.sc[s].unit_base = upi;
            units += (uint32_t)chunks * (uint32_t)B;
            upi += (uint32_t)chunks;
        }
        p.upi = (int32_t)upi;
        return units;                                            // one wavefront per workgroup
    }
    p.tile_rows = (yh::kTileFloats - 3) / p.K;
    if (p.tile_rows < 1) {                                       // a row wider than a tile: no plan, check_plan() refuses the call
        p.cuts = p.grid_blocks = 0;
        return 0;
    }
    uint64_t total_tiles = 0;
    for (int s = 0; s < p.S; ++s)
        total_tiles += ((uint64_t)p.sc[s].rows_img + p.tile_rows - 1) / p.tile_rows * (uint64_t)B;
    // tuning knob; by default at least kStreamTargetBlocks workgroups and at most kStreamTilesPerBlock tiles each
    uint64_t want = (uint64_t)knob(kTuneStreamBlocks, 0);
    if (want == 0)
        want = std::max<uint64_t>(kStreamTargetBlocks, total_tiles / (mode == yh::kModeConf ? kStreamTilesPerBlockConf : kStreamTilesPerBlock));
    uint64_t tpc = (total_tiles + want / 2) / want;              // tiles per workgroup, nearest
    if (tpc < 1) tpc = 1;
    if (tpc > 64) tpc = 64;
    uint32_t blocks = 0;
    for (int s = 0; s < p.S; ++s) {
        const uint64_t tiles_img = ((uint64_t)p.sc[s].rows_img + p.tile_rows - 1) / p.tile_rows;
        const uint64_t chunks = (tiles_img + tpc - 1) / tpc;
        p.sc[s].tiles_per_chunk = (uint32_t)((tiles_img + chunks - 1) / chunks);   // even split
        p.sc[s].chunks_per_img = (uint32_t)chunks;
        p.sc[s].blk_begin = blocks;
        blocks += (uint32_t)chunks * (uint32_t)B;
    }
    p.cuts = blocks;
    p.grid_blocks = blocks;
    const uint64_t resident = (uint64_t)knob(kTuneStreamResident, 0) * (uint64_t)compute_units();
    if (resident > 0 && resident < blocks) p.grid_blocks = (uint32_t)resident;
    return blocks;
}

// The workgroup-tile kernel holds whole rows in its LDS tile (three floats of it absorb the tile's 16-byte
// phase): wider rows have no plan there.  The objectness-first and the full-decode kernels take any row.
int check_plan(const yh::StreamParams &p, yh::StreamMode mode)
{
    if (mode != yh::kModeFull && !p.wave_ni && !p.row_gather && p.tile_rows < 1)
        return fail(YH_ERR_LIMIT, "rows of %d floats do not fit the %d-float tile of the streaming kernel this call selects "
                    "(%s): at most %d classes there", p.K, yh::kTileFloats,
                    mode == yh::kModeConf ? "yh_nms" : "max_candidates above 2048 or detect_kernel = 2",
                    yh::kTileFloats - 3 - (mode == yh::kModeConf ? 0 : 5));
    return YH_OK;
}

void fill_stream_from_config(yh::StreamParams &p, const yh_head_config *cfg, const float *const *heads, int B)
{
    std::memset(&p, 0, sizeof(p));
    p.S = cfg->num_scales; p.B = B; p.C = cfg->num_classes; p.P = cfg->num_priors;
    p.K = p.C + 5; p.N = (int32_t)boxes_per_image(cfg);
    p.magic_k = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)p.K) + 1u;
    p.scan_col_step = (4 * yh::kTileThreads) % p.K;
    uint32_t row_base = 0;
    for (int s = 0; s < p.S; ++s) {
        yh::StreamScale &sc = p.sc[s];
        sc.src = heads ? heads[s] : nullptr;
        sc.rows_img = (uint32_t)cfg->grid_h[s] * cfg->grid_w[s] * cfg->num_priors;
        sc.span = sc.rows_img * (uint32_t)p.K;
        sc.grid_w = (uint32_t)cfg->grid_w[s];
        // floor(2^32 / d), capped for d = 1: the kernels' quotient estimate stays at most one too small
        auto recip = [](uint64_t d) { const uint64_t m = ((uint64_t)1 << 32) / d; return (uint32_t)(m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m); };
        sc.magic_w = recip(sc.grid_w);
        sc.magic_p = recip((uint64_t)cfg->num_priors);
        sc.row_base = row_base;
        sc.mult = cfg->box_mult[s];
        std::memcpy(sc.prior_hw, cfg->prior_hw[s], sizeof(sc.prior_hw));
        row_base += sc.rows_img;
    }
}

int ensure_device()
{
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) {
        (void)hipGetLastError();
        return fail(YH_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    return YH_OK;
}

// Compute units of the current device (cached per device; 0 if the query fails).
int compute_units()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0; }
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 0) { (void)hipGetLastError(); n = 0; }
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

}  // namespace

extern "C" {

int yh_version(void) { return 200; }

int yh_set_tuning(const yh_tuning *t)
{
    const int32_t v[kTuneFields] = {t ? t->detect_kernel : 0, t ? t->row_steps : 0, t ? t->stream_tile_kib : 0,
                                    t ? t->stream_buffers : 0, t ? t->stream_units : 0, t ? t->stream_blocks : 0,
                                    t ? t->stream_resident : 0, t ? t->rows_dense_min : 0, t ? t->rows_probe : 0};
    for (int i = 0; i < kTuneFields; ++i)
        if (v[i] < 0) return fail(YH_ERR_INVALID_ARG, "tuning field %d is negative", i);
    for (int i = 0; i < kTuneFields; ++i) g_tuning[i].store(v[i], std::memory_order_relaxed);
    return YH_OK;
}

int yh_get_tuning(yh_tuning *t)
{
    if (!t) return fail(YH_ERR_INVALID_ARG, "tuning is null");
    t->detect_kernel = g_tuning[kTuneDetectKernel].load(std::memory_order_relaxed);
    t->row_steps = g_tuning[kTuneRowSteps].load(std::memory_order_relaxed);
    t->stream_tile_kib = g_tuning[kTuneStreamNi].load(std::memory_order_relaxed);
    t->stream_buffers = g_tuning[kTuneStreamNbuf].load(std::memory_order_relaxed);
    t->stream_units = g_tuning[kTuneStreamUnits].load(std::memory_order_relaxed);
    t->stream_blocks = g_tuning[kTuneStreamBlocks].load(std::memory_order_relaxed);
    t->stream_resident = g_tuning[kTuneStreamResident].load(std::memory_order_relaxed);
    t->rows_dense_min = g_tuning[kTuneRowsDenseMin].load(std::memory_order_relaxed);
    t->rows_probe = g_tuning[kTuneRowsProbe].load(std::memory_order_relaxed);
    return YH_OK;
}

const char *yh_last_error(void) { return g_err; }

int yh_device_count(int32_t *count)
{
    if (!count) return fail(YH_ERR_INVALID_ARG, "count is null");
    *count = 0;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) {
        (void)hipGetLastError();
        return fail(YH_ERR_NO_DEVICE, "no HIP device available (%s)",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    *count = n;
    return YH_OK;
}

int yh_probe_read(const void *src, size_t bytes, void *stream)
{
    if (!src && bytes) return fail(YH_ERR_INVALID_ARG, "src is null");
    if (reinterpret_cast<uintptr_t>(src) & 15u) return fail(YH_ERR_ALIGN, "src is not 16-byte aligned");
    const int rc = ensure_device();
    if (rc != YH_OK) return rc;
    const hipError_t e = yh::launch_probe_read(src, bytes, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "probe launch");
    return YH_OK;
}

int yh_boxes_per_image(const yh_head_config *cfg, int64_t *n)
{
    if (!n) return fail(YH_ERR_INVALID_ARG, "n is null");
    const int rc = check_config(cfg);
    if (rc != YH_OK) return rc;
    *n = boxes_per_image(cfg);
    return YH_OK;
}

int yh_decode(const yh_head_config *cfg, const float *const *heads, int32_t batch,
              float *boxes, float *conf, void *stream)
{
    int rc = check_config(cfg);
    if (rc != YH_OK) return rc;
    if (batch < 0) return fail(YH_ERR_INVALID_ARG, "batch %d is negative", batch);
    if (batch == 0) return YH_OK;
    if ((rc = check_heads(cfg, heads)) != YH_OK) return rc;
    if (!boxes) return fail(YH_ERR_INVALID_ARG, "boxes is null");
    if (reinterpret_cast<uintptr_t>(boxes) & 15u) return fail(YH_ERR_ALIGN, "boxes is not 16-byte aligned");
    if ((rc = ensure_device()) != YH_OK) return rc;
    yh::StreamParams p;
    fill_stream_from_config(p, cfg, heads, batch);
    p.boxes = boxes;
    p.conf = conf;
    const uint32_t blocks = plan_stream(p, batch, yh::kModeFull);
    const hipError_t e = yh::launch_stream(p, yh::kModeFull, blocks, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "decode launch");
    return YH_OK;
}

int yh_detect_workspace_bytes(const yh_head_config *cfg, int32_t batch, int32_t max_candidates,
                              int32_t nms_max_boxes, size_t *bytes)
{
    if (!bytes) return fail(YH_ERR_INVALID_ARG, "bytes is null");
    const int rc = check_config(cfg);
    if (rc != YH_OK) return rc;
    if (batch < 0 || max_candidates < 1 || nms_max_boxes < 1 || nms_max_boxes > YH_MAX_NMS_BOXES)
        return fail(YH_ERR_INVALID_ARG, "bad batch / max_candidates / nms_max_boxes");
    yh::StreamParams p;
    fill_stream_from_config(p, cfg, nullptr, batch);
    if (batch > 0) {
        plan_stream(p, batch, yh::kModeDetect, max_candidates);
        const int prc = check_plan(p, yh::kModeDetect);
        if (prc != YH_OK) return prc;
    }
    *bytes = carve_for_plan(nullptr, p, batch, max_candidates, nms_max_boxes).bytes;
    return YH_OK;
}

int yh_nms_workspace_bytes(int32_t batch, int64_t boxes_per_img, int32_t num_classes,
                           int32_t max_candidates, int32_t nms_max_boxes, size_t *bytes)
{
    if (!bytes) return fail(YH_ERR_INVALID_ARG, "bytes is null");
    if (batch < 0 || boxes_per_img < 1 || num_classes < 1 || max_candidates < 1 || nms_max_boxes < 1 ||
        nms_max_boxes > YH_MAX_NMS_BOXES)
        return fail(YH_ERR_INVALID_ARG, "bad size argument");
    if (num_classes > YH_MAX_CLASSES || num_classes > yh::kTileFloats - 3)
        return fail(YH_ERR_LIMIT, "num_classes %d: yh_nms streams rows of at most %d confidences", num_classes,
                    std::min<int>(YH_MAX_CLASSES, yh::kTileFloats - 3));
    *bytes = carve_detect(nullptr, batch, boxes_per_img, num_classes, max_candidates, nms_max_boxes, false, false, 0).bytes;
    return YH_OK;
}

static yh::NmsParams make_nms_params(const DetectWs &w, int32_t B, int32_t C, int32_t N, const float *boxes,
                         float iou_thr, int32_t max_boxes, int32_t max_cand,
                         int32_t *out_count, float *out_scores, float *out_boxes, int32_t *out_class,
                         int32_t *out_index, int32_t *out_class_candidates, int32_t *status)
{
    yh::NmsParams q;
    std::memset(&q, 0, sizeof(q));
    q.B = B; q.C = C; q.N = N; q.cand_cap = max_cand; q.max_boxes = max_boxes; q.iou_thr = iou_thr;
    q.cand = w.cand; q.seg_keys = w.seg_keys; q.img_count = w.img_count;
    q.seg_off = w.seg_off; q.seg_cnt = w.seg_cnt; q.kept_cnt = w.kept_cnt;
    q.class_hist = w.class_hist; q.class_cursor = w.class_cursor;
    q.kept_row = w.kept_row; q.kept_score = w.kept_score; q.kept_box = w.kept_box; q.boxes = boxes;
    q.out_count = out_count; q.out_scores = out_scores; q.out_boxes = out_boxes;
    q.out_class = out_class; q.out_index = out_index; q.out_class_candidates = out_class_candidates;
    q.status = status;
    return q;
}

// Streaming pass, then the candidates -> detections stage.
static int run_pipeline(const yh::StreamParams &p, yh::StreamMode mode, uint32_t blocks, const yh::NmsParams &q,
                        hipStream_t st, hipEvent_t ev_begin, hipEvent_t ev_end)
{
    hipError_t e;
    if (ev_begin && (e = hipEventRecord(ev_begin, st)) != hipSuccess) return hip_fail(e, "event record");
    e = yh::launch_stream(p, mode, blocks, st);
    if (e != hipSuccess) return hip_fail(e, "stream launch");
    if (ev_end && (e = hipEventRecord(ev_end, st)) != hipSuccess) return hip_fail(e, "event record");
    e = yh::launch_nms_pipeline(q, st);
    if (e != hipSuccess) return hip_fail(e, "nms launch");
    return YH_OK;
}

// Everything yh_detect derives from its arguments.  heads == nullptr: only the
// candidates -> detections stage is wanted; out_count == nullptr: only the
// streaming pass.
struct DetectPlan {
    yh::StreamParams p;
    yh::NmsParams q;
    uint32_t blocks;
    size_t zero_begin, zero_bytes;      // the workspace's counters (DetectWs)
};

static int detect_prepare(DetectPlan &d, const yh_head_config *cfg, const float *const *heads, bool want_stream,
                          bool want_select, int32_t batch, float score_threshold, float iou_threshold,
                          int32_t nms_max_boxes, int32_t max_candidates, void *workspace, size_t workspace_bytes,
                          int32_t *out_count, float *out_scores, float *out_boxes, int32_t *out_class,
                          int32_t *out_index, int32_t *out_class_candidates, int32_t *status)
{
    int rc = check_config(cfg);
    if (rc != YH_OK) return rc;
    if (want_select) {
        rc = check_nms_args(batch, iou_threshold, nms_max_boxes, max_candidates, workspace, out_count,
                            out_scores, out_boxes, out_class, out_index, status);
    } else {
        int32_t dummy = 0;                    // the output pointers are not part of this call
        rc = check_nms_args(batch, 0.5f, nms_max_boxes, max_candidates, workspace, &dummy, &dummy, &dummy,
                            &dummy, &dummy, &dummy);
    }
    if (rc != YH_OK) return rc;
    if (batch == 0) return YH_OK;
    if (want_stream && (rc = check_heads(cfg, heads)) != YH_OK) return rc;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return fail(YH_ERR_ALIGN, "workspace must be 256-byte aligned");
    if (want_select && (reinterpret_cast<uintptr_t>(out_boxes) & 15u))
        return fail(YH_ERR_ALIGN, "out_boxes must be 16-byte aligned");
    const int64_t N = boxes_per_image(cfg);
    yh::StreamParams &p = d.p;
    fill_stream_from_config(p, cfg, want_stream ? heads : nullptr, batch);
    d.blocks = plan_stream(p, batch, yh::kModeDetect, max_candidates);
    if ((rc = check_plan(p, yh::kModeDetect)) != YH_OK) return rc;
    const DetectWs w = carve_for_plan(workspace, p, batch, max_candidates, nms_max_boxes);
    if (workspace_bytes < w.bytes)
        return fail(YH_ERR_WORKSPACE, "workspace has %zu bytes, %zu needed", workspace_bytes, w.bytes);
    if ((rc = ensure_device()) != YH_OK) return rc;
    d.zero_begin = w.zero_begin;
    d.zero_bytes = w.zero_bytes;

    p.thr = score_threshold;
    p.prefilter = prefilter_logit(score_threshold);
    p.cand_cap = max_candidates;
    p.cand = w.cand;
    p.img_count = w.img_count;
    p.boxes = w.boxes;
    p.cand_box = w.cand_box;
    p.ukeys = w.ukeys;
    p.uboxes = w.uboxes;
    d.q = make_nms_params(w, batch, cfg->num_classes, (int32_t)N, w.boxes, iou_threshold,
                          nms_max_boxes, max_candidates, out_count, out_scores, out_boxes,
                          out_class, out_index, out_class_candidates, status);
    if (!w.boxes && !(p.wave_ni || p.row_gather)) {
        // boxes by row are decoded from the head tensors where they are looked at: those of this call, or of the
        // streaming call that filled this workspace
        d.q.src_P = p.P; d.q.src_K = p.K; d.q.src_C = p.C; d.q.src_S = p.S;
        std::memcpy(d.q.src, p.sc, sizeof(d.q.src));
        if (want_stream) {
            remember_heads(workspace, heads, p.S);
        } else {
            const float *prev[YH_MAX_SCALES];
            if (!recall_heads(workspace, prev))
                return fail(YH_ERR_INVALID_ARG, "no streaming call on this workspace is on record: its select stage reads the "
                            "boxes it needs from that call's head tensors");
            for (int s = 0; s < p.S; ++s) d.q.src[s].src = prev[s];
        }
    }
    if (!(p.wave_ni || p.row_gather) && yh::select_bins_by_slices(max_candidates, p.C, nms_max_boxes)) {
        // the workgroup-tile kernel counts the classes of the keys it lists: the select stage's first launch
        // (a pass over the list that counts them) falls away
        p.class_hist = w.class_hist;
        d.q.hist_ready = 1;
    }
    if (p.wave_ni || p.row_gather) {          // candidates arrive with slot-indexed boxes
        d.q.upi = p.upi;
        d.q.ukeys = w.ukeys;
        d.q.uboxes = w.uboxes;
        d.q.cand_box = w.cand_box;
        d.q.boxes_fill = w.boxes;
    }
    return YH_OK;
}

static int detect_impl(const yh_head_config *cfg, const float *const *heads, int32_t batch,
              float score_threshold, float iou_threshold, int32_t nms_max_boxes,
              int32_t max_candidates, void *workspace, size_t workspace_bytes,
              int32_t *out_count, float *out_scores, float *out_boxes,
              int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
              int32_t *status, void *stream, hipEvent_t ev_begin, hipEvent_t ev_end)
{
    DetectPlan d;
    const int rc = detect_prepare(d, cfg, heads, true, true, batch, score_threshold, iou_threshold, nms_max_boxes,
                                  max_candidates, workspace, workspace_bytes, out_count, out_scores, out_boxes,
                                  out_class, out_index, out_class_candidates, status);
    if (rc != YH_OK || batch == 0) return rc;
    return run_pipeline(d.p, yh::kModeDetect, d.blocks, d.q, static_cast<hipStream_t>(stream), ev_begin, ev_end);
}

int yh_detect(const yh_head_config *cfg, const float *const *heads, int32_t batch,
              float score_threshold, float iou_threshold, int32_t nms_max_boxes,
              int32_t max_candidates, void *workspace, size_t workspace_bytes,
              int32_t *out_count, float *out_scores, float *out_boxes,
              int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
              int32_t *status, void *stream)
{
    return detect_impl(cfg, heads, batch, score_threshold, iou_threshold, nms_max_boxes, max_candidates,
                       workspace, workspace_bytes, out_count, out_scores, out_boxes, out_class, out_index,
                       out_class_candidates, status, stream, nullptr, nullptr);
}

// ---- yh_detect_static: the workspace is a device array that lives in the library's code object.
// In host code `g_static_scratch` is only a handle for hipGetSymbolAddress: the array's device
// address differs per GPU and is looked up once per device, with that device current.
static __device__ __align__(256) unsigned char g_static_scratch[YH_STATIC_SCRATCH_BYTES];

static int static_scratch_of_current_device(void **out)
{
    constexpr int kDevices = 64;
    static std::mutex mu;
    static void *ptr_of[kDevices];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    if (dev < 0 || dev >= kDevices) return fail(YH_ERR_LIMIT, "device index %d outside [0, %d)", dev, kDevices);
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr_of[dev]) {
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_static_scratch));
        if (e != hipSuccess || !ptr) return hip_fail(e, "hipGetSymbolAddress of the static scratch");
        ptr_of[dev] = ptr;
    }
    *out = ptr_of[dev];
    return YH_OK;
}

int yh_detect_static(const yh_head_config *cfg, const float *const *heads, int32_t batch,
                     float score_threshold, float iou_threshold, int32_t nms_max_boxes,
                     int32_t max_candidates,
                     int32_t *out_count, float *out_scores, float *out_boxes,
                     int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
                     int32_t *status, void *stream)
{
    size_t need = 0;
    int rc = yh_detect_workspace_bytes(cfg, batch, max_candidates, nms_max_boxes, &need);
    if (rc != YH_OK) return rc;
    if (need > (size_t)YH_STATIC_SCRATCH_BYTES)
        return fail(YH_ERR_LIMIT, "this call needs %zu bytes of scratch, the static scratch holds %zu: lower batch or "
                    "max_candidates, or pass a workspace to yh_detect", need, (size_t)YH_STATIC_SCRATCH_BYTES);
    if (batch == 0)
        return detect_impl(cfg, heads, batch, score_threshold, iou_threshold, nms_max_boxes, max_candidates,
                           reinterpret_cast<void *>(256), need, out_count, out_scores, out_boxes, out_class, out_index,
                           out_class_candidates, status, stream, nullptr, nullptr);
    if ((rc = ensure_device()) != YH_OK) return rc;
    void *scratch = nullptr;
    if ((rc = static_scratch_of_current_device(&scratch)) != YH_OK) return rc;
    DetectPlan d;
    rc = detect_prepare(d, cfg, heads, true, true, batch, score_threshold, iou_threshold, nms_max_boxes,
                        max_candidates, scratch, (size_t)YH_STATIC_SCRATCH_BYTES, out_count, out_scores, out_boxes,
                        out_class, out_index, out_class_candidates, status);
    if (rc != YH_OK) return rc;
    // The array keeps whatever the last call (of any shape) left where this call's counters lie.
    hipStream_t st = static_cast<hipStream_t>(stream);
    const hipError_t e = hipMemsetAsync(static_cast<char *>(scratch) + d.zero_begin, 0, d.zero_bytes, st);
    if (e != hipSuccess) return hip_fail(e, "clearing the scratch counters");
    return run_pipeline(d.p, yh::kModeDetect, d.blocks, d.q, st, nullptr, nullptr);
}

int yh_detect_timed(const yh_head_config *cfg, const float *const *heads, int32_t batch,
                    float score_threshold, float iou_threshold, int32_t nms_max_boxes,
                    int32_t max_candidates, void *workspace, size_t workspace_bytes,
                    int32_t *out_count, float *out_scores, float *out_boxes,
                    int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
                    int32_t *status, void *stream, void *event_begin, void *event_end)
{
    return detect_impl(cfg, heads, batch, score_threshold, iou_threshold, nms_max_boxes, max_candidates,
                       workspace, workspace_bytes, out_count, out_scores, out_boxes, out_class, out_index,
                       out_class_candidates, status, stream, static_cast<hipEvent_t>(event_begin),
                       static_cast<hipEvent_t>(event_end));
}

int yh_detect_candidates(const yh_head_config *cfg, const float *const *heads, int32_t batch,
                         float score_threshold, int32_t nms_max_boxes, int32_t max_candidates,
                         void *workspace, size_t workspace_bytes, void *stream)
{
    DetectPlan d;
    const int rc = detect_prepare(d, cfg, heads, true, false, batch, score_threshold, 0.5f, nms_max_boxes,
                                  max_candidates, workspace, workspace_bytes, nullptr, nullptr, nullptr,
                                  nullptr, nullptr, nullptr, nullptr);
    if (rc != YH_OK || batch == 0) return rc;
    const hipError_t e = yh::launch_stream(d.p, yh::kModeDetect, d.blocks, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "stream launch");
    return YH_OK;
}

int yh_detect_select(const yh_head_config *cfg, int32_t batch,
                     float iou_threshold, int32_t nms_max_boxes, int32_t max_candidates,
                     void *workspace, size_t workspace_bytes,
                     int32_t *out_count, float *out_scores, float *out_boxes,
                     int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
                     int32_t *status, void *stream)
{
    DetectPlan d;
    const int rc = detect_prepare(d, cfg, nullptr, false, true, batch, 0.5f, iou_threshold, nms_max_boxes,
                                  max_candidates, workspace, workspace_bytes, out_count, out_scores, out_boxes,
                                  out_class, out_index, out_class_candidates, status);
    if (rc != YH_OK || batch == 0) return rc;
    const hipError_t e = yh::launch_nms_pipeline(d.q, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "nms launch");
    return YH_OK;
}

int yh_detect_pipelined(const yh_head_config *cfg, const float *const *heads, int32_t batch,
                        float score_threshold, float iou_threshold, int32_t nms_max_boxes,
                        int32_t max_candidates, void *workspace, void *prev_workspace, size_t workspace_bytes,
                        int32_t *out_count, float *out_scores, float *out_boxes,
                        int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
                        int32_t *status, void *stream, void *event_begin, void *event_end)
{
    if (heads && prev_workspace && workspace == prev_workspace)
        return fail(YH_ERR_INVALID_ARG, "workspace and prev_workspace must differ");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t ev_b = static_cast<hipEvent_t>(event_begin), ev_e = static_cast<hipEvent_t>(event_end);
    DetectPlan cur, prev;
    int rc;
    if (heads) {
        rc = detect_prepare(cur, cfg, heads, true, false, batch, score_threshold, 0.5f, nms_max_boxes,
                            max_candidates, workspace, workspace_bytes, nullptr, nullptr, nullptr,
                            nullptr, nullptr, nullptr, nullptr);
        if (rc != YH_OK) return rc;
    }
    if (prev_workspace) {
        rc = detect_prepare(prev, cfg, nullptr, false, true, batch, 0.5f, iou_threshold, nms_max_boxes,
                            max_candidates, prev_workspace, workspace_bytes, out_count, out_scores, out_boxes,
                            out_class, out_index, out_class_candidates, status);
        if (rc != YH_OK) return rc;
    }
    if (batch == 0 || (!heads && !prev_workspace)) return YH_OK;
    hipError_t e;
    const bool fuse = heads && prev_workspace &&
                      ((cur.p.wave_ni && yh::select_fits_stream(prev.q)) || (cur.p.row_gather && yh::select_fits_rows(prev.q)));
    if (prev_workspace && !fuse && (e = yh::launch_nms_pipeline(prev.q, st)) != hipSuccess) return hip_fail(e, "nms launch");
    if (heads) {
        if (ev_b && (e = hipEventRecord(ev_b, st)) != hipSuccess) return hip_fail(e, "event record");
        e = yh::launch_stream(cur.p, yh::kModeDetect, cur.blocks, st, fuse ? &prev.q : nullptr);
        if (e != hipSuccess) return hip_fail(e, "stream launch");
        if (ev_e && (e = hipEventRecord(ev_e, st)) != hipSuccess) return hip_fail(e, "event record");
    }
    return YH_OK;
}

static int nms_impl(int32_t batch, int64_t boxes_per_img, int32_t num_classes,
                    const float *boxes, const float *conf,
                    float score_threshold, float iou_threshold, int32_t nms_max_boxes,
                    int32_t max_candidates, void *workspace, size_t workspace_bytes,
                    int32_t *out_count, float *out_scores, float *out_boxes,
                    int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
                    int32_t *status, void *stream, hipEvent_t ev_begin, hipEvent_t ev_end)
{
    int rc = check_nms_args(batch, iou_threshold, nms_max_boxes, max_candidates, workspace, out_count,
                            out_scores, out_boxes, out_class, out_index, status);
    if (rc != YH_OK) return rc;
    if (num_classes < 1 || num_classes > YH_MAX_CLASSES)
        return fail(YH_ERR_LIMIT, "num_classes %d outside [1, %d]", num_classes, YH_MAX_CLASSES);
    if (boxes_per_img < 1 || boxes_per_img > YH_MAX_BOXES_PER_IMAGE)
        return fail(YH_ERR_LIMIT, "boxes_per_image %lld outside [1, %d]", (long long)boxes_per_img, YH_MAX_BOXES_PER_IMAGE);
    if ((boxes_per_img * num_classes + num_classes) * (int64_t)num_classes >= (int64_t)1 << 32)
        return fail(YH_ERR_LIMIT, "boxes_per_image * num_classes^2 exceeds 2^32");
    if (num_classes > yh::kTileFloats - 3)
        return fail(YH_ERR_LIMIT, "num_classes %d: yh_nms streams rows of at most %d confidences", num_classes, yh::kTileFloats - 3);
    if (batch == 0) return YH_OK;
    if (!boxes || !conf) return fail(YH_ERR_INVALID_ARG, "boxes or conf is null");
    if ((reinterpret_cast<uintptr_t>(boxes) & 15u) || (reinterpret_cast<uintptr_t>(conf) & 15u) ||
        (reinterpret_cast<uintptr_t>(out_boxes) & 15u) || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return fail(YH_ERR_ALIGN, "boxes / conf / out_boxes must be 16-byte and workspace 256-byte aligned");
    const DetectWs w = carve_detect(workspace, batch, boxes_per_img, num_classes, max_candidates, nms_max_boxes, false, false, 0);
    if (workspace_bytes < w.bytes)
        return fail(YH_ERR_WORKSPACE, "workspace has %zu bytes, %zu needed", workspace_bytes, w.bytes);
    if ((rc = ensure_device()) != YH_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);


    // the confidence matrix is streamed as a one-scale "head" whose rows hold C channels
    yh::StreamParams p;
    std::memset(&p, 0, sizeof(p));
    p.S = 1; p.B = batch; p.C = num_classes; p.P = 1; p.K = num_classes; p.N = (int32_t)boxes_per_img;
    p.magic_k = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)p.K) + 1u;
    p.scan_col_step = (4 * yh::kTileThreads) % p.K;
    p.thr = score_threshold;
    p.cand_cap = max_candidates;
    p.cand = w.cand;
    p.img_count = w.img_count;
    p.sc[0].src = conf;
    p.sc[0].rows_img = (uint32_t)boxes_per_img;
    p.sc[0].span = (uint32_t)(boxes_per_img * num_classes);
    p.sc[0].grid_w = 1;
    p.sc[0].magic_w = 0xFFFFFFFFu; p.sc[0].magic_p = 0xFFFFFFFFu;
    yh::NmsParams q = make_nms_params(w, batch, num_classes, (int32_t)boxes_per_img, boxes, iou_threshold,
                                      nms_max_boxes, max_candidates, out_count, out_scores, out_boxes,
                                      out_class, out_index, out_class_candidates, status);
    if (yh::select_bins_by_slices(max_candidates, num_classes, nms_max_boxes)) {
        p.class_hist = w.class_hist;              // counted where the keys are listed
        q.hist_ready = 1;
    }
    const uint32_t blocks = plan_stream(p, batch, yh::kModeConf);
    return run_pipeline(p, yh::kModeConf, blocks, q, st, ev_begin, ev_end);
}

int yh_nms(int32_t batch, int64_t boxes_per_img, int32_t num_classes,
           const float *boxes, const float *conf,
           float score_threshold, float iou_threshold, int32_t nms_max_boxes,
           int32_t max_candidates, void *workspace, size_t workspace_bytes,
           int32_t *out_count, float *out_scores, float *out_boxes,
           int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
           int32_t *status, void *stream)
{
    return nms_impl(batch, boxes_per_img, num_classes, boxes, conf, score_threshold, iou_threshold, nms_max_boxes,
                    max_candidates, workspace, workspace_bytes, out_count, out_scores, out_boxes, out_class, out_index,
                    out_class_candidates, status, stream, nullptr, nullptr);
}

int yh_nms_timed(int32_t batch, int64_t boxes_per_img, int32_t num_classes,
                 const float *boxes, const float *conf,
                 float score_threshold, float iou_threshold, int32_t nms_max_boxes,
                 int32_t max_candidates, void *workspace, size_t workspace_bytes,
                 int32_t *out_count, float *out_scores, float *out_boxes,
                 int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
                 int32_t *status, void *stream, void *event_begin, void *event_end)
{
    return nms_impl(batch, boxes_per_img, num_classes, boxes, conf, score_threshold, iou_threshold, nms_max_boxes,
                    max_candidates, workspace, workspace_bytes, out_count, out_scores, out_boxes, out_class, out_index,
                    out_class_candidates, status, stream, static_cast<hipEvent_t>(event_begin),
                    static_cast<hipEvent_t>(event_end));
}

static void plan_loss(yh::LossParams &p, const yh_head_config *cfg, int32_t batch, int32_t pad)
{
    std::memset(&p, 0, sizeof(p));
    p.S = cfg->num_scales; p.B = batch; p.C = cfg->num_classes; p.P = cfg->num_priors;
    p.K = p.C + 5; p.N = (int32_t)boxes_per_image(cfg); p.pad = pad;
    uint32_t blk = 0, row_base = 0;
    for (int s = 0; s < p.S; ++s) {
        yh::LossScale &sc = p.sc[s];
        sc.H = cfg->grid_h[s]; sc.W = cfg->grid_w[s];
        sc.rows_img = (uint32_t)sc.H * sc.W * cfg->num_priors;
        sc.row_base = row_base;
        sc.blk_begin = blk;
        sc.gt_rate = cfg->gt_rate[s];
        std::memcpy(sc.prior_hw, cfg->prior_hw[s], sizeof(sc.prior_hw));
        // q = umulhi(n, floor(2^32 / d) + 1) equals n / d for every n with n * d < 2^32
        auto magic = [](uint32_t d, uint64_t n_max) -> uint32_t {
            if (d <= 1u || n_max * d >= ((uint64_t)1 << 32)) return 0u;
            return (uint32_t)((((uint64_t)1 << 32) / d) + 1u);
        };
        sc.magic_p = magic((uint32_t)cfg->num_priors, sc.rows_img);
        sc.magic_w = magic((uint32_t)sc.W, (uint64_t)sc.H * sc.W);
        blk += (sc.rows_img + yh::kLossNegRows - 1) / yh::kLossNegRows;
        row_base += sc.rows_img;
    }
    // pairs per table: whole turns of the loop (two pairs), and a multiple of eight so that a table is a whole
    // number of 128-byte lines: the scalar data cache is not coherent with the vector stores that write the
    // tables, and a line shared by two tables could be cached before its second half is written
    p.pair_cap = 8 * ((2 * ((pad + 3) / 4) + 7) / 8);
    int occ_words = 0;
    for (int s = 0; s < p.S; ++s) occ_words = std::max(occ_words, (p.sc[s].H * p.sc[s].W + 31) >> 5);
    p.occ_cap = (occ_words + 31) / 32 * 32;
    p.blocks_per_img = (int32_t)blk;
    p.pos_blocks = (pad + yh::kLossPosGtPerBlock - 1) / yh::kLossPosGtPerBlock;
    if (p.pos_blocks < 1) p.pos_blocks = 1;
}

struct LossWs { double *pair; float *neg, *pos; int32_t *gcount; float *pair_tab; yh::PosRecord *pos_rec; uint32_t *occ_tab; int32_t *prep_flags; size_t bytes; };

static LossWs carve_loss(void *base, const yh::LossParams &p)
{
    LossWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const uintptr_t c = reinterpret_cast<uintptr_t>(base);      // may be null (a size query)
    const size_t o_pair = take(2 * sizeof(double));            // offset 0: documented in yolohead.h
    const size_t o_neg = take((size_t)p.B * p.blocks_per_img * 4);
    const size_t o_pos = take((size_t)p.B * p.pos_blocks * 3 * 4);
    const size_t o_g = take((size_t)p.B * 4);
    const size_t o_pp = take((size_t)p.B * p.S * p.pair_cap * 12 * 4);
    const size_t o_pr = take((size_t)p.B * p.pos_blocks * yh::kLossPosGtPerBlock * sizeof(yh::PosRecord));
    w.pair = reinterpret_cast<double *>(c + o_pair);
    w.neg = reinterpret_cast<float *>(c + o_neg);
    w.pos = reinterpret_cast<float *>(c + o_pos);
    w.gcount = reinterpret_cast<int32_t *>(c + o_g);
    w.pair_tab = reinterpret_cast<float *>(c + o_pp);
    const size_t o_occ = take((size_t)p.B * p.S * p.occ_cap * 4);
    const size_t o_flag = take((size_t)p.B * p.S * 4);
    w.pos_rec = reinterpret_cast<yh::PosRecord *>(c + o_pr);
    w.occ_tab = reinterpret_cast<uint32_t *>(c + o_occ);
    w.prep_flags = reinterpret_cast<int32_t *>(c + o_flag);
    w.bytes = off;
    return w;
}

// yh_loss_prepare's record: which workspace holds tables for which ground truth.  The next loss call on the
// workspace takes the record away, and uses the tables when it is a call for that very batch.
struct PreparedLoss { const void *ws; const float *gt; int32_t batch, pad; uint64_t cfg_hash; };
constexpr int kPreparedLosses = 64;
static std::mutex g_prepared_mu;
static PreparedLoss g_prepared[kPreparedLosses];
static unsigned g_prepared_next = 0;

static uint64_t config_hash(const yh_head_config *cfg)
{
    uint64_t h = 1469598103934665603ull;                      // FNV-1a over the bytes of the struct
    const unsigned char *c = reinterpret_cast<const unsigned char *>(cfg);
    for (size_t i = 0; i < sizeof(*cfg); ++i) h = (h ^ c[i]) * 1099511628211ull;
    return h;
}

static void record_prepared(const void *ws, const float *gt, int32_t batch, int32_t pad, uint64_t hash)
{
    std::lock_guard<std::mutex> lock(g_prepared_mu);
    int at = -1, empty = -1;
    for (int i = 0; i < kPreparedLosses && at < 0; ++i) {
        if (g_prepared[i].ws == ws) at = i;
        if (empty < 0 && g_prepared[i].ws == nullptr) empty = i;
    }
    // a free cell if there is one; with kPreparedLosses preparations outstanding one of them is forgotten (its loss
    // call then builds the tables itself)
    if (at < 0) at = empty >= 0 ? empty : (int)(g_prepared_next++ % kPreparedLosses);
    g_prepared[at] = PreparedLoss{ws, gt, batch, pad, hash};
}

static bool take_prepared(const void *ws, const float *gt, int32_t batch, int32_t pad, uint64_t hash)
{
    std::lock_guard<std::mutex> lock(g_prepared_mu);
    for (int i = 0; i < kPreparedLosses; ++i)
        if (g_prepared[i].ws == ws) {
            const PreparedLoss r = g_prepared[i];
            g_prepared[i] = PreparedLoss{nullptr, nullptr, 0, 0, 0};
            return r.gt == gt && r.batch == batch && r.pad == pad && r.cfg_hash == hash;
        }
    return false;
}

static int check_loss_sizes(const yh_head_config *cfg, int32_t batch, int32_t pad)
{
    if (batch < 0) return fail(YH_ERR_INVALID_ARG, "batch %d is negative", batch);
    if (pad < 1 || pad > yh::kLossMaxPad) return fail(YH_ERR_LIMIT, "pad %d outside [1, %d]", pad, yh::kLossMaxPad);
    int occ_words = 0;
    for (int s = 0; s < cfg->num_scales; ++s) {
        if (!(cfg->gt_rate[s] > 0.0f)) return fail(YH_ERR_INVALID_ARG, "gt_rate[%d] must be positive", s);
        if ((int64_t)cfg->grid_h[s] * cfg->grid_w[s] > (1 << 18))
            return fail(YH_ERR_LIMIT, "grid %d has more than 2^18 cells", s);
        const int w = (cfg->grid_h[s] * cfg->grid_w[s] + 31) >> 5;
        occ_words = w > occ_words ? w : occ_words;
    }
    // the ground-truth rows and the occupancy bitmap of the largest grid share a workgroup's LDS
    const size_t lds = yh::loss_lds_bytes(pad, occ_words) + 2048;
    if (lds > yh::kLdsPerWorkgroupMax)
        return fail(YH_ERR_LIMIT, "pad %d with a grid of %d cells needs %zu bytes of LDS, %zu available", pad,
                    occ_words * 32, lds, yh::kLdsPerWorkgroupMax);
    return YH_OK;
}

int yh_loss_workspace_bytes(const yh_head_config *cfg, int32_t batch, int32_t pad, size_t *bytes)
{
    if (!bytes) return fail(YH_ERR_INVALID_ARG, "bytes is null");
    int rc = check_config(cfg);
    if (rc != YH_OK) return rc;
    if ((rc = check_loss_sizes(cfg, batch, pad)) != YH_OK) return rc;
    yh::LossParams p;
    plan_loss(p, cfg, batch, pad);
    *bytes = carve_loss(nullptr, p).bytes;
    return YH_OK;
}

static int check_grad_heads(const yh_head_config *cfg, const float *const *heads, float *const *grad_heads)
{
    for (int s = 0; s < cfg->num_scales; ++s) {
        if (!grad_heads[s]) return fail(YH_ERR_INVALID_ARG, "grad_heads[%d] is null", s);
        if (grad_heads[s] == heads[s]) return fail(YH_ERR_INVALID_ARG, "grad_heads[%d] aliases heads[%d]", s, s);
    }
    return YH_OK;
}

// The forward pass; with grad_per_image and grad_heads, the gradient in the same pass.
static int loss_forward(const yh_head_config *cfg, const float *const *heads, const float *ground_truth,
                        int32_t batch, int32_t pad, const float *loss_scales, float ignore_threshold,
                        int32_t normalize_by_num_gt, void *workspace, size_t workspace_bytes,
                        float *loss_per_image, float *loss_parts, float *iou_max, float *loss_sum,
                        uint8_t *noobj_mask, const float *grad_per_image, float *const *grad_heads, void *stream)
{
    int rc = check_config(cfg);
    if (rc != YH_OK) return rc;
    if ((rc = check_loss_sizes(cfg, batch, pad)) != YH_OK) return rc;
    if (!loss_scales) return fail(YH_ERR_INVALID_ARG, "loss_scales is null");
    if (batch == 0) return YH_OK;
    if ((rc = check_heads(cfg, heads)) != YH_OK) return rc;
    if (!ground_truth || !workspace || !loss_per_image)
        return fail(YH_ERR_INVALID_ARG, "null ground_truth, workspace or loss_per_image");
    if (grad_per_image && (rc = check_grad_heads(cfg, heads, grad_heads)) != YH_OK) return rc;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return fail(YH_ERR_ALIGN, "workspace is not 256-byte aligned");
    yh::LossParams p;
    plan_loss(p, cfg, batch, pad);
    const LossWs w = carve_loss(workspace, p);
    if (workspace_bytes < w.bytes)
        return fail(YH_ERR_WORKSPACE, "workspace has %zu bytes, %zu needed", workspace_bytes, w.bytes);
    if ((rc = ensure_device()) != YH_OK) return rc;
    for (int s = 0; s < p.S; ++s) { p.sc[s].head = heads[s]; p.sc[s].grad = grad_per_image ? grad_heads[s] : nullptr; }
    p.grad_w = grad_per_image;
    p.w_coord = loss_scales[0]; p.w_noobj = loss_scales[1]; p.w_obj = loss_scales[2]; p.w_class = loss_scales[3];
    p.ignore_thr = ignore_threshold;
    p.normalize = normalize_by_num_gt ? 1 : 0;
    p.gt = ground_truth;
    p.neg_partial = w.neg; p.pos_partial = w.pos; p.gcount = w.gcount; p.reduce_pair = w.pair;
    p.pair_tab = w.pair_tab;
    p.occ_tab = w.occ_tab; p.prep_flags = w.prep_flags;
    p.prepared = take_prepared(workspace, ground_truth, batch, pad, config_hash(cfg)) ? 1 : 0;
    p.pos_rec = grad_per_image ? w.pos_rec : nullptr;
    p.cu_slots = compute_units();
    // The last workgroup of the main launch finalizes while the launch is small enough for the tickets
    // (one atomic per workgroup on one address) to cost less than a launch of their own; measured at
    // 896 workgroups: 0.9 us in favour of the tickets, at 1568: 8 us against them.
    static std::atomic<uint32_t> next_ticket{0};
    p.ticket_slot = (int64_t)batch * (p.pos_blocks + p.blocks_per_img) <= 1024
                        ? (int32_t)(next_ticket.fetch_add(1, std::memory_order_relaxed) % 4096u) : -1;
    p.loss_per_image = loss_per_image; p.loss_parts = loss_parts; p.iou_max = iou_max; p.loss_sum = loss_sum;
    p.noobj_mask = noobj_mask;
    const hipError_t e = yh::launch_loss(p, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "loss launch");
    return YH_OK;
}

int yh_loss_prepare(const yh_head_config *cfg, const float *ground_truth, int32_t batch, int32_t pad,
                    void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_config(cfg);
    if (rc != YH_OK) return rc;
    if ((rc = check_loss_sizes(cfg, batch, pad)) != YH_OK) return rc;
    if (batch == 0) return YH_OK;
    if (!ground_truth || !workspace) return fail(YH_ERR_INVALID_ARG, "null ground_truth or workspace");
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return fail(YH_ERR_ALIGN, "workspace is not 256-byte aligned");
    yh::LossParams p;
    plan_loss(p, cfg, batch, pad);
    const LossWs w = carve_loss(workspace, p);
    if (workspace_bytes < w.bytes)
        return fail(YH_ERR_WORKSPACE, "workspace has %zu bytes, %zu needed", workspace_bytes, w.bytes);
    if ((rc = ensure_device()) != YH_OK) return rc;
    p.gt = ground_truth;
    p.gcount = w.gcount; p.pair_tab = w.pair_tab; p.occ_tab = w.occ_tab; p.prep_flags = w.prep_flags;
    const hipError_t e = yh::launch_loss_prepare(p, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "loss prepare launch");
    record_prepared(workspace, ground_truth, batch, pad, config_hash(cfg));
    return YH_OK;
}

int yh_loss_forward(const yh_head_config *cfg, const float *const *heads, const float *ground_truth,
                    int32_t batch, int32_t pad, const float *loss_scales, float ignore_threshold,
                    int32_t normalize_by_num_gt, void *workspace, size_t workspace_bytes,
                    float *loss_per_image, float *loss_parts, float *iou_max, float *loss_sum,
                    uint8_t *noobj_mask, void *stream)
{
    return loss_forward(cfg, heads, ground_truth, batch, pad, loss_scales, ignore_threshold, normalize_by_num_gt,
                        workspace, workspace_bytes, loss_per_image, loss_parts, iou_max, loss_sum, noobj_mask,
                        nullptr, nullptr, stream);
}

int yh_loss(const yh_head_config *cfg, const float *const *heads, const float *ground_truth,
            int32_t batch, int32_t pad, const float *loss_scales, float ignore_threshold,
            int32_t normalize_by_num_gt, void *workspace, size_t workspace_bytes,
            float *loss_per_image, float *loss_parts, float *iou_max, float *loss_sum,
            void *stream)
{
    return loss_forward(cfg, heads, ground_truth, batch, pad, loss_scales, ignore_threshold, normalize_by_num_gt,
                        workspace, workspace_bytes, loss_per_image, loss_parts, iou_max, loss_sum, nullptr,
                        nullptr, nullptr, stream);
}

int yh_loss_with_grad(const yh_head_config *cfg, const float *const *heads, const float *ground_truth,
                      int32_t batch, int32_t pad, const float *loss_scales, float ignore_threshold,
                      int32_t normalize_by_num_gt, void *workspace, size_t workspace_bytes,
                      float *loss_per_image, float *loss_parts, float *iou_max, float *loss_sum,
                      const float *grad_per_image, float *const *grad_heads, void *stream)
{
    if (!grad_per_image || !grad_heads) return fail(YH_ERR_INVALID_ARG, "null grad_per_image or grad_heads");
    return loss_forward(cfg, heads, ground_truth, batch, pad, loss_scales, ignore_threshold, normalize_by_num_gt,
                        workspace, workspace_bytes, loss_per_image, loss_parts, iou_max, loss_sum, nullptr,
                        grad_per_image, grad_heads, stream);
}

int yh_loss_backward(const yh_head_config *cfg, const float *const *heads, const float *ground_truth,
                     int32_t batch, int32_t pad, const float *loss_scales, int32_t normalize_by_num_gt,
                     const uint8_t *noobj_mask, const float *grad_per_image, float *const *grad_heads,
                     void *stream)
{
    int rc = check_config(cfg);
    if (rc != YH_OK) return rc;
    if ((rc = check_loss_sizes(cfg, batch, pad)) != YH_OK) return rc;
    if (!loss_scales) return fail(YH_ERR_INVALID_ARG, "loss_scales is null");
    if (batch == 0) return YH_OK;
    if ((rc = check_heads(cfg, heads)) != YH_OK) return rc;
    if (!ground_truth || !noobj_mask || !grad_per_image || !grad_heads)
        return fail(YH_ERR_INVALID_ARG, "null ground_truth, noobj_mask, grad_per_image or grad_heads");
    if ((rc = check_grad_heads(cfg, heads, grad_heads)) != YH_OK) return rc;
    if ((rc = ensure_device()) != YH_OK) return rc;
    yh::LossParams p;
    plan_loss(p, cfg, batch, pad);
    for (int s = 0; s < p.S; ++s) { p.sc[s].head = heads[s]; p.sc[s].grad = grad_heads[s]; }
    p.w_coord = loss_scales[0]; p.w_noobj = loss_scales[1]; p.w_obj = loss_scales[2]; p.w_class = loss_scales[3];
    p.normalize = normalize_by_num_gt ? 1 : 0;
    p.gt = ground_truth;
    p.noobj_mask = const_cast<uint8_t *>(noobj_mask);
    p.grad_w = grad_per_image;
    const hipError_t e = yh::launch_loss_backward(p, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "loss backward launch");
    return YH_OK;
}

}  // extern "C"
