/* libyolohead — MI355X (gfx950) detection-head path of YOLO-TF behind a C ABI.
 *
 * The reference (Stick-To/YOLO-TF) has no FFI of its own: the path is inline
 * TensorFlow graph code inside YOLOv3._build_graph / YOLOv2._build_graph.  Each
 * entry point below names the graph region it replaces.
 *
 * Conventions
 *  - every function returns 0 on success and a negative yh_status otherwise;
 *    yh_last_error() gives the thread-local message of the last failure;
 *    nothing throws across this boundary;
 *  - every `const float*` / `float*` / `int32_t*` / `void* workspace` argument
 *    is a DEVICE pointer owned by the caller unless marked "host";
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *    all work is enqueued on it and no entry point synchronises with the host;
 *  - calls on different (device, stream, workspace) triples may run
 *    concurrently; the only process-wide setting is the launch-plan tuning of
 *    yh_set_tuning (never the process environment).  The library keeps two
 *    device arrays of its own per GPU: the scratch of yh_detect_static and a
 *    small ring of counters used by yh_loss (zero between calls, one cell per
 *    call in flight);
 *  - heads are the NHWC head tensors [B][H][W][P*(C+5)] fp32, one per scale,
 *    channel order per prior = C class logits, t_y, t_x, t_h, t_w, objectness
 *    (YOLOv3.py:95-102, YOLOv2.py:96-100); `heads` itself is a HOST array of
 *    num_scales device pointers; each must be 16-byte aligned;
 *  - box rows of an image are ordered scale 0, 1, 2 ... and (y*W + x)*P + p
 *    inside a scale (YOLOv3.py:320-349); boxes are (y1, x1, y2, x2).
 */
#ifndef YOLOHEAD_H
#define YOLOHEAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YH_MAX_SCALES 4
#define YH_MAX_PRIORS 16
#define YH_MAX_CLASSES 4096          /* class id takes 12 bits of a candidate key */
#define YH_MAX_BOXES_PER_IMAGE (1 << 20) /* box index takes 20 bits of a candidate key */
#define YH_MAX_NMS_BOXES 512         /* upper bound of nms_max_boxes */

typedef enum yh_status {
    YH_OK = 0,
    YH_ERR_INVALID_ARG = -1,
    YH_ERR_NO_DEVICE = -2,
    YH_ERR_HIP = -3,
    YH_ERR_WORKSPACE = -4,
    YH_ERR_LIMIT = -5,
    YH_ERR_ALIGN = -6
} yh_status;

/* bits of the per-image device-side status words written by yh_detect / yh_nms */
#define YH_STATUS_CANDIDATE_OVERFLOW 1 /* an image had more candidates than max_candidates; result invalid for it */
#define YH_STATUS_COUNT_SHIFT 8        /* with the overflow bit: bits 8..30 = the image's candidates / 256, rounded up
                                          (saturating), so that the caller can size max_candidates for a second call */

/* Geometry of the head, the data that distinguishes YOLOv3 from YOLOv2:
 *  prior_hw  : (h, w) in grid units.  YOLOv3: config priors / [8,16,32] with
 *              list entry 0 paired with the first grid (YOLOv3.py:37-41,
 *              111-113); YOLOv2: config priors unscaled (YOLOv2.py:33-35);
 *  box_mult  : decode multiplier, YOLOv3 32,32,16 (YOLOv3.py:346-348),
 *              YOLOv2 32 (YOLOv2.py:191);
 *  gt_rate   : ground-truth normalisation rate in the loss, YOLOv3 32,16,8
 *              (YOLOv3.py:118-120), YOLOv2 32 (YOLOv2.py:106). */
typedef struct yh_head_config {
    int32_t num_scales;
    int32_t num_priors;
    int32_t num_classes;
    int32_t grid_h[YH_MAX_SCALES];
    int32_t grid_w[YH_MAX_SCALES];
    float box_mult[YH_MAX_SCALES];
    float gt_rate[YH_MAX_SCALES];
    float prior_hw[YH_MAX_SCALES][YH_MAX_PRIORS][2];
} yh_head_config;

int yh_version(void);
const char *yh_last_error(void);

/* Launch-plan knobs of the streaming pass of yh_detect*, for benchmarks and
 * tests that compare the kernel variants; 0 in a field = the built-in default.
 * No counterpart in the reference.  Process-wide; set it while no call is in
 * progress, and query workspace sizes afterwards: yh_detect_workspace_bytes
 * depends on cfg, batch, the bounds and this tuning, and on nothing else.
 *   detect_kernel   1: objectness-first rows kernel (default); 2: streaming
 *                   kernels that read the whole head;
 *   row_steps       rows kernel: steps of 64 rows per work unit (default: the fewest
 *                   with which the launch is resident at once, 2..16);
 *   stream_tile_kib streaming kernel: wave-private tiles of 4 or 8 KiB
 *                   (default 8), anything else = workgroup tiles;
 *   stream_buffers  wave-private tiles in flight per wavefront, 2..4 (default 3);
 *   stream_units    wave-private variant: target number of work units per launch;
 *   stream_blocks   workgroup-tile variant: target number of workgroups' worth of
 *                   tiles the pass is cut into;
 *   stream_resident workgroup-tile variant: workgroups launched per compute unit,
 *                   each walking the cuts in turn (default: one workgroup per
 *                   cut, as many launched as there are cuts);
 *   rows_dense_min  rows kernel: a work unit of whose sixteen sampled rows at
 *                   least this many in 64 pass the objectness prefilter is read
 *                   whole, as contiguous spans, instead of row by row (default:
 *                   from the row width, 40 at 85 floats per row; 65 = never);
 *   rows_probe      rows kernel: 1 (default) units sample their rows as above;
 *                   2: no unit is read whole. */
typedef struct yh_tuning {
    int32_t detect_kernel;
    int32_t row_steps;
    int32_t stream_tile_kib;
    int32_t stream_buffers;
    int32_t stream_units;
    int32_t stream_blocks;
    int32_t stream_resident;
    int32_t rows_dense_min;
    int32_t rows_probe;
} yh_tuning;
int yh_set_tuning(const yh_tuning *tuning);   /* NULL restores the defaults */
int yh_get_tuning(yh_tuning *tuning);

/* Number of visible HIP devices (host out).  YH_ERR_NO_DEVICE when the HIP
 * runtime reports none: there is no CPU fallback. */
int yh_device_count(int32_t *count);

/* Streams `bytes` bytes at device pointer `src` (16-byte aligned) through a
 * kernel that only reads them.  Measurement hook (no counterpart in the
 * reference): timed by a benchmark it gives the read bandwidth this chip
 * delivers to a kernel, the yardstick next to the nominal HBM peak. */
int yh_probe_read(const void *src, size_t bytes, void *stream);

/* N = sum_s H_s * W_s * P (host out). */
int yh_boxes_per_image(const yh_head_config *cfg, int64_t *n);

/* Decode of every box: replaces YOLOv3.py:338-350 / YOLOv2.py:183-192, applied
 * to every image of the batch.
 *   boxes [B][N][4]; conf [B][N][C] = sigmoid(class) * sigmoid(obj), may be NULL. */
int yh_decode(const yh_head_config *cfg, const float *const *heads, int32_t batch,
              float *boxes, float *conf, void *stream);

/* Workspace size (host out) for yh_detect with these bounds.
 * The workspace of yh_detect / yh_nms must be ZERO-FILLED once after it is
 * allocated (hipMemset); every call leaves it in a state the next call, with
 * the same arguments to the size query, can start from.  The fill must be
 * complete, or ordered on `stream`, before the first call: a hipMemsetAsync on
 * another stream is not.  A workspace must not be shared by calls that overlap
 * in time.
 * max_candidates <= 2048 (with num_classes <= 1024, nms_max_boxes <= 128 and
 * num_classes * nms_max_boxes <= 4096) selects a single-launch form of the
 * candidates -> detections stage; larger capacities take a three-launch form.
 * Results are identical.  The size also depends on the launch plan of the
 * streaming pass, which the library derives from cfg, batch and yh_set_tuning. */
int yh_detect_workspace_bytes(const yh_head_config *cfg, int32_t batch,
                              int32_t max_candidates, int32_t nms_max_boxes, size_t *bytes);

/* test_one_image's graph from the head tensors to detection_pred, for every
 * image of the batch: replaces YOLOv3.py:320-368 / YOLOv2.py:183-210 (decode,
 * conf >= score_threshold filter, per-class tf.image.non_max_suppression,
 * gather, concat).  The confidence matrix is never materialised.
 *   max_candidates : capacity for (box, class) pairs passing the score filter,
 *                    per image; exceeding it sets YH_STATUS_CANDIDATE_OVERFLOW;
 *   out_count   [B]                       detections K_b of image b;
 *   out_scores  [B][C*nms_max_boxes]      first K_b valid; classes ascending,
 *   out_boxes   [B][C*nms_max_boxes][4]   selection order inside a class
 *   out_class   [B][C*nms_max_boxes]      (YOLOv3.py:362-367);
 *   out_index   [B][C*nms_max_boxes]      box row n of each detection;
 *   out_class_candidates [B][C]           per-class count after the score
 *                                         filter (YOLOv3.py:356); may be NULL;
 *   status      [B]                       per-image status word, written by the call.
 * Equal scores inside a class are taken in ascending box row. */
int yh_detect(const yh_head_config *cfg, const float *const *heads, int32_t batch,
              float score_threshold, float iou_threshold, int32_t nms_max_boxes,
              int32_t max_candidates, void *workspace, size_t workspace_bytes,
              int32_t *out_count, float *out_scores, float *out_boxes,
              int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
              int32_t *status, void *stream);

/* yh_detect without a workspace argument, for callers that cannot allocate or
 * carry one: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  It has no counterpart in the reference.  The workspace is
 * a device array of YH_STATIC_SCRATCH_BYTES inside the library's code object
 * (every GPU the library runs on holds its own copy until the process ends);
 * the call allocates and frees nothing.  Arguments and results are those of
 * yh_detect; a call whose yh_detect_workspace_bytes exceeds the array fails
 * with YH_ERR_LIMIT (e.g. it holds batch 64 of YOLOv3-608 at max_candidates
 * 512).  All calls on one GPU share the array: two calls that overlap in time
 * on one GPU (different streams or threads) race; calls on one stream, of any
 * shapes in any order, are fine (the call clears the counters it needs at
 * zero itself).  The device of `stream` must be the current device.  The
 * first call on a device looks the array's address up (hipGetSymbolAddress):
 * make it before a graph capture begins. */
#define YH_STATIC_SCRATCH_BYTES (8u << 20)
int yh_detect_static(const yh_head_config *cfg, const float *const *heads, int32_t batch,
                     float score_threshold, float iou_threshold, int32_t nms_max_boxes,
                     int32_t max_candidates,
                     int32_t *out_count, float *out_scores, float *out_boxes,
                     int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
                     int32_t *status, void *stream);

/* yh_detect with two caller-created hipEvent_t (as void*, either may be NULL)
 * recorded on `stream` immediately before and after the streaming kernel
 * (decode + score filter + compaction), so that a benchmark can time the
 * dominant kernel alone.  Measurement hook: no counterpart in the reference. */
int yh_detect_timed(const yh_head_config *cfg, const float *const *heads, int32_t batch,
                    float score_threshold, float iou_threshold, int32_t nms_max_boxes,
                    int32_t max_candidates, void *workspace, size_t workspace_bytes,
                    int32_t *out_count, float *out_scores, float *out_boxes,
                    int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
                    int32_t *status, void *stream, void *event_begin, void *event_end);

/* yh_detect as two calls that may go to two streams, so that the short
 * candidates -> detections stage of one batch overlaps the streaming pass of
 * the next (the caller orders them with events and gives batches that overlap
 * different workspaces), or run on one stream back to back:
 *   yh_detect_candidates  YOLOv3.py:338-357 / YOLOv2.py:183-199: decode and
 *                         conf >= score_threshold filter; leaves the candidates
 *                         (key and box) in the workspace;
 *   yh_detect_select      YOLOv3.py:355-368 / YOLOv2.py:197-210: per-class NMS,
 *                         gather, concat, from that workspace.
 * Both take the arguments of the same name of yh_detect, with the same values
 * for the pair; results are those of yh_detect.  With max_candidates beyond the
 * single-launch form yh_detect_select reads the boxes it looks at from the head
 * tensors of the yh_detect_candidates call on that workspace (leave them as
 * they are until it has run); the library keeps that call on record for the 64
 * workspaces streamed into most recently and fails with YH_ERR_INVALID_ARG
 * for a workspace that is not among them. */
int yh_detect_candidates(const yh_head_config *cfg, const float *const *heads, int32_t batch,
                         float score_threshold, int32_t nms_max_boxes, int32_t max_candidates,
                         void *workspace, size_t workspace_bytes, void *stream);
int yh_detect_select(const yh_head_config *cfg, int32_t batch,
                     float iou_threshold, int32_t nms_max_boxes, int32_t max_candidates,
                     void *workspace, size_t workspace_bytes,
                     int32_t *out_count, float *out_scores, float *out_boxes,
                     int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
                     int32_t *status, void *stream);

/* yh_detect software-pipelined over a sequence of equally shaped batches on ONE
 * stream: call i streams batch i (yh_detect_candidates, into `workspace`) and
 * selects batch i-1 (yh_detect_select, from `prev_workspace`, the workspace of
 * call i-1, into the outputs) inside the same kernel launch, where the short
 * latency-bound select stage hides behind the bandwidth-bound streaming pass.
 *   prev_workspace == NULL : nothing to select (first call);
 *   heads == NULL          : nothing to stream (drains the last batch).
 * The outputs and status belong to batch i-1.  Both workspaces have
 * workspace_bytes bytes; the caller alternates between two.  When the select
 * stage's bounds do not allow it to ride in the streaming launch (see
 * yh_detect_workspace_bytes: the single-launch form, and about 40 KiB of LDS,
 * e.g. max_candidates <= 704 at 80 classes x 10 boxes) the two stages run one
 * after the other, with the same results.  event_begin / event_end as in
 * yh_detect_timed.  No counterpart in the reference, whose session.run is
 * synchronous; results per batch are those of yh_detect. */
int yh_detect_pipelined(const yh_head_config *cfg, const float *const *heads, int32_t batch,
                        float score_threshold, float iou_threshold, int32_t nms_max_boxes,
                        int32_t max_candidates, void *workspace, void *prev_workspace,
                        size_t workspace_bytes,
                        int32_t *out_count, float *out_scores, float *out_boxes,
                        int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
                        int32_t *status, void *stream, void *event_begin, void *event_end);

/* The same pipeline entered after the decode: replaces YOLOv3.py:351-368 /
 * YOLOv2.py:193-210 on caller-supplied boxes [B][N][4] and conf [B][N][C].
 * Workspace: yh_nms_workspace_bytes. */
int yh_nms_workspace_bytes(int32_t batch, int64_t boxes_per_image, int32_t num_classes,
                           int32_t max_candidates, int32_t nms_max_boxes, size_t *bytes);
int yh_nms(int32_t batch, int64_t boxes_per_image, int32_t num_classes,
           const float *boxes, const float *conf,
           float score_threshold, float iou_threshold, int32_t nms_max_boxes,
           int32_t max_candidates, void *workspace, size_t workspace_bytes,
           int32_t *out_count, float *out_scores, float *out_boxes,
           int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
           int32_t *status, void *stream);

/* yh_nms with two caller-created hipEvent_t (as void*, either may be NULL)
 * recorded on `stream` immediately before and after the pass over the
 * confidence matrix (score filter + candidate compaction), as yh_detect_timed
 * does for yh_detect.  Measurement hook: no counterpart in the reference. */
int yh_nms_timed(int32_t batch, int64_t boxes_per_image, int32_t num_classes,
                 const float *boxes, const float *conf,
                 float score_threshold, float iou_threshold, int32_t nms_max_boxes,
                 int32_t max_candidates, void *workspace, size_t workspace_bytes,
                 int32_t *out_count, float *out_scores, float *out_boxes,
                 int32_t *out_class, int32_t *out_index, int32_t *out_class_candidates,
                 int32_t *status, void *stream, void *event_begin, void *event_end);

/* Training loss of the batch up to the per-image values: replaces
 * YOLOv3.py:117-310 / YOLOv2.py:105-173 (GT normalisation, best-prior and
 * scale assignment, coordinate / class / object sums, the IoU ignore mask and
 * the no-object sum).
 *   ground_truth [B][pad][5]  (yc, xc, h, w, class) in pixels, -1 padded
 *                             (utils/image_augmentor.py:205-244);
 *   loss_scales (host) [4]    coord, noobj, obj, class (config order);
 *   ignore_threshold          0.5 (YOLOv3.py:301) or 0.6 (YOLOv2.py:171);
 *   normalize_by_num_gt       1: divide by the GT count (YOLOv3.py:308-309),
 *                             0: do not (YOLOv2.py:172);
 *   loss_per_image [B]; loss_parts [B][4] = coord, class, obj, noobj sums, may
 *   be NULL; iou_max [B][N] = ignore-mask IoU maximum of every box (NaN in
 *   cells holding a GT centre), may be NULL; loss_sum [2], may be NULL:
 *   [0] = sum over the batch's images (the value a data-parallel caller
 *   all-reduces before dividing by the global batch), [1] = that sum divided
 *   by `batch` in fp32 (reduce_mean over this call's images, YOLOv3.py:311:
 *   the total_loss of a single-process caller).
 *   After the call the first 16 bytes of `workspace` hold two doubles: the sum
 *   of the per-image losses before its rounding to fp32, and `batch`.  A
 *   data-parallel caller all-reduces (sum) that pair in place and divides:
 *   reduce_mean over the global batch without a staging buffer of its own. */
int yh_loss_workspace_bytes(const yh_head_config *cfg, int32_t batch, int32_t pad, size_t *bytes);

/* Optional first half of a loss call: everything the loss takes from the ground
 * truth alone (YOLOv3.py:118-131, 246-257, 435-442: the rows' count, their
 * corners and areas in every scale's grid units, the cells that hold a centre),
 * written into `workspace` by a small launch of its own.  A training loop
 * issues it as soon as a batch's ground truth is on the device, ahead of the
 * convolutions, on the stream of the loss call; the NEXT yh_loss /
 * yh_loss_forward / yh_loss_with_grad call on that workspace then skips that
 * work in every one of its workgroups if it is a call for the same cfg,
 * ground_truth pointer, batch and pad (and takes its usual path, same results
 * bit for bit, if it is not).  The ground truth must not change in between,
 * and neither may the workspace's memory be released or rewritten: the record
 * is by address, and a loss call that matches it trusts what it finds there.
 * The library keeps the record of prepared workspaces (a small table, the
 * oldest entry is dropped); the same workspace size as yh_loss. */
int yh_loss_prepare(const yh_head_config *cfg, const float *ground_truth, int32_t batch, int32_t pad,
                    void *workspace, size_t workspace_bytes, void *stream);
int yh_loss(const yh_head_config *cfg, const float *const *heads, const float *ground_truth,
            int32_t batch, int32_t pad, const float *loss_scales, float ignore_threshold,
            int32_t normalize_by_num_gt, void *workspace, size_t workspace_bytes,
            float *loss_per_image, float *loss_parts, float *iou_max, float *loss_sum,
            void *stream);

/* The loss with its backward: what `optimizer.minimize(loss)` differentiates at
 * YOLOv3.py:312-318 / YOLOv2.py:175-181, up to the head tensors (the network
 * behind them is outside this path).
 *   yh_loss_forward  = yh_loss, and records in noobj_mask [B][N] (uint8) which
 *                      boxes entered the no-object sum (cell without a GT centre
 *                      and IoU maximum <= ignore_threshold; YOLOv3.py:298-307);
 *   yh_loss_backward   writes d(sum_b grad_per_image[b] * loss_per_image[b]) /
 *                      d(head) into grad_heads (host array of num_scales device
 *                      pointers, shaped like heads, completely overwritten; must
 *                      not alias heads).  heads, ground_truth, loss_scales,
 *                      normalize_by_num_gt and noobj_mask are those of the
 *                      forward call.  The ignore mask and the target assignment
 *                      are piecewise constant in the head, as in the reference's
 *                      graph (they depend on priors and ground truth only), so
 *                      the gradient is closed-form: w * (sigmoid(x) - target)
 *                      for the cross-entropy terms, w * (t - log(g / prior)) for
 *                      the size terms, divided by the GT count when
 *                      normalize_by_num_gt is set. */
int yh_loss_forward(const yh_head_config *cfg, const float *const *heads, const float *ground_truth,
                    int32_t batch, int32_t pad, const float *loss_scales, float ignore_threshold,
                    int32_t normalize_by_num_gt, void *workspace, size_t workspace_bytes,
                    float *loss_per_image, float *loss_parts, float *iou_max, float *loss_sum,
                    uint8_t *noobj_mask, void *stream);
int yh_loss_backward(const yh_head_config *cfg, const float *const *heads, const float *ground_truth,
                     int32_t batch, int32_t pad, const float *loss_scales, int32_t normalize_by_num_gt,
                     const uint8_t *noobj_mask, const float *grad_per_image, float *const *grad_heads,
                     void *stream);

/* Both at once, for a training step whose upstream gradient is known before the
 * forward pass (grad_per_image [B]: the constant 1/batch of `reduce_mean`,
 * YOLOv3.py:311-318, times any loss scaling): yh_loss's outputs and
 * yh_loss_backward's grad_heads from one pass over the rows.  The gradient
 * tensor's zeros are written while the ignore-mask loop computes, instead of by a
 * launch of their own afterwards; results are those of yh_loss_forward followed
 * by yh_loss_backward, bit for bit.  grad_heads must not alias heads. */
int yh_loss_with_grad(const yh_head_config *cfg, const float *const *heads, const float *ground_truth,
                      int32_t batch, int32_t pad, const float *loss_scales, float ignore_threshold,
                      int32_t normalize_by_num_gt, void *workspace, size_t workspace_bytes,
                      float *loss_per_image, float *loss_parts, float *iou_max, float *loss_sum,
                      const float *grad_per_image, float *const *grad_heads, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLOHEAD_H */
