/* CPU restatement (plain C, fp32) of the YOLO-TF detection-head path.
 *
 * TEST INFRASTRUCTURE ONLY: linked by tests/, by the smoke check and by the
 * cpu_baseline leg of bench.py; never by the shipped library.
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors for this path
 * and its arithmetic runs inside TensorFlow 1.12 (absent here).  This file is a
 * second restatement, written as scalar loops independently of the NumPy one
 * (yolo_head_oracle.py); the two are cross-checked against each other.
 *
 * Reference lines followed are cited per function (paths relative to the
 * reference checkout).  Build: see oracle/Makefile (-O2 -ffp-contract=off, no
 * fast-math, so + - * / min max round exactly as IEEE fp32 says).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define YO_MAX_SCALES 4
#define YO_MAX_PRIORS 16

/* Same field order as the shipped library's head description, so that one
 * ctypes structure serves both in the tests. */
typedef struct {
    int32_t num_scales;
    int32_t num_priors;
    int32_t num_classes;
    int32_t grid_h[YO_MAX_SCALES];
    int32_t grid_w[YO_MAX_SCALES];
    float box_mult[YO_MAX_SCALES];
    float gt_rate[YO_MAX_SCALES];
    float prior_hw[YO_MAX_SCALES][YO_MAX_PRIORS][2];
} yo_head_config;

static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log(1+exp(-|x|)). */
static float sigmoid_ce_(float x, float z)
{
    return (fmaxf(x, 0.0f) - x * z) + log1pf(expf(-fabsf(x)));
}

static int64_t boxes_per_image_(const yo_head_config *c)
{
    int64_t n = 0;
    for (int s = 0; s < c->num_scales; ++s)
        n += (int64_t)c->grid_h[s] * c->grid_w[s] * c->num_priors;
    return n;
}

int64_t yo_boxes_per_image(const yo_head_config *c) { return boxes_per_image_(c); }

/* ---------------------------------------------------------------- decode */

/* YOLOv3.py:338-350, YOLOv2.py:183-192 for image b.  boxes [N][4], conf [N][C]
 * (conf may be NULL). */
static void decode_image_(const yo_head_config *c, const float *const *heads, int b,
                          float *boxes, float *conf)
{
    const int C = c->num_classes, P = c->num_priors, K = C + 5;
    int64_t n = 0;
    for (int s = 0; s < c->num_scales; ++s) {
        const int H = c->grid_h[s], W = c->grid_w[s];
        const float *img = heads[s] + (int64_t)b * H * W * P * K;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int p = 0; p < P; ++p, ++n) {
                    const float *r = img + (((int64_t)y * W + x) * P + p) * K;
                    const float ay = (float)y + 0.5f, ax = (float)x + 0.5f;
                    const float by = ay + sigmoidf_(r[C]);
                    const float bx = ax + sigmoidf_(r[C + 1]);
                    const float bh = c->prior_hw[s][p][0] + expf(r[C + 2]);
                    const float bw = c->prior_hw[s][p][1] + expf(r[C + 3]);
                    const float m = c->box_mult[s];
                    boxes[n * 4 + 0] = (by - bh / 2.0f) * m;
                    boxes[n * 4 + 1] = (bx - bw / 2.0f) * m;
                    boxes[n * 4 + 2] = (by + bh / 2.0f) * m;
                    boxes[n * 4 + 3] = (bx + bw / 2.0f) * m;
                    if (conf) {
                        const float so = sigmoidf_(r[C + 4]);
                        for (int k = 0; k < C; ++k)
                            conf[n * C + k] = sigmoidf_(r[k]) * so;
                    }
                }
    }
}

int yo_decode(const yo_head_config *c, const float *const *heads, int B,
              float *boxes, float *conf, int threads)
{
    const int64_t N = boxes_per_image_(c);
    const int C = c->num_classes;
    (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
        decode_image_(c, heads, b, boxes + (int64_t)b * N * 4,
                      conf ? conf + (int64_t)b * N * C : NULL);
    return 0;
}

/* ------------------------------------------------------------------- NMS */

/* NonMaxSuppressionV3's overlap test (tensorflow/core/kernels/
 * non_max_suppression_op.cc, TF 1.12). */
static int nms_suppresses_(const float *bi, const float *bj, float thr)
{
    const float ymin_i = fminf(bi[0], bi[2]), xmin_i = fminf(bi[1], bi[3]);
    const float ymax_i = fmaxf(bi[0], bi[2]), xmax_i = fmaxf(bi[1], bi[3]);
    const float ymin_j = fminf(bj[0], bj[2]), xmin_j = fminf(bj[1], bj[3]);
    const float ymax_j = fmaxf(bj[0], bj[2]), xmax_j = fmaxf(bj[1], bj[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.0f || area_j <= 0.0f) return 0;
    const float iy = fmaxf(fminf(ymax_i, ymax_j) - fmaxf(ymin_i, ymin_j), 0.0f);
    const float ix = fmaxf(fminf(xmax_i, xmax_j) - fmaxf(xmin_i, xmin_j), 0.0f);
    const float inter = iy * ix;
    const float iou = inter / ((area_i + area_j) - inter);
    return iou > thr;
}

typedef struct { float score; int32_t idx; } cand_t;

/* descending score, then ascending index (the declared tie rule) */
static int cand_cmp_(const void *a, const void *b)
{
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* YOLOv3.py:351-368 / YOLOv2.py:193-210 for one image, given decoded boxes
 * [N][4] and conf [N][C].  Outputs hold C*max_boxes slots; returns K. */
static int nms_image_(int64_t N, int C, const float *boxes, const float *conf,
                      float score_thr, float iou_thr, int max_boxes,
                      float *o_scores, float *o_boxes, int32_t *o_class, int32_t *o_index,
                      cand_t *cand, int32_t *sel)
{
    int K = 0;
    for (int k = 0; k < C; ++k) {
        int64_t M = 0;
        for (int64_t n = 0; n < N; ++n)
            if (conf[n * C + k] >= score_thr) { cand[M].score = conf[n * C + k]; cand[M].idx = (int32_t)n; ++M; }
        qsort(cand, (size_t)M, sizeof(cand_t), cand_cmp_);
        int nsel = 0;
        const int out_size = max_boxes < M ? max_boxes : (int)M;
        for (int64_t i = 0; i < M && nsel < out_size; ++i) {
            const float *bi = boxes + (int64_t)cand[i].idx * 4;
            int keep = 1;
            for (int j = nsel - 1; j >= 0; --j)
                if (nms_suppresses_(bi, boxes + (int64_t)sel[j] * 4, iou_thr)) { keep = 0; break; }
            if (keep) {
                sel[nsel++] = cand[i].idx;
                o_scores[K] = cand[i].score;
                memcpy(o_boxes + (int64_t)K * 4, bi, 4 * sizeof(float));
                o_class[K] = k;
                o_index[K] = cand[i].idx;
                ++K;
            }
        }
    }
    return K;
}

/* Per-class NMS of already decoded inputs: boxes [B][N][4], conf [B][N][C]. */
int yo_nms(int B, int64_t N, int C, const float *boxes, const float *conf,
           float score_thr, float iou_thr, int max_boxes,
           int32_t *out_count, float *out_scores, float *out_boxes,
           int32_t *out_class, int32_t *out_index, int threads)
{
    if (!(iou_thr >= 0.0f && iou_thr <= 1.0f)) return -1;
    const int64_t cap = (int64_t)C * max_boxes;
    (void)threads;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)(N > 0 ? N : 1));
        int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_boxes > 0 ? max_boxes : 1));
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b)
            out_count[b] = nms_image_(N, C, boxes + (int64_t)b * N * 4, conf + (int64_t)b * N * C,
                                      score_thr, iou_thr, max_boxes,
                                      out_scores + b * cap, out_boxes + b * cap * 4,
                                      out_class + b * cap, out_index + b * cap, cand, sel);
        free(cand); free(sel);
    }
    return 0;
}

/* test_one_image's graph, image by image: decode then per-class NMS. */
int yo_detect(const yo_head_config *c, const float *const *heads, int B,
              float score_thr, float iou_thr, int max_boxes,
              int32_t *out_count, float *out_scores, float *out_boxes,
              int32_t *out_class, int32_t *out_index, int threads)
{
    if (!(iou_thr >= 0.0f && iou_thr <= 1.0f)) return -1;
    const int64_t N = boxes_per_image_(c);
    const int C = c->num_classes;
    const int64_t cap = (int64_t)C * max_boxes;
    (void)threads;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        float *boxes = (float *)malloc(sizeof(float) * (size_t)N * 4);
        float *conf = (float *)malloc(sizeof(float) * (size_t)N * C);
        cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)(N > 0 ? N : 1));
        int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_boxes > 0 ? max_boxes : 1));
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            decode_image_(c, heads, b, boxes, conf);
            out_count[b] = nms_image_(N, C, boxes, conf, score_thr, iou_thr, max_boxes,
                                      out_scores + b * cap, out_boxes + b * cap * 4,
                                      out_class + b * cap, out_index + b * cap, cand, sel);
        }
        free(boxes); free(conf); free(cand); free(sel);
    }
    return 0;
}

/* ------------------------------------------------------------------ loss */

/* _get_normlized_gn's row count (YOLOv3.py:437, YOLOv2.py:248): the first
 * index of the minimum of column 0. */
static int gt_count_(const float *gt, int pad)
{
    int best = 0;
    for (int j = 1; j < pad; ++j)
        if (gt[j * 5] < gt[best * 5]) best = j;
    return pad > 0 ? best : 0;
}

/* One image of YOLOv3.py:117-310 (normalize_by_num_gt = 1, three scales) or
 * YOLOv2.py:105-173 (normalize_by_num_gt = 0, one scale), with the scale
 * count, rates, priors and the ignore threshold taken from the arguments.
 * A GT goes to scale s < S-1 when its best IoU there beats every other
 * scale's strictly, otherwise to the last scale (YOLOv3.py:185-189).
 * parts[4] = coord, class, obj, noobj sums; iou_max [N] (optional) receives the
 * ignore-mask IoU maximum (NaN in cells that hold a GT centre). */
static float loss_image_(const yo_head_config *c, const float *const *heads, int b,
                         const float *gt, int pad, const float wts[4], float ignore_thr,
                         int normalize_by_num_gt, float *parts, float *iou_max_out)
{
    const int S = c->num_scales, C = c->num_classes, P = c->num_priors, K = C + 5;
    const int G = gt_count_(gt, pad);
    float *best_iou = (float *)malloc(sizeof(float) * (size_t)(G > 0 ? G : 1) * S);
    int *best_p = (int *)malloc(sizeof(int) * (size_t)(G > 0 ? G : 1) * S);
    float *g1 = (float *)malloc(sizeof(float) * (size_t)(G > 0 ? G : 1) * 5);

    /* positive branch, part 1: best prior of every GT at every scale
     * (YOLOv3.py:132-184) */
    for (int s = 0; s < S; ++s) {
        const float rate = c->gt_rate[s];
        for (int g = 0; g < G; ++g) {
            const float gy = gt[g * 5 + 0] / rate, gx = gt[g * 5 + 1] / rate;
            const float gh = gt[g * 5 + 2] / rate, gw = gt[g * 5 + 3] / rate;
            const float cy = floorf(gy), cx = floorf(gx);
            const float ay = cy + 0.5f, ax = cx + 0.5f;
            const float gy1 = gy - gh / 2.0f, gx1 = gx - gw / 2.0f;
            const float gy2 = gy + gh / 2.0f, gx2 = gx + gw / 2.0f;
            const float garea = (gy2 - gy1) * (gx2 - gx1);
            float bi = 0.0f; int bp = 0;
            for (int p = 0; p < P; ++p) {
                const float ah = c->prior_hw[s][p][0], aw = c->prior_hw[s][p][1];
                const float ay1 = ay - ah / 2.0f, ax1 = ax - aw / 2.0f;
                const float ay2 = ay + ah / 2.0f, ax2 = ax + aw / 2.0f;
                const float inter = (fminf(gy2, ay2) - fmaxf(gy1, ay1)) * (fminf(gx2, ax2) - fmaxf(gx1, ax1));
                const float iou = inter / ((ah * aw + garea) - inter);
                if (p == 0 || iou > bi) { bi = iou; bp = p; }   /* first maximum */
            }
            best_iou[s * G + g] = bi; best_p[s * G + g] = bp;
        }
    }

    double coord = 0.0, cls = 0.0, obj = 0.0, noobj = 0.0;
    /* positive branch, part 2: scale choice and the four sums
     * (YOLOv3.py:185-241, YOLOv2.py:128-145) */
    for (int g = 0; g < G; ++g) {
        int s_sel = S - 1;
        for (int s = 0; s < S - 1; ++s) {
            int wins = 1;
            for (int t = 0; t < S; ++t)
                if (t != s && !(best_iou[s * G + g] > best_iou[t * G + g])) wins = 0;
            if (wins) { s_sel = s; break; }
        }
        const int s = s_sel, p = best_p[s * G + g];
        const float rate = c->gt_rate[s];
        const int H = c->grid_h[s], W = c->grid_w[s];
        const float gy = gt[g * 5 + 0] / rate, gx = gt[g * 5 + 1] / rate;
        const float gh = gt[g * 5 + 2] / rate, gw = gt[g * 5 + 3] / rate;
        const int label = (int)(gt[g * 5 + 4]);               /* tf.cast float->int32 truncates */
        const int cy = (int)floorf(gy), cx = (int)floorf(gx);
        if (cy < 0 || cy >= H || cx < 0 || cx >= W) continue;  /* outside the grid: the reference's gather would fail */
        const float *r = heads[s] + ((((int64_t)b * H + cy) * W + cx) * P + p) * K;
        const float ty = gy - floorf(gy), tx = gx - floorf(gx);
        const float th = logf(gh / c->prior_hw[s][p][0]), tw = logf(gw / c->prior_hw[s][p][1]);
        coord += sigmoid_ce_(r[C], ty); coord += sigmoid_ce_(r[C + 1], tx);
        const float dh = r[C + 2] - th, dw = r[C + 3] - tw;
        coord += 0.5f * (dh * dh); coord += 0.5f * (dw * dw);
        for (int k = 0; k < C; ++k) cls += sigmoid_ce_(r[k], k == label ? 1.0f : 0.0f);
        obj += sigmoid_ce_(r[C + 4], 1.0f);
    }

    /* negative branch (YOLOv3.py:242-307, YOLOv2.py:146-171) */
    int64_t nbase = 0;
    for (int s = 0; s < S; ++s) {
        const float rate = c->gt_rate[s];
        const int H = c->grid_h[s], W = c->grid_w[s];
        unsigned char *occ = (unsigned char *)calloc((size_t)H * W, 1);
        for (int g = 0; g < G; ++g) {
            const float gy = gt[g * 5 + 0] / rate, gx = gt[g * 5 + 1] / rate;
            const float gh = gt[g * 5 + 2] / rate, gw = gt[g * 5 + 3] / rate;
            const int cy = (int)floorf(gy), cx = (int)floorf(gx);
            if (cy >= 0 && cy < H && cx >= 0 && cx < W) occ[cy * W + cx] = 1;
            g1[g * 5 + 0] = gy - gh / 2.0f; g1[g * 5 + 1] = gx - gw / 2.0f;
            g1[g * 5 + 2] = gy + gh / 2.0f; g1[g * 5 + 3] = gx + gw / 2.0f;
            g1[g * 5 + 4] = (g1[g * 5 + 2] - g1[g * 5 + 0]) * (g1[g * 5 + 3] - g1[g * 5 + 1]);
        }
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int p = 0; p < P; ++p) {
                    const int64_t n = nbase + ((int64_t)y * W + x) * P + p;
                    if (occ[y * W + x]) { if (iou_max_out) iou_max_out[n] = NAN; continue; }
                    const float ay = (float)y + 0.5f, ax = (float)x + 0.5f;
                    const float ah = c->prior_hw[s][p][0], aw = c->prior_hw[s][p][1];
                    const float ly = ay - ah / 2.0f, lx = ax - aw / 2.0f;   /* "yx_nobest" */
                    const float uy = ay + ah / 2.0f, ux = ax + aw / 2.0f;   /* "hw_nobest" */
                    const float by1 = ly - uy / 2.0f, bx1 = lx - ux / 2.0f;
                    const float by2 = ly + uy / 2.0f, bx2 = lx + ux / 2.0f;
                    const float aarea = (by2 - by1) * (bx2 - bx1);
                    float m = -INFINITY; int isnan_ = 0;
                    for (int g = 0; g < G; ++g) {
                        const float inter = (fminf(g1[g * 5 + 2], by2) - fmaxf(g1[g * 5 + 0], by1)) *
                                            (fminf(g1[g * 5 + 3], bx2) - fmaxf(g1[g * 5 + 1], bx1));
                        const float iou = inter / ((aarea + g1[g * 5 + 4]) - inter);
                        if (iou != iou) isnan_ = 1;
                        else if (iou > m) m = iou;
                    }
                    if (isnan_) m = NAN;
                    if (iou_max_out) iou_max_out[n] = m;
                    if (m <= ignore_thr) {
                        const float *r = heads[s] + ((((int64_t)b * H + y) * W + x) * P + p) * K;
                        noobj += sigmoid_ce_(r[C + 4], 0.0f);
                    }
                }
        free(occ);
        nbase += (int64_t)H * W * P;
    }
    free(best_iou); free(best_p); free(g1);

    const float fc = (float)coord, fk = (float)cls, fo = (float)obj, fn = (float)noobj;
    if (parts) { parts[0] = fc; parts[1] = fk; parts[2] = fo; parts[3] = fn; }
    float pos = (wts[0] * fc + wts[3] * fk) + wts[2] * fo;   /* coord, class, obj weights */
    float neg = wts[1] * fn;                                  /* noobj weight */
    if (normalize_by_num_gt) { pos = pos / (float)G; neg = neg / (float)G; }
    return pos + neg;
}

/* wts = {coord_scale, noobj_scale, obj_scale, class_scale}.  loss_per_image [B];
 * parts [B][4] optional; iou_max [B][N] optional; *loss_mean = reduce_mean. */
int yo_loss(const yo_head_config *c, const float *const *heads, const float *gt,
            int B, int pad, const float *wts, float ignore_thr, int normalize_by_num_gt,
            float *loss_per_image, float *parts, float *iou_max, float *loss_mean, int threads)
{
    const int64_t N = boxes_per_image_(c);
    (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
        loss_per_image[b] = loss_image_(c, heads, b, gt + (int64_t)b * pad * 5, pad, wts, ignore_thr,
                                        normalize_by_num_gt, parts ? parts + b * 4 : NULL,
                                        iou_max ? iou_max + (int64_t)b * N : NULL);
    double acc = 0.0;
    for (int b = 0; b < B; ++b) acc += loss_per_image[b];
    if (loss_mean) *loss_mean = (float)(acc / (double)B);
    return 0;
}

int yo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
