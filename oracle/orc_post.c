/*
 * orc_post.c -- CPU oracle: preprocess, box/keypoint decode, NMS, parse_output, PnP.
 *
 * TEST INFRASTRUCTURE ONLY (see irmv_oracle.h).  PARITY UNPINNED against NPP,
 * EfficientNMS_TRT and OpenCV (closed or absent; the reference tests pin no
 * values).  Compiled with -ffp-contract=off: every fp32 expression below is
 * evaluated exactly as written (fmaf only where spelled out), so the HIP
 * kernels, which spell the same expressions under `#pragma clang fp
 * contract(off)`, can be compared bit for bit on integer/index results and on
 * the fp32 decode.
 */
#include "irmv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================= */
/* preprocess: reference src/yolo_engine.cpp:179-200                        */
/*   K1 nppiMirror both axes  -> rotate180 (folded into the sampling coords) */
/*   K2 nppiResize LINEAR     -> fixed-point bilinear, half-pixel centres    */
/*   K3 nppiScale 8u->32f 0..1-> v / 255                                     */
/*   K4 packed->planar        -> CHW output                                  */
/* NPP's exact rounding is closed source; the spec here (OpenCV-style 11-bit */
/* coefficients, round-half-up, edge clamp) is this build's own and is what  */
/* the HIP kernel must reproduce exactly.                                    */
/* ======================================================================= */
#define COEF_BITS 11
#define COEF_ONE (1 << COEF_BITS)

/* source index / weight of destination coordinate d along an axis of source
 * length sn mapped onto dn destination samples */
static void axis_tap(int d, int sn, int dn, int *i0, int *i1, int *w1)
{
    /* f = (d + 0.5) * sn / dn - 0.5 = ((2d+1)*sn - dn) / (2*dn) */
    long long num = (long long)(2 * d + 1) * sn - dn;
    long long den = 2LL * dn;
    long long fl = num >= 0 ? num / den : -((-num + den - 1) / den);
    long long frac = num - fl * den; /* in [0, den) */
    int w = (int)((frac * COEF_ONE + dn) / den); /* round half up */
    int a = (int)fl, b = (int)fl + 1;
    if (a < 0) { a = 0; b = 0; w = 0; }
    if (a >= sn - 1) { a = sn - 1; b = sn - 1; w = 0; }
    *i0 = a; *i1 = b; *w1 = w;
}

static void letterbox_geom(int sw, int sh, int net, int *nw, int *nh, int *px, int *py)
{
    /* r = min(net/sw, net/sh); new = round(s * r) */
    double r = (double)net / sw < (double)net / sh ? (double)net / sw : (double)net / sh;
    *nw = (int)floor(sw * r + 0.5);
    *nh = (int)floor(sh * r + 0.5);
    if (*nw > net) *nw = net;
    if (*nh > net) *nh = net;
    *px = (net - *nw) / 2;
    *py = (net - *nh) / 2;
}

void orc_preprocess(const uint8_t *src, int sw, int sh, int net, int mode, int rotate180,
                    int swap_rb, float *out_chw, uint8_t *out_u8)
{
    int nw = net, nh = net, px = 0, py = 0;
    if (mode == ORC_RESIZE_LETTERBOX) letterbox_geom(sw, sh, net, &nw, &nh, &px, &py);
    for (int dy = 0; dy < net; dy++) {
        for (int dx = 0; dx < net; dx++) {
            int v[3];
            int ry = dy - py, rx = dx - px;
            if (ry < 0 || ry >= nh || rx < 0 || rx >= nw) {
                v[0] = v[1] = v[2] = 114; /* letterbox pad colour */
            } else {
                int y0, y1, wy, x0, x1, wx;
                axis_tap(ry, sh, nh, &y0, &y1, &wy);
                axis_tap(rx, sw, nw, &x0, &x1, &wx);
                if (rotate180) { /* rotated(x, y) = src(sw-1-x, sh-1-y) */
                    y0 = sh - 1 - y0; y1 = sh - 1 - y1;
                    x0 = sw - 1 - x0; x1 = sw - 1 - x1;
                }
                for (int c = 0; c < 3; c++) {
                    int sc = swap_rb ? 2 - c : c;
                    uint32_t p00 = src[((size_t)y0 * sw + x0) * 3 + sc];
                    uint32_t p01 = src[((size_t)y0 * sw + x1) * 3 + sc];
                    uint32_t p10 = src[((size_t)y1 * sw + x0) * 3 + sc];
                    uint32_t p11 = src[((size_t)y1 * sw + x1) * 3 + sc];
                    uint32_t top = (uint32_t)(COEF_ONE - wx) * p00 + (uint32_t)wx * p01;
                    uint32_t bot = (uint32_t)(COEF_ONE - wx) * p10 + (uint32_t)wx * p11;
                    uint32_t acc = (uint32_t)(COEF_ONE - wy) * top + (uint32_t)wy * bot;
                    v[c] = (int)((acc + (1u << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS));
                }
            }
            for (int c = 0; c < 3; c++) {
                out_chw[((size_t)c * net + dy) * net + dx] = (float)v[c] / 255.0f;
                if (out_u8) out_u8[((size_t)dy * net + dx) * 3 + c] = (uint8_t)v[c];
            }
        }
    }
}

void orc_rotate180(const uint8_t *src, int sw, int sh, uint8_t *dst)
{
    for (int y = 0; y < sh; y++)
        for (int x = 0; x < sw; x++)
            memcpy(dst + ((size_t)y * sw + x) * 3, src + ((size_t)(sh - 1 - y) * sw + (sw - 1 - x)) * 3, 3);
}

/* ======================================================================= */
/* decode + NMS                                                            */
/* ======================================================================= */

/* exp(x) as a fixed sequence of fp32 operations (Cody-Waite reduction,
 * degree-6 Taylor, exponent insertion): identical bits on CPU and GPU. */
float orc_expf(float x)
{
    if (x < -87.0f) x = -87.0f;
    if (x > 88.0f) x = 88.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 720.0f;
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int32_t bits;
    memcpy(&bits, &p, 4);
    bits += (int32_t)n << 23;
    memcpy(&p, &bits, 4);
    return p;
}

/* score > thr  <=>  logit > log(thr / (1 - thr)); the comparison is done on the
 * logit so that it involves no transcendental on the data path. */
float orc_logit_threshold(float score_thr)
{
    double t = (double)score_thr;
    return (float)log(t / (1.0 - t));
}

static inline uint32_t orderable(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

static inline float unorderable(uint32_t u)
{
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static void level_of_anchor(int a, int net, int *ix, int *iy, int *stride)
{
    int base = 0;
    for (int l = 0; l < 3; l++) {
        int s = 8 << l, w = net / s, cnt = w * w;
        if (a < base + cnt) {
            int r = a - base;
            *ix = r % w; *iy = r / w; *stride = s;
            return;
        }
        base += cnt;
    }
    *ix = *iy = 0; *stride = 0;
}

/* DFL expectation over 16 bins */
static float dfl_side(const float *l)
{
    float m = l[0];
    for (int j = 1; j < 16; j++) m = l[j] > m ? l[j] : m;
    float se = 0.0f, sj = 0.0f;
    for (int j = 0; j < 16; j++) {
        float e = orc_expf(l[j] - m);
        se = se + e;
        sj = sj + e * (float)j;
    }
    return sj / se;
}

static void decode_box(const float *rec, int ix, int iy, int stride, float box[4])
{
    float ax = (float)ix + 0.5f, ay = (float)iy + 0.5f, s = (float)stride;
    float dl = dfl_side(rec + 0), dt = dfl_side(rec + 16), dr = dfl_side(rec + 32), db = dfl_side(rec + 48);
    box[0] = (ax - dl) * s;
    box[1] = (ay - dt) * s;
    box[2] = (ax + dr) * s;
    box[3] = (ay + db) * s;
}

static int cmp_key_desc(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? 1 : (x > y ? -1 : 0);
}

/* Candidate key: high 32 bits = order-preserving image of the class logit,
 * low 32 bits = ~(anchor * nc + class), so that a descending sort yields
 * score-descending order with ties broken by lower anchor, then lower class. */
int orc_decode_candidates(const float *head, int net, int nc, int nk, float score_thr,
                          float *boxes, uint64_t *keys, int keys_cap)
{
    const int A = orc_num_anchors(net), no = 64 + nc + nk;
    const float lt = orc_logit_threshold(score_thr);
    int n = 0;
    for (int a = 0; a < A; a++) {
        const float *rec = head + (size_t)a * no;
        int ix, iy, s;
        level_of_anchor(a, net, &ix, &iy, &s);
        if (boxes) decode_box(rec, ix, iy, s, boxes + (size_t)a * 4);
        for (int c = 0; c < nc; c++) {
            if (rec[64 + c] > lt) {
                if (keys && n < keys_cap)
                    keys[n] = ((uint64_t)orderable(rec[64 + c]) << 32) |
                              (uint64_t)(0xffffffffu - (uint32_t)(a * nc + c));
                n++;
            }
        }
    }
    if (keys) qsort(keys, n < keys_cap ? n : keys_cap, sizeof(uint64_t), cmp_key_desc);
    return n;
}

/* "IoU(a, b) > thr" without the division: inter > thr * union.  (Boxes decoded from
 * DFL distances have x2 >= x1, y2 >= y1, so union >= inter >= 0; union == 0 gives
 * false, as the quotient's NaN would.)  The HIP kernel evaluates the same
 * expression, so survivor sets are bit-reproducible. */
static inline int iou_gt(const float *a, const float *b, float thr)
{
    float ix1 = a[0] > b[0] ? a[0] : b[0];
    float iy1 = a[1] > b[1] ? a[1] : b[1];
    float ix2 = a[2] < b[2] ? a[2] : b[2];
    float iy2 = a[3] < b[3] ? a[3] : b[3];
    float iw = ix2 - ix1, ih = iy2 - iy1;
    iw = iw > 0.0f ? iw : 0.0f;
    ih = ih > 0.0f ? ih : 0.0f;
    float inter = iw * ih;
    float aa = (a[2] - a[0]) * (a[3] - a[1]);
    float ab = (b[2] - b[0]) * (b[3] - b[1]);
    float uni = (aa + ab) - inter;
    return inter > thr * uni;
}

/* EfficientNMS walk: emit unless an already emitted box
 * of the same class has IoU > iou_thr; stop at max_det. */
int orc_nms_sorted(const float *boxes, const int *classes, int n, float iou_thr, int max_det,
                   int *keep_idx)
{
    int kept = 0;
    for (int i = 0; i < n && kept < max_det; i++) {
        int ok = 1;
        for (int j = 0; j < kept; j++) {
            int q = keep_idx[j];
            if (classes[q] == classes[i] && iou_gt(boxes + (size_t)q * 4, boxes + (size_t)i * 4, iou_thr)) {
                ok = 0;
                break;
            }
        }
        if (ok) keep_idx[kept++] = i;
    }
    return kept;
}

int orc_decode_nms(const float *head, int net, int nc, int nk, float score_thr, float iou_thr,
                   int max_det, int pre_nms_cap, float *det_boxes, float *det_scores,
                   int *det_classes, int *det_anchor, float *det_kpts, int *n_candidates)
{
    const int A = orc_num_anchors(net), no = 64 + nc + nk;
    float *boxes = malloc((size_t)A * 4 * sizeof(float));
    uint64_t *keys = malloc((size_t)A * nc * sizeof(uint64_t));
    int n = orc_decode_candidates(head, net, nc, nk, score_thr, boxes, keys, A * nc);
    if (n_candidates) *n_candidates = n;
    if (n > pre_nms_cap) n = pre_nms_cap; /* keys are sorted: keeps the top pre_nms_cap */
    float *cb = malloc((size_t)(n > 0 ? n : 1) * 4 * sizeof(float));
    int *cc = malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    int *ca = malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    for (int i = 0; i < n; i++) {
        uint32_t id = 0xffffffffu - (uint32_t)(keys[i] & 0xffffffffu);
        ca[i] = (int)(id / (uint32_t)nc);
        cc[i] = (int)(id % (uint32_t)nc);
        memcpy(cb + (size_t)i * 4, boxes + (size_t)ca[i] * 4, 16);
    }
    int *keep = malloc((size_t)(max_det > 0 ? max_det : 1) * sizeof(int));
    int kept = orc_nms_sorted(cb, cc, n, iou_thr, max_det, keep);
    for (int j = 0; j < kept; j++) {
        int i = keep[j];
        memcpy(det_boxes + (size_t)j * 4, cb + (size_t)i * 4, 16);
        float logit = unorderable((uint32_t)(keys[i] >> 32));
        det_scores[j] = 1.0f / (1.0f + orc_expf(-logit));
        det_classes[j] = cc[i];
        if (det_anchor) det_anchor[j] = ca[i];
        if (det_kpts && nk > 0) {
            int ix, iy, s;
            level_of_anchor(ca[i], net, &ix, &iy, &s);
            const float *kp = head + (size_t)ca[i] * no + 64 + nc;
            /* k = (2 v + (anchor - 0.5)) * stride */
            for (int q = 0; q < nk / 2; q++) {
                float axm = ((float)ix + 0.5f) - 0.5f, aym = ((float)iy + 0.5f) - 0.5f;
                det_kpts[(size_t)j * nk + 2 * q] = (2.0f * kp[2 * q] + axm) * (float)s;
                det_kpts[(size_t)j * nk + 2 * q + 1] = (2.0f * kp[2 * q + 1] + aym) * (float)s;
            }
        }
    }
    free(boxes); free(keys); free(cb); free(cc); free(ca); free(keep);
    return kept;
}

/* reference src/yolo_engine.cpp:155-156, :202-220: xyxy * (W/640, H/640).
 * Letterbox mode (north-star variant): x = (x - pad_x) / r. */
void orc_parse_output(const float *det_boxes, int n, int src_w, int src_h, int net, int mode,
                      float *out_xyxy)
{
    if (mode == ORC_RESIZE_STRETCH) {
        float sx = (float)src_w / (float)net, sy = (float)src_h / (float)net;
        for (int i = 0; i < n; i++) {
            out_xyxy[i * 4 + 0] = det_boxes[i * 4 + 0] * sx;
            out_xyxy[i * 4 + 1] = det_boxes[i * 4 + 1] * sy;
            out_xyxy[i * 4 + 2] = det_boxes[i * 4 + 2] * sx;
            out_xyxy[i * 4 + 3] = det_boxes[i * 4 + 3] * sy;
        }
    } else {
        int nw, nh, px, py;
        letterbox_geom(src_w, src_h, net, &nw, &nh, &px, &py);
        float sx = (float)src_w / (float)nw, sy = (float)src_h / (float)nh;
        for (int i = 0; i < n; i++) {
            out_xyxy[i * 4 + 0] = (det_boxes[i * 4 + 0] - (float)px) * sx;
            out_xyxy[i * 4 + 1] = (det_boxes[i * 4 + 1] - (float)py) * sy;
            out_xyxy[i * 4 + 2] = (det_boxes[i * 4 + 2] - (float)px) * sx;
            out_xyxy[i * 4 + 3] = (det_boxes[i * 4 + 3] - (float)py) * sy;
        }
    }
}

/* ======================================================================= */
/* PnP: cv::solvePnP(..., SOLVEPNP_IPPE) restated     */
/* ======================================================================= */

/* reference src/pnp_solver.cpp:18-33: metres; model frame x forward, y left,
 * z up; order bottom-left, top-left, top-right, bottom-right. */
void orc_armor_object_points(int armor_size, double obj[12])
{
    const double hy = (armor_size == 1 ? 225.0 : 135.0) / 2.0 / 1000.0;
    const double hz = 55.0 / 2.0 / 1000.0;
    const double p[12] = { 0, hy, -hz, 0, hy, hz, 0, -hy, hz, 0, -hy, -hz };
    memcpy(obj, p, sizeof p);
}

/* cv::undistortPoints, plumb_bob 5 coefficients [k1 k2 p1 p2 k3], 5 fixed
 * iterations, no rectification / new camera matrix */
void orc_undistort_points(const double K[9], const double D[5], const float *pts, int n,
                          double *out)
{
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
    for (int i = 0; i < n; i++) {
        double x0 = ((double)pts[2 * i] - cx) / fx, y0 = ((double)pts[2 * i + 1] - cy) / fy;
        double x = x0, y = y0;
        for (int it = 0; it < 5; it++) {
            double r2 = x * x + y * y;
            double icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2);
            double dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
            double dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        out[2 * i] = x;
        out[2 * i + 1] = y;
    }
}

void orc_rodrigues(const double r[3], double R[9])
{
    double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < 1e-14) {
        const double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        memcpy(R, I, sizeof I);
        return;
    }
    double kx = r[0] / th, ky = r[1] / th, kz = r[2] / th, c = cos(th), s = sin(th), v = 1.0 - c;
    R[0] = c + kx * kx * v;      R[1] = kx * ky * v - kz * s; R[2] = kx * kz * v + ky * s;
    R[3] = ky * kx * v + kz * s; R[4] = c + ky * ky * v;      R[5] = ky * kz * v - kx * s;
    R[6] = kz * kx * v - ky * s; R[7] = kz * ky * v + kx * s; R[8] = c + kz * kz * v;
}

static void rot_to_rvec(const double R[9], double r[3])
{
    double tr = R[0] + R[4] + R[8];
    double c = (tr - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    double ax = R[7] - R[5], ay = R[2] - R[6], az = R[3] - R[1];
    double s = 0.5 * sqrt(ax * ax + ay * ay + az * az);
    double th = atan2(s, c);
    if (s > 1e-9) {
        double k = th / (2.0 * s);
        r[0] = ax * k; r[1] = ay * k; r[2] = az * k;
    } else if (c > 0.0) {
        r[0] = r[1] = r[2] = 0.0;
    } else { /* theta ~ pi: axis from the diagonal */
        double xx = sqrt(fmax((R[0] + 1.0) * 0.5, 0.0));
        double yy = sqrt(fmax((R[4] + 1.0) * 0.5, 0.0));
        double zz = sqrt(fmax((R[8] + 1.0) * 0.5, 0.0));
        if (R[1] + R[3] < 0.0) yy = -yy;
        if (R[2] + R[6] < 0.0) zz = -zz;
        if (xx == 0.0 && R[5] + R[7] < 0.0) zz = -zz;
        double nn = sqrt(xx * xx + yy * yy + zz * zz);
        r[0] = th * xx / nn; r[1] = th * yy / nn; r[2] = th * zz / nn;
    }
}

/* rvec -> rotation matrix -> quaternion (x, y, z, w), the consumer step at
 * reference src/irm_detector.cpp:218-226 (tf2::Matrix3x3::getRotation). */
void orc_rvec_to_quat(const double rvec[3], double q[4])
{
    double R[9];
    orc_rodrigues(rvec, R);
    double tr = R[0] + R[4] + R[8];
    if (tr > 0.0) {
        double s = sqrt(tr + 1.0);
        q[3] = s * 0.5;
        s = 0.5 / s;
        q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
    } else {
        int i = R[0] < R[4] ? (R[4] < R[8] ? 2 : 1) : (R[0] < R[8] ? 2 : 0);
        int j = (i + 1) % 3, k = (i + 2) % 3;
        double s = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[i] = s * 0.5;
        s = 0.5 / s;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * s;
        q[j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
        q[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
    }
}

void orc_project_points(const double K[9], const double D[5], const double rvec[3],
                        const double tvec[3], const double *obj, int n, double *out)
{
    double R[9];
    orc_rodrigues(rvec, R);
    for (int i = 0; i < n; i++) {
        const double *p = obj + 3 * i;
        double X = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + tvec[0];
        double Y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + tvec[1];
        double Z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + tvec[2];
        double x = X / Z, y = Y / Z, r2 = x * x + y * y;
        double cd = 1.0 + ((D[4] * r2 + D[1]) * r2 + D[0]) * r2;
        double xd = x * cd + 2.0 * D[2] * x * y + D[3] * (r2 + 2.0 * x * x);
        double yd = y * cd + D[2] * (r2 + 2.0 * y * y) + 2.0 * D[3] * x * y;
        out[2 * i] = K[0] * xd + K[2];
        out[2 * i + 1] = K[4] * yd + K[5];
    }
}

/* Jacobi eigen-decomposition of a symmetric 3x3; eigenvalues descending,
 * eigenvectors in the columns of V. */
static void eig3_sym(const double A_[9], double w[3], double V[9])
{
    double A[9];
    memcpy(A, A_, sizeof A);
    const double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(V, I, sizeof I);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = A[p * 3 + q];
                if (fabs(apq) < 1e-300) continue;
                double th = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
                double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) { /* A = A * J */
                    double akp = A[k * 3 + p], akq = A[k * 3 + q];
                    A[k * 3 + p] = c * akp - s * akq;
                    A[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) { /* A = J^T * A */
                    double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                    A[p * 3 + k] = c * apk - s * aqk;
                    A[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - s * vkq;
                    V[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
    for (int i = 0; i < 2; i++) /* sort descending */
        for (int j = i + 1; j < 3; j++)
            if (w[j] > w[i]) {
                double t = w[i]; w[i] = w[j]; w[j] = t;
                for (int k = 0; k < 3; k++) { t = V[k * 3 + i]; V[k * 3 + i] = V[k * 3 + j]; V[k * 3 + j] = t; }
            }
}

/* 8x8 linear solve, Gaussian elimination with partial pivoting */
static int solve8(double A[8][9])
{
    for (int c = 0; c < 8; c++) {
        int piv = c;
        for (int r = c + 1; r < 8; r++)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (fabs(A[piv][c]) < 1e-300) return 0;
        if (piv != c)
            for (int k = 0; k < 9; k++) { double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
        for (int r = 0; r < 8; r++) {
            if (r == c) continue;
            double f = A[r][c] / A[c][c];
            for (int k = c; k < 9; k++) A[r][k] -= f * A[c][k];
        }
    }
    for (int c = 0; c < 8; c++) A[c][8] /= A[c][c];
    return 1;
}

/* IPPE computeRotations (Collins & Bartoli; OpenCV calib3d ippe.cpp): the two
 * rotations consistent with the homography Jacobian J at the origin and the
 * image (p, q) of the origin. */
static int ippe_rotations(double j00, double j01, double j10, double j11, double p, double q,
                          double R1[9], double R2[9])
{
    double rv[9];
    double s = sqrt(p * p + q * q + 1.0), t = sqrt(p * p + q * q);
    double costh = 1.0 / s, sinth = sqrt(1.0 - 1.0 / (s * s));
    if (t < 1e-300) {
        const double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        memcpy(rv, I, sizeof I);
    } else {
        double k0 = p / t, k1 = q / t;
        rv[0] = (costh - 1.0) * k0 * k0 + 1.0; rv[1] = k0 * k1 * (costh - 1.0); rv[2] = k0 * sinth;
        rv[3] = k0 * k1 * (costh - 1.0); rv[4] = (costh - 1.0) * k1 * k1 + 1.0; rv[5] = k1 * sinth;
        rv[6] = -k0 * sinth; rv[7] = -k1 * sinth; rv[8] = (costh - 1.0) * (k0 * k0 + k1 * k1) + 1.0;
    }
    double b00 = rv[0] - p * rv[6], b01 = rv[1] - p * rv[7];
    double b10 = rv[3] - q * rv[6], b11 = rv[4] - q * rv[7];
    double det = b00 * b11 - b01 * b10;
    if (fabs(det) < 1e-300) return 0;
    double dti = 1.0 / det;
    double bi00 = dti * b11, bi01 = -dti * b01, bi10 = -dti * b10, bi11 = dti * b00;
    double a00 = bi00 * j00 + bi01 * j10, a01 = bi00 * j01 + bi01 * j11;
    double a10 = bi10 * j00 + bi11 * j10, a11 = bi10 * j01 + bi11 * j11;
    double ata00 = a00 * a00 + a01 * a01, ata01 = a00 * a10 + a01 * a11, ata11 = a10 * a10 + a11 * a11;
    double g2 = 0.5 * (ata00 + ata11 + sqrt((ata00 - ata11) * (ata00 - ata11) + 4.0 * ata01 * ata01));
    if (g2 <= 0.0) return 0;
    double g = sqrt(g2);
    double r00 = a00 / g, r01 = a01 / g, r10 = a10 / g, r11 = a11 / g;
    double b0 = sqrt(fmax(1.0 - r00 * r00 - r10 * r10, 0.0));
    double b1 = sqrt(fmax(1.0 - r01 * r01 - r11 * r11, 0.0));
    double sp = -r00 * r01 - r10 * r11;
    if (sp < 0.0) b1 = -b1;
    for (int sol = 0; sol < 2; sol++) {
        double c0 = sol ? -b0 : b0, c1 = sol ? -b1 : b1;
        /* Rt = [ r00 r01 x ; r10 r11 y ; c0 c1 z ], third column = col0 x col1 */
        double m[9] = { r00, r01, r10 * c1 - c0 * r11,
                        r10, r11, c0 * r01 - r00 * c1,
                        c0,  c1,  r00 * r11 - r01 * r10 };
        double *R = sol ? R2 : R1;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R[i * 3 + j] = rv[i * 3 + 0] * m[0 * 3 + j] + rv[i * 3 + 1] * m[1 * 3 + j] + rv[i * 3 + 2] * m[2 * 3 + j];
    }
    return 1;
}

/* IPPE computeTranslation: least squares over the points of
 *   tx - x tz = x rz - rx ; ty - y tz = y rz - ry   with (rx,ry,rz) = R (X,Y,0) */
static int ippe_translation(const double *cxy, const double *nxy, int n, const double R[9], double t[3])
{
    double A[9] = { 0 }, b[3] = { 0 };
    for (int i = 0; i < n; i++) {
        double X = cxy[2 * i], Y = cxy[2 * i + 1], x = nxy[2 * i], y = nxy[2 * i + 1];
        double rx = R[0] * X + R[1] * Y, ry = R[3] * X + R[4] * Y, rz = R[6] * X + R[7] * Y;
        double e1 = x * rz - rx, e2 = y * rz - ry;
        A[0] += 1.0; A[2] += -x; A[4] += 1.0; A[5] += -y; A[8] += x * x + y * y;
        b[0] += e1; b[1] += e2; b[2] += -x * e1 - y * e2;
    }
    A[6] = A[2]; A[7] = A[5];
    double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (fabs(det) < 1e-300) return 0;
    double inv[9] = {
        (A[4] * A[8] - A[5] * A[7]), -(A[1] * A[8] - A[2] * A[7]), (A[1] * A[5] - A[2] * A[4]),
        -(A[3] * A[8] - A[5] * A[6]), (A[0] * A[8] - A[2] * A[6]), -(A[0] * A[5] - A[2] * A[3]),
        (A[3] * A[7] - A[4] * A[6]), -(A[0] * A[7] - A[1] * A[6]), (A[0] * A[4] - A[1] * A[3]) };
    for (int i = 0; i < 3; i++) t[i] = (inv[i * 3] * b[0] + inv[i * 3 + 1] * b[1] + inv[i * 3 + 2] * b[2]) / det;
    return 1;
}

static double reproj_err(const double *obj, const double *nxy, int n, const double R[9], const double t[3])
{
    double e = 0.0;
    for (int i = 0; i < n; i++) {
        const double *p = obj + 3 * i;
        double X = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0];
        double Y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1];
        double Z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2];
        double dx = X / Z - nxy[2 * i], dy = Y / Z - nxy[2 * i + 1];
        e += dx * dx + dy * dy;
    }
    return sqrt(e / (2.0 * n));
}

int orc_solve_pnp_ippe(const double K[9], const double D[5], const float img_pts[8], int armor_size,
                       double rvec[3], double tvec[3], double rvec2[3], double tvec2[3], double err[2])
{
    enum { N = 4 };
    double obj[3 * N], nxy[2 * N];
    orc_armor_object_points(armor_size, obj);
    orc_undistort_points(K, D, img_pts, N, nxy);

    /* makeCanonicalObjectPoints: centre, rotate the plane onto z = 0 */
    double cen[3] = { 0, 0, 0 };
    for (int i = 0; i < N; i++) for (int k = 0; k < 3; k++) cen[k] += obj[3 * i + k] / N;
    double S[9] = { 0 }, U[9], w[3];
    for (int i = 0; i < N; i++)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                S[a * 3 + b] += (obj[3 * i + a] - cen[a]) * (obj[3 * i + b] - cen[b]);
    eig3_sym(S, w, U);
    double Rc[9]; /* R = U^T, det forced to +1 */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rc[i * 3 + j] = U[j * 3 + i];
    double det = Rc[0] * (Rc[4] * Rc[8] - Rc[5] * Rc[7]) - Rc[1] * (Rc[3] * Rc[8] - Rc[5] * Rc[6]) + Rc[2] * (Rc[3] * Rc[7] - Rc[4] * Rc[6]);
    if (det < 0) for (int j = 0; j < 3; j++) Rc[6 + j] = -Rc[6 + j];
    double cxy[2 * N];
    for (int i = 0; i < N; i++) {
        double d[3] = { obj[3 * i] - cen[0], obj[3 * i + 1] - cen[1], obj[3 * i + 2] - cen[2] };
        cxy[2 * i] = Rc[0] * d[0] + Rc[1] * d[1] + Rc[2] * d[2];
        cxy[2 * i + 1] = Rc[3] * d[0] + Rc[4] * d[1] + Rc[5] * d[2];
    }

    /* homography canonical (X, Y) -> normalised image (x, y), h22 = 1 */
    double M[8][9];
    for (int i = 0; i < N; i++) {
        double X = cxy[2 * i], Y = cxy[2 * i + 1], x = nxy[2 * i], y = nxy[2 * i + 1];
        double r0[9] = { X, Y, 1, 0, 0, 0, -x * X, -x * Y, x };
        double r1[9] = { 0, 0, 0, X, Y, 1, -y * X, -y * Y, y };
        memcpy(M[2 * i], r0, sizeof r0);
        memcpy(M[2 * i + 1], r1, sizeof r1);
    }
    if (!solve8(M)) return 0;
    double H[9] = { M[0][8], M[1][8], M[2][8], M[3][8], M[4][8], M[5][8], M[6][8], M[7][8], 1.0 };

    /* Jacobian of the homography at the origin, and the origin's image */
    double j00 = H[0] - H[6] * H[2], j01 = H[1] - H[7] * H[2];
    double j10 = H[3] - H[6] * H[5], j11 = H[4] - H[7] * H[5];
    double Ra[9], Rb[9], ta[3], tb[3];
    if (!ippe_rotations(j00, j01, j10, j11, H[2], H[5], Ra, Rb)) return 0;
    if (!ippe_translation(cxy, nxy, N, Ra, ta) || !ippe_translation(cxy, nxy, N, Rb, tb)) return 0;

    /* back to the model frame: [R|t] = [Ra|ta] * [Rc | -Rc cen] */
    double R1[9], R2[9], t1[3], t2[3], mc[3];
    for (int i = 0; i < 3; i++) mc[i] = -(Rc[i * 3] * cen[0] + Rc[i * 3 + 1] * cen[1] + Rc[i * 3 + 2] * cen[2]);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            R1[i * 3 + j] = Ra[i * 3] * Rc[j] + Ra[i * 3 + 1] * Rc[3 + j] + Ra[i * 3 + 2] * Rc[6 + j];
            R2[i * 3 + j] = Rb[i * 3] * Rc[j] + Rb[i * 3 + 1] * Rc[3 + j] + Rb[i * 3 + 2] * Rc[6 + j];
        }
        t1[i] = Ra[i * 3] * mc[0] + Ra[i * 3 + 1] * mc[1] + Ra[i * 3 + 2] * mc[2] + ta[i];
        t2[i] = Rb[i * 3] * mc[0] + Rb[i * 3 + 1] * mc[1] + Rb[i * 3 + 2] * mc[2] + tb[i];
    }
    double e1 = reproj_err(obj, nxy, N, R1, t1), e2 = reproj_err(obj, nxy, N, R2, t2);
    double r1[3], r2[3];
    rot_to_rvec(R1, r1);
    rot_to_rvec(R2, r2);
    if (e1 <= e2) {
        memcpy(rvec, r1, 24); memcpy(tvec, t1, 24);
        if (rvec2) memcpy(rvec2, r2, 24);
        if (tvec2) memcpy(tvec2, t2, 24);
        if (err) { err[0] = e1; err[1] = e2; }
    } else {
        memcpy(rvec, r2, 24); memcpy(tvec, t2, 24);
        if (rvec2) memcpy(rvec2, r1, 24);
        if (tvec2) memcpy(tvec2, t1, 24);
        if (err) { err[0] = e2; err[1] = e1; }
    }
    return isfinite(rvec[0] + rvec[1] + rvec[2] + tvec[0] + tvec[1] + tvec[2]) ? 1 : 0;
}
