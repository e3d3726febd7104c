/*
 * orc_light.c -- CPU oracle for the classical armor-point extraction.
 *
 * TEST INFRASTRUCTURE ONLY (see irmv_oracle.h).  Restates
 *   IrmDetector::extract_armors      reference src/irm_detector.cpp:292-355
 *   Light::Light / is_light          reference include/irmv_detection/armor.hpp:14-36
 *   Armor::Armor                     reference include/irmv_detection/armor.hpp:58-68
 * and the OpenCV calls they make, from OpenCV's published algorithms [external]:
 *   cv::cvtColor(BGR2GRAY, 8u)   fixed point, 15-bit coefficients 3735 / 19235 / 9798, round to nearest
 *   cv::threshold(THRESH_BINARY) v > thr ? 255 : 0
 *   cv::findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)
 *                                Suzuki-Abe border following as in the legacy C implementation
 *                                (raster scan with LNBD, +2 / -126 border labels, external test on the
 *                                sign of the last border label in the row, clockwise start search,
 *                                counter-clockwise following, a point is emitted when the step
 *                                direction changes); contours are returned LAST FOUND FIRST
 *   cv::minAreaRect              convex hull (monotone chain) + minimum-area enclosing rectangle over the
 *                                hull edges; only the four corners are used downstream
 * PARITY UNPINNED: OpenCV is not in this image and the reference tests pin no value; every
 * choice above that OpenCV's source would settle is this build's own and is stated here.
 * Compiled with -ffp-contract=off like orc_post.c.
 */
#include "irmv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LBL_POS 2
#define LBL_NEG (-126)

static const int DX[16] = { 1, 1, 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1 };
static const int DY[16] = { 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1, 1, 1 };

/* gray = (c0 * 3735 + c1 * 19235 + c2 * 9798 + 2^14) >> 15, the frame treated as "BGR"
 * exactly as the reference does (src/irm_detector.cpp:309) whatever its real channel order */
static inline int gray_of(const uint8_t *px) { return (px[0] * 3735 + px[1] * 19235 + px[2] * 9798 + (1 << 14)) >> 15; }

/* Border following from the start pixel (x0, y0) of an outer border in the padded label image.
 * Emits CHAIN_APPROX_SIMPLE points (ROI coordinates = padded - 1).  Returns the number of points. */
static int trace_border(signed char *img, int step, int x0, int y0, short *pts, int cap)
{
    int n = 0;
    signed char *i0 = img + (size_t)y0 * step + x0;
    int s = 4, s_end = 4;
    signed char *i1;
    do {
        s = (s - 1) & 7;
        i1 = i0 + DY[s] * step + DX[s];
    } while (*i1 == 0 && s != s_end);
    if (s == s_end) { /* isolated pixel */
        *i0 = (signed char)LBL_NEG;
        if (n < cap) { pts[2 * n] = (short)(x0 - 1); pts[2 * n + 1] = (short)(y0 - 1); }
        return n + 1;
    }
    signed char *i3 = i0, *i4;
    int px = x0, py = y0, prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        for (;;) {
            ++s;
            i4 = i3 + DY[s] * step + DX[s];
            if (*i4 != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)LBL_NEG;   /* the east neighbour was examined and is background */
        else if (*i3 == 1) *i3 = LBL_POS;
        if (s != prev_s) {
            if (n < cap) { pts[2 * n] = (short)(px - 1); pts[2 * n + 1] = (short)(py - 1); }
            n++;
            prev_s = s;
        }
        px += DX[s]; py += DY[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
    return n;
}

/* External contours of a binary ROI (bin: 0 / nonzero, w x h).  Contours come back in OpenCV's
 * order (last found first): offsets[i]..offsets[i+1] index pts (x, y shorts).  Returns the count. */
int orc_find_external_contours(const uint8_t *bin, int w, int h, short *pts, int pts_cap, int *offsets, int max_contours)
{
    const int step = w + 2;
    signed char *img = calloc((size_t)(h + 2) * step, 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)(y + 1) * step + x + 1] = bin[(size_t)y * w + x] ? 1 : 0;
    int nfound = 0, npts = 0;
    int *starts = malloc(sizeof(int) * (size_t)(max_contours + 1));
    for (int y = 1; y <= h; y++) {
        int lnbd_x = 0;          /* lnbd = (0, y): frame background */
        int prev = 0;
        signed char *row = img + (size_t)y * step;
        for (int x = 1; x <= w + 1; x++) {
            const int p = row[x];
            if (p == prev) continue;
            int is_hole = 0;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) goto resume;
                if (prev & -2) lnbd_x = x - 1;
                is_hole = 1;
            }
            if (is_hole || row[lnbd_x] > 0) goto resume;   /* RETR_EXTERNAL: skip holes and anything inside a component */
            if (nfound < max_contours) {
                starts[nfound] = npts;
                npts += trace_border(img, step, x, y, pts + 2 * (size_t)npts, pts_cap - npts > 0 ? pts_cap - npts : 0);
                nfound++;
            } else {
                short dummy[2];
                trace_border(img, step, x, y, dummy, 0);
            }
            lnbd_x = x;
            prev = row[x];
            continue;
        resume:
            prev = p;
            if (prev & -2) lnbd_x = x;
        }
    }
    starts[nfound] = npts;
    /* reverse to OpenCV's order, compacting the points accordingly */
    short *tmp = malloc(sizeof(short) * 2 * (size_t)(npts > 0 ? npts : 1));
    int o = 0;
    for (int i = nfound - 1, k = 0; i >= 0; i--, k++) {
        const int a = starts[i], b = starts[i + 1];
        offsets[k] = o;
        for (int j = a; j < b && j < pts_cap; j++) { tmp[2 * o] = pts[2 * j]; tmp[2 * o + 1] = pts[2 * j + 1]; o++; }
    }
    offsets[nfound] = o;
    memcpy(pts, tmp, sizeof(short) * 2 * (size_t)o);
    free(tmp); free(starts); free(img);
    return nfound;
}

/* ---- minAreaRect: hull + best edge ------------------------------------------------------- */
static long long cross_ll(const short *o, const short *a, const short *b)
{
    return (long long)(a[0] - o[0]) * (b[1] - o[1]) - (long long)(a[1] - o[1]) * (b[0] - o[0]);
}

static int cmp_pt(const void *a, const void *b)
{
    const short *p = a, *q = b;
    if (p[0] != q[0]) return p[0] < q[0] ? -1 : 1;
    return p[1] < q[1] ? -1 : (p[1] > q[1] ? 1 : 0);
}

/* Andrew monotone chain; hull in counter-clockwise order (y down: clockwise on screen), no collinear points */
static int convex_hull(const short *pts, int n, short *hull)
{
    short *s = malloc(sizeof(short) * 2 * (size_t)n);
    memcpy(s, pts, sizeof(short) * 2 * (size_t)n);
    qsort(s, n, 2 * sizeof(short), cmp_pt);
    int m = 0;
    for (int i = 0; i < n; i++) {   /* unique */
        if (m && s[2 * (m - 1)] == s[2 * i] && s[2 * (m - 1) + 1] == s[2 * i + 1]) continue;
        s[2 * m] = s[2 * i]; s[2 * m + 1] = s[2 * i + 1]; m++;
    }
    if (m < 3) { memcpy(hull, s, sizeof(short) * 2 * (size_t)m); free(s); return m; }
    int k = 0;
    for (int i = 0; i < m; i++) {
        while (k >= 2 && cross_ll(hull + 2 * (k - 2), hull + 2 * (k - 1), s + 2 * i) <= 0) k--;
        hull[2 * k] = s[2 * i]; hull[2 * k + 1] = s[2 * i + 1]; k++;
    }
    for (int i = m - 2, t = k + 1; i >= 0; i--) {
        while (k >= t && cross_ll(hull + 2 * (k - 2), hull + 2 * (k - 1), s + 2 * i) <= 0) k--;
        hull[2 * k] = s[2 * i]; hull[2 * k + 1] = s[2 * i + 1]; k++;
    }
    free(s);
    return k - 1;
}

/* Minimum-area enclosing rectangle of a point set: corners[8] (4 points, float).  For every hull edge
 * the rectangle aligned with it; the first edge with the smallest area wins. */
void orc_min_area_rect(const short *pts, int n, float corners[8])
{
    short *hull = malloc(sizeof(short) * 2 * (size_t)(2 * n + 2));
    const int h = convex_hull(pts, n, hull);
    if (h == 1) {
        for (int i = 0; i < 4; i++) { corners[2 * i] = hull[0]; corners[2 * i + 1] = hull[1]; }
    } else if (h == 2) {
        corners[0] = hull[0]; corners[1] = hull[1]; corners[2] = hull[0]; corners[3] = hull[1];
        corners[4] = hull[2]; corners[5] = hull[3]; corners[6] = hull[2]; corners[7] = hull[3];
    } else {
        double best = 1e300, bc[8] = { 0 };
        for (int i = 0; i < h; i++) {
            const short *a = hull + 2 * i, *b = hull + 2 * ((i + 1) % h);
            double ux = b[0] - a[0], uy = b[1] - a[1];
            const double len = sqrt(ux * ux + uy * uy);
            ux /= len; uy /= len;
            double smin = 1e300, smax = -1e300, tmin = 1e300, tmax = -1e300;
            for (int j = 0; j < h; j++) {
                const double dx = hull[2 * j] - a[0], dy = hull[2 * j + 1] - a[1];
                const double s = dx * ux + dy * uy, t = -dx * uy + dy * ux;
                if (s < smin) smin = s;
                if (s > smax) smax = s;
                if (t < tmin) tmin = t;
                if (t > tmax) tmax = t;
            }
            const double area = (smax - smin) * (tmax - tmin);
            if (area < best) {
                best = area;
                const double sx[4] = { smin, smax, smax, smin }, tx[4] = { tmin, tmin, tmax, tmax };
                for (int c = 0; c < 4; c++) {
                    bc[2 * c] = a[0] + sx[c] * ux - tx[c] * uy;
                    bc[2 * c + 1] = a[1] + sx[c] * uy + tx[c] * ux;
                }
            }
        }
        for (int c = 0; c < 8; c++) corners[c] = (float)bc[c];
    }
    free(hull);
}

/* ---- Light / Armor (reference include/irmv_detection/armor.hpp) ---------------------------- */
typedef struct { float top[2], bottom[2], center[2]; double length, width, tilt; } light_t;

static void make_light(const float c[8], light_t *L)
{
    float p[4][2];
    for (int i = 0; i < 4; i++) { p[i][0] = c[2 * i]; p[i][1] = c[2 * i + 1]; }
    for (int i = 1; i < 4; i++)   /* sort by y (stable insertion: ties keep corner order) */
        for (int j = i; j > 0 && p[j][1] < p[j - 1][1]; j--) {
            float t0 = p[j][0], t1 = p[j][1];
            p[j][0] = p[j - 1][0]; p[j][1] = p[j - 1][1]; p[j - 1][0] = t0; p[j - 1][1] = t1;
        }
    L->top[0] = (p[0][0] + p[1][0]) / 2; L->top[1] = (p[0][1] + p[1][1]) / 2;
    L->bottom[0] = (p[2][0] + p[3][0]) / 2; L->bottom[1] = (p[2][1] + p[3][1]) / 2;
    L->center[0] = (c[0] + c[2] + c[4] + c[6]) / 4; L->center[1] = (c[1] + c[3] + c[5] + c[7]) / 4;
    const double dx = (double)L->top[0] - L->bottom[0], dy = (double)L->top[1] - L->bottom[1];
    L->length = sqrt(dx * dx + dy * dy);
    const double wx = (double)p[0][0] - p[1][0], wy = (double)p[0][1] - p[1][1];
    L->width = sqrt(wx * wx + wy * wy);
    L->tilt = atan2(fabs(dx), fabs(dy)) / 3.14159265358979323846 * 180.0;
}

/* One bbox -> at most one armor.  out: valid, size (0 small / 1 large), pts[8] = LB, LT, RT, RB
 * (src/pnp_solver.cpp:41-44), center[2].  Returns 1 if an armor was produced. */
int orc_extract_armor(const uint8_t *img, int cols, int rows, const float xyxy[4], const orc_light_params *P,
                      int *size, float pts[8], float center[2], int *n_lights_out)
{
    float min_x = xyxy[0] > 0.0f ? xyxy[0] : 0.0f, min_y = xyxy[1] > 0.0f ? xyxy[1] : 0.0f;
    float max_x = xyxy[2] < (float)cols ? xyxy[2] : (float)cols, max_y = xyxy[3] < (float)rows ? xyxy[3] : (float)rows;
    if (n_lights_out) *n_lights_out = 0;
    if (min_x >= max_x || min_y >= max_y) return 0;
    const int rx = (int)min_x, ry = (int)min_y, rw = (int)(max_x - min_x), rh = (int)(max_y - min_y);   /* cv::Rect(float...) truncates */
    if (rw <= 0 || rh <= 0) return 0;   /* (the reference would hand an empty Mat to cvtColor and throw) */
    uint8_t *bin = malloc((size_t)rw * rh);
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++)
            bin[(size_t)y * rw + x] = gray_of(img + ((size_t)(ry + y) * cols + rx + x) * 3) > P->binary_threshold ? 255 : 0;
    const int pts_cap = 4 * rw * rh + 64, max_c = rw * rh / 2 + 2;   /* no cap, as in OpenCV: every border pixel is emitted at most 4 times */
    short *cp = malloc(sizeof(short) * 2 * (size_t)pts_cap);
    int *off = malloc(sizeof(int) * (max_c + 1));
    const int nc = orc_find_external_contours(bin, rw, rh, cp, pts_cap, off, max_c);
    light_t lights[2];
    int nl = 0, total_lights = 0;
    for (int i = 0; i < nc; i++) {
        const int n = off[i + 1] - off[i];
        if (n < 5) continue;
        float c[8];
        orc_min_area_rect(cp + 2 * off[i], n, c);
        light_t L;
        make_light(c, &L);
        const double ratio = L.width / L.length;
        if (!(P->light_min_ratio < ratio && ratio < P->light_max_ratio && L.tilt < P->light_max_angle)) continue;
        L.center[0] += min_x; L.center[1] += min_y; L.top[0] += min_x; L.top[1] += min_y; L.bottom[0] += min_x; L.bottom[1] += min_y;
        if (nl < 2) lights[nl++] = L;
        total_lights++;
    }
    free(bin); free(cp); free(off);
    if (n_lights_out) *n_lights_out = total_lights;
    if (total_lights < 2) return 0;
    const light_t *l = lights[0].center[0] < lights[1].center[0] ? &lights[0] : &lights[1];
    const light_t *r = l == &lights[0] ? &lights[1] : &lights[0];
    const double avg = (lights[0].length + lights[1].length) / 2;
    const double cdx = (double)l->center[0] - r->center[0], cdy = (double)l->center[1] - r->center[1];
    const double cd = sqrt(cdx * cdx + cdy * cdy) / avg;
    const int large = cd > P->armor_min_large_center_distance;
    if (!large && (P->armor_min_small_center_distance > cd || P->armor_max_small_center_distance < cd)) return 0;
    if (large && (P->armor_min_large_center_distance > cd || P->armor_max_large_center_distance < cd)) return 0;
    *size = large;
    pts[0] = l->bottom[0]; pts[1] = l->bottom[1]; pts[2] = l->top[0]; pts[3] = l->top[1];
    pts[4] = r->top[0]; pts[5] = r->top[1]; pts[6] = r->bottom[0]; pts[7] = r->bottom[1];
    center[0] = (l->center[0] + r->center[0]) / 2; center[1] = (l->center[1] + r->center[1]) / 2;
    return 1;
}

void orc_light_params_default(orc_light_params *P)
{
    /* reference src/irm_detector.cpp:152-173 */
    P->binary_threshold = 150;
    P->light_min_ratio = 0.1f; P->light_max_ratio = 0.4f; P->light_max_angle = 40.0f;
    P->armor_min_small_center_distance = 0.8; P->armor_max_small_center_distance = 3.2;
    P->armor_min_large_center_distance = 3.2; P->armor_max_large_center_distance = 5.5;
}
