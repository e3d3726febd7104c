// Classical armor-point extraction on the GPU: the
// reference's actual source of the four PnP points for bbox-only models --
//   IrmDetector::extract_armors      reference src/irm_detector.cpp:292-355
//   Light::Light / is_light, Armor   reference include/irmv_detection/armor.hpp:11-77
// i.e. per detection: ROI -> gray -> threshold -> external contours -> minAreaRect ->
// light gating -> first two lights -> armor gating, then the same fp64 IPPE PnP as
// the keypoint path.  The OpenCV pieces (cvtColor, threshold, findContours
// RETR_EXTERNAL / CHAIN_APPROX_SIMPLE, minAreaRect) are restated exactly as in
// oracle/orc_light.c, which documents the choices; this file is compiled with
// -ffp-contract=off so that the float geometry matches it bit for bit.
//
// One workgroup per detection.  The frame stays in HBM (no 3.9 MB D2H of the
// rotated image as on the reference's CPU path): the 180-degree rotation is folded
// into the pixel fetch.  Thresholding and the per-contour geometry run lane-
// parallel; the Suzuki-Abe border following is sequential by nature and runs on one
// lane per detection.  Scratch is bounded (label pool per frame, 1024 contours and
// points_cap contour points per detection); a detection that exhausts any of it gets
// armor_valid = -1 -- "no answer" -- never a truncated, silently different result.
#include "irmv_common.hpp"
#include "pnp_device.hpp"

namespace irmv {

constexpr int kLblPos = 2, kLblNeg = -126;
__device__ const int kDX[16] = {1, 1, 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1};
__device__ const int kDY[16] = {0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1, 1, 1};

struct LightRec { float top[2], bottom[2], center[2]; double length; int ok; };

// Border following from the outer-border start (x0, y0) of the padded label image; emits the
// CHAIN_APPROX_SIMPLE points in ROI coordinates.  Returns the number of points (counted past cap).
__device__ int trace_border(signed char *img, int step, int x0, int y0, short *pts, int cap)
{
    int n = 0;
    signed char *i0 = img + (size_t)y0 * step + x0;
    int s = 4, s_end = 4;
    signed char *i1;
    do {
        s = (s - 1) & 7;
        i1 = i0 + kDY[s] * step + kDX[s];
    } while (*i1 == 0 && s != s_end);
    if (s == s_end) {
        *i0 = (signed char)kLblNeg;
        if (n < cap) { pts[0] = (short)(x0 - 1); pts[1] = (short)(y0 - 1); }
        return 1;
    }
    signed char *i3 = i0, *i4;
    int px = x0, py = y0, prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        for (;;) {
            ++s;
            i4 = i3 + kDY[s] * step + kDX[s];
            if (*i4 != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)kLblNeg;
        else if (*i3 == 1) *i3 = (signed char)kLblPos;
        if (s != prev_s) {
            if (n < cap) { pts[2 * n] = (short)(px - 1); pts[2 * n + 1] = (short)(py - 1); }
            n++;
            prev_s = s;
        }
        px += kDX[s]; py += kDY[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
    return n;
}

__device__ __forceinline__ int cross_i(const short *o, const short *a, const short *b)
{
    return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0]);   // |coords| < 2^12: fits int32
}

// minAreaRect of one contour (points sorted in place, hull built in `hull`), then Light + gating.
__device__ void contour_to_light(short *p, int n, short *hull, const LightArgs &a, float min_x, float min_y, LightRec &L)
{
    L.ok = 0;
    // sort by (x, y): insertion sort, contours are a few dozen points
    for (int i = 1; i < n; i++) {
        const short x = p[2 * i], y = p[2 * i + 1];
        int j = i;
        while (j > 0 && (p[2 * (j - 1)] > x || (p[2 * (j - 1)] == x && p[2 * (j - 1) + 1] > y))) {
            p[2 * j] = p[2 * (j - 1)]; p[2 * j + 1] = p[2 * (j - 1) + 1];
            j--;
        }
        p[2 * j] = x; p[2 * j + 1] = y;
    }
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (m && p[2 * (m - 1)] == p[2 * i] && p[2 * (m - 1) + 1] == p[2 * i + 1]) continue;
        p[2 * m] = p[2 * i]; p[2 * m + 1] = p[2 * i + 1]; m++;
    }
    float c[8];
    if (m == 1) {
        for (int i = 0; i < 4; i++) { c[2 * i] = p[0]; c[2 * i + 1] = p[1]; }
    } else if (m == 2) {
        c[0] = p[0]; c[1] = p[1]; c[2] = p[0]; c[3] = p[1]; c[4] = p[2]; c[5] = p[3]; c[6] = p[2]; c[7] = p[3];
    } else {
        int k = 0;
        for (int i = 0; i < m; i++) {
            while (k >= 2 && cross_i(hull + 2 * (k - 2), hull + 2 * (k - 1), p + 2 * i) <= 0) k--;
            hull[2 * k] = p[2 * i]; hull[2 * k + 1] = p[2 * i + 1]; k++;
        }
        for (int i = m - 2, t = k + 1; i >= 0; i--) {
            while (k >= t && cross_i(hull + 2 * (k - 2), hull + 2 * (k - 1), p + 2 * i) <= 0) k--;
            hull[2 * k] = p[2 * i]; hull[2 * k + 1] = p[2 * i + 1]; k++;
        }
        const int h = k - 1;
        if (h == 2) {
            c[0] = hull[0]; c[1] = hull[1]; c[2] = hull[0]; c[3] = hull[1]; c[4] = hull[2]; c[5] = hull[3]; c[6] = hull[2]; c[7] = hull[3];
        } else {
            double best = 1e300, bc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < h; i++) {
                const short *pa = hull + 2 * i, *pb = hull + 2 * ((i + 1) % h);
                double ux = pb[0] - pa[0], uy = pb[1] - pa[1];
                const double len = sqrt(ux * ux + uy * uy);
                ux /= len; uy /= len;
                double smin = 1e300, smax = -1e300, tmin = 1e300, tmax = -1e300;
                for (int j = 0; j < h; j++) {
                    const double dx = hull[2 * j] - pa[0], dy = hull[2 * j + 1] - pa[1];
                    const double s = dx * ux + dy * uy, t = -dx * uy + dy * ux;
                    if (s < smin) smin = s;
                    if (s > smax) smax = s;
                    if (t < tmin) tmin = t;
                    if (t > tmax) tmax = t;
                }
                const double area = (smax - smin) * (tmax - tmin);
                if (area < best) {
                    best = area;
                    const double sx[4] = {smin, smax, smax, smin}, tx[4] = {tmin, tmin, tmax, tmax};
                    for (int q = 0; q < 4; q++) {
                        bc[2 * q] = pa[0] + sx[q] * ux - tx[q] * uy;
                        bc[2 * q + 1] = pa[1] + sx[q] * uy + tx[q] * ux;
                    }
                }
            }
            for (int q = 0; q < 8; q++) c[q] = (float)bc[q];
        }
    }
    // Light(box): corners sorted by y, top / bottom mid-points, length, width, tilt (armor.hpp:14-27)
    float q[4][2];
    for (int i = 0; i < 4; i++) { q[i][0] = c[2 * i]; q[i][1] = c[2 * i + 1]; }
    for (int i = 1; i < 4; i++)
        for (int j = i; j > 0 && q[j][1] < q[j - 1][1]; j--) {
            const float t0 = q[j][0], t1 = q[j][1];
            q[j][0] = q[j - 1][0]; q[j][1] = q[j - 1][1]; q[j - 1][0] = t0; q[j - 1][1] = t1;
        }
    L.top[0] = (q[0][0] + q[1][0]) / 2; L.top[1] = (q[0][1] + q[1][1]) / 2;
    L.bottom[0] = (q[2][0] + q[3][0]) / 2; L.bottom[1] = (q[2][1] + q[3][1]) / 2;
    L.center[0] = (c[0] + c[2] + c[4] + c[6]) / 4; L.center[1] = (c[1] + c[3] + c[5] + c[7]) / 4;
    const double dx = (double)L.top[0] - L.bottom[0], dy = (double)L.top[1] - L.bottom[1];
    L.length = sqrt(dx * dx + dy * dy);
    const double wx = (double)q[0][0] - q[1][0], wy = (double)q[0][1] - q[1][1];
    const double width = sqrt(wx * wx + wy * wy);
    const double tilt = atan2(fabs(dx), fabs(dy)) / 3.14159265358979323846 * 180.0;
    const double ratio = width / L.length;
    if (!(a.light_min_ratio < ratio && ratio < a.light_max_ratio && tilt < a.light_max_angle)) return;
    L.center[0] += min_x; L.center[1] += min_y; L.top[0] += min_x; L.top[1] += min_y; L.bottom[0] += min_x; L.bottom[1] += min_y;
    L.ok = 1;
}

// cv::Rect(Point2f...) of the clamped bbox (src/irm_detector.cpp:299-307): truncation, not rounding
struct Roi { float min_x, min_y; int rx, ry, rw, rh; };
__device__ __forceinline__ bool roi_of(const float *xyxy, int cols, int rows, Roi &r)
{
    const float fx1 = xyxy[0], fy1 = xyxy[1], fx2 = xyxy[2], fy2 = xyxy[3];
    r.min_x = fx1 > 0.0f ? fx1 : 0.0f; r.min_y = fy1 > 0.0f ? fy1 : 0.0f;
    const float max_x = fx2 < (float)cols ? fx2 : (float)cols, max_y = fy2 < (float)rows ? fy2 : (float)rows;
    r.rx = r.ry = r.rw = r.rh = 0;
    if (r.min_x >= max_x || r.min_y >= max_y) return false;
    r.rx = (int)r.min_x; r.ry = (int)r.min_y; r.rw = (int)(max_x - r.min_x); r.rh = (int)(max_y - r.min_y);
    return r.rw > 0 && r.rh > 0;
}
__device__ __forceinline__ unsigned long long label_bytes(const Roi &r)
{
    return ((unsigned long long)(r.rw + 2) * (r.rh + 2) + 15ull) & ~15ull;
}

__global__ __launch_bounds__(256) void light_extract_kernel(LightArgs a)
{
    __shared__ int s_nfound, s_toolarge;
    __shared__ unsigned long long s_prefix;
    __shared__ int s_start[kLightMaxContours + 1];
    __shared__ LightRec s_light[kLightMaxContours];
    const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ndet = a.num_dets ? min(a.num_dets[b * a.num_dets_stride], a.max_det) : a.n_boxes;
    DevDet *d = a.dets + (size_t)b * a.max_det + j;
    if (j >= ndet) return;
    Roi R;
    const bool empty = !roi_of(a.boxes ? a.boxes + 4 * j : d->xyxy, a.cols, a.rows, R);
    const float min_x = R.min_x, min_y = R.min_y;
    const int rx = R.rx, ry = R.ry, rw = R.rw, rh = R.rh, step = rw + 2;
    // label image carved from the frame's pool in detection order (deterministic: a full pool drops the
    // lowest-score boxes): offset = sum of the needs of the detections before this one
    if (tid == 0) { s_prefix = 0ull; s_nfound = 0; }
    __syncthreads();
    unsigned long long mine = 0;
    for (int i = tid; i < j; i += blockDim.x) {
        Roi Q;
        if (roi_of(a.boxes ? a.boxes + 4 * i : a.dets[(size_t)b * a.max_det + i].xyxy, a.cols, a.rows, Q)) mine += label_bytes(Q);
    }
    if (mine) atomicAdd(&s_prefix, mine);
    __syncthreads();
    const bool fits = !empty && s_prefix + label_bytes(R) <= a.label_pool;
    if (tid == 0) s_toolarge = (!empty && !fits) ? 1 : 0;
    const bool skip = empty || !fits;
    signed char *img = a.labels + (size_t)b * a.label_pool + (skip ? 0 : s_prefix);
    short *pts = a.points + ((size_t)b * a.max_det + j) * a.points_cap * 2;
    const uint8_t *frame = a.frames + (size_t)b * a.frame_bytes;

    if (!skip) {
        // 1. gray + threshold into the zero-bordered label image (rotation folded into the fetch)
        const int total = step * (rh + 2);
        for (int i = tid; i < total; i += blockDim.x) {
            const int y = i / step, x = i - y * step;
            signed char v = 0;
            if (x >= 1 && x <= rw && y >= 1 && y <= rh) {
                int sx = rx + x - 1, sy = ry + y - 1;
                if (a.rotate180) { sx = a.cols - 1 - sx; sy = a.rows - 1 - sy; }
                const uint8_t *px = frame + ((size_t)sy * a.cols + sx) * 3;
                const int gray = (px[0] * 3735 + px[1] * 19235 + px[2] * 9798 + (1 << 14)) >> 15;
                v = gray > a.binary_threshold ? 1 : 0;
            }
            img[i] = v;
        }
    }
    __syncthreads();
    // 2. raster scan + border following (one lane)
    if (!skip && tid == 0) {
        int nfound = 0, npts = 0;
        for (int y = 1; y <= rh; y++) {
            int lnbd_x = 0, prev = 0;
            signed char *row = img + (size_t)y * step;
            for (int x = 1; x <= rw + 1; x++) {
                const int p = row[x];
                if (p == prev) continue;
                int is_hole = 0;
                bool start = true;
                if (!(prev == 0 && p == 1)) {
                    if (p != 0 || prev < 1) start = false;
                    else {
                        if (prev & -2) lnbd_x = x - 1;
                        is_hole = 1;
                    }
                }
                if (start && !(is_hole || row[lnbd_x] > 0)) {
                    if (nfound < kLightMaxContours) {
                        s_start[nfound] = npts;
                        const int room = a.points_cap - npts;
                        npts += trace_border(img, step, x, y, pts + 2 * (size_t)(room > 0 ? npts : 0), room > 0 ? room : 0);
                        nfound++;
                    } else {
                        short dummy[2];
                        trace_border(img, step, x, y, dummy, 0);
                        nfound = kLightMaxContours + 1;
                    }
                    lnbd_x = x;
                    prev = row[x];
                    continue;
                }
                prev = p;
                if (prev & -2) lnbd_x = x;
            }
        }
        s_start[nfound] = npts;
        s_nfound = nfound < kLightMaxContours ? nfound : kLightMaxContours;
        if (nfound > kLightMaxContours || npts > a.points_cap) s_toolarge = 1;
    }
    __syncthreads();
    // 3. one lane per contour: minAreaRect -> Light -> gating
    const int nfound = s_nfound;
    const bool pts_ok = !s_toolarge;
    for (int c = tid; c < nfound; c += blockDim.x) {
        s_light[c].ok = 0;
        const int n = s_start[c + 1] - s_start[c];
        if (n >= 5 && pts_ok) contour_to_light(pts + 2 * (size_t)s_start[c], n, a.hulls + (((size_t)b * a.max_det + j) * a.points_cap + s_start[c]) * 2 * 2, a, min_x, min_y, s_light[c]);
    }
    __syncthreads();
    // 4. first two lights in OpenCV's contour order (last found first) -> Armor -> PnP
    if (tid == 0) {
        int nl = 0, total = 0, idx[2] = {0, 0};
        for (int c = nfound - 1; c >= 0; c--)
            if (s_light[c].ok) { if (nl < 2) idx[nl++] = c; total++; }
        int valid = 0, size = 0;
        float kp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (total >= 2 && !s_toolarge) {
            const LightRec &A = s_light[idx[0]], &B = s_light[idx[1]];
            const LightRec &l = A.center[0] < B.center[0] ? A : B, &r = A.center[0] < B.center[0] ? B : A;
            const double avg = (A.length + B.length) / 2;
            const double cdx = (double)l.center[0] - r.center[0], cdy = (double)l.center[1] - r.center[1];
            const double cd = sqrt(cdx * cdx + cdy * cdy) / avg;
            size = cd > a.min_large_cd ? 1 : 0;
            valid = 1;
            if (!size && (a.min_small_cd > cd || a.max_small_cd < cd)) valid = 0;
            if (size && (a.min_large_cd > cd || a.max_large_cd < cd)) valid = 0;
            if (valid) {
                kp[0] = l.bottom[0]; kp[1] = l.bottom[1]; kp[2] = l.top[0]; kp[3] = l.top[1];
                kp[4] = r.top[0]; kp[5] = r.top[1]; kp[6] = r.bottom[0]; kp[7] = r.bottom[1];
            }
        }
        for (int i = 0; i < 8; i++) { d->kpts[i] = kp[i]; d->kpts_net[i] = 0.f; }
        d->armor_valid = s_toolarge ? -1 : valid;
        d->armor_size = size;
        d->n_lights = total;
        for (int i = 0; i < 3; i++) { d->rvec[i] = 0.0; d->tvec[i] = 0.0; }
        d->quat[0] = d->quat[1] = d->quat[2] = 0.0; d->quat[3] = 1.0;
        d->pnp_ok = 0;
        // the reference always solves with the SMALL model (src/pnp_solver.cpp:47-48)
        if (valid) d->pnp_ok = solve_pnp_ippe(*a.pnp, d->kpts, a.pnp_armor_size, d->rvec, d->tvec, d->quat) ? 1 : 0;
    }
}

void launch_light_extract(const LightArgs &a, int n_boxes_max, int batch, hipStream_t s)
{
    if (n_boxes_max <= 0 || batch <= 0) return;
    hipLaunchKernelGGL(light_extract_kernel, dim3(n_boxes_max, batch), dim3(256), 0, s, a);
}

}  // namespace irmv
