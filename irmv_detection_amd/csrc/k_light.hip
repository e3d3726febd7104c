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
// parallel; the Suzuki-Abe raster scan runs on one wave per detection (transitions found
// 64 pixels at a time, the sequential state walked over them), the border following
// itself on one lane.  Scratch is bounded (label pool per frame, 1024 contours and
// points_cap contour points per detection); a detection that exhausts any of it gets
// armor_valid = -1 -- "no answer" -- never a truncated, silently different result.
#include "irmv_common.hpp"
#include "pnp_device.hpp"

namespace irmv {

constexpr int kLblPos = 2, kLblNeg = -126;
// Freeman direction s (0 = east, counter-clockwise in image coordinates) -> (dx, dy), from 2-bit fields: a table in
// memory would put two dependent loads into every step of the border following
__device__ __forceinline__ int dir_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }   // 1, 1, 0, -1, -1, -1, 0, 1
__device__ __forceinline__ int dir_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }   // 0, -1, -1, -1, 0, 1, 1, 1

struct LightRec { float top[2], bottom[2], center[2]; double length; int ok; };

// The label image is addressed through an address-space-typed pointer (LDS for ROIs that fit, else global): the
// border following and the scan are chains of dependent accesses, and a generic pointer would make every one of them a
// flat access that waits on both the LDS and the vector-memory counter.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

using lds_i8 = __attribute__((address_space(3))) signed char;
using glb_i8 = __attribute__((address_space(1))) signed char;

// Border following from the outer-border start (x0, y0) of the padded label image; emits the
// CHAIN_APPROX_SIMPLE points in ROI coordinates.  Returns the number of points (counted past cap).
template <typename P>
__device__ int trace_border(P *img, int step, int x0, int y0, short *pts, int cap, int max_steps)
{
    // all eight neighbours of the current pixel are fetched together (one memory latency per border pixel instead of one
    // per probe); the direction search runs on the resulting bit mask and the next pixel's own label rides along
    const int d0 = 1, d1 = 1 - step, d2 = -step, d3 = -1 - step, d4 = -1, d5 = step - 1, d6 = step, d7 = step + 1;
    unsigned long long nv = 0;   // the eight neighbour labels, one byte each
    auto neighbours = [&](const P *c) -> unsigned {
        const unsigned n0 = (unsigned char)c[d0], n1 = (unsigned char)c[d1], n2 = (unsigned char)c[d2], n3 = (unsigned char)c[d3];
        const unsigned n4 = (unsigned char)c[d4], n5 = (unsigned char)c[d5], n6 = (unsigned char)c[d6], n7 = (unsigned char)c[d7];
        nv = (unsigned long long)(n0 | (n1 << 8) | (n2 << 16) | (n3 << 24)) | ((unsigned long long)(n4 | (n5 << 8) | (n6 << 16) | (n7 << 24)) << 32);
        return (n0 != 0) | ((n1 != 0) << 1) | ((n2 != 0) << 2) | ((n3 != 0) << 3) | ((n4 != 0) << 4) | ((n5 != 0) << 5) | ((n6 != 0) << 6) |
               ((n7 != 0) << 7);
    };
    int n = 0;
    P *i0 = img + (size_t)y0 * step + x0;
    // clockwise from direction 3 (3, 2, 1, 0, 7, 6, 5, 4): first non-zero neighbour; none (direction 4, the left one, is
    // background at an outer-border start) = a single-pixel component
    unsigned nz = neighbours(i0);
    int s = -1;
    for (int k = 0; k < 8; k++) {
        const int dir = (3 - k) & 7;
        if (nz & (1u << dir)) { s = dir; break; }
    }
    if (s < 0 || s == 4) {
        *i0 = (signed char)kLblNeg;
        if (n < cap) { pts[0] = (short)(x0 - 1); pts[1] = (short)(y0 - 1); }
        return 1;
    }
    P *const i1 = i0 + dir_dy(s) * step + dir_dx(s);
    P *i3 = i0, *i4;
    int px = x0, py = y0, prev_s = s ^ 4;
    int cur = 1;   // label of i3: the start pixel is unmarked foreground
    // A border visits each of its pixels at most four times, so the walk ends within 4 x (image pixels) steps; the bound
    // only exists so that a corrupted label image could never keep a wave spinning on the GPU.
    for (int guard = 0; guard < max_steps; guard++) {
        const int s_end = s;
        // counter-clockwise from s + 1: first non-zero neighbour of i3 (there is one: we came from it)
        const unsigned rot = ((nz | (nz << 8)) >> (s + 1)) & 0xffu;
        s = (s + 1 + (__ffs((int)rot) - 1)) & 7;
        i4 = i3 + dir_dy(s) * step + dir_dx(s);
        const int next = (int)(signed char)((nv >> (8 * s)) & 0xffull);   // label of i4 as just read
        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)kLblNeg;
        else if (cur == 1) *i3 = (signed char)kLblPos;
        if (s != prev_s) {
            if (n < cap) { pts[2 * n] = (short)(px - 1); pts[2 * n + 1] = (short)(py - 1); }
            n++;
            prev_s = s;
        }
        px += dir_dx(s); py += dir_dy(s);
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        cur = next;
        s = (s + 4) & 7;
        nz = neighbours(i3);
    }
    return n;
}

template <typename S>
__device__ __forceinline__ int cross_i(const S *o, const S *a, const S *b)
{
    return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0]);   // |coords| < 2^12: fits int32
}

// minAreaRect of one contour, then Light + gating -- executed by a whole wave (all 64 lanes call it together).
// p: the contour's n points (sorted and de-duplicated in place); scratch: room for 2n points (sort buffer, then hull).
// Same arithmetic as oracle/orc_light.c: the sort is a rank sort and every hull edge's bounding rectangle is measured
// by its own lane, but each number is produced by the same operations, and the winner is the first minimal edge.
template <typename S>
__device__ void contour_to_light(S *p, int n, S *scratch, const LightArgs &a, float min_x, float min_y, int lane, LightRec &L)
{
    L.ok = 0;
    // 1. sort by (x, y): rank of a point = points with a smaller key, plus equal keys before it
    S *sorted = scratch + 2 * (size_t)n;
    for (int i = lane; i < n; i += 64) {
        const int xi = p[2 * i], yi = p[2 * i + 1];
        const int ki = (xi << 16) | (yi & 0xffff);
        int rank = 0;
        for (int j = 0; j < n; j++) {
            const int kj = ((int)p[2 * j] << 16) | ((int)p[2 * j + 1] & 0xffff);
            rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
        }
        sorted[2 * rank] = (short)xi; sorted[2 * rank + 1] = (short)yi;
    }
    __threadfence_block();
    wave_lds_sync();
    S *hull = scratch;
    float c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int h = 0, special = 1;   // special: corners already final (degenerate contours)
    if (lane == 0) {
        int m = 0;
        for (int i = 0; i < n; i++) {
            if (m && p[2 * (m - 1)] == sorted[2 * i] && p[2 * (m - 1) + 1] == sorted[2 * i + 1]) continue;
            p[2 * m] = sorted[2 * i]; p[2 * m + 1] = sorted[2 * i + 1]; m++;
        }
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
            h = k - 1;
            if (h == 2) {
                c[0] = hull[0]; c[1] = hull[1]; c[2] = hull[0]; c[3] = hull[1]; c[4] = hull[2]; c[5] = hull[3]; c[6] = hull[2]; c[7] = hull[3];
            } else {
                special = 0;
            }
        }
    }
    __threadfence_block();
    wave_lds_sync();
    h = __shfl(h, 0);
    special = __shfl(special, 0);
    if (!special) {
        // 2. one hull edge per lane: area of the bounding rectangle aligned with it
        auto measure = [&](int i, double &ux, double &uy, double &smin, double &smax, double &tmin, double &tmax) {
            const S *pa = hull + 2 * i, *pb = hull + 2 * ((i + 1) % h);
            ux = pb[0] - pa[0]; uy = pb[1] - pa[1];
            const double len = sqrt(ux * ux + uy * uy);
            ux /= len; uy /= len;
            smin = 1e300; smax = -1e300; tmin = 1e300; tmax = -1e300;
            for (int j = 0; j < h; j++) {
                const double dx = hull[2 * j] - pa[0], dy = hull[2 * j + 1] - pa[1];
                const double sv = dx * ux + dy * uy, tv = -dx * uy + dy * ux;
                if (sv < smin) smin = sv;
                if (sv > smax) smax = sv;
                if (tv < tmin) tmin = tv;
                if (tv > tmax) tmax = tv;
            }
        };
        double best = 1e300;
        int bi = 0x7fffffff;
        for (int i = lane; i < h; i += 64) {
            double ux, uy, smin, smax, tmin, tmax;
            measure(i, ux, uy, smin, smax, tmin, tmax);
            const double area = (smax - smin) * (tmax - tmin);
            if (area < best) { best = area; bi = i; }
        }
        for (int off = 32; off; off >>= 1) {   // first edge of minimal area, as the sequential `area < best` scan picks it
            const double ob = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0 && bi < h) {
            double ux, uy, smin, smax, tmin, tmax;
            measure(bi, ux, uy, smin, smax, tmin, tmax);
            const S *pa = hull + 2 * bi;
            const double sx[4] = {smin, smax, smax, smin}, tx[4] = {tmin, tmin, tmax, tmax};
            for (int q = 0; q < 4; q++) {
                c[2 * q] = (float)(pa[0] + sx[q] * ux - tx[q] * uy);
                c[2 * q + 1] = (float)(pa[1] + sx[q] * uy + tx[q] * ux);
            }
        }
    }
    if (lane != 0) return;
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

// Raster scan + border following of one ROI by ONE wave (all 64 lanes call it).
//
// The sequential scan (oracle/orc_light.c) acts only where a pixel's label differs from its left neighbour's, and of
// its state only one thing is ever tested: whether the pixel `lnbd_x` points at carries the positive border mark.
// lnbd_x moves to x when the pixel entered is marked, and to x - 1 when a positively marked pixel is left for
// background; a border is started at an unmarked foreground pixel entered from background iff the last such event
// in the row was not "positive".  So per 64-pixel block three ballots (start candidates, positive events, negative
// events) and a few scalar bit operations reproduce every decision of the scan; only accepted starts are walked
// (lane 0 follows the border), after which the labels to the right are re-read because they now carry marks.
template <typename P>
__device__ void scan_external(P *img, int step, int rw, int rh, int lane, short *pts, int points_cap, int *s_start, int *s_nfound, int *s_toolarge)
{
    int nfound = 0, npts = 0;
    for (int y = 1; y <= rh; y++) {
        P *row = img + (size_t)y * step;
        bool last_pos = false;   // label[lnbd_x] > 0 (lnbd_x starts on the zero border)
        for (int cx = 1; cx <= rw + 1; cx += 256) {
            // lane L owns pixels cx + L + 64 k (k = 0..3): four consecutive 64-pixel blocks, all loads issued together
            int cur[4], left[4];
            auto load_labels = [&]() {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int x = cx + lane + 64 * k;
                    cur[k] = row[min(x, rw + 1)];
                    left[k] = row[min(x, rw + 1) - 1];
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (cx + lane + 64 * k > rw + 1) { cur[k] = 0; left[k] = 0; }
            };
            load_labels();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (cx + 64 * k > rw + 1) break;                            // past the ROI
                if (!__ballot((cur[k] | left[k]) != 0)) continue;          // background only: no candidate, no event
                unsigned long long cand = __ballot(cur[k] == 1 && left[k] == 0);
                unsigned long long ev_pos = __ballot(cur[k] == kLblPos || (cur[k] == 0 && left[k] == kLblPos));
                unsigned long long ev_neg = __ballot(cur[k] == kLblNeg);
                while (cand) {
                    const int bit = __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    const unsigned long long below = (1ull << bit) - 1ull;
                    const unsigned long long e = (ev_pos | ev_neg) & below;
                    const bool pos = e ? (((ev_pos & below) >> (63 - __clzll((long long)e))) & 1ull) != 0 : last_pos;
                    if (pos) continue;   // inside a component that has already been followed: not an outer border
                    const int xx = cx + bit + 64 * k;
                    const bool keep = nfound < kLightMaxContours;
                    const int room = keep ? points_cap - npts : 0;
                    int n = 0;
                    if (lane == 0) {
                        short dummy[2];
                        n = trace_border(img, step, xx, y, room > 0 ? pts + 2 * (size_t)npts : dummy, room > 0 ? room : 0, 4 * step * (rh + 2) + 16);
                    }
                    n = __shfl(n, 0);
                    if (keep) {
                        if (lane == 0) s_start[nfound] = npts;
                        npts += n;
                        nfound++;
                    } else {
                        nfound = kLightMaxContours + 1;
                    }
                    __threadfence_block();
                    load_labels();   // marks to the right (this block and the following ones of the chunk)
                    const unsigned long long from = ~((2ull << bit) - 1ull);   // positions > bit; the start pixel itself is now an event
                    cand = __ballot(cur[k] == 1 && left[k] == 0) & from;
                    ev_pos = __ballot(cur[k] == kLblPos || (cur[k] == 0 && left[k] == kLblPos));
                    ev_neg = __ballot(cur[k] == kLblNeg);
                }
                const unsigned long long e = ev_pos | ev_neg;
                if (e) last_pos = ((ev_pos >> (63 - __clzll((long long)e))) & 1ull) != 0;
            }
        }
    }
    if (lane == 0) {
        const int nf = nfound < kLightMaxContours ? nfound : kLightMaxContours;
        s_start[nf] = npts;
        *s_nfound = nf;
        if (nfound > kLightMaxContours || npts > points_cap) *s_toolarge = 1;
    }
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
    __shared__ short s_cpts[4][2 * kLightLdsPoints * 3 + 4];   // per wave: a contour's points + 2n points of sort / hull scratch
    __shared__ LightRec s_top[4][2];   // per wave: the last two gated lights it measured
    __shared__ int s_topc[4][2], s_nlights;
    __shared__ __attribute__((aligned(16))) signed char s_img[kLightLdsImage];   // label image of ROIs up to ~200 x 200 (else: the HBM pool)
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
    if (tid == 0) { s_prefix = 0ull; s_nfound = 0; s_nlights = 0; }
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
    // small ROIs keep their label image in LDS (the scan and the border following are latency chains); the pool slice stays
    // reserved either way, so which detections fit does not depend on this
    const bool in_lds = !skip && label_bytes(R) <= (unsigned long long)kLightLdsImage;
    signed char *img = in_lds ? s_img : a.labels + (size_t)b * a.label_pool + (skip ? 0 : s_prefix);
    short *pts = a.points + ((size_t)b * a.max_det + j) * a.points_cap * 2;
    const uint8_t *frame = a.frames + (size_t)b * a.frame_bytes;

    if (!skip) {
        // 1. gray + threshold into the zero-bordered label image (rotation folded into the fetch): a wave per row,
        // several pixels of a lane in flight at once (the frame bytes come straight from HBM)
        const int wave = tid >> 6, lane = tid & 63;
        constexpr int RU = 8;   // rows of one wave in flight together
        for (int y0 = wave; y0 < rh + 2; y0 += 4 * RU) {
            for (int x = lane; x < step; x += 64) {
                const bool xin = x >= 1 && x <= rw;
                int sx = rx + min(max(x, 1), rw) - 1;          // clamped: the loads below are unconditional
                if (a.rotate180) sx = a.cols - 1 - sx;
                int gray[RU];
#pragma unroll
                for (int r = 0; r < RU; r++) {
                    const int y = y0 + 4 * r;
                    int sy = ry + min(max(y, 1), rh) - 1;
                    if (a.rotate180) sy = a.rows - 1 - sy;
                    const uint8_t *px = frame + ((size_t)sy * a.cols + sx) * 3;
                    gray[r] = (px[0] * 3735 + px[1] * 19235 + px[2] * 9798 + (1 << 14)) >> 15;
                }
#pragma unroll
                for (int r = 0; r < RU; r++) {
                    const int y = y0 + 4 * r;
                    if (y < rh + 2) img[(size_t)y * step + x] = (xin && y >= 1 && y <= rh && gray[r] > a.binary_threshold) ? 1 : 0;
                }
            }
        }
    }
    __syncthreads();
    // 2. raster scan + border following, wave 0.  The scan only ever acts where a pixel's label differs from its left
    // neighbour's, so the wave finds those transitions 64 pixels at a time (one byte compare per lane + a ballot) and
    // walks just the set bits with the sequential Suzuki-Abe state (lnbd_x); the walk, its loads and its decisions are
    // wave-uniform.  A border is followed by lane 0; its marks change labels to the right, so the rest of the chunk is
    // re-examined afterwards.  Same decisions in the same order as the one-pixel-at-a-time scan of oracle/orc_light.c.
    if (!skip && tid < 64) {
        if (in_lds) scan_external((lds_i8 *)s_img, step, rw, rh, tid, pts, a.points_cap, s_start, &s_nfound, &s_toolarge);
        else scan_external((glb_i8 *)img, step, rw, rh, tid, pts, a.points_cap, s_start, &s_nfound, &s_toolarge);
    }
    __syncthreads();
    // 3. a wave per contour: minAreaRect -> Light -> gating.  Contours are visited in discovery order, so the two lights
    // OpenCV's order (last found first) puts in front are the last two a wave keeps; the waves' pairs are merged below.
    const int nfound = s_nfound;
    const bool pts_ok = !s_toolarge;
    {
        const int wave = tid >> 6, lane = tid & 63;
        short *hulls = a.hulls + ((size_t)b * a.max_det + j) * a.points_cap * 2 * 2;
        LightRec r0, r1;
        int c0 = -1, c1 = -1, cnt = 0;
        r0.ok = r1.ok = 0;
        for (int c = wave; c < nfound && pts_ok; c += 4) {
            const int n = s_start[c + 1] - s_start[c];
            if (n < 5) continue;
            LightRec L;
            if (n <= kLightLdsPoints) {
                // the sort / hull walks are chains of dependent accesses: contours of ordinary size are measured in LDS
                using lds_i16 = __attribute__((address_space(3))) short;
                lds_i16 *lp = (lds_i16 *)s_cpts[wave];
                const short *gp = pts + 2 * (size_t)s_start[c];
                for (int i = lane; i < 2 * n; i += 64) lp[i] = gp[i];
                wave_lds_sync();
                contour_to_light(lp, n, lp + 2 * kLightLdsPoints, a, min_x, min_y, lane, L);
                wave_lds_sync();
            } else {
                contour_to_light(pts + 2 * (size_t)s_start[c], n, hulls + (size_t)s_start[c] * 2 * 2, a, min_x, min_y, lane, L);
            }
            if (lane == 0 && L.ok) { r1 = r0; c1 = c0; r0 = L; c0 = c; cnt++; }
        }
        if (lane == 0) {
            s_top[wave][0] = r0; s_top[wave][1] = r1;
            s_topc[wave][0] = c0; s_topc[wave][1] = c1;
            if (cnt) atomicAdd(&s_nlights, cnt);
        }
    }
    __syncthreads();
    // 4. first two lights in OpenCV's contour order -> Armor -> PnP
    if (tid == 0) {
        const int total = s_nlights;
        int valid = 0, size = 0;
        float kp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (total >= 2 && !s_toolarge) {
            int w0 = 0, k0 = 0, w1 = 0, k1 = 0, b0 = -1, b1 = -1;   // the two candidates with the highest contour index
            for (int w = 0; w < 4; w++)
                for (int k = 0; k < 2; k++) {
                    const int c = s_topc[w][k];
                    if (c > b0) { b1 = b0; w1 = w0; k1 = k0; b0 = c; w0 = w; k0 = k; }
                    else if (c > b1) { b1 = c; w1 = w; k1 = k; }
                }
            const LightRec &A = s_top[w0][k0], &B = s_top[w1][k1];
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
