// fp64 planar PnP (IPPE) device functions, shared by k_post.hip (keypoint path) and k_light.hip
// (classical path); both translation units are compiled with -ffp-contract=off.
#pragma once

#include "irmv_common.hpp"

namespace irmv {

// ---------------------------------------------------------------------------
// Planar PnP (IPPE, Collins & Bartoli 2014) for the armor rectangle, fp64, one
// lane per armor.  The object points (src/pnp_solver.cpp:18-33) are
// (0, +-hy, +-hz): already centred, plane normal = model x.  Canonical frame:
// Xc = y_model, Yc = z_model, Zc = x_model (a proper rotation).  The homography
// canonical plane -> normalised image is the closed-form rectangle->quad map.
// ---------------------------------------------------------------------------
struct Pose { double R[9]; double t[3]; double err; };

__device__ inline void undistort4(const PnpConst &c, const float *pts, double *nxy)
{
    for (int i = 0; i < 4; i++) {
        const double x0 = ((double)pts[2 * i] - c.cx) / c.fx, y0 = ((double)pts[2 * i + 1] - c.cy) / c.fy;
        double x = x0, y = y0;
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y;
            const double icd = 1.0 / (1.0 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
            const double dx = 2.0 * c.p1 * x * y + c.p2 * (r2 + 2.0 * x * x);
            const double dy = c.p1 * (r2 + 2.0 * y * y) + 2.0 * c.p2 * x * y;
            x = (x0 - dx) * icd;
            y = (y0 - dy) * icd;
        }
        nxy[2 * i] = x;
        nxy[2 * i + 1] = y;
    }
}

__device__ inline bool ippe_translation(const double *cx, const double *cy, const double *nxy, const double *R, double *t)
{
    double a02 = 0, a12 = 0, a22 = 0, b0 = 0, b1 = 0, b2 = 0;
    for (int i = 0; i < 4; i++) {
        const double X = cx[i], Y = cy[i], x = nxy[2 * i], y = nxy[2 * i + 1];
        const double rx = R[0] * X + R[1] * Y, ry = R[3] * X + R[4] * Y, rz = R[6] * X + R[7] * Y;
        const double e1 = x * rz - rx, e2 = y * rz - ry;
        a02 -= x; a12 -= y; a22 += x * x + y * y;
        b0 += e1; b1 += e2; b2 += -x * e1 - y * e2;
    }
    // A = [[4,0,a02],[0,4,a12],[a02,a12,a22]]
    const double det = 4.0 * (4.0 * a22 - a12 * a12) - a02 * (4.0 * a02);
    if (!(fabs(det) > 1e-300)) return false;
    const double i00 = 4.0 * a22 - a12 * a12, i01 = a02 * a12, i02 = -4.0 * a02;
    const double i11 = 4.0 * a22 - a02 * a02, i12 = -4.0 * a12, i22 = 16.0;
    t[0] = (i00 * b0 + i01 * b1 + i02 * b2) / det;
    t[1] = (i01 * b0 + i11 * b1 + i12 * b2) / det;
    t[2] = (i02 * b0 + i12 * b1 + i22 * b2) / det;
    return true;
}

__device__ inline void rot_to_rvec(const double *R, double *r)
{
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double ax = R[7] - R[5], ay = R[2] - R[6], az = R[3] - R[1];
    const double s = 0.5 * sqrt(ax * ax + ay * ay + az * az);
    const double th = atan2(s, c);
    if (s > 1e-9) {
        const double k = th / (2.0 * s);
        r[0] = ax * k; r[1] = ay * k; r[2] = az * k;
    } else if (c > 0.0) {
        r[0] = r[1] = r[2] = 0.0;
    } else {
        double xx = sqrt(fmax((R[0] + 1.0) * 0.5, 0.0));
        double yy = sqrt(fmax((R[4] + 1.0) * 0.5, 0.0));
        double zz = sqrt(fmax((R[8] + 1.0) * 0.5, 0.0));
        if (R[1] + R[3] < 0.0) yy = -yy;
        if (R[2] + R[6] < 0.0) zz = -zz;
        if (xx == 0.0 && R[5] + R[7] < 0.0) zz = -zz;
        const double nn = sqrt(xx * xx + yy * yy + zz * zz);
        r[0] = th * xx / nn; r[1] = th * yy / nn; r[2] = th * zz / nn;
    }
}

__device__ inline void rot_to_quat(const double *R, double *q)
{
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0.0) {
        double s = sqrt(tr + 1.0);
        q[3] = s * 0.5;
        s = 0.5 / s;
        q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {
        double s = sqrt(R[0] - R[4] - R[8] + 1.0);
        q[0] = s * 0.5; s = 0.5 / s;
        q[3] = (R[7] - R[5]) * s; q[1] = (R[3] + R[1]) * s; q[2] = (R[6] + R[2]) * s;
    } else if (R[4] >= R[8]) {
        double s = sqrt(R[4] - R[8] - R[0] + 1.0);
        q[1] = s * 0.5; s = 0.5 / s;
        q[3] = (R[2] - R[6]) * s; q[2] = (R[7] + R[5]) * s; q[0] = (R[1] + R[3]) * s;
    } else {
        double s = sqrt(R[8] - R[0] - R[4] + 1.0);
        q[2] = s * 0.5; s = 0.5 / s;
        q[3] = (R[3] - R[1]) * s; q[0] = (R[2] + R[6]) * s; q[1] = (R[5] + R[7]) * s;
    }
}

// pts: LB, LT, RT, RB in source-frame pixels (src/pnp_solver.cpp:41-44)
__device__ inline bool solve_pnp_ippe(const PnpConst &c, const float *pts, int armor_size, double *rvec, double *tvec, double *quat)
{
    const double hy = c.hy[armor_size], hz = c.hz[armor_size];
    double nxy[8];
    undistort4(c, pts, nxy);
    // canonical (Xc, Yc) of LB, LT, RT, RB = (y_model, z_model)
    const double cX[4] = {hy, hy, -hy, -hy}, cY[4] = {-hz, hz, hz, -hz};
    // unit square (u, v) = ((hy - Xc) / 2hy, (Yc + hz) / 2hz): (0,0)=LB (1,0)=RB (1,1)=RT (0,1)=LT
    const double x0 = nxy[0], y0 = nxy[1], x1 = nxy[6], y1 = nxy[7], x2 = nxy[4], y2 = nxy[5], x3 = nxy[2], y3 = nxy[3];
    const double dx1 = x1 - x2, dx2 = x3 - x2, sx = x0 - x1 + x2 - x3;
    const double dy1 = y1 - y2, dy2 = y3 - y2, sy = y0 - y1 + y2 - y3;
    const double den = dx1 * dy2 - dy1 * dx2;
    if (!(fabs(den) > 1e-300)) return false;
    const double gg = (sx * dy2 - dx2 * sy) / den, hh = (dx1 * sy - sx * dy1) / den;
    const double sa = x1 - x0 + gg * x1, sb = x3 - x0 + hh * x3, sc = x0;
    const double sd = y1 - y0 + gg * y1, se = y3 - y0 + hh * y3, sf = y0;
    // H = Hs * [[-1/2hy, 0, 1/2], [0, 1/2hz, 1/2], [0, 0, 1]]
    const double iu = -0.5 / hy, iv = 0.5 / hz;
    double H[9] = {sa * iu, sb * iv, 0.5 * sa + 0.5 * sb + sc,
                   sd * iu, se * iv, 0.5 * sd + 0.5 * se + sf,
                   gg * iu, hh * iv, 0.5 * gg + 0.5 * hh + 1.0};
    if (!(fabs(H[8]) > 1e-300)) return false;
    const double ih = 1.0 / H[8];
    for (int i = 0; i < 9; i++) H[i] *= ih;
    const double p = H[2], q = H[5];
    const double j00 = H[0] - H[6] * p, j01 = H[1] - H[7] * p, j10 = H[3] - H[6] * q, j11 = H[4] - H[7] * q;

    // rotation taking the optical axis onto the ray through the plane origin
    double rv[9];
    const double s = sqrt(p * p + q * q + 1.0), t = sqrt(p * p + q * q);
    const double costh = 1.0 / s, sinth = sqrt(1.0 - 1.0 / (s * s));
    if (t < 1e-300) {
        rv[0] = 1; rv[1] = 0; rv[2] = 0; rv[3] = 0; rv[4] = 1; rv[5] = 0; rv[6] = 0; rv[7] = 0; rv[8] = 1;
    } else {
        const double k0 = p / t, k1 = q / t;
        rv[0] = (costh - 1.0) * k0 * k0 + 1.0; rv[1] = k0 * k1 * (costh - 1.0); rv[2] = k0 * sinth;
        rv[3] = rv[1]; rv[4] = (costh - 1.0) * k1 * k1 + 1.0; rv[5] = k1 * sinth;
        rv[6] = -k0 * sinth; rv[7] = -k1 * sinth; rv[8] = (costh - 1.0) * (k0 * k0 + k1 * k1) + 1.0;
    }
    const double b00 = rv[0] - p * rv[6], b01 = rv[1] - p * rv[7], b10 = rv[3] - q * rv[6], b11 = rv[4] - q * rv[7];
    const double bdet = b00 * b11 - b01 * b10;
    if (!(fabs(bdet) > 1e-300)) return false;
    const double dti = 1.0 / bdet;
    const double a00 = dti * (b11 * j00 - b01 * j10), a01 = dti * (b11 * j01 - b01 * j11);
    const double a10 = dti * (-b10 * j00 + b00 * j10), a11 = dti * (-b10 * j01 + b00 * j11);
    const double ata00 = a00 * a00 + a01 * a01, ata01 = a00 * a10 + a01 * a11, ata11 = a10 * a10 + a11 * a11;
    const double g2 = 0.5 * (ata00 + ata11 + sqrt((ata00 - ata11) * (ata00 - ata11) + 4.0 * ata01 * ata01));
    if (!(g2 > 0.0)) return false;
    const double gam = sqrt(g2);
    const double r00 = a00 / gam, r01 = a01 / gam, r10 = a10 / gam, r11 = a11 / gam;
    const double bb0 = sqrt(fmax(1.0 - r00 * r00 - r10 * r10, 0.0));
    double bb1 = sqrt(fmax(1.0 - r01 * r01 - r11 * r11, 0.0));
    if (-r00 * r01 - r10 * r11 < 0.0) bb1 = -bb1;

    Pose best, other;
    for (int sol = 0; sol < 2; sol++) {
        const double c0 = sol ? -bb0 : bb0, c1 = sol ? -bb1 : bb1;
        const double m[9] = {r00, r01, r10 * c1 - c0 * r11, r10, r11, c0 * r01 - r00 * c1, c0, c1, r00 * r11 - r01 * r10};
        Pose ps;
        double Rc[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rc[i * 3 + j] = rv[i * 3] * m[j] + rv[i * 3 + 1] * m[3 + j] + rv[i * 3 + 2] * m[6 + j];
        if (!ippe_translation(cX, cY, nxy, Rc, ps.t)) return false;
        double e = 0.0;
        for (int i = 0; i < 4; i++) {
            const double X = Rc[0] * cX[i] + Rc[1] * cY[i] + ps.t[0];
            const double Y = Rc[3] * cX[i] + Rc[4] * cY[i] + ps.t[1];
            const double Z = Rc[6] * cX[i] + Rc[7] * cY[i] + ps.t[2];
            const double ex = X / Z - nxy[2 * i], ey = Y / Z - nxy[2 * i + 1];
            e += ex * ex + ey * ey;
        }
        ps.err = sqrt(e / 8.0);
        // model = canonical^T: columns (x_m, y_m, z_m) = (Zc, Xc, Yc)
        for (int i = 0; i < 3; i++) {
            ps.R[i * 3 + 0] = Rc[i * 3 + 2];
            ps.R[i * 3 + 1] = Rc[i * 3 + 0];
            ps.R[i * 3 + 2] = Rc[i * 3 + 1];
        }
        if (sol == 0) best = ps; else other = ps;
    }
    if (!(best.err <= other.err)) best = other;
    rot_to_rvec(best.R, rvec);
    tvec[0] = best.t[0]; tvec[1] = best.t[1]; tvec[2] = best.t[2];
    if (quat) rot_to_quat(best.R, quat);
    const double chk = rvec[0] + rvec[1] + rvec[2] + tvec[0] + tvec[1] + tvec[2];
    return chk == chk && fabs(chk) < 1e300;
}

// The same solver spread over a PAIR of lanes (2 consecutive lanes, both active, same arguments; the result is valid on both).
// The two data-parallel parts run split over the pair -- points q and q + 2 of the iterative undistortion, IPPE solution q --
// everything else is computed redundantly in lockstep.  Per point / per solution the operation sequence is the one of
// solve_pnp_ippe(), so the results are bit-identical to it; degenerate inputs are carried as a flag instead of early returns
// so that the shuffles stay converged.
// (Round 4: a pair, not a quad.  The solver is ~ 1 300 fp64 instructions at 4 issue cycles each: ISSUE bound, so what counts is
// how many waves share a SIMD.  A hundred survivors on quads are seven waves, two to a SIMD: 12 k cycles; on pairs they are
// four waves, one per SIMD, each ~ 150 instructions longer: 7 k.)
__device__ inline double quad_bcast(double v, int src_lane) { return __shfl(v, src_lane); }

// (px[h], py[h]): point q + 2 h of the armor (LB, LT, RT, RB) in source-frame pixels -- BY VALUE: a pointer indexed by the lane
// parity had put the caller's array (and with it the whole record) into scratch memory
__device__ inline bool solve_pnp_ippe_pair(const PnpConst &c, float px0, float py0, float px1, float py1, int armor_size, double *rvec, double *tvec, double *quat)
{
    const int lane = threadIdx.x & 63, base = lane & ~1, q = lane & 1;
    const double hy = c.hy[armor_size], hz = c.hz[armor_size];
    bool ok = true;
    double nxy[8];
    {   // points q and q + 2 on this lane (undistort4's loop body, the two chains interleaved by the scheduler)
        double xs[2], ys[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float px = h ? px1 : px0, py = h ? py1 : py0;
            const double x0 = ((double)px - c.cx) / c.fx, y0 = ((double)py - c.cy) / c.fy;
            double x = x0, y = y0;
            for (int it = 0; it < 5; it++) {
                const double r2 = x * x + y * y;
                const double icd = 1.0 / (1.0 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
                const double dx = 2.0 * c.p1 * x * y + c.p2 * (r2 + 2.0 * x * x);
                const double dy = c.p1 * (r2 + 2.0 * y * y) + 2.0 * c.p2 * x * y;
                x = (x0 - dx) * icd;
                y = (y0 - dy) * icd;
            }
            xs[h] = x;
            ys[h] = y;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {           // point i sits on lane (i & 1), slot (i >> 1)
            nxy[2 * i] = quad_bcast(xs[i >> 1], base + (i & 1));
            nxy[2 * i + 1] = quad_bcast(ys[i >> 1], base + (i & 1));
        }
    }
    const double cX[4] = {hy, hy, -hy, -hy}, cY[4] = {-hz, hz, hz, -hz};
    const double x0 = nxy[0], y0 = nxy[1], x1 = nxy[6], y1 = nxy[7], x2 = nxy[4], y2 = nxy[5], x3 = nxy[2], y3 = nxy[3];
    const double dx1 = x1 - x2, dx2 = x3 - x2, sx = x0 - x1 + x2 - x3;
    const double dy1 = y1 - y2, dy2 = y3 - y2, sy = y0 - y1 + y2 - y3;
    double den = dx1 * dy2 - dy1 * dx2;
    if (!(fabs(den) > 1e-300)) { ok = false; den = 1.0; }
    const double gg = (sx * dy2 - dx2 * sy) / den, hh = (dx1 * sy - sx * dy1) / den;
    const double sa = x1 - x0 + gg * x1, sb = x3 - x0 + hh * x3, sc = x0;
    const double sd = y1 - y0 + gg * y1, se = y3 - y0 + hh * y3, sf = y0;
    const double iu = -0.5 / hy, iv = 0.5 / hz;
    double H[9] = {sa * iu, sb * iv, 0.5 * sa + 0.5 * sb + sc,
                   sd * iu, se * iv, 0.5 * sd + 0.5 * se + sf,
                   gg * iu, hh * iv, 0.5 * gg + 0.5 * hh + 1.0};
    if (!(fabs(H[8]) > 1e-300)) { ok = false; H[8] = 1.0; }
    const double ih = 1.0 / H[8];
    for (int i = 0; i < 9; i++) H[i] *= ih;
    const double p = H[2], qq = H[5];
    const double j00 = H[0] - H[6] * p, j01 = H[1] - H[7] * p, j10 = H[3] - H[6] * qq, j11 = H[4] - H[7] * qq;
    double rv[9];
    const double s = sqrt(p * p + qq * qq + 1.0), t = sqrt(p * p + qq * qq);
    const double costh = 1.0 / s, sinth = sqrt(1.0 - 1.0 / (s * s));
    if (t < 1e-300) {
        rv[0] = 1; rv[1] = 0; rv[2] = 0; rv[3] = 0; rv[4] = 1; rv[5] = 0; rv[6] = 0; rv[7] = 0; rv[8] = 1;
    } else {
        const double k0 = p / t, k1 = qq / t;
        rv[0] = (costh - 1.0) * k0 * k0 + 1.0; rv[1] = k0 * k1 * (costh - 1.0); rv[2] = k0 * sinth;
        rv[3] = rv[1]; rv[4] = (costh - 1.0) * k1 * k1 + 1.0; rv[5] = k1 * sinth;
        rv[6] = -k0 * sinth; rv[7] = -k1 * sinth; rv[8] = (costh - 1.0) * (k0 * k0 + k1 * k1) + 1.0;
    }
    const double b00 = rv[0] - p * rv[6], b01 = rv[1] - p * rv[7], b10 = rv[3] - qq * rv[6], b11 = rv[4] - qq * rv[7];
    double bdet = b00 * b11 - b01 * b10;
    if (!(fabs(bdet) > 1e-300)) { ok = false; bdet = 1.0; }
    const double dti = 1.0 / bdet;
    const double a00 = dti * (b11 * j00 - b01 * j10), a01 = dti * (b11 * j01 - b01 * j11);
    const double a10 = dti * (-b10 * j00 + b00 * j10), a11 = dti * (-b10 * j01 + b00 * j11);
    const double ata00 = a00 * a00 + a01 * a01, ata01 = a00 * a10 + a01 * a11, ata11 = a10 * a10 + a11 * a11;
    double g2 = 0.5 * (ata00 + ata11 + sqrt((ata00 - ata11) * (ata00 - ata11) + 4.0 * ata01 * ata01));
    if (!(g2 > 0.0)) { ok = false; g2 = 1.0; }
    const double gam = sqrt(g2);
    const double r00 = a00 / gam, r01 = a01 / gam, r10 = a10 / gam, r11 = a11 / gam;
    const double bb0 = sqrt(fmax(1.0 - r00 * r00 - r10 * r10, 0.0));
    double bb1 = sqrt(fmax(1.0 - r01 * r01 - r11 * r11, 0.0));
    if (-r00 * r01 - r10 * r11 < 0.0) bb1 = -bb1;

    // solution q on this lane
    Pose ps;
    bool sol_ok;
    {
        const int sol = q;
        const double c0 = sol ? -bb0 : bb0, c1 = sol ? -bb1 : bb1;
        const double m[9] = {r00, r01, r10 * c1 - c0 * r11, r10, r11, c0 * r01 - r00 * c1, c0, c1, r00 * r11 - r01 * r10};
        double Rc[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rc[i * 3 + j] = rv[i * 3] * m[j] + rv[i * 3 + 1] * m[3 + j] + rv[i * 3 + 2] * m[6 + j];
        sol_ok = ippe_translation(cX, cY, nxy, Rc, ps.t);
        if (!sol_ok) { ps.t[0] = ps.t[1] = 0.0; ps.t[2] = 1.0; }
        double e = 0.0;
        for (int i = 0; i < 4; i++) {
            const double X = Rc[0] * cX[i] + Rc[1] * cY[i] + ps.t[0];
            const double Y = Rc[3] * cX[i] + Rc[4] * cY[i] + ps.t[1];
            const double Z = Rc[6] * cX[i] + Rc[7] * cY[i] + ps.t[2];
            const double ex = X / Z - nxy[2 * i], ey = Y / Z - nxy[2 * i + 1];
            e += ex * ex + ey * ey;
        }
        ps.err = sqrt(e / 8.0);
        for (int i = 0; i < 3; i++) {
            ps.R[i * 3 + 0] = Rc[i * 3 + 2];
            ps.R[i * 3 + 1] = Rc[i * 3 + 0];
            ps.R[i * 3 + 2] = Rc[i * 3 + 1];
        }
    }
    // solve_pnp_ippe(): a failed translation of EITHER solution fails the call; else `best` = solution 0 unless !(err0 <= err1)
    const int ok0 = __shfl((int)sol_ok, base), ok1 = __shfl((int)sol_ok, base + 1);
    const double e0 = quad_bcast(ps.err, base), e1 = quad_bcast(ps.err, base + 1);
    ok = ok && ok0 && ok1;
    const int win = base + ((e0 <= e1) ? 0 : 1);
    Pose best;
#pragma unroll
    for (int i = 0; i < 9; i++) best.R[i] = quad_bcast(ps.R[i], win);
#pragma unroll
    for (int i = 0; i < 3; i++) best.t[i] = quad_bcast(ps.t[i], win);
    rot_to_rvec(best.R, rvec);
    tvec[0] = best.t[0]; tvec[1] = best.t[1]; tvec[2] = best.t[2];
    if (quat) rot_to_quat(best.R, quat);
    const double chk = rvec[0] + rvec[1] + rvec[2] + tvec[0] + tvec[1] + tvec[2];
    return ok && chk == chk && fabs(chk) < 1e300;
}

}  // namespace irmv
