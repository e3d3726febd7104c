/*
 * orc_net.c -- CPU oracle, network part: YOLOv8n forward in fp32 from an .irmw blob.
 *
 * TEST INFRASTRUCTURE ONLY (see irmv_oracle.h).  PARITY UNPINNED: the reference
 * runs an opaque TensorRT engine (src/yolo_engine.cpp:105) that is not in the
 * repository; the graph below is the published Ultralytics YOLOv8n (scale n)
 * definition as tabulated in SURVEY.md Appendix A, with nc = 14
 * (include/irmv_detection/armor.hpp:7) and an optional pose-style keypoint
 * branch.  Checked against torch-CPU (F.conv2d / max_pool2d / interpolate) in
 * tests/test_oracle_net.py.
 *
 * Written for obviousness, not speed: concat and upsample are explicit copies,
 * every conv is a direct loop nest (vectorised over output channels, OpenMP
 * over output pixels).
 */
#include "irmv_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- fp16 round trip (IEEE binary16, round-to-nearest-even) ------------- */
static uint16_t f32_to_f16_bits(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u : 0));
    if (absx >= 0x477ff000u) /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    if (absx < 0x38800000u) { /* subnormal half or zero */
        if (absx < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 */
        uint32_t exp = absx >> 23;
        uint32_t man = (absx & 0x7fffffu) | 0x800000u;
        uint32_t shift = 126u - exp; /* 14..24 */
        uint32_t half = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half & 1u))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t exp = (absx >> 23) - 112u;
    uint32_t man = absx & 0x7fffffu;
    uint32_t half = (exp << 10) | (man >> 13);
    uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;
    return (uint16_t)(sign | half);
}

static float f16_bits_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t x;
    if (exp == 0) {
        if (man == 0) {
            x = sign;
        } else {
            int e = -1;
            do { e++; man <<= 1; } while (!(man & 0x400u));
            x = sign | ((uint32_t)(112 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        x = sign | 0x7f800000u | (man << 13);
    } else {
        x = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

static inline float round_f16(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }

/* exported for tests */
float orc_round_f16(float f) { return round_f16(f); }

/* ---- blob --------------------------------------------------------------- */
struct orc_conv {
    char name[32];
    int cin, cout, k, stride, act;
    int groups; /* > 1: depthwise (groups == cout, cin == 1) */
    float *w; /* [k*k][cin][cout] */
    float *b; /* [cout] */
};

struct orc_net {
    int nc, nk, n;
    int backbone; /* 0: YOLOv8n C2f stages; 1: ShuffleNetV2 stages (irmv_detection_amd/arch.py) */
    struct orc_conv *L;
};

#pragma pack(push, 1)
struct blob_header { char magic[4]; uint32_t version, nc, nk, reg_max, n_layers, dtype, reserved; };
struct blob_layer { char name[32]; uint32_t cin, cout, k, stride, act, pad; uint64_t w_off, b_off; };
#pragma pack(pop)

orc_net *orc_net_load(const uint8_t *blob, size_t bytes)
{
    if (bytes < sizeof(struct blob_header)) return NULL;
    struct blob_header h;
    memcpy(&h, blob, sizeof h);
    /* dtype 1: fp16 weights; dtype 2: int8 weights + fp32 per-output-channel scales, used as w = fp16(q * scale) */
    if (memcmp(h.magic, "IRMW", 4) != 0 || h.version != 1 || (h.dtype != 1 && h.dtype != 2) || h.reg_max != 16)
        return NULL;
    orc_net *n = calloc(1, sizeof *n);
    n->nc = (int)h.nc;
    n->nk = (int)h.nk;
    n->n = (int)h.n_layers;
    n->backbone = (int)h.reserved;
    n->L = calloc(n->n, sizeof *n->L);
    for (int i = 0; i < n->n; i++) {
        struct blob_layer l;
        memcpy(&l, blob + sizeof h + (size_t)i * sizeof l, sizeof l);
        struct orc_conv *c = &n->L[i];
        memcpy(c->name, l.name, 32);
        c->name[31] = 0;
        c->cin = l.cin; c->cout = l.cout; c->k = l.k; c->stride = l.stride; c->act = l.act;
        c->groups = l.pad > 1 ? (int)l.pad : 1;
        size_t nw = (size_t)c->cout * c->k * c->k * c->cin;
        const size_t w_bytes = h.dtype == 2 ? ((nw + 3) & ~(size_t)3) + (size_t)c->cout * 4 : nw * 2;
        if (l.w_off + w_bytes > bytes || l.b_off + (size_t)c->cout * 4 > bytes) {
            orc_net_free(n);
            return NULL;
        }
        c->w = malloc(nw * sizeof(float));
        c->b = malloc((size_t)c->cout * sizeof(float));
        const uint16_t *wh = (const uint16_t *)(blob + l.w_off); /* OHWI */
        const int8_t *wq = (const int8_t *)(blob + l.w_off);
        const float *scale = (const float *)(blob + l.w_off + ((nw + 3) & ~(size_t)3));
        int taps = c->k * c->k;
        for (int o = 0; o < c->cout; o++)
            for (int t = 0; t < taps; t++)
                for (int ci = 0; ci < c->cin; ci++) {
                    const size_t src = ((size_t)o * taps + t) * c->cin + ci;
                    c->w[((size_t)t * c->cin + ci) * c->cout + o] =
                        h.dtype == 2 ? orc_round_f16((float)wq[src] * scale[o]) : f16_bits_to_f32(wh[src]);
                }
        memcpy(c->b, blob + l.b_off, (size_t)c->cout * 4);
    }
    return n;
}

void orc_net_free(orc_net *n)
{
    if (!n) return;
    for (int i = 0; i < n->n; i++) { free(n->L[i].w); free(n->L[i].b); }
    free(n->L);
    free(n);
}

int orc_net_nc(const orc_net *n) { return n->nc; }
int orc_net_nk(const orc_net *n) { return n->nk; }
int orc_head_channels(const orc_net *n) { return 64 + n->nc + n->nk; }
/* worker threads of the forward pass (bench.py's cpu_baseline reports the 1-thread and the all-core rate) */
void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_anchors(int net) { return (net / 8) * (net / 8) + (net / 16) * (net / 16) + (net / 32) * (net / 32); }

static const struct orc_conv *find_layer(const orc_net *n, const char *name)
{
    for (int i = 0; i < n->n; i++)
        if (strcmp(n->L[i].name, name) == 0) return &n->L[i];
    return NULL;
}

/* ---- tensors ------------------------------------------------------------ */
typedef struct { float *d; int H, W, C; int owned; } T;

static T talloc(int H, int W, int C)
{
    T t = { calloc((size_t)H * W * C, sizeof(float)), H, W, C, 1 };
    return t;
}
static void tfree(T *t) { if (t->owned) free(t->d); t->d = NULL; }

static inline float silu(float v) { return v / (1.0f + expf(-v)); }

/* y[:, :, yoff:yoff+cout] = act(conv(x[:, :, xoff:xoff+cin]) + b) (+ res[:, :, roff:..]) */
static int conv(const orc_net *n, const char *name, const T *x, int xoff, T *y, int yoff,
                const T *res, int roff, int emu)
{
    const struct orc_conv *c = find_layer(n, name);
    if (!c) { fprintf(stderr, "oracle: no layer %s\n", name); return -1; }
    const int k = c->k, s = c->stride, pad = k / 2, cin = c->cin, cout = c->cout;
    const int Ho = (x->H + 2 * pad - k) / s + 1, Wo = (x->W + 2 * pad - k) / s + 1;
    if (Ho != y->H || Wo != y->W || xoff + cin > x->C || yoff + cout > y->C || cout > 256) {
        fprintf(stderr, "oracle: shape mismatch at %s\n", name);
        return -1;
    }
#pragma omp parallel for schedule(static)
    for (int p = 0; p < Ho * Wo; p++) {
        const int oy = p / Wo, ox = p % Wo;
        float acc[256];
        for (int o = 0; o < cout; o++) acc[o] = c->b[o];
        for (int kh = 0; kh < k; kh++) {
            const int iy = oy * s - pad + kh;
            if (iy < 0 || iy >= x->H) continue;
            for (int kw = 0; kw < k; kw++) {
                const int ix = ox * s - pad + kw;
                if (ix < 0 || ix >= x->W) continue;
                const float *xp = x->d + ((size_t)iy * x->W + ix) * x->C + xoff;
                const float *wp = c->w + (size_t)(kh * k + kw) * cin * cout;
                for (int ci = 0; ci < cin; ci++) {
                    const float xv = xp[ci];
                    const float *wr = wp + (size_t)ci * cout;
                    for (int o = 0; o < cout; o++) acc[o] += xv * wr[o];
                }
            }
        }
        float *yp = y->d + (size_t)p * y->C + yoff;
        const float *rp = res ? res->d + (size_t)p * res->C + roff : NULL;
        for (int o = 0; o < cout; o++) {
            float v = acc[o];
            if (c->act == 1) v = silu(v);
            if (rp) v += rp[o];
            yp[o] = (emu && c->act == 1) ? round_f16(v) : v;
        }
    }
    return 0;
}

/* depthwise 3x3 (groups == channels), bias, no activation: y[:, :, yoff + c] = b[c] + sum_taps x[:, :, xoff + c] * w[tap][c].
 * emu: the engine stores this tensor in fp16 like every other activation. */
static int dwconv(const orc_net *n, const char *name, const T *x, int xoff, T *y, int yoff, int emu)
{
    const struct orc_conv *c = find_layer(n, name);
    if (!c) { fprintf(stderr, "oracle: no layer %s\n", name); return -1; }
    const int k = c->k, s = c->stride, pad = k / 2, C = c->cout;
    const int Ho = (x->H + 2 * pad - k) / s + 1, Wo = (x->W + 2 * pad - k) / s + 1;
    if (c->groups != C || c->cin != 1 || Ho != y->H || Wo != y->W || xoff + C > x->C || yoff + C > y->C) {
        fprintf(stderr, "oracle: shape mismatch at %s\n", name);
        return -1;
    }
#pragma omp parallel for schedule(static)
    for (int p = 0; p < Ho * Wo; p++) {
        const int oy = p / Wo, ox = p % Wo;
        float *yp = y->d + (size_t)p * y->C + yoff;
        for (int ch = 0; ch < C; ch++) {
            float acc = c->b[ch];
            for (int kh = 0; kh < k; kh++) {
                const int iy = oy * s - pad + kh;
                if (iy < 0 || iy >= x->H) continue;
                for (int kw = 0; kw < k; kw++) {
                    const int ix = ox * s - pad + kw;
                    if (ix < 0 || ix >= x->W) continue;
                    acc += x->d[((size_t)iy * x->W + ix) * x->C + xoff + ch] * c->w[(size_t)(kh * k + kw) * C + ch];
                }
            }
            if (c->act == 1) acc = silu(acc);
            yp[ch] = emu ? round_f16(acc) : acc;
        }
    }
    return 0;
}

/* out[:, :, 2 i] = a[:, :, aoff + i], out[:, :, 2 i + 1] = b[:, :, boff + i]: concat + channel shuffle with two groups */
static T shuffle_cat(const T *a, int aoff, const T *b, int boff, int bc)
{
    T o = talloc(a->H, a->W, 2 * bc);
    for (int p = 0; p < o.H * o.W; p++)
        for (int i = 0; i < bc; i++) {
            o.d[(size_t)p * o.C + 2 * i] = a->d[(size_t)p * a->C + aoff + i];
            o.d[(size_t)p * o.C + 2 * i + 1] = b->d[(size_t)p * b->C + boff + i];
        }
    return o;
}

/* ShuffleNetV2 stride-2 block (arch.py _shuffle_down): out = shuffle(cat(pw(dw(x)), pw2(dw(pw1(x))))) */
static int shuffle_down(const orc_net *net, const char *prefix, const T *in, int c2, int emu, T *out)
{
    char nm[64];
    const int bc = c2 / 2, Ho = in->H / 2, Wo = in->W / 2;
    T d1 = talloc(Ho, Wo, in->C), b1 = talloc(Ho, Wo, bc), p1 = talloc(in->H, in->W, bc), d2 = talloc(Ho, Wo, bc), b2 = talloc(Ho, Wo, bc);
    snprintf(nm, sizeof nm, "%s.b1.dw", prefix);  int rc = dwconv(net, nm, in, 0, &d1, 0, emu);
    snprintf(nm, sizeof nm, "%s.b1.pw", prefix);  rc |= conv(net, nm, &d1, 0, &b1, 0, NULL, 0, emu);
    snprintf(nm, sizeof nm, "%s.b2.pw1", prefix); rc |= conv(net, nm, in, 0, &p1, 0, NULL, 0, emu);
    snprintf(nm, sizeof nm, "%s.b2.dw", prefix);  rc |= dwconv(net, nm, &p1, 0, &d2, 0, emu);
    snprintf(nm, sizeof nm, "%s.b2.pw2", prefix); rc |= conv(net, nm, &d2, 0, &b2, 0, NULL, 0, emu);
    *out = shuffle_cat(&b1, 0, &b2, 0, bc);
    tfree(&d1); tfree(&b1); tfree(&p1); tfree(&d2); tfree(&b2);
    return rc;
}

/* ShuffleNetV2 stride-1 unit (arch.py _shuffle_unit): x1, x2 = split(x); out = shuffle(cat(x1, pw2(dw(pw1(x2))))) */
static int shuffle_unit(const orc_net *net, const char *prefix, const T *in, int emu, T *out)
{
    char nm[64];
    const int bc = in->C / 2;
    T p1 = talloc(in->H, in->W, bc), d2 = talloc(in->H, in->W, bc), b2 = talloc(in->H, in->W, bc);
    snprintf(nm, sizeof nm, "%s.b2.pw1", prefix); int rc = conv(net, nm, in, bc, &p1, 0, NULL, 0, emu);
    snprintf(nm, sizeof nm, "%s.b2.dw", prefix);  rc |= dwconv(net, nm, &p1, 0, &d2, 0, emu);
    snprintf(nm, sizeof nm, "%s.b2.pw2", prefix); rc |= conv(net, nm, &d2, 0, &b2, 0, NULL, 0, emu);
    *out = shuffle_cat(in, 0, &b2, 0, bc);
    tfree(&p1); tfree(&d2); tfree(&b2);
    return rc;
}

/* 5x5 stride-1 pad-2 max pool, slice -> slice (padding never wins: -inf) */
static void maxpool5(const T *x, int xoff, T *y, int yoff, int C)
{
#pragma omp parallel for schedule(static)
    for (int p = 0; p < x->H * x->W; p++) {
        const int oy = p / x->W, ox = p % x->W;
        for (int ch = 0; ch < C; ch++) {
            float m = -INFINITY;
            for (int dy = -2; dy <= 2; dy++) {
                const int iy = oy + dy;
                if (iy < 0 || iy >= x->H) continue;
                for (int dx = -2; dx <= 2; dx++) {
                    const int ix = ox + dx;
                    if (ix < 0 || ix >= x->W) continue;
                    const float v = x->d[((size_t)iy * x->W + ix) * x->C + xoff + ch];
                    if (v > m) m = v;
                }
            }
            y->d[(size_t)p * y->C + yoff + ch] = m;
        }
    }
}

/* out = concat(nearest_upsample2x(a), b) along channels */
static T up_cat(const T *a, const T *b)
{
    T o = talloc(b->H, b->W, a->C + b->C);
    for (int y = 0; y < o.H; y++)
        for (int x = 0; x < o.W; x++) {
            float *op = o.d + ((size_t)y * o.W + x) * o.C;
            memcpy(op, a->d + ((size_t)(y / 2) * a->W + x / 2) * a->C, (size_t)a->C * 4);
            memcpy(op + a->C, b->d + ((size_t)y * b->W + x) * b->C, (size_t)b->C * 4);
        }
    return o;
}

static T cat2(const T *a, const T *b)
{
    T o = talloc(a->H, a->W, a->C + b->C);
    for (size_t p = 0; p < (size_t)o.H * o.W; p++) {
        memcpy(o.d + p * o.C, a->d + p * a->C, (size_t)a->C * 4);
        memcpy(o.d + p * o.C + a->C, b->d + p * b->C, (size_t)b->C * 4);
    }
    return o;
}

struct tapctx { const char *want; float *out; int *shape; };

static void tap(struct tapctx *tc, const char *name, const T *t)
{
    if (!tc->want || !tc->out || strcmp(tc->want, name) != 0) return;
    memcpy(tc->out, t->d, (size_t)t->H * t->W * t->C * 4);
    if (tc->shape) { tc->shape[0] = t->H; tc->shape[1] = t->W; tc->shape[2] = t->C; }
}

/* C2f(c1, c2, n, shortcut): SURVEY.md Appendix A "Blocks" */
static int c2f(const orc_net *net, const char *prefix, const T *in, int c2, int nb, int shortcut,
               int emu, T *out)
{
    char nm[64];
    const int c = c2 / 2;
    T cat = talloc(in->H, in->W, (2 + nb) * c);
    snprintf(nm, sizeof nm, "%s.cv1", prefix);
    int rc = conv(net, nm, in, 0, &cat, 0, NULL, 0, emu);
    for (int i = 0; i < nb && !rc; i++) {
        T tmp = talloc(in->H, in->W, c);
        snprintf(nm, sizeof nm, "%s.m.%d.cv1", prefix, i);
        rc |= conv(net, nm, &cat, (1 + i) * c, &tmp, 0, NULL, 0, emu);
        snprintf(nm, sizeof nm, "%s.m.%d.cv2", prefix, i);
        rc |= conv(net, nm, &tmp, 0, &cat, (2 + i) * c, shortcut ? &cat : NULL, (1 + i) * c, emu);
        tfree(&tmp);
    }
    *out = talloc(in->H, in->W, c2);
    snprintf(nm, sizeof nm, "%s.cv2", prefix);
    rc |= conv(net, nm, &cat, 0, out, 0, NULL, 0, emu);
    tfree(&cat);
    return rc;
}

static int conv_new(const orc_net *net, const char *name, const T *in, int cout, int stride, int emu,
                    T *out)
{
    *out = talloc(in->H / stride, in->W / stride, cout);
    return conv(net, name, in, 0, out, 0, NULL, 0, emu);
}

int orc_net_forward(const orc_net *n, const float *in_chw, int net, int emu, float *head,
                    const char *tapname, float *tap_out, int *tap_shape)
{
    if (net % 32 != 0) return -1;
    struct tapctx tc = { tapname, tap_out, tap_shape };
    int rc = 0;
    T x = talloc(net, net, 3);
    for (int c = 0; c < 3; c++)
        for (int p = 0; p < net * net; p++) {
            float v = in_chw[(size_t)c * net * net + p];
            x.d[(size_t)p * 3 + c] = emu ? round_f16(v) : v;
        }
    T a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a12, a15, a16, a18, a19, a21;
    rc |= conv_new(n, "model.0.conv", &x, 16, 2, emu, &a0);  tap(&tc, "0", &a0);
    rc |= conv_new(n, "model.1.conv", &a0, 32, 2, emu, &a1); tap(&tc, "1", &a1);
    if (n->backbone == 1) {
        /* ShuffleNetV2 stages of the same widths: a4 (P3), a6 (P4), a8 (P5) keep their meaning; a2, a3, a5, a7 are the
         * blocks in between (tensor names = block indices) */
        rc |= shuffle_down(n, "model.2", &a1, 64, emu, &a2);  tap(&tc, "2", &a2);
        rc |= shuffle_unit(n, "model.3", &a2, emu, &a4);      tap(&tc, "3", &a4);
        rc |= shuffle_down(n, "model.4", &a4, 128, emu, &a3); tap(&tc, "4", &a3);
        rc |= shuffle_unit(n, "model.5", &a3, emu, &a5);      tap(&tc, "5", &a5);
        rc |= shuffle_unit(n, "model.6", &a5, emu, &a6);      tap(&tc, "6", &a6);
        rc |= shuffle_down(n, "model.7", &a6, 256, emu, &a7); tap(&tc, "7", &a7);
        rc |= shuffle_unit(n, "model.8", &a7, emu, &a8);      tap(&tc, "8", &a8);
    } else {
    rc |= c2f(n, "model.2", &a1, 32, 1, 1, emu, &a2);        tap(&tc, "2", &a2);
    rc |= conv_new(n, "model.3.conv", &a2, 64, 2, emu, &a3); tap(&tc, "3", &a3);
    rc |= c2f(n, "model.4", &a3, 64, 2, 1, emu, &a4);        tap(&tc, "4", &a4);
    rc |= conv_new(n, "model.5.conv", &a4, 128, 2, emu, &a5); tap(&tc, "5", &a5);
    rc |= c2f(n, "model.6", &a5, 128, 2, 1, emu, &a6);       tap(&tc, "6", &a6);
    rc |= conv_new(n, "model.7.conv", &a6, 256, 2, emu, &a7); tap(&tc, "7", &a7);
    rc |= c2f(n, "model.8", &a7, 256, 1, 1, emu, &a8);       tap(&tc, "8", &a8);
    }
    { /* SPPF */
        T cat = talloc(a8.H, a8.W, 512);
        rc |= conv(n, "model.9.cv1", &a8, 0, &cat, 0, NULL, 0, emu);
        maxpool5(&cat, 0, &cat, 128, 128);
        maxpool5(&cat, 128, &cat, 256, 128);
        maxpool5(&cat, 256, &cat, 384, 128);
        a9 = talloc(a8.H, a8.W, 256);
        rc |= conv(n, "model.9.cv2", &cat, 0, &a9, 0, NULL, 0, emu);
        tfree(&cat);
    }
    tap(&tc, "9", &a9);
    { T c11 = up_cat(&a9, &a6);  rc |= c2f(n, "model.12", &c11, 128, 1, 0, emu, &a12); tfree(&c11); }
    tap(&tc, "12", &a12);
    { T c14 = up_cat(&a12, &a4); rc |= c2f(n, "model.15", &c14, 64, 1, 0, emu, &a15); tfree(&c14); }
    tap(&tc, "15", &a15);
    rc |= conv_new(n, "model.16.conv", &a15, 64, 2, emu, &a16); tap(&tc, "16", &a16);
    { T c17 = cat2(&a16, &a12);  rc |= c2f(n, "model.18", &c17, 128, 1, 0, emu, &a18); tfree(&c17); }
    tap(&tc, "18", &a18);
    rc |= conv_new(n, "model.19.conv", &a18, 128, 2, emu, &a19); tap(&tc, "19", &a19);
    { T c20 = cat2(&a19, &a9);   rc |= c2f(n, "model.21", &c20, 256, 1, 0, emu, &a21); tfree(&c20); }
    tap(&tc, "21", &a21);

    /* Detect head: per level, box (cv2) / cls (cv3) / kpt (cv4) branches write
     * channel slices of the level's [H*W][no] block of `head`. */
    const int no = orc_head_channels(n);
    const T *P[3] = { &a15, &a18, &a21 };
    size_t base = 0;
    for (int i = 0; i < 3 && !rc; i++) {
        T lvl = { head + base * no, P[i]->H, P[i]->W, no, 0 };
        const char *br[3] = { "cv2", "cv3", "cv4" };
        const int mid[3] = { 64, 64, 16 };
        const int off[3] = { 0, 64, 64 + n->nc };
        for (int b = 0; b < (n->nk > 0 ? 3 : 2); b++) {
            char nm[64];
            T t1 = talloc(lvl.H, lvl.W, mid[b]), t2 = talloc(lvl.H, lvl.W, mid[b]);
            snprintf(nm, sizeof nm, "model.22.%s.%d.0", br[b], i);
            rc |= conv(n, nm, P[i], 0, &t1, 0, NULL, 0, emu);
            snprintf(nm, sizeof nm, "22.%s.%d.0", br[b], i); tap(&tc, nm, &t1);
            snprintf(nm, sizeof nm, "model.22.%s.%d.1", br[b], i);
            rc |= conv(n, nm, &t1, 0, &t2, 0, NULL, 0, emu);
            snprintf(nm, sizeof nm, "22.%s.%d.1", br[b], i); tap(&tc, nm, &t2);
            snprintf(nm, sizeof nm, "model.22.%s.%d.2", br[b], i);
            rc |= conv(n, nm, &t2, 0, &lvl, off[b], NULL, 0, emu);
            tfree(&t1); tfree(&t2);
        }
        base += (size_t)lvl.H * lvl.W;
    }
    T *all[] = { &x, &a0, &a1, &a2, &a3, &a4, &a5, &a6, &a7, &a8, &a9, &a12, &a15, &a16, &a18, &a19, &a21 };
    for (size_t i = 0; i < sizeof all / sizeof *all; i++) tfree(all[i]);
    return rc;
}

int orc_conv_layer(const orc_net *n, const char *layer_name, const float *x_nhwc, int H, int W,
                   float *y_nhwc)
{
    const struct orc_conv *c = find_layer(n, layer_name);
    if (!c) return -1;
    T x = { (float *)x_nhwc, H, W, c->cin, 0 };
    T y = { y_nhwc, H / c->stride, W / c->stride, c->cout, 0 };
    return conv(n, layer_name, &x, 0, &y, 0, NULL, 0, 0);
}
