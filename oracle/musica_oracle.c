/*
 * musica_oracle.c — CPU restatement of the reference's MUSICA path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see musica_oracle.h).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root). Emulation rules fixed here:
 *   Q1 out-of-bounds imageLoad returns 0, out-of-bounds imageStore /
 *      imageAtomicAdd is dropped;
 *   Q2 never-written texels read as 0 (images are calloc'ed);
 *   Q3 bare `clamp(...)` statements are no-ops;
 *   Q4 uvec4(float) stored to an r32f image == float(uint(value));
 *   Q5 pow(x, 2) == x * x; pow(r, 5.0) in img_relevant == ((r*r)*(r*r))*r
 *      (GLSL leaves pow's precision undefined; a multiplication chain is the
 *      restatement both this file and the HIP kernels use);
 *   Q6 IEEE binary32 everywhere, float->int truncates, float->uint of a
 *      negative value saturates to 0, NaN -> dropped where it would index;
 *   Q7 the hard-coded constants.
 * Extra decisions where the reference is undefined:
 *   - constant arrays such as weight[5] are folded by glslang in double and
 *     narrowed once: {0.1f, 0.25f, 0.3f, 0.25f, 0.1f};
 *   - meanCount / meanSum with meanSum == 0 (gradation_curve_generate.comp:74)
 *     yields 0;
 *   - L == 4 makes the reference's exponent 0/0 (src/vk_processing.cpp:270-274);
 *     the exponent is taken as 0 (factor 1) when L - 3 <= 1.
 * Compile with -ffp-contract=off and without -ffast-math.
 */
#include "musica_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define WG 32            /* WORKGROUP_SIZE in every 2-D shader */
#define REDUCE_AREA 8    /* img_max_reduce.comp:5 */
#define HIST_AREA 16     /* noise_hist.comp:5 */
#define MAX_NOISE_VALUE 0.1f /* noise_hist.comp:7 */
#define MAX_CNR_VALUE 256.0f /* img_cnr.comp:6 */

static int g_threads = 1;
void musica_oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int musica_oracle_get_threads(void) { return g_threads; }

#ifdef _OPENMP
#define PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(g_threads)")
#else
#define PAR_FOR
#endif

/* imageLoad with Q1. */
static inline float ld(const float* im, int side, int x, int y) {
    return (x >= 0 && y >= 0 && x < side && y < side) ? im[(size_t)y * side + x] : 0.0f;
}

static inline uint32_t ceil_div_u(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

/* Q4/Q6: float -> uint as the GPU converts (truncate, negative and NaN -> 0, saturate high). */
static inline uint32_t f2u(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

/* ---------------------------------------------------------------------- */
/* K1 img_sqrt.comp:10-18 */
void musica_oracle_k_sqrt(const uint16_t* in, uint32_t side, float* out) {
    size_t n = (size_t)side * side;
    PAR_FOR
    for (size_t i = 0; i < n; i++) out[i] = sqrtf((float)in[i]);
}

/* K2 img_max_reduce.comp:12-56 (one link; out side = ceil(side / 8), src/vk_processing.cpp:52-54) */
void musica_oracle_k_max_reduce(const float* in, uint32_t side, float* out) {
    int os = (int)ceil_div_u(side, REDUCE_AREA);
    PAR_FOR
    for (int iy = 0; iy < os; iy++)
        for (int ix = 0; ix < os; ix++) {
            float maxValue = 0.0f;                                   /* :19 */
            for (int m = 0; m < REDUCE_AREA; m++)                    /* :25 */
                for (int n = 0; n < REDUCE_AREA; n++) {              /* :26 */
                    float p = ld(in, (int)side, ix * REDUCE_AREA + m, iy * REDUCE_AREA + n);
                    maxValue = p > maxValue ? p : maxValue;          /* :39 */
                }
            out[(size_t)iy * os + ix] = (float)f2u(maxValue);        /* :53-55, Q4 */
        }
}

/* K3 min_reduce.comp:12-32 */
void musica_oracle_k_min_reduce(const float* in, uint32_t side, float* out) {
    int os = (int)ceil_div_u(side, REDUCE_AREA);
    PAR_FOR
    for (int iy = 0; iy < os; iy++)
        for (int ix = 0; ix < os; ix++) {
            float minValue = ld(in, (int)side, ix, iy);              /* :19 — invocationCoord, not the block base */
            for (int m = 0; m < REDUCE_AREA; m++)
                for (int n = 0; n < REDUCE_AREA; n++) {
                    float p = ld(in, (int)side, ix * REDUCE_AREA + m, iy * REDUCE_AREA + n);
                    minValue = p < minValue ? p : minValue;          /* :26 */
                }
            out[(size_t)iy * os + ix] = (float)f2u(minValue);        /* :30-31, Q4 */
        }
}

/* K4 img_normalize.comp:13-35 (the clamp at :27 discards its result, Q3) */
void musica_oracle_k_normalize(const float* in, uint32_t side, float minv, float maxv, float* out) {
    size_t n = (size_t)side * side;
    PAR_FOR
    for (size_t i = 0; i < n; i++) out[i] = (in[i] - minv) / (maxv - minv); /* :24 */
}

/* mirror() of img_smooth.comp:10-16 (the clamp at :14 is a no-op, Q3) */
static inline int mirror(int n, int lo, int hi) {
    int v = n;
    if (v > hi) v = hi - (v - hi);
    else if (v < lo) v = lo + (lo - v);
    return v;
}

static const float W5[5] = {0.1f, 0.25f, 0.3f, 0.25f, 0.1f}; /* img_smooth.comp:23-30 */

/* 5x5 Burt-Adelson smooth with per-tap gain (gain == 0 => no gain factor in the product).
 * ORDER_REFERENCE: img_smooth.comp:32-45 / img_smooth_upsampled.comp:32-45 literally.
 * ORDER_FAST: vertical pass v(x) = sum_n w[n] s(x, y+n-2), then horizontal
 * pass sum_m w[m] v(x+m-2), each a left-to-right chain starting from the
 * first product; the x4 gain is applied once to the final sum (exact, a
 * power of two). */
static void smooth5(const float* in, int side, float* out, int order, int gain4) {
    int hi = side - 1;
    if (order == MUSICA_ORDER_REFERENCE) {
        PAR_FOR
        for (int y = 0; y < side; y++)
            for (int x = 0; x < side; x++) {
                float pixel = 0.0f;
                for (int m = 0; m < 5; m++)
                    for (int n = 0; n < 5; n++) {
                        float c = ld(in, side, mirror(x + m - 2, 0, hi), mirror(y + n - 2, 0, hi));
                        if (gain4) pixel += W5[m] * W5[n] * 4.0f * c;  /* img_smooth_upsampled.comp:43 */
                        else pixel += W5[m] * W5[n] * c;               /* img_smooth.comp:43 */
                    }
                out[(size_t)y * side + x] = pixel;
            }
    } else {
        float* v = (float*)malloc((size_t)side * side * sizeof(float));
        PAR_FOR
        for (int y = 0; y < side; y++)
            for (int x = 0; x < side; x++) {
                float acc = W5[0] * ld(in, side, x, mirror(y - 2, 0, hi));
                acc = acc + W5[1] * ld(in, side, x, mirror(y - 1, 0, hi));
                acc = acc + W5[2] * ld(in, side, x, mirror(y, 0, hi));
                acc = acc + W5[3] * ld(in, side, x, mirror(y + 1, 0, hi));
                acc = acc + W5[4] * ld(in, side, x, mirror(y + 2, 0, hi));
                v[(size_t)y * side + x] = acc;
            }
        PAR_FOR
        for (int y = 0; y < side; y++)
            for (int x = 0; x < side; x++) {
                float acc = W5[0] * ld(v, side, mirror(x - 2, 0, hi), y);
                acc = acc + W5[1] * ld(v, side, mirror(x - 1, 0, hi), y);
                acc = acc + W5[2] * ld(v, side, mirror(x, 0, hi), y);
                acc = acc + W5[3] * ld(v, side, mirror(x + 1, 0, hi), y);
                acc = acc + W5[4] * ld(v, side, mirror(x + 2, 0, hi), y);
                out[(size_t)y * side + x] = gain4 ? 4.0f * acc : acc;
            }
        free(v);
    }
}

/* K5 img_smooth.comp:18-50 */
void musica_oracle_k_smooth(const float* in, uint32_t side, float* out, int order) {
    smooth5(in, (int)side, out, order, 0);
}

/* K8 img_smooth_upsampled.comp:18-50 */
void musica_oracle_k_smooth_upsampled(const float* in, uint32_t side, float* out, int order) {
    smooth5(in, (int)side, out, order, 1);
}

/* K6 img_downsample.comp:10-20 (out side = ceil(side / 2), src/vk_processing.cpp:116) */
void musica_oracle_k_downsample(const float* in, uint32_t side, float* out) {
    int os = (int)ceil_div_u(side, 2);
    PAR_FOR
    for (int y = 0; y < os; y++)
        for (int x = 0; x < os; x++) out[(size_t)y * os + x] = ld(in, (int)side, 2 * x, 2 * y); /* :15 */
}

/* K7 img_upsample.comp:10-20: out(2x, 2y) = in(x, y); other texels keep their value (Q2: zero). */
void musica_oracle_k_upsample(const float* in, uint32_t in_side, float* out, uint32_t out_side) {
    PAR_FOR
    for (int y = 0; y < (int)in_side; y++)
        for (int x = 0; x < (int)in_side; x++) {
            int ox = 2 * x, oy = 2 * y;
            if (ox < (int)out_side && oy < (int)out_side)            /* Q1: dropped store */
                out[(size_t)oy * out_side + ox] = in[(size_t)y * in_side + x];
        }
}

/* K9 img_difference.comp:11-20 */
void musica_oracle_k_difference(const float* a, const float* b, uint32_t side, float* out) {
    size_t n = (size_t)side * side;
    PAR_FOR
    for (size_t i = 0; i < n; i++) out[i] = a[i] - b[i];
}

/* K17 img_addition.comp:11-20 */
void musica_oracle_k_addition(const float* a, const float* b, uint32_t side, float* out) {
    size_t n = (size_t)side * side;
    PAR_FOR
    for (size_t i = 0; i < n; i++) out[i] = a[i] + b[i];
}

/* K10 img_sdev.comp:10-35 (pow(pixel, 2) -> pixel * pixel, Q5; OOB taps are 0, divisor stays 25) */
void musica_oracle_k_sdev(const float* in, uint32_t side_u, float* out, int order) {
    int side = (int)side_u;
    if (order == MUSICA_ORDER_REFERENCE) {
        PAR_FOR
        for (int y = 0; y < side; y++)
            for (int x = 0; x < side; x++) {
                float sum = 0.0f;
                for (int m = 0; m < 5; m++)
                    for (int n = 0; n < 5; n++) {
                        float p = ld(in, side, x + m - 2, y + n - 2);  /* :19-21 */
                        sum += p * p;                                  /* :23 */
                    }
                out[(size_t)y * side + x] = sqrtf(sum / 25.0f);        /* :30 */
            }
    } else {
        /* ORDER_FAST: vertical sums of squares, then horizontal sums, left-to-right chains. */
        float* q = (float*)malloc((size_t)side * side * sizeof(float));
        PAR_FOR
        for (int y = 0; y < side; y++)
            for (int x = 0; x < side; x++) {
                float a = ld(in, side, x, y - 2), b = ld(in, side, x, y - 1), c = ld(in, side, x, y);
                float d = ld(in, side, x, y + 1), e = ld(in, side, x, y + 2);
                float acc = a * a;
                acc = acc + b * b;
                acc = acc + c * c;
                acc = acc + d * d;
                acc = acc + e * e;
                q[(size_t)y * side + x] = acc;
            }
        PAR_FOR
        for (int y = 0; y < side; y++)
            for (int x = 0; x < side; x++) {
                float acc = ld(q, side, x - 2, y);
                acc = acc + ld(q, side, x - 1, y);
                acc = acc + ld(q, side, x, y);
                acc = acc + ld(q, side, x + 1, y);
                acc = acc + ld(q, side, x + 2, y);
                out[(size_t)y * side + x] = sqrtf(acc / 25.0f);
            }
        free(q);
    }
}

/* K11 noise_hist.comp:14-49. `groups` workgroups of 32x32 threads per axis; each
 * thread scans a 16x16 area, m (x) outer, n (y) inner; `break` leaves only the n loop. */
void musica_oracle_k_noise_hist(const float* sdev, uint32_t side, uint32_t groups, uint32_t* hist) {
    int threads = (int)groups * WG;
    for (int gy = 0; gy < threads; gy++)
        for (int gx = 0; gx < threads; gx++) {
            int bx = gx * HIST_AREA, by = gy * HIST_AREA;             /* :15-18 */
            if (bx >= (int)side || by >= (int)side) continue;          /* every load would be 0 -> break */
            for (int m = 0; m < HIST_AREA; m++)
                for (int n = 0; n < HIST_AREA; n++) {
                    float cur = ld(sdev, (int)side, bx + m, by + n);  /* :22-23 */
                    if (cur == 0.0f) break;                            /* :29 */
                    if (cur != cur) break;                             /* int(NaN) undefined: restated as bin 0 -> break (:39) */
                    float adj = cur / MAX_NOISE_VALUE;                 /* :31 */
                    if (adj > 1.0f) break;                             /* :33 */
                    int bin = (int)(adj * (float)MUSICA_NOISE_BINS + 0.5f); /* :35 */
                    if (bin == 0) break;                               /* :39 */
                    if (bin >= 0 && bin < MUSICA_NOISE_BINS) hist[bin] += 1u; /* :45, Q1 */
                }
        }
}

/* K12 img_histogram_max.comp:14-32 */
void musica_oracle_k_histogram_max(const uint32_t* hist, uint32_t bins, musica_hist_max_point* out) {
    out->maxValue = 0;
    out->maxBin = 0;
    for (uint32_t i = 0; i < bins; i++)
        if (hist[i] > out->maxValue) {                                 /* :25 — first maximum wins */
            out->maxValue = hist[i];
            out->maxBin = i;
        }
}

/* interpolate() contrast_curve_generate.comp:28-31 */
static inline float interpolate(float from, float to, float percent) {
    float difference = to - from;
    return from + (difference * percent);
}

/* generateCurve() contrast_curve_generate.comp:39-54 (steps = 11, i <= 10) and
 * gradation_curve_generate.comp:30-46 (steps = 10, i < 10). */
static void generate_curve(musica_point* pts, uint32_t* count, musica_point s, musica_point mid, musica_point e, uint32_t steps) {
    for (uint32_t i = 0; i < steps; i++) {
        float t = (float)i / 10.0f;
        float xa = interpolate(s.x, mid.x, t);
        float ya = interpolate(s.y, mid.y, t);
        float xb = interpolate(mid.x, e.x, t);
        float yb = interpolate(mid.y, e.y, t);
        float x = interpolate(xa, xb, t);
        float y = interpolate(ya, yb, t);
        pts[*count].x = x;
        pts[*count].y = y;
        (*count)++;
    }
}

static inline musica_point P(float x, float y) {
    musica_point p;
    p.x = x;
    p.y = y;
    return p;
}

/* K13 contrast_curve_generate.comp:56-94. Stale points beyond pointsCount stay as they are
 * (Q2: a fresh buffer is zero). */
void musica_oracle_k_contrast_curve_generate(musica_hist_max_point mp, musica_contrast_params cp, musica_contrast_curve* out) {
    uint32_t cnt = 0;
    float low = cp.lowContrastFactor, high = cp.highContrastFactor;
    if (low == 1.0f) {                                                 /* :59 */
        out->points[cnt++] = P(0.0f, high);                            /* :68 */
        out->points[cnt++] = P(1.0f, high);                            /* :69 */
    } else {
        float p = (float)mp.maxBin * (1.0f / (float)MUSICA_NOISE_BINS) * MAX_NOISE_VALUE; /* :71 */
        generate_curve(out->points, &cnt, P(0.0f, 1.0f), P(p * 4.0f / 5.0f, low), P(p, low), 11);          /* :72-76 */
        generate_curve(out->points, &cnt, P(p, low), P(p * 6.0f / 5.0f, low), P(p * 7.0f / 5.0f, low * 4.0f / 5.0f), 11); /* :77-81 */
        generate_curve(out->points, &cnt, P(p * 7.0f / 5.0f, low * 4.0f / 5.0f), P(p * 2.0f, 1.0f), P(1.0f, 1.0f), 11);   /* :82-86 */
    }
    out->pointsCount = cnt;
}

/* getY() contrast_curve_apply.comp:27-36 == img_apply_gradation_curve.comp:27-36.
 * points[i + 1] one past pointsCount reads whatever the buffer holds (Q2: zero for a fresh one);
 * callers pass arrays with at least count + 1 entries. */
float musica_oracle_get_y(const musica_point* points, uint32_t count, float x) {
    for (uint32_t i = 0; i < count; i++) {
        if (points[i].x == x) return points[i].y;                      /* :29 */
        if (points[i].x <= x && points[i + 1].x >= x) {                /* :31 */
            float m = (points[i + 1].y - points[i].y) / (points[i + 1].x - points[i].x); /* linearFunction :22-25 */
            return m * (x - points[i].x) + points[i].y;
        }
    }
    return 0.0f;                                                       /* :35 */
}

/* K14 contrast_curve_apply.comp:38-66 */
void musica_oracle_k_contrast_curve_apply(const float* band, const float* sdev, uint32_t side, const musica_contrast_curve* curve, float* out) {
    size_t n = (size_t)side * side;
    PAR_FOR
    for (size_t i = 0; i < n; i++) {
        float s = sdev ? sdev[i] : 0.0f;                               /* never-written sdev image (levels >= 4), Q2 */
        out[i] = band[i] * musica_oracle_get_y(curve->points, curve->pointsCount, s); /* :61 */
    }
}

/* K15 img_cnr.comp:18-48 */
void musica_oracle_k_cnr(const float* sdev, uint32_t side, musica_hist_max_point mp, float* out) {
    float ref = (float)mp.maxBin * (1.0f / (float)MUSICA_NOISE_BINS) * MAX_NOISE_VALUE; /* :22 */
    if (ref == 0.0f) ref = (1.0f / (float)MUSICA_NOISE_BINS) * MAX_NOISE_VALUE;       /* :25 */
    size_t n = (size_t)side * side;
    PAR_FOR
    for (size_t i = 0; i < n; i++) {
        float cnr = sdev[i] / ref;                                     /* :31 */
        out[i] = cnr / MAX_CNR_VALUE;                                  /* :43 */
    }
}

/* linearFunction() noise_reduction.comp:24-31 — note m * x, not m * (x - p1.x). */
static inline float nr_factor(musica_nr_params p, float cnr) {
    if (cnr < p.lowCnr) return p.lowFactor;
    else if (cnr > p.highCnr) return p.highFactor;
    else {
        float m = (p.highFactor - p.lowFactor) / (p.highCnr - p.lowCnr);
        return m * cnr + p.lowFactor;
    }
}

/* K16 noise_reduction.comp:33-60 */
void musica_oracle_k_noise_reduction(const float* band, uint32_t side, const float* cnr, uint32_t cnr_side, musica_nr_params p, float* out) {
    uint32_t scale = f2u(ceilf((float)side / (float)cnr_side));       /* :38 (int / float) */
    PAR_FOR
    for (int y = 0; y < (int)side; y++)
        for (int x = 0; x < (int)side; x++) {
            float c = ld(cnr, (int)cnr_side, (int)((uint32_t)x / scale), (int)((uint32_t)y / scale)) * MAX_CNR_VALUE; /* :39-45 */
            float f = nr_factor(p, c);                                 /* :47 */
            out[(size_t)y * side + x] = band[(size_t)y * side + x] * f; /* :57 */
        }
}

/* K18 img_relevant.comp:28-64 */
void musica_oracle_k_relevant(const float* normalized, uint32_t side, const float* cnr, uint32_t cnr_side, float* out) {
    const uint32_t border = 100;                                       /* :21 */
    const float lowLimit = 1.0f, ramp = 5.0f, highLimit = MAX_CNR_VALUE; /* :23-25 */
    uint32_t scale = f2u(ceilf((float)side / (float)cnr_side));       /* :32 */
    PAR_FOR
    for (int y = 0; y < (int)side; y++)
        for (int x = 0; x < (int)side; x++) {
            float pixel = normalized[(size_t)y * side + x];
            float c = ld(cnr, (int)cnr_side, (int)((uint32_t)x / scale), (int)((uint32_t)y / scale)) * MAX_CNR_VALUE;
            /* int vs uint comparisons of :46-49 are done in uint (size.x - border wraps when side < 100) */
            uint32_t ux = (uint32_t)x, uy = (uint32_t)y, lim = side - border;
            int inside = ux > border && ux < lim && uy > border && uy < lim;
            float v;
            if (c >= lowLimit && c <= lowLimit + ramp && inside) {     /* :44-50 */
                float r = c / (lowLimit + ramp);
                v = ((r * r) * (r * r)) * r;                           /* pow(r, 5.0) :51, Q5 */
            } else if (c >= lowLimit + ramp && c <= highLimit && pixel <= 0.90f && inside) { /* :53-61 */
                v = 1.0f;
            } else v = 0.0f;
            out[(size_t)y * side + x] = v;
        }
}

/* K19 gradation_histogram.comp:14-34: `return` ends the whole thread at the first zero pixel. */
void musica_oracle_k_gradation_histogram(const float* img, const float* relevant, uint32_t side, uint32_t groups, uint32_t* hist) {
    int threads = (int)groups * WG;
    for (int gy = 0; gy < threads; gy++)
        for (int gx = 0; gx < threads; gx++) {
            int bx = gx * HIST_AREA, by = gy * HIST_AREA;
            int stop = 0;
            for (int m = 0; m < HIST_AREA && !stop; m++)
                for (int n = 0; n < HIST_AREA; n++) {
                    float cur = ld(img, (int)side, bx + m, by + n);   /* :22 */
                    if (cur == 0.0f) { stop = 1; break; }              /* :24 return */
                    if (cur != cur) continue;                          /* Q6: NaN never indexes */
                    float scaled = cur * (float)MUSICA_GRAD_BINS;      /* :26 */
                    if (!(scaled > -2147483648.0f && scaled < 2147483648.0f)) continue;
                    int bin = (int)scaled;
                    float rel = ld(relevant, (int)side, bx + m, by + n); /* :28 */
                    if (bin >= 0 && bin < MUSICA_GRAD_BINS) hist[bin] += f2u(rel * 100.0f); /* :30, Q1 */
                }
        }
}

/* K20 gradation_curve_generate.comp:50-193. All integer arithmetic is uint32 and wraps. */
void musica_oracle_k_gradation_curve_generate(const uint32_t* hist, musica_grad_curve* out) {
    const uint32_t lowestRelevantGradBin = 10;                         /* :48 */
    float m = 3.0f;                                                    /* :52 */
    float t0 = 0.0f, t1 = 0.0f, ta = 0.0f, y_m = 0.5f;                 /* :54-57 */
    uint32_t maxCount = 0, maxPosition = 0, meanCount = 0, meanSum = 0;
    for (uint32_t i = lowestRelevantGradBin; i < MUSICA_GRAD_BINS; i++) { /* :67-72 */
        uint32_t count = hist[i] / 100u;
        meanCount += count * i;
        meanSum += count;
    }
    uint32_t meanQuot = meanSum ? meanCount / meanSum : 0u;            /* :74 (x / 0 restated as 0) */
    float meanHistPos = (float)meanQuot / (float)MUSICA_GRAD_BINS;
    uint32_t upper = f2u(meanHistPos * (float)MUSICA_GRAD_BINS);       /* :77 */
    for (uint32_t i = lowestRelevantGradBin; i < upper; i++) {
        uint32_t count = hist[i < MUSICA_GRAD_BINS ? i : 0] / 100u;
        if (i >= MUSICA_GRAD_BINS) count = 0;                          /* Q1 */
        if (count > maxCount) {                                        /* :80 */
            maxCount = count;
            maxPosition = i;
        }
    }
    uint32_t lowThreshold = f2u((float)maxCount * 0.05f);              /* :88 */
    for (uint32_t i = maxPosition; i > 0; i--) {                       /* :94-105 */
        uint32_t count = hist[i] / 100u;
        float position = (float)i * (1.0f / (float)MUSICA_GRAD_BINS);
        if (count >= lowThreshold && position > 0.0f) t0 = position;
        else break;
    }
    for (uint32_t i = maxPosition; i < MUSICA_GRAD_BINS; i++) {        /* :108-119 */
        uint32_t count = hist[i] / 100u;
        float position = (float)i * (1.0f / (float)MUSICA_GRAD_BINS);
        if (count > 0) t1 = position;
        else break;
    }
    float maxHistPos = (float)maxPosition * (1.0f / (float)MUSICA_GRAD_BINS); /* :121 */
    ta = maxHistPos;                                                   /* :133 */
    t0 -= 0.01f;                                                       /* :140 */
    if (t0 < 0.0f) t0 = 0.0f;
    if (t1 > 1.0f) t1 = 1.0f;                                          /* :144 */
    float tf = -(0.5f / m) + ta;                                       /* :146 */
    if (tf < t0) tf = t0;                                              /* :149 */
    uint32_t idx = 0;
    out->points[idx++] = P(0.0f, 0.0f);                                /* :152 */
    generate_curve(out->points, &idx, P(t0, 0.0f), P(tf, 0.0f), P(ta, y_m), 10); /* :156-160 */
    if (tf == t0) m = y_m / (ta - tf);                                 /* :162-163 */
    float ts = (y_m / m) + ta;                                         /* :165 */
    generate_curve(out->points, &idx, P(ta, y_m), P(ts, 1.0f), P(t1, 1.0f), 10); /* :169-173 */
    out->points[idx++] = P(1.0f, 1.0f);                                /* :179 */
    out->pointsCount = idx;                                            /* :181 */
    out->t0 = t0;
    out->ta = ta;
    out->t1 = t1;
}

/* K21 img_apply_gradation_curve.comp:38-47 */
void musica_oracle_k_apply_gradation_curve(const float* in, uint32_t side, const musica_grad_curve* curve, float* out) {
    size_t n = (size_t)side * side;
    PAR_FOR
    for (size_t i = 0; i < n; i++) out[i] = musica_oracle_get_y(curve->points, curve->pointsCount, in[i]);
}

/* ---- CLAHE trio (reference: disabled, restated from shader text) -------- */

/* K22 clahe_histogram.comp:13-45. hist layout [tx][ty][bin]. */
void musica_oracle_k_clahe_histogram(const float* img, const float* relevant, uint32_t side, uint32_t* hist) {
    const int T = MUSICA_CLAHE_TILES, B = MUSICA_CLAHE_BINS;
    for (int y = 0; y < (int)side; y++)
        for (int x = 0; x < (int)side; x++) {
            float cur = img[(size_t)y * side + x];
            if (cur != cur) continue;
            float scaled = cur * (float)(B - 1) + 0.5f;                /* :20 */
            if (!(scaled > -2147483648.0f && scaled < 2147483648.0f)) continue;
            int bin = (int)scaled;
            uint32_t tx = f2u((float)x / (float)side * (float)T);      /* :34 */
            uint32_t ty = f2u((float)y / (float)side * (float)T);      /* :35 */
            float rel = relevant[(size_t)y * side + x];
            if (rel == 1.0f && bin >= 0 && bin < B && tx < (uint32_t)T && ty < (uint32_t)T) /* :39-44, Q1 */
                hist[((size_t)tx * T + ty) * B + bin] += 1u;
        }
}

/* K23 clahe_grad_curve.comp:21-100. points layout [tx][ty][i]. */
void musica_oracle_k_clahe_grad_curve(const uint32_t* hist, musica_point* points) {
    const int T = MUSICA_CLAHE_TILES, B = MUSICA_CLAHE_BINS;
    for (int tx = 0; tx < T; tx++)
        for (int ty = 0; ty < T; ty++) {
            const uint32_t* h = hist + ((size_t)tx * T + ty) * B;
            musica_point* pts = points + ((size_t)tx * T + ty) * B;
            float ny[MUSICA_CLAHE_BINS];
            uint32_t count = 0;
            for (int i = 0; i < B; i++) count += h[i];                 /* :31-43 */
            for (int i = 0; i < B; i++) ny[i] = (float)h[i] / (float)count; /* :47-57 */
            float clipLimit = 1.0f / 32.0f, clipCount = 0.0f;          /* :60-61 */
            for (int i = 0; i < B; i++)
                if (ny[i] > clipLimit) {                               /* :63-69 */
                    float diff = ny[i] - clipLimit;
                    clipCount += diff;
                    ny[i] = clipLimit;
                }
            float clipAdd = clipCount / (float)B;                      /* :76 */
            for (int i = 0; i < B; i++) ny[i] += clipAdd;              /* :78-80 */
            float curr = 0.0f;
            for (int i = 0; i < B; i++) {                              /* :85-93 */
                curr += ny[i];
                float posX = (float)i * (1.0f / (float)B);
                if (i == B - 1) posX = 1.0f;
                pts[i] = P(posX, curr);
            }
        }
}

/* getY() of clahe_grad_curve_apply.comp:27-36 — 256 points, points[256] restated as (0, 0). */
static float clahe_get_y(const musica_point* pts, float x) {
    const int B = MUSICA_CLAHE_BINS;
    for (int i = 0; i < B; i++) {
        if (pts[i].x == x) return pts[i].y;
        musica_point nx = (i + 1 < B) ? pts[i + 1] : P(0.0f, 0.0f);
        if (pts[i].x <= x && nx.x >= x) {
            float m = (nx.y - pts[i].y) / (nx.x - pts[i].x);
            return m * (x - pts[i].x) + pts[i].y;
        }
    }
    return 0.0f;
}

static inline float signf_(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

/* K24 clahe_grad_curve_apply.comp:38-161 */
void musica_oracle_k_clahe_grad_curve_apply(const float* in, uint32_t side, const musica_point* points, float* out) {
    const int T = MUSICA_CLAHE_TILES, B = MUSICA_CLAHE_BINS;
    uint32_t G = side / (uint32_t)T;                                   /* :43 */
    PAR_FOR
    for (int y = 0; y < (int)side; y++)
        for (int x = 0; x < (int)side; x++) {
            float pixel = in[(size_t)y * side + x];
            float px = (float)x / (float)G, py = (float)y / (float)G;  /* :45-48 */
            float bx = (float)f2u(px) + 0.5f, by = (float)f2u(py) + 0.5f; /* :50-53 */
            float dx = px - bx, dy = py - by;                          /* :55-58 */
            float cx[4], cy[4];
            int cnt, usex, usey;
            if (dx == 0.0f && dy == 0.0f) { cnt = 1; usex = 0; usey = 0; cx[0] = bx; cy[0] = by; }
            else if (dx == 0.0f) { cnt = 2; usex = 0; usey = 1; cx[0] = bx; cy[0] = by; cx[1] = bx; cy[1] = by + signf_(dy); }
            else if (dy == 0.0f) { cnt = 2; usex = 1; usey = 0; cx[0] = bx; cy[0] = by; cx[1] = bx + signf_(dx); cy[1] = by; }
            else {
                cnt = 4; usex = 1; usey = 1;
                cx[0] = bx; cy[0] = by;
                cx[1] = bx + signf_(dx); cy[1] = by;
                cx[2] = bx; cy[2] = by + signf_(dy);
                cx[3] = bx + signf_(dx); cy[3] = by + signf_(dy);
            }
            float combined = 0.0f;
            if (cnt == 1) {
                uint32_t tx = f2u(floorf(bx)), ty = f2u(floorf(by));   /* :63 (no clip there) */
                if (tx < (uint32_t)T && ty < (uint32_t)T) combined = clahe_get_y(points + ((size_t)tx * T + ty) * B, pixel);
            } else {
                for (int i = 0; i < cnt; i++) {
                    float tdx = cx[i] - px, tdy = cy[i] - py;
                    uint32_t tx = f2u(floorf(cx[i])), ty = f2u(floorf(cy[i]));
                    if (tx > (uint32_t)T - 1) tx = T - 1;              /* :78-79 */
                    if (ty > (uint32_t)T - 1) ty = T - 1;
                    float g = clahe_get_y(points + ((size_t)tx * T + ty) * B, pixel);
                    if (usex && usey) combined += (1.0f - fabsf(tdx)) * (1.0f - fabsf(tdy)) * g; /* :138-146 */
                    else if (usey) combined += (1.0f - fabsf(tdy)) * g; /* :81-88 */
                    else combined += (1.0f - fabsf(tdx)) * g;          /* :107-114 */
                }
            }
            out[(size_t)y * side + x] = combined;
        }
}

/* ---- host parameter formulas ------------------------------------------ */

/* include/vk_processing.h:39-49 and the two #defines of :16-17, as data (musica_tunables, include/musica.h) */
void musica_oracle_tunables_default(musica_tunables* t) {
    t->nr_high_cnr = 9.0f; t->nr_max_high_factor = 1.2f; t->nr_low_cnr = 3.0f; t->nr_min_low_factor = 0.6f;
    t->high_contrast_max_reduction = 0.2f; t->low_contrast_max_enhancement = 3.0f;
    t->linear_low_contrast = 0u; t->linear_high_contrast = 0u;
}

/* src/vk_processing.cpp:259-293, every branch (tunables vk_processing.h:44-49) */
musica_contrast_params musica_oracle_host_contrast_params_ex(uint32_t i, uint32_t levels, const musica_tunables* t) {
    const uint32_t coarserLevelsStart = MUSICA_COARSER_LEVELS_START;
    const float highContrastMaxReduction = t->high_contrast_max_reduction, lowContrastMaxEnhancment = t->low_contrast_max_enhancement;
    musica_contrast_params cp;
    uint32_t coarserLevelsCount = levels - coarserLevelsStart;
    if (i < coarserLevelsStart) cp.highContrastFactor = 1.0f;
    else if (t->linear_high_contrast) {                                 /* #ifdef LINEAR_HIGH_CONTRAST_LEVELS_REDUCTION, :263-268 */
        cp.highContrastFactor = coarserLevelsCount > 1
            ? 1.0f - (float)(i - coarserLevelsStart) * (1.0f - highContrastMaxReduction) / (float)(levels - coarserLevelsStart - 1)
            : 1.0f;                                                     /* 0 / 0 at L = 4: taken as "no reduction" (the power form's exponent 0) */
    } else {
        float e = coarserLevelsCount > 1 ? (float)(i - coarserLevelsStart) / (float)(coarserLevelsCount - 1) : 0.0f;
        cp.highContrastFactor = powf(highContrastMaxReduction, e);     /* :270-274 */
    }
    if (i >= coarserLevelsStart) cp.lowContrastFactor = 1.0f;
    else if (t->linear_low_contrast)                                    /* #ifdef LINEAR_LOW_CONTRAST_LEVELS_REDUCTION, :282-287 */
        cp.lowContrastFactor = lowContrastMaxEnhancment - (float)i * ((lowContrastMaxEnhancment - 1.0f) / (float)coarserLevelsStart);
    else
        cp.lowContrastFactor = powf(lowContrastMaxEnhancment, 1.0f - ((float)i / (float)coarserLevelsStart)); /* :288-292 */
    return cp;
}
musica_contrast_params musica_oracle_host_contrast_params(uint32_t i, uint32_t levels) {
    musica_tunables t;
    musica_oracle_tunables_default(&t);
    return musica_oracle_host_contrast_params_ex(i, levels, &t);
}

/* src/vk_processing.cpp:321-325 (tunables vk_processing.h:39-42) */
musica_nr_params musica_oracle_host_nr_params_ex(uint32_t i, const musica_tunables* t) {
    const float nrHighCnr = t->nr_high_cnr, nrMaxHighFactor = t->nr_max_high_factor, nrLowCnr = t->nr_low_cnr, nrMinLowFactor = t->nr_min_low_factor;
    musica_nr_params p;
    p.highCnr = nrHighCnr;
    p.highFactor = nrMaxHighFactor - (nrMaxHighFactor - 1.0f) * ((float)i / (float)MUSICA_CNR_LEVEL);
    p.lowCnr = nrLowCnr;
    p.lowFactor = nrMinLowFactor + (1.0f - nrMinLowFactor) * ((float)i / (float)MUSICA_CNR_LEVEL);
    return p;
}
musica_nr_params musica_oracle_host_nr_params(uint32_t i) {
    musica_tunables t;
    musica_oracle_tunables_default(&t);
    return musica_oracle_host_nr_params_ex(i, &t);
}

/* ---- whole pipeline --------------------------------------------------- */

struct musica_oracle {
    uint32_t N, L;
    int order;
    uint32_t flags;
    uint32_t S[MUSICA_MAX_LEVELS + 1];
    /* norm stage */
    float* sqrt_img;
    float* normalized;
    float minv, maxv;
    /* reduce stage (index = level) */
    float* smooth[MUSICA_MAX_LEVELS];
    float* down[MUSICA_MAX_LEVELS];
    float* upsampled[MUSICA_MAX_LEVELS];
    float* lowpass[MUSICA_MAX_LEVELS];
    float* band[MUSICA_MAX_LEVELS];
    /* analysis */
    float* sdev[MUSICA_MAX_LEVELS];
    uint32_t noise_hist[MUSICA_MAX_LEVELS][MUSICA_NOISE_BINS];
    musica_hist_max_point noise_max[MUSICA_MAX_LEVELS];
    musica_contrast_params cparams[MUSICA_MAX_LEVELS];
    struct { musica_contrast_curve c; musica_point guard; } curve[MUSICA_MAX_LEVELS];
    float* cnr;
    musica_nr_params nr[3];
    /* apply + expand (index = level) */
    float* contrast_band[MUSICA_MAX_LEVELS];
    float* nr_band[3];
    float* exp_up[MUSICA_MAX_LEVELS];
    float* exp_low[MUSICA_MAX_LEVELS];
    float* expand[MUSICA_MAX_LEVELS];
    /* gradation */
    float* relevant;
    uint32_t grad_hist[MUSICA_GRAD_BINS];
    musica_hist_max_point grad_max;
    struct { musica_grad_curve c; musica_point guard; } gcurve;
    float* graded;
    /* clahe */
    uint32_t* clahe_hist;
    musica_point* clahe_pts;
    float* clahe_graded;
};

static float* zalloc(uint32_t side) { return (float*)calloc((size_t)side * side, sizeof(float)); }

musica_oracle* musica_oracle_create(uint32_t N, uint32_t levels, int order, uint32_t flags) { return musica_oracle_create_ex(N, levels, order, flags, NULL); }
musica_oracle* musica_oracle_create_ex(uint32_t N, uint32_t levels, int order, uint32_t flags, const musica_tunables* tunables) {
    musica_tunables tun;
    musica_oracle_tunables_default(&tun);
    if (tunables) tun = *tunables;
    if (N < 16) return NULL;
    uint32_t Lref = 0;
    while ((1u << Lref) < N) Lref++;                                   /* ceil(log2 N), src/vk_processing.cpp:1989 */
    uint32_t L = levels ? levels : Lref;
    if (L < MUSICA_MIN_LEVELS || L > Lref || L > MUSICA_MAX_LEVELS) return NULL;
    musica_oracle* o = (musica_oracle*)calloc(1, sizeof(*o));
    o->N = N; o->L = L; o->order = order; o->flags = flags;
    o->S[0] = N;
    for (uint32_t i = 0; i < L; i++) o->S[i + 1] = ceil_div_u(o->S[i], 2); /* :116,:150 */
    o->sqrt_img = zalloc(N);
    o->normalized = zalloc(N);
    for (uint32_t i = 0; i < L; i++) {
        o->smooth[i] = zalloc(o->S[i]);
        o->down[i] = zalloc(o->S[i + 1]);
        o->upsampled[i] = zalloc(o->S[i]);
        o->lowpass[i] = zalloc(o->S[i]);
        o->band[i] = zalloc(o->S[i]);
        o->sdev[i] = zalloc(o->S[i]);
        o->contrast_band[i] = zalloc(o->S[i]);
        o->exp_up[i] = zalloc(o->S[i]);
        o->exp_low[i] = zalloc(o->S[i]);
        o->expand[i] = zalloc(o->S[i]);
        o->cparams[i] = musica_oracle_host_contrast_params_ex(i, L, &tun);
    }
    for (uint32_t i = 0; i < 3; i++) {
        o->nr_band[i] = zalloc(o->S[i]);
        o->nr[i] = musica_oracle_host_nr_params_ex(i, &tun);
    }
    o->cnr = zalloc(o->S[MUSICA_CNR_LEVEL]);
    o->relevant = zalloc(N);
    o->graded = zalloc(N);
    if (flags & MUSICA_ORACLE_FLAG_CLAHE) {
        o->clahe_hist = (uint32_t*)calloc(MUSICA_CLAHE_TILES * MUSICA_CLAHE_TILES * MUSICA_CLAHE_BINS, sizeof(uint32_t));
        o->clahe_pts = (musica_point*)calloc(MUSICA_CLAHE_TILES * MUSICA_CLAHE_TILES * MUSICA_CLAHE_BINS, sizeof(musica_point));
        o->clahe_graded = zalloc(N);
    }
    return o;
}

void musica_oracle_destroy(musica_oracle* o) {
    if (!o) return;
    free(o->sqrt_img); free(o->normalized);
    for (uint32_t i = 0; i < o->L; i++) {
        free(o->smooth[i]); free(o->down[i]); free(o->upsampled[i]); free(o->lowpass[i]); free(o->band[i]);
        free(o->sdev[i]); free(o->contrast_band[i]); free(o->exp_up[i]); free(o->exp_low[i]); free(o->expand[i]);
    }
    for (uint32_t i = 0; i < 3; i++) free(o->nr_band[i]);
    free(o->cnr); free(o->relevant); free(o->graded);
    free(o->clahe_hist); free(o->clahe_pts); free(o->clahe_graded);
    free(o);
}

uint32_t musica_oracle_levels(const musica_oracle* o) { return o->L; }
uint32_t musica_oracle_level_size(const musica_oracle* o, uint32_t level) { return level <= o->L ? o->S[level] : 0; }

/* stage "norm": src/vk_processing.cpp:2182-2222 */
static void stage_norm(musica_oracle* o, const uint16_t* px) {
    uint32_t N = o->N;
    musica_oracle_k_sqrt(px, N, o->sqrt_img);                          /* :2182 */
    /* max chain :2190-2199, sizes from :52-83 */
    {
        uint32_t s = N;
        float* cur = o->sqrt_img;
        float* owned = NULL;
        while (s > 1) {
            uint32_t ns = ceil_div_u(s, REDUCE_AREA);
            float* nxt = zalloc(ns);
            musica_oracle_k_max_reduce(cur, s, nxt);
            free(owned);
            owned = nxt; cur = nxt; s = ns;
        }
        o->maxv = cur[0];
        free(owned);
    }
    /* min chain :2202-2211 */
    {
        uint32_t s = N;
        float* cur = o->sqrt_img;
        float* owned = NULL;
        while (s > 1) {
            uint32_t ns = ceil_div_u(s, REDUCE_AREA);
            float* nxt = zalloc(ns);
            musica_oracle_k_min_reduce(cur, s, nxt);
            free(owned);
            owned = nxt; cur = nxt; s = ns;
        }
        o->minv = cur[0];
        free(owned);
    }
    musica_oracle_k_normalize(o->sqrt_img, N, o->minv, o->maxv, o->normalized); /* :2213 */
}

/* stage "red": src/vk_processing.cpp:2233-2273; wiring :758-761, :790-793, :821-824, :852-855, :891-896 */
static void stage_reduce(musica_oracle* o) {
    for (uint32_t i = 0; i < o->L; i++) {
        const float* in = i == 0 ? o->normalized : o->down[i - 1];
        uint32_t s = o->S[i];
        musica_oracle_k_smooth(in, s, o->smooth[i], o->order);
        musica_oracle_k_downsample(o->smooth[i], s, o->down[i]);
        musica_oracle_k_upsample(o->down[i], o->S[i + 1], o->upsampled[i], s); /* odd texels stay 0 (Q2) */
        musica_oracle_k_smooth_upsampled(o->upsampled[i], s, o->lowpass[i], o->order);
        musica_oracle_k_difference(in, o->lowpass[i], s, o->band[i]);
    }
}

/* stage "anly": src/vk_processing.cpp:2284-2357 */
static void stage_analysis(musica_oracle* o) {
    for (uint32_t i = 0; i < o->L; i++) {
        if (i < MUSICA_COARSER_LEVELS_START || i <= MUSICA_CNR_LEVEL) {  /* :2285 */
            musica_oracle_k_sdev(o->band[i], o->S[i], o->sdev[i], o->order);
            memset(o->noise_hist[i], 0, sizeof(o->noise_hist[i]));     /* clear, :2153-2158 */
            musica_oracle_k_noise_hist(o->sdev[i], o->S[i], o->N / (WG * HIST_AREA), o->noise_hist[i]); /* :2293-2295 */
            musica_oracle_k_histogram_max(o->noise_hist[i], MUSICA_NOISE_BINS, &o->noise_max[i]);
        }
        musica_oracle_k_contrast_curve_generate(o->noise_max[i], o->cparams[i], &o->curve[i].c); /* :2310 */
    }
    musica_oracle_k_cnr(o->sdev[MUSICA_CNR_LEVEL], o->S[MUSICA_CNR_LEVEL], o->noise_max[MUSICA_CNR_LEVEL], o->cnr); /* :2353 */
}

/* stages "aply" + "exp": src/vk_processing.cpp:2361-2431; wiring :1099-1111, :1508-1520, :930-934, :1002-1016 */
static void stage_expand(musica_oracle* o) {
    uint32_t L = o->L;
    for (uint32_t slot = 0; slot < L; slot++) {
        uint32_t lvl = L - 1 - slot;                                   /* :1102-1110 */
        const float* sd = lvl <= MUSICA_CNR_LEVEL ? o->sdev[lvl] : NULL; /* levels >= 4: never-written sdev (Q2) */
        musica_oracle_k_contrast_curve_apply(o->band[lvl], sd, o->S[lvl], &o->curve[lvl].c, o->contrast_band[lvl]);
    }
    for (uint32_t i = 0; i < MUSICA_CNR_LEVEL; i++) {                  /* :2373-2385 */
        uint32_t lvl = MUSICA_CNR_LEVEL - 1 - i;                       /* input expandBandpass[L-3+i] == level 2-i (:1511-1514) */
        /* params buffer index cnrLevel - i - 1 == lvl (:1518-1520) */
        musica_oracle_k_noise_reduction(o->contrast_band[lvl], o->S[lvl], o->cnr, o->S[MUSICA_CNR_LEVEL], o->nr[lvl], o->nr_band[lvl]);
    }
    for (uint32_t slot = 0; slot < L; slot++) {                        /* :2396-2431 */
        uint32_t lvl = L - 1 - slot;
        const float* src = slot == 0 ? o->down[L - 1] : o->expand[lvl + 1]; /* :930-934 */
        memset(o->exp_up[lvl], 0, (size_t)o->S[lvl] * o->S[lvl] * sizeof(float));
        musica_oracle_k_upsample(src, o->S[lvl + 1], o->exp_up[lvl], o->S[lvl]);
        musica_oracle_k_smooth_upsampled(o->exp_up[lvl], o->S[lvl], o->exp_low[lvl], o->order);
        const float* b = lvl < MUSICA_CNR_LEVEL - 1 ? o->nr_band[lvl] : o->contrast_band[lvl]; /* :1009-1016 */
        musica_oracle_k_addition(o->exp_low[lvl], b, o->S[lvl], o->expand[lvl]);
    }
}

/* stage "grad": src/vk_processing.cpp:2456-2518 */
static void stage_gradation(musica_oracle* o) {
    uint32_t N = o->N;
    musica_oracle_k_relevant(o->normalized, N, o->cnr, o->S[MUSICA_CNR_LEVEL], o->relevant); /* :2456, wiring :1585-1589 */
    if (o->flags & MUSICA_ORACLE_FLAG_CLAHE) {                         /* :2471-2489 */
        memset(o->clahe_hist, 0, MUSICA_CLAHE_TILES * MUSICA_CLAHE_TILES * MUSICA_CLAHE_BINS * sizeof(uint32_t));
        musica_oracle_k_clahe_histogram(o->expand[0], o->relevant, N, o->clahe_hist);
        musica_oracle_k_clahe_grad_curve(o->clahe_hist, o->clahe_pts);
        musica_oracle_k_clahe_grad_curve_apply(o->expand[0], N, o->clahe_pts, o->clahe_graded);
    }
    memset(o->grad_hist, 0, sizeof(o->grad_hist));                     /* clear, :2159-2162 */
    musica_oracle_k_gradation_histogram(o->expand[0], o->relevant, N, ceil_div_u(N, WG * HIST_AREA), o->grad_hist); /* :2492-2494, wiring :1622-1627 */
    musica_oracle_k_histogram_max(o->grad_hist, MUSICA_GRAD_BINS, &o->grad_max); /* :2499 */
    musica_oracle_k_gradation_curve_generate(o->grad_hist, &o->gcurve.c); /* :2503 */
    musica_oracle_k_apply_gradation_curve(o->expand[0], N, &o->gcurve.c, o->graded); /* :2513, wiring :1767 */
}

int musica_oracle_execute(musica_oracle* o, const uint16_t* pixels) {
    if (!o || !pixels) return 0;
    stage_norm(o, pixels);
    stage_reduce(o);
    stage_analysis(o);
    stage_expand(o);
    stage_gradation(o);
    return 1;
}

int musica_oracle_run_stage(musica_oracle* o, int stage) {
    switch (stage) {
        case MUSICA_STAGE_REDUCE: stage_reduce(o); return 1;
        case MUSICA_STAGE_ANALYSIS: stage_analysis(o); return 1;
        case MUSICA_STAGE_EXPAND: stage_expand(o); return 1;
        case MUSICA_STAGE_GRADATION: stage_gradation(o); return 1;
        default: return 0; /* NORM needs the pixels: use execute */
    }
}

static float* image_ptr(const musica_oracle* o, int kind, uint32_t level, uint32_t* side) {
    uint32_t L = o->L;
    switch (kind) {
        case MUSICA_IMG_NORMALIZED: *side = o->N; return level == 0 ? o->normalized : NULL;
        case MUSICA_IMG_SQRT: *side = o->N; return level == 0 ? o->sqrt_img : NULL;
        case MUSICA_IMG_GRADED: *side = o->N; return level == 0 ? o->graded : NULL;
        case MUSICA_IMG_CLAHE_GRADED: *side = o->N; return level == 0 ? o->clahe_graded : NULL;
        case MUSICA_IMG_RELEVANT: *side = o->N; return level == 0 ? o->relevant : NULL;
        case MUSICA_IMG_CNR: *side = o->S[MUSICA_CNR_LEVEL]; return level == MUSICA_CNR_LEVEL ? o->cnr : NULL;
        default: break;
    }
    if (level >= L) return NULL;
    *side = o->S[level];
    switch (kind) {
        case MUSICA_IMG_DOWNSAMPLED: *side = o->S[level + 1]; return o->down[level];
        case MUSICA_IMG_BANDPASS: return o->band[level];
        case MUSICA_IMG_SDEV: return o->sdev[level];
        case MUSICA_IMG_EXPAND: return o->expand[level];
        case MUSICA_IMG_LOWPASS: return o->lowpass[level];
        case MUSICA_IMG_EXP_BANDPASS: return level < MUSICA_CNR_LEVEL - 1 ? o->nr_band[level] : o->contrast_band[level];
        case MUSICA_ORACLE_IMG_SMOOTH: return o->smooth[level];
        case MUSICA_ORACLE_IMG_UPSAMPLED: return o->upsampled[level];
        case MUSICA_ORACLE_IMG_EXP_UPSAMPLED: return o->exp_up[level];
        case MUSICA_ORACLE_IMG_EXP_LOWPASS: return o->exp_low[level];
        case MUSICA_ORACLE_IMG_CONTRAST_BAND: case MUSICA_IMG_CONTRAST_BAND: return o->contrast_band[level];
        case MUSICA_ORACLE_IMG_NR_BAND: return level < 3 ? o->nr_band[level] : NULL;
        default: return NULL;
    }
}

const float* musica_oracle_image(const musica_oracle* o, int kind, uint32_t level, uint32_t* side) {
    uint32_t s = 0;
    float* p = image_ptr(o, kind, level, &s);
    if (side) *side = p ? s : 0;
    return p;
}

int musica_oracle_set_image(musica_oracle* o, int kind, uint32_t level, const float* src) {
    uint32_t s = 0;
    float* p = image_ptr(o, kind, level, &s);
    if (!p) return 0;
    memcpy(p, src, (size_t)s * s * sizeof(float));
    return 1;
}

const uint32_t* musica_oracle_noise_hist(const musica_oracle* o, uint32_t level) { return level < o->L ? o->noise_hist[level] : NULL; }
const uint32_t* musica_oracle_grad_hist(const musica_oracle* o) { return o->grad_hist; }
musica_hist_max_point musica_oracle_noise_hist_max(const musica_oracle* o, uint32_t level) { return o->noise_max[level < o->L ? level : 0]; }
musica_hist_max_point musica_oracle_grad_hist_max(const musica_oracle* o) { return o->grad_max; }
const musica_contrast_curve* musica_oracle_contrast_curve(const musica_oracle* o, uint32_t level) { return level < o->L ? &o->curve[level].c : NULL; }
const musica_grad_curve* musica_oracle_grad_curve(const musica_oracle* o) { return &o->gcurve.c; }
musica_contrast_params musica_oracle_contrast_params(const musica_oracle* o, uint32_t level) { return o->cparams[level < o->L ? level : 0]; }
musica_nr_params musica_oracle_nr_params(const musica_oracle* o, uint32_t level) { return o->nr[level < 3 ? level : 0]; }
void musica_oracle_minmax(const musica_oracle* o, float* mn, float* mx) { *mn = o->minv; *mx = o->maxv; }
const uint32_t* musica_oracle_clahe_hist(const musica_oracle* o) { return o->clahe_hist; }
const musica_point* musica_oracle_clahe_curves(const musica_oracle* o) { return o->clahe_pts; }

void musica_oracle_stats(const musica_oracle* o, musica_stats* d) {
    memset(d, 0, sizeof(*d));
    d->min_sqrt = o->minv;
    d->max_sqrt = o->maxv;
    for (int i = 0; i < 4; i++) {
        d->noise_max_bin[i] = o->noise_max[i].maxBin;
        d->noise_max_value[i] = o->noise_max[i].maxValue;
    }
    d->grad_max_bin = o->grad_max.maxBin;
    d->grad_max_value = o->grad_max.maxValue;
    uint32_t s = o->S[MUSICA_CNR_LEVEL];
    double acc = 0.0;
    for (size_t i = 0; i < (size_t)s * s; i++) acc += (double)o->cnr[i];
    d->mean_cnr = (float)(acc / ((double)s * s) * 256.0);              /* test/mean_cnr/script.py:13-24 on cnr.bmp */
    d->t0 = o->gcurve.c.t0;
    d->ta = o->gcurve.c.ta;
    d->t1 = o->gcurve.c.t1;
}

/* saveOutImage src/vk_processing.cpp:2603-2645: crop 10, (uint8_t)(255.0f * (v - 0) / (1 - 0)). */
int musica_oracle_out_pixels(const musica_oracle* o, uint8_t* dst) {
    uint32_t N = o->N, margin = MUSICA_OUT_MARGIN;
    if (N <= 2 * margin) return 0;
    uint32_t nw = N - 2 * margin;
    for (uint32_t y = 0; y < nw; y++)
        for (uint32_t x = 0; x < nw; x++) {
            float v = o->graded[(size_t)(y + margin) * N + x + margin];
            float q = 255.0f * (v - 0.0f) / (1.0f - 0.0f);             /* :2632 */
            dst[(size_t)y * nw + x] = (uint8_t)(int32_t)q;             /* graded is always in [0, 1] */
        }
    return 1;
}

static void put_u16(FILE* f, uint32_t v) { fputc(v & 0xFF, f); fputc((v >> 8) & 0xFF, f); }
static void put_u32(FILE* f, uint32_t v) { put_u16(f, v & 0xFFFF); put_u16(f, v >> 16); }

/* stbi_write_bmp(path, w, h, comp = 1, data): dependencies/stb/stb_image_write.h:492-500
 * (header "11 4 22 4" "4 44 22 444444"), pixels :451-476 (bottom-up, gray replicated, row padding). */
int musica_oracle_write_bmp_gray(const char* path, uint32_t w, uint32_t h, const uint8_t* data) {
    FILE* f = fopen(path, "wb");
    if (!f) return 0;
    uint32_t pad = (uint32_t)(-(int32_t)(w * 3)) & 3u;
    fputc('B', f); fputc('M', f);
    put_u32(f, 14 + 40 + (w * 3 + pad) * h);
    put_u16(f, 0); put_u16(f, 0);
    put_u32(f, 14 + 40);
    put_u32(f, 40); put_u32(f, w); put_u32(f, h);
    put_u16(f, 1); put_u16(f, 24);
    for (int i = 0; i < 6; i++) put_u32(f, 0);
    for (int32_t j = (int32_t)h - 1; j >= 0; j--) {
        for (uint32_t i = 0; i < w; i++) {
            uint8_t g = data[(size_t)j * w + i];
            fputc(g, f); fputc(g, f); fputc(g, f);
        }
        for (uint32_t k = 0; k < pad; k++) fputc(0, f);
    }
    fclose(f);
    return 1;
}

/* stbi_write_bmp(path, w, h, comp = 4, data): dependencies/stb/stb_image_write.h:501-509 — a V4 header (108 bytes, BI_BITFIELDS,
 * masks 0xff0000 / 0xff00 / 0xff / 0xff000000), pixels :451-476 bottom-up as B, G, R, A, no row padding. What debugProcess
 * writes for its two RGBA plots (src/vk_processing.cpp:2758-2806). */
int musica_oracle_write_bmp_rgba(const char* path, uint32_t w, uint32_t h, const uint8_t* data) {
    FILE* f = fopen(path, "wb");
    if (!f) return 0;
    fputc('B', f); fputc('M', f);
    put_u32(f, 14 + 108 + w * h * 4);
    put_u16(f, 0); put_u16(f, 0);
    put_u32(f, 14 + 108);
    put_u32(f, 108); put_u32(f, w); put_u32(f, h);
    put_u16(f, 1); put_u16(f, 32);
    put_u32(f, 3);
    for (int i = 0; i < 5; i++) put_u32(f, 0);
    put_u32(f, 0xff0000u); put_u32(f, 0xff00u); put_u32(f, 0xffu); put_u32(f, 0xff000000u);
    for (int i = 0; i < 13; i++) put_u32(f, 0);          /* cstype, 9 endpoint words, 3 gamma words */
    for (int32_t j = (int32_t)h - 1; j >= 0; j--)
        for (uint32_t i = 0; i < w; i++) {
            const uint8_t* q = data + ((size_t)j * w + i) * 4;
            fputc(q[2], f); fputc(q[1], f); fputc(q[0], f); fputc(q[3], f);
        }
    fclose(f);
    return 1;
}

/* ---- the two RGBA plots the reference renders on every execute (#define RENDER_HISTS, include/vk_processing.h:22) ---------
 * Both shaders run as ONE workgroup of 512 invocations (execute(1, 1), src/vk_processing.cpp:2347, :2508) on a
 * histRenderWidth x histRenderHeight = 512 x 128 rgba8 image (include/vk_processing.h:31-32); invocation x draws column x.
 * imageStore outside the image is dropped (Q1); a stored vec4 component 0 / 1 is the byte 0 / 255. */
static void plot_px(uint8_t* img, uint32_t x, uint32_t y, uint8_t r, uint8_t g, uint8_t b) {
    if (x >= MUSICA_HIST_RENDER_W || y >= MUSICA_HIST_RENDER_H) return;
    uint8_t* q = img + ((size_t)y * MUSICA_HIST_RENDER_W + x) * 4;
    q[0] = r; q[1] = g; q[2] = b; q[3] = 255;
}

/* noise_hist_render.comp:17-76 bound to the histogram and the argmax of cnrLevel (src/vk_processing.cpp:1260-1266):
 * positionConversionFactor is 1.0, so only bins 0..511 of the 2048 are drawn, one per column. */
void musica_oracle_render_noise_hist(const musica_oracle* o, uint8_t* rgba) {
    const uint32_t H = MUSICA_HIST_RENDER_H;
    const uint32_t maxValue = o->noise_max[MUSICA_CNR_LEVEL].maxValue, maxBin = o->noise_max[MUSICA_CNR_LEVEL].maxBin;
    memset(rgba, 0, (size_t)MUSICA_HIST_RENDER_W * H * 4);
    for (uint32_t pos = 0; pos < MUSICA_HIST_RENDER_W; pos++) {
        const float factor = 1.0f;                                                    /* :19 */
        const uint32_t bin = f2u((float)pos * factor);                                /* :23 */
        const uint32_t value = bin < MUSICA_NOISE_BINS ? o->noise_hist[MUSICA_CNR_LEVEL][bin] : 0u;   /* :41 */
        uint32_t barHeight = f2u((float)value * ((float)H / (float)(maxValue + 1u))); /* :49 */
        if (barHeight > H) barHeight = H - 1u;                                        /* :50 */
        const uint32_t startY = H - barHeight - 1u;                                   /* :58 (uint arithmetic) */
        for (uint32_t y = 0; y < H; y++) plot_px(rgba, pos, y, 0, 0, 0);              /* :62-64 */
        plot_px(rgba, pos, H - 1u, 255, 0, 0);                                        /* :66 */
        for (uint32_t y = startY; y < startY + barHeight; y++) {                      /* :68-76, barWidth = 1 */
            if (bin <= maxBin && (float)bin + factor > (float)maxBin) plot_px(rgba, pos, y, 0, 255, 0);
            else plot_px(rgba, pos, y, 255, 255, 255);
        }
    }
}

/* getY() of gradation_curve_debug_render.comp:31-46: first i < pointsCount with points[i].x == x, or points[i].x <= x <=
 * points[i + 1].x (points[pointsCount] is the never-written entry behind the curve: 0, 0), m * (x - x_i) + y_i. */
static float plot_get_y(const musica_point* pts, uint32_t count, float x) {
    for (uint32_t i = 0; i < count; i++) {
        if (pts[i].x == x) return pts[i].y;
        if (pts[i].x <= x && pts[i + 1].x >= x) {
            const float m = (pts[i + 1].y - pts[i].y) / (pts[i + 1].x - pts[i].x);
            return m * (x - pts[i].x) + pts[i].y;
        }
    }
    return 0.0f;
}

/* gradation_curve_debug_render.comp:48-123 bound to the gradation histogram, its argmax and the tone curve
 * (src/vk_processing.cpp:1668-1675): every SECOND histogram bin (factor 1024 / 512 = 2), the t0 / ta / t1 columns, the curve. */
void musica_oracle_render_grad_hist(const musica_oracle* o, uint8_t* rgba) {
    const uint32_t H = MUSICA_HIST_RENDER_H, W = MUSICA_HIST_RENDER_W;
    const uint32_t maxValue = o->grad_max.maxValue, maxBin = o->grad_max.maxBin;
    const musica_grad_curve* gc = &o->gcurve.c;
    memset(rgba, 0, (size_t)W * H * 4);
    for (uint32_t pos = 0; pos < W; pos++) {
        const float factor = (float)MUSICA_GRAD_BINS / 512.0f;                        /* :52 */
        const uint32_t bin = f2u((float)pos * factor);                                /* :54 */
        const uint32_t value = bin < MUSICA_GRAD_BINS ? o->grad_hist[bin] : 0u;       /* :56 */
        uint32_t barHeight = f2u((float)value * ((float)H / (float)(maxValue + 1u))); /* :64 */
        if (barHeight > H) barHeight = H - 1u;                                        /* :65 */
        const uint32_t startY = H - barHeight - 1u;                                   /* :73 */
        plot_px(rgba, pos, H - 1u, 255, 0, 0);                                        /* :77 (overwritten by the loop below) */
        for (uint32_t y = 0; y < H; y++) {                                            /* :79-91 */
            if (y >= startY && y < startY + barHeight) {
                if (bin <= maxBin && (float)bin + factor > (float)maxBin) plot_px(rgba, pos, y, 255, 0, 255);
                else plot_px(rgba, pos, y, 255, 255, 255);
            } else {
                plot_px(rgba, pos, y, 0, 0, 0);
            }
        }
        const float step = 1.0f / 512.0f;
        const float cp = (float)pos * step;                                           /* :94 */
        const uint32_t posX = f2u(cp * 512.0f * ((float)W / 512.0f));                 /* :99 */
        const uint32_t posY = (H - 1u) - f2u(plot_get_y(gc->points, gc->pointsCount, cp) * (float)(H - 1u));   /* :100 */
        const float next = (float)(pos + 1u) * step;
        if (cp <= gc->t0 && gc->t0 < next)                                            /* :103-107: i runs to imageSize.x, rows >= H are dropped */
            for (uint32_t i = 0; i < W; i++) plot_px(rgba, posX, i, 255, 0, 0);
        if (cp <= gc->ta && gc->ta < next)                                            /* :110-114 */
            for (uint32_t i = 0; i < W; i++) plot_px(rgba, posX, i, 0, 255, 0);
        if (cp <= gc->t1 && gc->t1 < next)                                            /* :117-121 */
            for (uint32_t i = 0; i < W; i++) plot_px(rgba, posX, i, 255, 0, 0);
        plot_px(rgba, posX, posY, 0, 0, 255);                                         /* :123 */
    }
}

int musica_oracle_save_out_image(const musica_oracle* o, const char* path) {
    uint32_t nw = o->N - 2 * MUSICA_OUT_MARGIN;
    uint8_t* buf = (uint8_t*)malloc((size_t)nw * nw);
    int ok = musica_oracle_out_pixels(o, buf) && musica_oracle_write_bmp_gray(path, nw, nw, buf);
    free(buf);
    return ok;
}

/* VulkanState::downloadAndSaveImage, src/vk_state.cpp:809-855: every texel as (uint8_t)(255.0f * (v - min) / (max - min))
 * (:834), written by stbi_write_bmp with one component (:848-854). The C cast is undefined outside [0, 256); restated as the
 * x86 lowering the reference's MSVC build gets (cvttss2si to int32, low byte kept; NaN and out-of-int32 values give
 * 0x80000000 -> 0) — the same statement as csrc/musica_ctx.hip dump_image. */
static int dump_image(const float* img, uint32_t side, const char* dir, const char* name, float maxValue, float minValue) {
    if (!img) return 0;
    size_t n = (size_t)side * side;
    uint8_t* out = (uint8_t*)malloc(n ? n : 1);
    for (size_t i = 0; i < n; i++) {
        float q = 255.0f * (img[i] - minValue) / (maxValue - minValue);
        out[i] = (q == q && q > -2147483648.0f && q < 2147483648.0f) ? (uint8_t)(int32_t)q : 0;
    }
    char path[4096];
    snprintf(path, sizeof(path), "%s/%s", dir && *dir ? dir : ".", name);
    int ok = musica_oracle_write_bmp_gray(path, side, side, out);
    free(out);
    return ok;
}

/* VulkanProcessing::debugProcess, src/vk_processing.cpp:2661-2756: the image dumps, in the reference's order and with its
 * (max, min) pairs. Slot i of the expand-side arrays is level L-1-i (src/vk_processing.cpp:930-934, 1099-1111);
 * expandBandpassImageStates is the output of contrast_curve_apply (before noise reduction). The two RGBA plots
 * (noise_hist.bmp, grad_hist.bmp, :2758-2806) are the render shaders' images of the last execute. */
int musica_oracle_debug_process(const musica_oracle* o, const char* dir) {
    char name[64];
    int ok = dump_image(o->normalized, o->N, dir, "norm.bmp", 1.0f, 0.0f);                          /* :2664-2671 */
    for (uint32_t i = 0; i < o->L && ok; i++) {                                                       /* :2673-2690 */
        snprintf(name, sizeof(name), "red_bandpass_%u.bmp", i);
        ok = ok && dump_image(o->band[i], o->S[i], dir, name, 1.0f, -1.0f);
        snprintf(name, sizeof(name), "red_lowpass_%u.bmp", i);
        ok = ok && dump_image(o->lowpass[i], o->S[i], dir, name, 1.0f, 0.0f);
    }
    ok = ok && dump_image(o->sdev[MUSICA_CNR_LEVEL], o->S[MUSICA_CNR_LEVEL], dir, "sdev.bmp", 1.0f, -1.0f);   /* :2692-2699 */
    ok = ok && dump_image(o->cnr, o->S[MUSICA_CNR_LEVEL], dir, "cnr.bmp", 1.0f, 0.0f);                        /* :2701-2708 */
    for (uint32_t i = 0; i < o->L && ok; i++) {                                                       /* :2710-2727 */
        uint32_t lvl = o->L - 1 - i;
        snprintf(name, sizeof(name), "exp_bandpass_%u.bmp", i);
        ok = ok && dump_image(o->contrast_band[lvl], o->S[lvl], dir, name, 1.0f, -1.0f);
        snprintf(name, sizeof(name), "exp_lowpass_%u.bmp", i);
        ok = ok && dump_image(o->exp_low[lvl], o->S[lvl], dir, name, 1.0f, 0.0f);
    }
    ok = ok && dump_image(o->relevant, o->N, dir, "relevant.bmp", 1.0f, 0.0f);                     /* :2729-2736 */
    ok = ok && dump_image(o->graded, o->N, dir, "graded.bmp", 1.0f, 0.0f);                         /* :2749-2756 */
    if (ok) {                                                                                       /* :2758-2806 */
        uint8_t* plot = (uint8_t*)malloc((size_t)MUSICA_HIST_RENDER_W * MUSICA_HIST_RENDER_H * 4);
        char path[4096];
        musica_oracle_render_noise_hist(o, plot);
        snprintf(path, sizeof(path), "%s/noise_hist.bmp", dir && *dir ? dir : ".");
        ok = musica_oracle_write_bmp_rgba(path, MUSICA_HIST_RENDER_W, MUSICA_HIST_RENDER_H, plot);
        musica_oracle_render_grad_hist(o, plot);
        snprintf(path, sizeof(path), "%s/grad_hist.bmp", dir && *dir ? dir : ".");
        ok = ok && musica_oracle_write_bmp_rgba(path, MUSICA_HIST_RENDER_W, MUSICA_HIST_RENDER_H, plot);
        free(plot);
    }
    return ok;
}

/* test/standalone/main.cpp:54-75 */
int musica_oracle_read_raw(const char* path, uint32_t N, uint16_t* dst) {
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    const long offset = 256;
    if (size != offset + (long)N * N * 2) { fclose(f); return 0; }     /* :57-60 */
    uint8_t* buf = (uint8_t*)malloc((size_t)size);
    if (fread(buf, 1, (size_t)size, f) != (size_t)size) { free(buf); fclose(f); return 0; }
    fclose(f);
    for (size_t i = 0; i < (size_t)N * N; i++)
        dst[i] = (uint16_t)((buf[offset + 2 * i + 1] << 8) | buf[offset + 2 * i]); /* :71-72 */
    free(buf);
    return 1;
}
