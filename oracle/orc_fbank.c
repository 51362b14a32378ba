/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the reference's online log-mel filterbank front end.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything under oracle/.  The shipped path is the HIP kernel in
 * april_asr_amd/csrc/kernels_fbank.hip.
 *
 * Follows (reference file:line):
 *   src/fbank.c:49-55     povey window           -> orc_make_window
 *   src/fbank.c:61-95     mel bank table         -> orc_make_melbank
 *   src/fbank.c:174-306   accept_waveform        -> orc_fbank_accept (+ frame math in orc_fbank_frame)
 *   src/fbank.c:308-325   flush padding          -> orc_fbank_flush
 *   src/fbank.c:327-349   pull_segments          -> orc_fbank_pull
 *   src/fft/pocketfft.c:65-228   twiddle generation (sincos_2pibyn_half: all three branches, n % 4 == 0, n even, n odd)
 *   src/fft/pocketfft.c:1111-1134 radf2, :1136-1168 radf3, :1170-1209 radf4, :1211-1260 radf5, :1266-1409 radfg (any other factor),
 *   :1730-1764 rfftp_forward, :1798-1827 factorisation, :1843-1881 twiddle layout, :2155-2182 the choice between this plan and Bluestein.
 *
 * Pinned: bit-exact against the reference's own fbank.c/pocketfft.c compiled
 * into oracle/_ref/libaprilref.so (tests/test_oracle_fbank.py) and against the
 * committed golden vectors tests/golden/fbank_*.npz generated from that build.
 *
 * Design differences from the reference (behaviour-preserving): the reference
 * carries a "previous leftover" array and three copy cases; here the stream is
 * a plain FIFO of samples and frame k is cut at stream offset k*shift.  FFT
 * lengths: whatever pocketfft runs through its radix passes (4 / 2 / 3 / 5 and
 * the generic pass for any other factor; every exported model uses round_pow2 = 1
 * = a power of two, a model with round_pow2 = 0 has the frame length itself, e.g.
 * 400, or 882 = 2 3 3 7 7 at 44.1 kHz / 20 ms).  Lengths for which pocketfft
 * picks Bluestein's algorithm instead (a large prime factor) are refused.
 *
 * Build with -ffp-contract=off: the reference is built for baseline x86-64
 * (no FMA contraction) and bit-exactness depends on it.
  *
 * Third-party notice: the FFT pass structure and twiddle polynomial coefficients
 * restated below follow pocketfft, Copyright (C) 2010-2019 Max-Planck-Society,
 * BSD 3-Clause License (full text in THIRD_PARTY_NOTICES.md).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

/* ------------------------------------------------------------------ */
/* twiddles: pocketfft.c:65-228 restated for n % 4 == 0               */
/* ------------------------------------------------------------------ */

/* cos(pi a)-1 and sin(pi a) for |a| <= 0.25 (pocketfft.c:65-90) */
static void cm1_sin_pi(double a, double *cm1, double *sn)
{
    const double s = a * a;
    double r = -1.0369917389758117e-4;
    r = fma(r, s, 1.9294935641298806e-3);
    r = fma(r, s, -2.5806887942825395e-2);
    r = fma(r, s, 2.3533063028328211e-1);
    r = fma(r, s, -1.3352627688538006e+0);
    r = fma(r, s, 4.0587121264167623e+0);
    r = fma(r, s, -4.9348022005446790e+0);
    *cm1 = r * s;
    r = 4.6151442520157035e-4;
    r = fma(r, s, -7.3700183130883555e-3);
    r = fma(r, s, 8.2145868949323936e-2);
    r = fma(r, s, -5.9926452893214921e-1);
    r = fma(r, s, 2.5501640398732688e+0);
    r = fma(r, s, -5.1677127800499516e+0);
    const double s3 = s * a;
    r = r * s3;
    *sn = fma(a, 3.1415926535897931e+0, r);
}

/* (cos,sin)(2 pi i / den) for the first octant, interleaved (pocketfft.c:92-123) */
static void first_octant(size_t den, double *res)
{
    const size_t n = (den + 4) >> 3;
    if (n == 0) return;
    res[0] = 1.0; res[1] = 0.0;
    if (n == 1) return;
    const size_t blk = (size_t)sqrt((double)n);
    for (size_t i = 1; i < blk; ++i)
        cm1_sin_pi((2.0 * (double)i) / (double)den, &res[2 * i], &res[2 * i + 1]);
    for (size_t start = blk; start < n; start += blk) {
        double c0, s0;
        cm1_sin_pi((2.0 * (double)start) / (double)den, &c0, &s0);
        res[2 * start] = c0 + 1.0;
        res[2 * start + 1] = s0;
        size_t end = blk;
        if (start + end > n) end = n - start;
        for (size_t i = 1; i < end; ++i) {
            const double cx = res[2 * i], sx = res[2 * i + 1];
            res[2 * (start + i)]     = ((c0 * cx - s0 * sx + c0) + cx) + 1.0;
            res[2 * (start + i) + 1] = (c0 * sx + s0 * cx) + s0 + sx;
        }
    }
    for (size_t i = 1; i < blk; ++i) res[2 * i] += 1.0;
}

/* table of (cos,sin)(2 pi i / n), i in [0, n/2); n % 4 == 0  (pocketfft.c:172-183,205-214) */
static void half_circle_table(size_t n, double *res /* 2n doubles of room */)
{
    first_octant(n, res);
    const size_t quart = n >> 2;
    if ((n & 7) == 0) res[quart] = res[quart + 1] = 0.707106781186547524400844362104849;
    for (size_t i = 2, j = 2 * quart - 2; i < quart; i += 2, j -= 2) {
        res[j] = res[i + 1];
        res[j + 1] = res[i];
    }
    const size_t half = n >> 1;
    for (size_t i = 0; i < half; i += 2) {
        res[i + half] = -res[i + 1];
        res[i + half + 1] = res[i];
    }
}

/* table of (cos,sin)(2 pi i / n), i in [0, n/2], for any n: pocketfft builds it from the first octant of n, 2n or 4n by exact
 * symmetries, depending on n mod 4 (pocketfft.c:121-226).  res: 2n doubles of room. */
static int half_circle_table_any(size_t n, double *res)
{
    if ((n & 3) == 0) { half_circle_table(n, res); return 0; }
    const size_t den = (n & 1) ? 4 * n : 2 * n;
    double *oct = (double *)malloc((2 * ((den + 4) >> 3) + 2) * sizeof(double));
    if (!oct) return -1;
    first_octant(den, oct);
    if ((n & 1) == 0) {
        /* n = 4q + 2: the quarter circle [0, q] from the octant of 2n (even multiples directly, the upper part mirrored at pi/4 from the
         * odd ones), then the second quarter mirrored at pi/2 */
        const size_t cnt = (n + 2) >> 2;
        for (size_t e = 0; e < cnt; ++e) {
            if (e < (cnt + 1) / 2) { res[2 * e] = oct[4 * e]; res[2 * e + 1] = oct[4 * e + 1]; }
            else { const size_t m = 2 * (cnt - 1 - e) + 1; res[2 * e] = oct[2 * m + 1]; res[2 * e + 1] = oct[2 * m]; }
        }
        const size_t half = n >> 1;
        for (size_t e = 1; 2 * e < half; ++e) { res[2 * (half - e)] = -res[2 * e]; res[2 * (half - e) + 1] = res[2 * e + 1]; }
    } else {
        /* n odd: entry i is the point 4 i of the circle of 4n, folded into its first octant */
        const size_t cnt = (n + 1) >> 1;
        for (size_t i = 0; i < cnt; ++i) {
            const size_t i4 = 4 * i;
            if (2 * i4 <= n) { res[2 * i] = oct[2 * i4]; res[2 * i + 1] = oct[2 * i4 + 1]; }
            else if (i4 <= n) { const size_t m = n - i4; res[2 * i] = oct[2 * m + 1]; res[2 * i + 1] = oct[2 * m]; }
            else if (2 * i4 <= 3 * n) { const size_t m = i4 - n; res[2 * i] = -oct[2 * m + 1]; res[2 * i + 1] = oct[2 * m]; }
            else { const size_t m = 2 * n - i4; res[2 * i] = -oct[2 * m]; res[2 * i + 1] = oct[2 * m + 1]; }
        }
    }
    free(oct);
    return 0;
}

/* pocketfft.c:234-274, 2155-2182: would make_rfft_plan() run this length through the radix passes (1) or through Bluestein (0)? */
static size_t largest_prime_factor(size_t n)
{
    size_t res = 1;
    while ((n & 1) == 0) { res = 2; n >>= 1; }
    size_t limit = (size_t)sqrt((double)n + 0.01);
    for (size_t x = 3; x <= limit; x += 2)
        while (n % x == 0) { res = x; n /= x; limit = (size_t)sqrt((double)n + 0.01); }
    if (n > 1) res = n;
    return res;
}
static double cost_guess(size_t n)
{
    const double lfp = 1.1;
    const size_t ni = n;
    double result = 0.0;
    while ((n & 1) == 0) { result += 2; n >>= 1; }
    size_t limit = (size_t)sqrt((double)n + 0.01);
    for (size_t x = 3; x <= limit; x += 2)
        while (n % x == 0) { result += (x <= 5) ? (double)x : lfp * (double)x; n /= x; limit = (size_t)sqrt((double)n + 0.01); }
    if (n > 1) result += (n <= 5) ? (double)n : lfp * (double)n;
    return result * (double)ni;
}
static size_t good_size(size_t n)
{
    if (n <= 6) return n;
    size_t best = 2 * n;
    for (size_t f2 = 1; f2 < best; f2 *= 2)
        for (size_t f23 = f2; f23 < best; f23 *= 3)
            for (size_t f235 = f23; f235 < best; f235 *= 5)
                for (size_t f2357 = f235; f2357 < best; f2357 *= 7)
                    for (size_t f = f2357; f < best; f *= 11)
                        if (f >= n) best = f;
    return best;
}
static int radix_plan_chosen(size_t n)
{
    if (n < 50 || (double)largest_prime_factor(n) <= sqrt((double)n)) return 1;
    const double comp1 = 0.5 * cost_guess(n);
    double comp2 = 2 * cost_guess(good_size(2 * n - 1));
    comp2 *= 1.5;
    return !(comp2 < comp1);
}

/* ------------------------------------------------------------------ */
/* real forward FFT plan (factors 4, 2, 3, 5, anything else)          */
/* ------------------------------------------------------------------ */

int orc_rfft_plan_init(OrcRfftPlan *p, size_t n)
{
    memset(p, 0, sizeof(*p));
    if (n < 2 || n > ORC_FFT_MAX || !radix_plan_chosen(n)) return -1;      /* (Bluestein lengths: not restated) */
    p->n = n;
    /* pocketfft.c:1798-1827: strip 4s, then one 2 which is swapped to the front, then the odd divisors in rising order, then what is left */
    size_t len = n, nf = 0;
    while ((len % 4) == 0) { if (nf >= 16) return -1; p->fct[nf++] = 4; len >>= 2; }
    if ((len % 2) == 0) {
        len >>= 1;
        if (nf >= 16) return -1;
        p->fct[nf++] = 2;
        size_t t = p->fct[0]; p->fct[0] = p->fct[nf - 1]; p->fct[nf - 1] = t;
    }
    size_t maxl = (size_t)sqrt((double)len) + 1;
    for (size_t divisor = 3; len > 1 && divisor < maxl; divisor += 2)
        if ((len % divisor) == 0) {
            while ((len % divisor) == 0) { if (nf >= 16) return -1; p->fct[nf++] = divisor; len /= divisor; }
            maxl = (size_t)sqrt((double)len) + 1;
        }
    if (len > 1) { if (nf >= 16) return -1; p->fct[nf++] = len; }
    p->nfct = nf;
    /* twiddle layout, pocketfft.c:1843-1881 */
    double *circle = (double *)malloc(2 * n * sizeof(double) + 64);
    if (!circle) return -1;
    if (half_circle_table_any(n, circle)) { free(circle); return -1; }
    size_t total = 0, l1 = 1;
    for (size_t k = 0; k < nf; ++k) {
        size_t ip = p->fct[k], ido = n / (l1 * ip);
        total += (ip - 1) * (ido - 1);
        if (ip > 5) total += 2 * ip;
        l1 *= ip;
    }
    p->tw_store = (double *)calloc(total ? total : 1, sizeof(double));
    double *ptr = p->tw_store;
    l1 = 1;
    for (size_t k = 0; k < nf; ++k) {
        size_t ip = p->fct[k], ido = n / (l1 * ip);
        if (k < nf - 1) {
            p->tw[k] = ptr;
            ptr += (ip - 1) * (ido - 1);
            for (size_t j = 1; j < ip; ++j)
                for (size_t i = 1; i <= (ido - 1) / 2; ++i) {
                    p->tw[k][(j - 1) * (ido - 1) + 2 * i - 2] = circle[2 * j * l1 * i];
                    p->tw[k][(j - 1) * (ido - 1) + 2 * i - 1] = circle[2 * j * l1 * i + 1];
                }
        }
        if (ip > 5) {       /* the generic pass's own roots of unity: (cos, sin)(2 pi i / ip), the upper half by conjugation */
            double *r = p->tws[k] = ptr;
            ptr += 2 * ip;
            r[0] = 1.0; r[1] = 0.0;
            for (size_t i = 1; i <= (ip >> 1); ++i) {
                const double c = circle[2 * i * (n / ip)], sn = circle[2 * i * (n / ip) + 1];
                r[2 * i] = c; r[2 * i + 1] = sn;
                r[2 * (ip - i)] = c; r[2 * (ip - i) + 1] = -sn;
            }
        }
        l1 *= ip;
    }
    free(circle);
    return 0;
}

void orc_rfft_plan_free(OrcRfftPlan *p) { free(p->tw_store); p->tw_store = NULL; }

/* radix-2 real butterfly pass, pocketfft.c:1111-1134 */
static void pass2(size_t ido, size_t l1, const double *in, double *out, const double *w)
{
#define IN2(a, b, c) in[(a) + ido * ((b) + l1 * (c))]
#define OUT2(a, b, c) out[(a) + ido * ((b) + 2 * (c))]
    for (size_t k = 0; k < l1; ++k) {
        OUT2(0, 0, k) = IN2(0, k, 0) + IN2(0, k, 1);
        OUT2(ido - 1, 1, k) = IN2(0, k, 0) - IN2(0, k, 1);
    }
    if ((ido & 1) == 0)
        for (size_t k = 0; k < l1; ++k) {
            OUT2(0, 1, k) = -IN2(ido - 1, k, 1);
            OUT2(ido - 1, 0, k) = IN2(ido - 1, k, 0);
        }
    if (ido <= 2) return;
    for (size_t k = 0; k < l1; ++k)
        for (size_t i = 2; i < ido; i += 2) {
            const size_t ic = ido - i;
            const double wr = w[i - 2], wi = w[i - 1];
            const double xr = IN2(i - 1, k, 1), xi = IN2(i, k, 1);
            const double tr2 = wr * xr + wi * xi;
            const double ti2 = wr * xi - wi * xr;
            OUT2(i - 1, 0, k) = IN2(i - 1, k, 0) + tr2;
            OUT2(ic - 1, 1, k) = IN2(i - 1, k, 0) - tr2;
            OUT2(i, 0, k) = ti2 + IN2(i, k, 0);
            OUT2(ic, 1, k) = ti2 - IN2(i, k, 0);
        }
#undef IN2
#undef OUT2
}

/* radix-4 real butterfly pass, pocketfft.c:1170-1209 */
static void pass4(size_t ido, size_t l1, const double *in, double *out, const double *w)
{
    static const double hsqt2 = 0.70710678118654752440;
#define IN4(a, b, c) in[(a) + ido * ((b) + l1 * (c))]
#define OUT4(a, b, c) out[(a) + ido * ((b) + 4 * (c))]
#define W4(x, i) w[(i) + (x) * (ido - 1)]
    for (size_t k = 0; k < l1; ++k) {
        const double tr1 = IN4(0, k, 3) + IN4(0, k, 1);
        OUT4(0, 2, k) = IN4(0, k, 3) - IN4(0, k, 1);
        const double tr2 = IN4(0, k, 0) + IN4(0, k, 2);
        OUT4(ido - 1, 1, k) = IN4(0, k, 0) - IN4(0, k, 2);
        OUT4(0, 0, k) = tr2 + tr1;
        OUT4(ido - 1, 3, k) = tr2 - tr1;
    }
    if ((ido & 1) == 0)
        for (size_t k = 0; k < l1; ++k) {
            const double ti1 = -hsqt2 * (IN4(ido - 1, k, 1) + IN4(ido - 1, k, 3));
            const double tr1 = hsqt2 * (IN4(ido - 1, k, 1) - IN4(ido - 1, k, 3));
            OUT4(ido - 1, 0, k) = IN4(ido - 1, k, 0) + tr1;
            OUT4(ido - 1, 2, k) = IN4(ido - 1, k, 0) - tr1;
            OUT4(0, 3, k) = ti1 + IN4(ido - 1, k, 2);
            OUT4(0, 1, k) = ti1 - IN4(ido - 1, k, 2);
        }
    if (ido <= 2) return;
    for (size_t k = 0; k < l1; ++k)
        for (size_t i = 2; i < ido; i += 2) {
            const size_t ic = ido - i;
            /* c_j = conj(w_j) * x_j */
            const double cr2 = W4(0, i - 2) * IN4(i - 1, k, 1) + W4(0, i - 1) * IN4(i, k, 1);
            const double ci2 = W4(0, i - 2) * IN4(i, k, 1) - W4(0, i - 1) * IN4(i - 1, k, 1);
            const double cr3 = W4(1, i - 2) * IN4(i - 1, k, 2) + W4(1, i - 1) * IN4(i, k, 2);
            const double ci3 = W4(1, i - 2) * IN4(i, k, 2) - W4(1, i - 1) * IN4(i - 1, k, 2);
            const double cr4 = W4(2, i - 2) * IN4(i - 1, k, 3) + W4(2, i - 1) * IN4(i, k, 3);
            const double ci4 = W4(2, i - 2) * IN4(i, k, 3) - W4(2, i - 1) * IN4(i - 1, k, 3);
            const double tr1 = cr4 + cr2, tr4 = cr4 - cr2;
            const double ti1 = ci2 + ci4, ti4 = ci2 - ci4;
            const double tr2 = IN4(i - 1, k, 0) + cr3, tr3 = IN4(i - 1, k, 0) - cr3;
            const double ti2 = IN4(i, k, 0) + ci3, ti3 = IN4(i, k, 0) - ci3;
            OUT4(i - 1, 0, k) = tr2 + tr1;  OUT4(ic - 1, 3, k) = tr2 - tr1;
            OUT4(i, 0, k) = ti1 + ti2;      OUT4(ic, 3, k) = ti1 - ti2;
            OUT4(i - 1, 2, k) = tr3 + ti4;  OUT4(ic - 1, 1, k) = tr3 - ti4;
            OUT4(i, 2, k) = tr4 + ti3;      OUT4(ic, 1, k) = tr4 - ti3;
        }
#undef IN4
#undef OUT4
#undef W4
}

/* radix-3 real butterfly pass, pocketfft.c:1136-1168 */
static void pass3(size_t ido, size_t l1, const double *in, double *out, const double *w)
{
    static const double taur = -0.5, taui = 0.86602540378443864676;
#define IN3(a, b, c) in[(a) + ido * ((b) + l1 * (c))]
#define OUT3(a, b, c) out[(a) + ido * ((b) + 3 * (c))]
#define W3(x, i) w[(i) + (x) * (ido - 1)]
    for (size_t k = 0; k < l1; ++k) {
        const double cr2 = IN3(0, k, 1) + IN3(0, k, 2);
        OUT3(0, 0, k) = IN3(0, k, 0) + cr2;
        OUT3(0, 2, k) = taui * (IN3(0, k, 2) - IN3(0, k, 1));
        OUT3(ido - 1, 1, k) = IN3(0, k, 0) + taur * cr2;
    }
    if (ido == 1) return;
    for (size_t k = 0; k < l1; ++k)
        for (size_t i = 2; i < ido; i += 2) {
            const size_t ic = ido - i;
            /* d_j = conj(w_j) * x_j */
            const double dr2 = W3(0, i - 2) * IN3(i - 1, k, 1) + W3(0, i - 1) * IN3(i, k, 1);
            const double di2 = W3(0, i - 2) * IN3(i, k, 1) - W3(0, i - 1) * IN3(i - 1, k, 1);
            const double dr3 = W3(1, i - 2) * IN3(i - 1, k, 2) + W3(1, i - 1) * IN3(i, k, 2);
            const double di3 = W3(1, i - 2) * IN3(i, k, 2) - W3(1, i - 1) * IN3(i - 1, k, 2);
            const double cr2 = dr2 + dr3, ci2 = di2 + di3;
            OUT3(i - 1, 0, k) = IN3(i - 1, k, 0) + cr2;
            OUT3(i, 0, k) = IN3(i, k, 0) + ci2;
            const double tr2 = IN3(i - 1, k, 0) + taur * cr2;
            const double ti2 = IN3(i, k, 0) + taur * ci2;
            const double tr3 = taui * (di2 - di3);
            const double ti3 = taui * (dr3 - dr2);
            OUT3(i - 1, 2, k) = tr2 + tr3;  OUT3(ic - 1, 1, k) = tr2 - tr3;
            OUT3(i, 2, k) = ti3 + ti2;      OUT3(ic, 1, k) = ti3 - ti2;
        }
#undef IN3
#undef OUT3
#undef W3
}

/* radix-5 real butterfly pass, pocketfft.c:1211-1260 */
static void pass5(size_t ido, size_t l1, const double *in, double *out, const double *w)
{
    static const double tr11 = 0.3090169943749474241, ti11 = 0.95105651629515357212,
                        tr12 = -0.8090169943749474241, ti12 = 0.58778525229247312917;
#define IN5(a, b, c) in[(a) + ido * ((b) + l1 * (c))]
#define OUT5(a, b, c) out[(a) + ido * ((b) + 5 * (c))]
#define W5(x, i) w[(i) + (x) * (ido - 1)]
    for (size_t k = 0; k < l1; ++k) {
        const double cr2 = IN5(0, k, 4) + IN5(0, k, 1), ci5 = IN5(0, k, 4) - IN5(0, k, 1);
        const double cr3 = IN5(0, k, 3) + IN5(0, k, 2), ci4 = IN5(0, k, 3) - IN5(0, k, 2);
        OUT5(0, 0, k) = IN5(0, k, 0) + cr2 + cr3;
        OUT5(ido - 1, 1, k) = IN5(0, k, 0) + tr11 * cr2 + tr12 * cr3;
        OUT5(0, 2, k) = ti11 * ci5 + ti12 * ci4;
        OUT5(ido - 1, 3, k) = IN5(0, k, 0) + tr12 * cr2 + tr11 * cr3;
        OUT5(0, 4, k) = ti12 * ci5 - ti11 * ci4;
    }
    if (ido == 1) return;
    for (size_t k = 0; k < l1; ++k)
        for (size_t i = 2; i < ido; i += 2) {
            const size_t ic = ido - i;
            const double dr2 = W5(0, i - 2) * IN5(i - 1, k, 1) + W5(0, i - 1) * IN5(i, k, 1);
            const double di2 = W5(0, i - 2) * IN5(i, k, 1) - W5(0, i - 1) * IN5(i - 1, k, 1);
            const double dr3 = W5(1, i - 2) * IN5(i - 1, k, 2) + W5(1, i - 1) * IN5(i, k, 2);
            const double di3 = W5(1, i - 2) * IN5(i, k, 2) - W5(1, i - 1) * IN5(i - 1, k, 2);
            const double dr4 = W5(2, i - 2) * IN5(i - 1, k, 3) + W5(2, i - 1) * IN5(i, k, 3);
            const double di4 = W5(2, i - 2) * IN5(i, k, 3) - W5(2, i - 1) * IN5(i - 1, k, 3);
            const double dr5 = W5(3, i - 2) * IN5(i - 1, k, 4) + W5(3, i - 1) * IN5(i, k, 4);
            const double di5 = W5(3, i - 2) * IN5(i, k, 4) - W5(3, i - 1) * IN5(i - 1, k, 4);
            const double cr2 = dr5 + dr2, ci5 = dr5 - dr2;
            const double ci2 = di2 + di5, cr5 = di2 - di5;
            const double cr3 = dr4 + dr3, ci4 = dr4 - dr3;
            const double ci3 = di3 + di4, cr4 = di3 - di4;
            OUT5(i - 1, 0, k) = IN5(i - 1, k, 0) + cr2 + cr3;
            OUT5(i, 0, k) = IN5(i, k, 0) + ci2 + ci3;
            const double tr2 = IN5(i - 1, k, 0) + tr11 * cr2 + tr12 * cr3;
            const double ti2 = IN5(i, k, 0) + tr11 * ci2 + tr12 * ci3;
            const double tr3 = IN5(i - 1, k, 0) + tr12 * cr2 + tr11 * cr3;
            const double ti3 = IN5(i, k, 0) + tr12 * ci2 + tr11 * ci3;
            const double tr5 = cr5 * ti11 + cr4 * ti12, tr4 = cr5 * ti12 - cr4 * ti11;
            const double ti5 = ci5 * ti11 + ci4 * ti12, ti4 = ci5 * ti12 - ci4 * ti11;
            OUT5(i - 1, 2, k) = tr2 + tr5;  OUT5(ic - 1, 1, k) = tr2 - tr5;
            OUT5(i, 2, k) = ti5 + ti2;      OUT5(ic, 1, k) = ti5 - ti2;
            OUT5(i - 1, 4, k) = tr3 + tr4;  OUT5(ic - 1, 3, k) = tr3 - tr4;
            OUT5(i, 4, k) = ti4 + ti3;      OUT5(ic, 3, k) = ti4 - ti3;
        }
#undef IN5
#undef OUT5
#undef W5
}

/* generic real butterfly pass for any odd factor ip > 5, pocketfft.c:1266-1409 (radfg).  Three sweeps; every element of a sweep is
 * an independent task with a fixed operation order (the device kernel runs the same tasks one per lane, a barrier between sweeps):
 *   1. in place on x: the twiddle products of the columns j / ip - j and their sum / difference pairs (the i = 0 column without twiddles);
 *   2. x -> y: row l of the ip x ip real DFT over the columns (cosine sums into y[.][l], sine sums into y[.][ip - l]; the terms are
 *      added three at first, then in groups of four, two, one -- the grouping is part of the result), and the plain sum into y[.][0];
 *   3. y -> x: the half-complex interleave.
 * The result is in x (the caller's INPUT buffer); y is scratch. */
static void passg(size_t ido, size_t ip, size_t l1, double *x, double *y, const double *w, const double *cs)
{
    const size_t half = (ip + 1) / 2, idl1 = ido * l1, nq = (ido - 1) / 2;
#define X1(a, b, c) x[(a) + ido * ((b) + l1 * (c))]
#define X2(a, b) x[(a) + idl1 * (b)]
#define Y2(a, b) y[(a) + idl1 * (b)]
#define Y1(a, b, c) y[(a) + ido * ((b) + l1 * (c))]
#define XO(a, b, c) x[(a) + ido * ((b) + ip * (c))]
    /* sweep 1 */
    for (size_t j = 1; j < half; ++j) {
        const size_t jc = ip - j;
        for (size_t k = 0; k < l1; ++k) {
            for (size_t q = 0; q < nq; ++q) {
                const size_t i = 1 + 2 * q;
                const double *wj = w + (j - 1) * (ido - 1) + 2 * q, *wc = w + (jc - 1) * (ido - 1) + 2 * q;
                const double t1 = X1(i, k, j), t2 = X1(i + 1, k, j), t3 = X1(i, k, jc), t4 = X1(i + 1, k, jc);
                const double x1 = wj[0] * t1 + wj[1] * t2, x2 = wj[0] * t2 - wj[1] * t1;
                const double x3 = wc[0] * t3 + wc[1] * t4, x4 = wc[0] * t4 - wc[1] * t3;
                X1(i, k, j) = x1 + x3;  X1(i, k, jc) = x2 - x4;
                X1(i + 1, k, j) = x2 + x4;  X1(i + 1, k, jc) = x3 - x1;
            }
            const double a = X1(0, k, j), b = X1(0, k, jc);
            X1(0, k, j) = a + b;
            X1(0, k, jc) = b - a;
        }
    }
    /* sweep 2 */
    for (size_t l = 1; l < half; ++l) {
        const size_t lc = ip - l;
        for (size_t ik = 0; ik < idl1; ++ik) {
            double re = X2(ik, 0) + cs[2 * l] * X2(ik, 1) + cs[4 * l] * X2(ik, 2);
            double im = cs[2 * l + 1] * X2(ik, ip - 1) + cs[4 * l + 1] * X2(ik, ip - 2);
            size_t ang = 2 * l, j = 3, jc = ip - 3;
            for (; j + 3 < half; j += 4, jc -= 4) {
                size_t a1 = ang + l; if (a1 >= ip) a1 -= ip;
                size_t a2 = a1 + l; if (a2 >= ip) a2 -= ip;
                size_t a3 = a2 + l; if (a3 >= ip) a3 -= ip;
                size_t a4 = a3 + l; if (a4 >= ip) a4 -= ip;
                ang = a4;
                re += cs[2 * a1] * X2(ik, j) + cs[2 * a2] * X2(ik, j + 1) + cs[2 * a3] * X2(ik, j + 2) + cs[2 * a4] * X2(ik, j + 3);
                im += cs[2 * a1 + 1] * X2(ik, jc) + cs[2 * a2 + 1] * X2(ik, jc - 1) + cs[2 * a3 + 1] * X2(ik, jc - 2) + cs[2 * a4 + 1] * X2(ik, jc - 3);
            }
            for (; j + 1 < half; j += 2, jc -= 2) {
                size_t a1 = ang + l; if (a1 >= ip) a1 -= ip;
                size_t a2 = a1 + l; if (a2 >= ip) a2 -= ip;
                ang = a2;
                re += cs[2 * a1] * X2(ik, j) + cs[2 * a2] * X2(ik, j + 1);
                im += cs[2 * a1 + 1] * X2(ik, jc) + cs[2 * a2 + 1] * X2(ik, jc - 1);
            }
            for (; j < half; ++j, --jc) {
                ang += l; if (ang >= ip) ang -= ip;
                re += cs[2 * ang] * X2(ik, j);
                im += cs[2 * ang + 1] * X2(ik, jc);
            }
            Y2(ik, l) = re;
            Y2(ik, lc) = im;
        }
    }
    for (size_t ik = 0; ik < idl1; ++ik) {
        double s = X2(ik, 0);
        for (size_t j = 1; j < half; ++j) s += X2(ik, j);
        Y2(ik, 0) = s;
    }
    /* sweep 3 */
    for (size_t k = 0; k < l1; ++k)
        for (size_t i = 0; i < ido; ++i) XO(i, 0, k) = Y1(i, k, 0);
    for (size_t j = 1; j < half; ++j) {
        const size_t jc = ip - j, j2 = 2 * j - 1;
        for (size_t k = 0; k < l1; ++k) {
            XO(ido - 1, j2, k) = Y1(0, k, j);
            XO(0, j2 + 1, k) = Y1(0, k, jc);
            for (size_t q = 0; q < nq; ++q) {
                const size_t i = 1 + 2 * q, ic = ido - i - 2;
                XO(i, j2 + 1, k) = Y1(i, k, j) + Y1(i, k, jc);
                XO(ic, j2, k) = Y1(i, k, j) - Y1(i, k, jc);
                XO(i + 1, j2 + 1, k) = Y1(i + 1, k, j) + Y1(i + 1, k, jc);
                XO(ic + 1, j2, k) = Y1(i + 1, k, jc) - Y1(i + 1, k, j);
            }
        }
    }
#undef X1
#undef X2
#undef Y2
#undef Y1
#undef XO
}

/* in-place forward real FFT, FFTPACK half-complex output (pocketfft.c:1730-1764) */
void orc_rfft_forward(const OrcRfftPlan *p, double *c, double *scratch)
{
    const size_t n = p->n;
    size_t l1 = n;
    double *a = c, *b = scratch;
    for (size_t k1 = 0; k1 < p->nfct; ++k1) {
        const size_t k = p->nfct - k1 - 1;
        const size_t ip = p->fct[k];
        const size_t ido = n / l1;
        l1 /= ip;
        if (ip == 4) pass4(ido, l1, a, b, p->tw[k]);
        else if (ip == 2) pass2(ido, l1, a, b, p->tw[k]);
        else if (ip == 3) pass3(ido, l1, a, b, p->tw[k]);
        else if (ip == 5) pass5(ido, l1, a, b, p->tw[k]);
        else { passg(ido, ip, l1, a, b, p->tw[k], p->tws[k]); continue; }      /* (its result is in a) */
        double *t = a; a = b; b = t;
    }
    if (a != c) memcpy(c, a, n * sizeof(double));
}

/* ------------------------------------------------------------------ */
/* tables (fbank.c:49-95)                                             */
/* ------------------------------------------------------------------ */

void orc_make_window(float *out, int n)
{
    const double nf = (double)n;
    for (int i = 0; i < n; ++i)
        out[i] = (float)pow(0.5 - 0.5 * cos((double)i / nf * 6.283185307), 0.85);
}

static double mel_of_hz(double f) { return 1127.0 * log(1.0 + f / 700.0); }

void orc_make_melbank(float *tab, int nbins, int nfft_bins, int padded, int rate, int lo_hz, int hi_hz)
{
    if (hi_hz == 0) hi_hz = rate / 2;
    const float bin_hz = (float)rate / (float)padded;
    const float mlo = (float)mel_of_hz((double)lo_hz);
    const float mhi = (float)mel_of_hz((double)hi_hz);
    const float step = (mhi - mlo) / ((float)nbins + 1.0f);
    for (int m = 0; m < nbins; ++m) {
        const float left = mlo + (float)m * step;
        const float center = left + step;
        const float right = center + step;
        for (int j = 0; j < nfft_bins; ++j) {
            const float hz = bin_hz * (float)j;
            const float mel = (float)mel_of_hz((double)hz);
            float w = 0.0f;
            if (mel > left && mel < right)
                w = (mel <= center) ? (mel - left) / (center - left) : (right - mel) / (right - center);
            tab[m * nfft_bins + j] = w;
        }
    }
}

/* ------------------------------------------------------------------ */
/* one frame: fbank.c:228-296                                         */
/* ------------------------------------------------------------------ */

static const float kFloor = 1.1920928955078125e-07f; /* fbank.c:37 */

void orc_fbank_frame(const OrcFbank *fb, const float *frame /* padded samples */, float *out /* nbins */)
{
    const int n = fb->padded;
    double *d = fb->work;          /* n doubles            */
    double *r = fb->work + n;      /* n + 1 doubles        */
    double *scr = fb->work + 2 * n + 1;
    for (int j = 0; j < n; ++j) d[j] = (double)frame[j];
    /* DC removal with a float running sum (fbank.c:241-246) */
    {
        float sum = 0;
        for (int j = 0; j < n; ++j) sum += d[j];
        float mean = sum / n;
        for (int j = 0; j < n; ++j) d[j] -= mean;
    }
    /* pre-emphasis back to front, then the j=0 self term (fbank.c:249-253) */
    {
        const float pe = 0.97f;
        for (int j = n - 1; j > 0; --j) d[j] -= pe * d[j - 1];
        d[0] -= pe * d[0];
    }
    for (int j = 0; j < n; ++j) d[j] *= fb->window[j];
    /* rfft into r[1..n], then shift so bin k = (r[2k], r[2k+1]) (fbank.c:259-270) */
    memcpy(r + 1, d, (size_t)n * sizeof(double));
    orc_rfft_forward(&fb->plan, r + 1, scr);
    r[0] = r[1];
    r[1] = 0.0;
    for (int k = 0; k < fb->nfft_bins; ++k) {
        float re = (float)r[2 * k], im = (float)r[2 * k + 1];
        d[k] = re * re + im * im;
    }
    for (int m = 0; m < fb->nbins; ++m) {
        float v = 0.0f;
        const float *wrow = fb->mel + (size_t)m * fb->nfft_bins;
        for (int k = 0; k < fb->nfft_bins; ++k) {
            float p = (float)d[k];
            v += p * wrow[k];
        }
        out[m] = v;
    }
    for (int m = 0; m < fb->nbins; ++m) {
        float v = out[m] > kFloor ? out[m] : kFloor;
        out[m] = (float)log((double)v);
    }
}

/* ------------------------------------------------------------------ */
/* online object                                                      */
/* ------------------------------------------------------------------ */

OrcFbank *orc_fbank_new(int rate, int shift_ms, int len_ms, int nbins, int round_pow2,
                        int mel_lo, int mel_hi, int seg_count, int seg_step)
{
    OrcFbank *fb = (OrcFbank *)calloc(1, sizeof(*fb));
    fb->shift = shift_ms * rate / 1000;
    int win = len_ms * rate / 1000;
    int padded = win;
    if (round_pow2) { padded = 1; while (padded < win) padded <<= 1; }
    fb->padded = padded;
    fb->nfft_bins = padded / 2;
    fb->nbins = nbins;
    fb->seg_count = seg_count;
    fb->seg_step = seg_step;
    fb->shift_ms = shift_ms;
    if (orc_rfft_plan_init(&fb->plan, (size_t)padded) != 0) { free(fb); return NULL; }
    fb->window = (float *)calloc((size_t)padded, sizeof(float));
    orc_make_window(fb->window, padded);
    fb->mel = (float *)calloc((size_t)nbins * fb->nfft_bins, sizeof(float));
    orc_make_melbank(fb->mel, nbins, fb->nfft_bins, padded, rate, mel_lo, mel_hi);
    fb->ring_frames = seg_count * 32;                 /* fbank.c:147 */
    fb->ring = (float *)calloc((size_t)fb->ring_frames * nbins, sizeof(float));
    fb->fifo_cap = 4 * padded + 65536;
    fb->fifo = (float *)calloc((size_t)fb->fifo_cap, sizeof(float));
    fb->work = (double *)calloc((size_t)(3 * padded + 2), sizeof(double));
    return fb;
}

void orc_fbank_free(OrcFbank *fb)
{
    if (!fb) return;
    orc_rfft_plan_free(&fb->plan);
    free(fb->window); free(fb->mel); free(fb->ring); free(fb->fifo); free(fb->work);
    free(fb);
}

static void push_row(OrcFbank *fb, const float *row, int is_real)
{
    memcpy(fb->ring + (size_t)fb->head * fb->nbins, row, (size_t)fb->nbins * sizeof(float));
    fb->head = (fb->head + 1) % fb->ring_frames;
    fb->avail += 1;
    if (is_real) fb->avail_shadow = (long)fb->avail;   /* fbank.c:300 */
}

/* fbank.c:174-306.  wave == NULL means zeros (fbank.c:175). */
void orc_fbank_accept(OrcFbank *fb, const float *wave, size_t count)
{
    /* The reference processes the call's samples frame by frame and stops
       (dropping the rest of THIS call) when the ring is full (fbank.c:190-193).
       FIFO restatement: append, then cut frames while a whole frame is there. */
    size_t done = 0;
    float row[256];
    if (fb->avail + 1 > (size_t)fb->ring_frames) { fb->dropped = 1; return; } /* whole call dropped */
    while (done < count) {
        size_t room = (size_t)fb->fifo_cap - fb->fifo_len;
        size_t take = count - done < room ? count - done : room;
        if (wave) memcpy(fb->fifo + fb->fifo_len, wave + done, take * sizeof(float));
        else      memset(fb->fifo + fb->fifo_len, 0, take * sizeof(float));
        fb->fifo_len += take;
        done += take;
        size_t pos = 0;
        while (fb->fifo_len - pos >= (size_t)fb->padded) {
            if (fb->avail + 1 > (size_t)fb->ring_frames) {
                /* ring full: reference warns and returns without saving leftover */
                fb->fifo_len = 0;
                fb->dropped = 1;
                return;
            }
            orc_fbank_frame(fb, fb->fifo + pos, row);
            push_row(fb, row, 1);
            pos += (size_t)fb->shift;
        }
        memmove(fb->fifo, fb->fifo + pos, (fb->fifo_len - pos) * sizeof(float));
        fb->fifo_len -= pos;
    }
}

/* fbank.c:308-325 */
int orc_fbank_flush(OrcFbank *fb)
{
    long lim = -(long)(fb->seg_count * 3);
    if (fb->avail_shadow < lim) return 0;
    float row[256];
    for (int m = 0; m < fb->nbins; ++m) row[m] = (float)log((double)kFloor);
    while (fb->avail < (size_t)fb->seg_count) push_row(fb, row, 0);
    return 1;
}

/* fbank.c:327-349 */
int orc_fbank_pull(OrcFbank *fb, float *out)
{
    if (fb->avail < (size_t)fb->seg_count) return 0;
    for (int i = 0; i < fb->seg_count; ++i) {
        int idx = (fb->tail + i) % fb->ring_frames;
        memcpy(out + (size_t)i * fb->nbins, fb->ring + (size_t)idx * fb->nbins,
               (size_t)fb->nbins * sizeof(float));
    }
    fb->tail = (fb->tail + fb->seg_step) % fb->ring_frames;
    fb->avail -= (size_t)fb->seg_step;
    fb->avail_shadow -= fb->seg_step;
    return 1;
}

int orc_fbank_stride_ms(const OrcFbank *fb) { return fb->seg_step * fb->shift_ms; }
const float *orc_fbank_window_ptr(const OrcFbank *fb) { return fb->window; }
const float *orc_fbank_mel_ptr(const OrcFbank *fb) { return fb->mel; }
int orc_fbank_padded(const OrcFbank *fb) { return fb->padded; }
