// See fbank_tables.h.
//
// Third-party notice.  The radix-4 / 2 / 3 / 5 real-FFT pass structure and the twiddle-factor polynomial
// coefficients restated here follow pocketfft (the FFT the reference links, src/fft/pocketfft.c):
//   Copyright (C) 2010-2019 Max-Planck-Society.  All rights reserved.  BSD 3-Clause License
//   (https://gitlab.mpcdf.mpg.de/mtr/pocketfft/-/blob/81d171a6/LICENSE.md); the full text, including the
//   disclaimer, is in THIRD_PARTY_NOTICES.md at the repository root.
#include "fbank_tables.h"
#include <cmath>

namespace aprilx {
namespace {

// minimax polynomials for cos(pi a) - 1 and sin(pi a), |a| <= 1/4 -- the coefficients and
// the Horner/fma evaluation order are those of pocketfft.c:65-90 (bitwise reproducibility of
// the twiddles is what makes the GPU fbank bit-compatible with the reference).
struct CosM1Sin { double cm1, sn; };
CosM1Sin eval_octant_poly(double a)
{
    const double s = a * a;
    double r = -1.0369917389758117e-4;
    r = std::fma(r, s, 1.9294935641298806e-3);
    r = std::fma(r, s, -2.5806887942825395e-2);
    r = std::fma(r, s, 2.3533063028328211e-1);
    r = std::fma(r, s, -1.3352627688538006e+0);
    r = std::fma(r, s, 4.0587121264167623e+0);
    r = std::fma(r, s, -4.9348022005446790e+0);
    CosM1Sin out;
    out.cm1 = r * s;
    r = 4.6151442520157035e-4;
    r = std::fma(r, s, -7.3700183130883555e-3);
    r = std::fma(r, s, 8.2145868949323936e-2);
    r = std::fma(r, s, -5.9926452893214921e-1);
    r = std::fma(r, s, 2.5501640398732688e+0);
    r = std::fma(r, s, -5.1677127800499516e+0);
    const double a3 = s * a;
    r = r * a3;
    out.sn = std::fma(a, 3.1415926535897931e+0, r);
    return out;
}

// (cos, sin)(2 pi m / n) for m in [0, (n+4)/8): blocked angle addition, pocketfft.c:92-123
void first_octant(size_t n, std::vector<double> &cs)
{
    const size_t cnt = (n + 4) >> 3;
    cs.assign(2 * (cnt ? cnt : 1), 0.0);
    cs[0] = 1.0; cs[1] = 0.0;
    if (cnt <= 1) return;
    const size_t blk = (size_t)std::sqrt((double)cnt);
    std::vector<CosM1Sin> small(blk);
    for (size_t i = 1; i < blk; ++i) small[i] = eval_octant_poly((2.0 * (double)i) / (double)n);
    for (size_t start = blk; start < cnt; start += blk) {
        const CosM1Sin b = eval_octant_poly((2.0 * (double)start) / (double)n);
        cs[2 * start] = b.cm1 + 1.0;
        cs[2 * start + 1] = b.sn;
        size_t end = blk;
        if (start + end > cnt) end = cnt - start;
        for (size_t i = 1; i < end; ++i) {
            const double cx = small[i].cm1, sx = small[i].sn;
            cs[2 * (start + i)] = ((b.cm1 * cx - b.sn * sx + b.cm1) + cx) + 1.0;
            cs[2 * (start + i) + 1] = (b.cm1 * sx + b.sn * cx) + b.sn + sx;
        }
    }
    for (size_t i = 1; i < blk; ++i) { cs[2 * i] = small[i].cm1 + 1.0; cs[2 * i + 1] = small[i].sn; }
}

struct Circle {
    size_t n;
    std::vector<double> oct;
    explicit Circle(size_t n_) : n(n_) { first_octant(n, oct); }
    // exact symmetries used by pocketfft.c:172-214 for n % 4 == 0
    void point(size_t m, double &c, double &s) const {
        const size_t q = n >> 2, e = n >> 3;
        if (m >= q) { double c1, s1; point(m - q, c1, s1); c = -s1; s = c1; return; }
        if ((n & 7) == 0 && m == e) { c = s = 0.707106781186547524400844362104849; return; }
        if (m > e) { double c1, s1; point(q - m, c1, s1); c = s1; s = c1; return; }
        c = oct[2 * m]; s = oct[2 * m + 1];
    }
};

// (cos, sin)(2 pi m / n) for m in [0, n / 2], any n: pocketfft takes the first octant of n, 2 n or 4 n (n mod 4 = 0, 2, odd) and reaches
// the rest by exact symmetries (pocketfft.c:121-226); which octant sample a point comes from is part of the bit pattern
std::vector<double> half_circle(size_t n)
{
    std::vector<double> t(2 * (n / 2 + 1), 0.0);
    if ((n & 3) == 0) {
        Circle c(n);
        for (size_t m = 0; m < n / 2; ++m) c.point(m, t[2 * m], t[2 * m + 1]);
        return t;
    }
    std::vector<double> oct;
    if ((n & 1) == 0) {                                  // n = 4 q + 2: octant of 2 n
        first_octant(2 * n, oct);
        const size_t cnt = (n + 2) >> 2, half = n >> 1;
        for (size_t e = 0; e < cnt; ++e) {
            if (e < (cnt + 1) / 2) { t[2 * e] = oct[4 * e]; t[2 * e + 1] = oct[4 * e + 1]; }
            else { const size_t m = 2 * (cnt - 1 - e) + 1; t[2 * e] = oct[2 * m + 1]; t[2 * e + 1] = oct[2 * m]; }      // mirrored at pi / 4
        }
        for (size_t e = 1; 2 * e < half; ++e) { t[2 * (half - e)] = -t[2 * e]; t[2 * (half - e) + 1] = t[2 * e + 1]; }   // mirrored at pi / 2
    } else {                                             // n odd: point 4 i of the circle of 4 n, folded into its first octant
        first_octant(4 * n, oct);
        const size_t cnt = (n + 1) >> 1;
        for (size_t i = 0; i < cnt; ++i) {
            const size_t i4 = 4 * i;
            if (2 * i4 <= n) { t[2 * i] = oct[2 * i4]; t[2 * i + 1] = oct[2 * i4 + 1]; }
            else if (i4 <= n) { const size_t m = n - i4; t[2 * i] = oct[2 * m + 1]; t[2 * i + 1] = oct[2 * m]; }
            else if (2 * i4 <= 3 * n) { const size_t m = i4 - n; t[2 * i] = -oct[2 * m + 1]; t[2 * i + 1] = oct[2 * m]; }
            else { const size_t m = 2 * n - i4; t[2 * i] = -oct[2 * m]; t[2 * i + 1] = oct[2 * m + 1]; }
        }
    }
    return t;
}

// pocketfft.c:234-289, 2155-2182: does make_rfft_plan() run length n through its radix passes, or through Bluestein's algorithm?
size_t largest_prime_factor(size_t n)
{
    size_t res = 1;
    while ((n & 1) == 0) { res = 2; n >>= 1; }
    size_t limit = (size_t)std::sqrt((double)n + 0.01);
    for (size_t x = 3; x <= limit; x += 2)
        while (n % x == 0) { res = x; n /= x; limit = (size_t)std::sqrt((double)n + 0.01); }
    return n > 1 ? n : res;
}
double cost_guess(size_t n)
{
    const size_t n0 = n;
    double cost = 0.0;
    auto price = [](size_t f) { return f <= 5 ? (double)f : 1.1 * (double)f; };      // factors without a hard-coded pass cost more
    while ((n & 1) == 0) { cost += 2; n >>= 1; }
    size_t limit = (size_t)std::sqrt((double)n + 0.01);
    for (size_t x = 3; x <= limit; x += 2)
        while (n % x == 0) { cost += price(x); n /= x; limit = (size_t)std::sqrt((double)n + 0.01); }
    if (n > 1) cost += price(n);
    return cost * (double)n0;
}
size_t good_size(size_t n)                               // the smallest 2-3-5-7-11-smooth number >= n
{
    if (n <= 6) return n;
    size_t best = 2 * n;
    for (size_t a = 1; a < best; a *= 2)
        for (size_t b = a; b < best; b *= 3)
            for (size_t c = b; c < best; c *= 5)
                for (size_t d = c; d < best; d *= 7)
                    for (size_t e = d; e < best; e *= 11)
                        if (e >= n) best = e;
    return best;
}
bool radix_plan_chosen(size_t n)
{
    if (n < 50 || (double)largest_prime_factor(n) <= std::sqrt((double)n)) return true;
    const double radix = 0.5 * cost_guess(n);
    double blue = 2 * cost_guess(good_size(2 * n - 1));
    blue *= 1.5;
    return !(blue < radix);
}

double mel_of(double hz) { return 1127.0 * std::log(1.0 + hz / 700.0); }

}  // namespace

bool build_fbank_tables(int sample_rate, int frame_shift_ms, int frame_length_ms, int nbins, bool round_pow2,
                        int mel_low, int mel_high, FbankHostTables &t)
{
    t.sample_rate = sample_rate;
    t.shift = frame_shift_ms * sample_rate / 1000;
    t.window_size = frame_length_ms * sample_rate / 1000;
    int padded = t.window_size;
    if (round_pow2) { padded = 1; while (padded < t.window_size) padded <<= 1; }
    // FFT lengths: whatever pocketfft runs through its radix passes -- 4 / 2 / 3 / 5 and the generic pass for any other factor
    // (kernels_fbank.hip); round_pow2 = 0 models have the frame length itself, e.g. 400 at 16 kHz / 25 ms, 882 = 2 3 3 7 7 at 44.1 kHz /
    // 20 ms.  Lengths pocketfft would hand to Bluestein's algorithm (a large prime factor) are refused.
    if (padded < 8 || padded > 8192 || !radix_plan_chosen((size_t)padded)) return false;
    t.padded = padded; t.nfft_bins = padded / 2; t.nbins = nbins;

    // window: fbank.c:49-55 (N = padded length, denominator N)
    t.window.resize((size_t)padded);
    for (int i = 0; i < padded; ++i)
        t.window[(size_t)i] = (float)std::pow(0.5 - 0.5 * std::cos((double)i / (double)padded * 6.283185307), 0.85);

    // mel bank: fbank.c:65-95 (float arithmetic on float-rounded mel values)
    if (mel_high == 0) mel_high = sample_rate / 2;
    const float bin_hz = (float)sample_rate / (float)padded;
    const float mlo = (float)mel_of((double)mel_low), mhi = (float)mel_of((double)mel_high);
    const float step = (mhi - mlo) / ((float)nbins + 1.0f);
    t.mel.assign((size_t)nbins * t.nfft_bins, 0.0f);
    t.mel_lo.assign((size_t)nbins, 0); t.mel_hi.assign((size_t)nbins, 0);
    for (int m = 0; m < nbins; ++m) {
        const float left = mlo + (float)m * step, center = left + step, right = center + step;
        int lo = t.nfft_bins, hi = 0;
        for (int j = 0; j < t.nfft_bins; ++j) {
            const float hz = bin_hz * (float)j;
            const float mel = (float)mel_of((double)hz);
            float w = 0.0f;
            if (mel > left && mel < right) w = mel <= center ? (mel - left) / (center - left) : (right - mel) / (right - center);
            t.mel[(size_t)m * t.nfft_bins + j] = w;
            if (w != 0.0f) { if (j < lo) lo = j; hi = j + 1; }
        }
        if (hi == 0) lo = 0;
        t.mel_lo[(size_t)m] = lo; t.mel_hi[(size_t)m] = hi;
    }

    // factors: all 4s, then at most one 2 moved to the front, then the odd divisors in rising order, then what is left (pocketfft.c:1798-1827)
    t.factors.clear();
    size_t len = (size_t)padded;
    while (len % 4 == 0) { t.factors.push_back(4); len >>= 2; }
    if (len % 2 == 0) { len >>= 1; t.factors.push_back(2); std::swap(t.factors.front(), t.factors.back()); }
    size_t maxl = (size_t)std::sqrt((double)len) + 1;
    for (size_t divisor = 3; len > 1 && divisor < maxl; divisor += 2)
        if (len % divisor == 0) {
            while (len % divisor == 0) { t.factors.push_back((int)divisor); len /= divisor; }
            maxl = (size_t)std::sqrt((double)len) + 1;
        }
    if (len > 1) t.factors.push_back((int)len);
    if (t.factors.size() > 16) return false;

    // twiddles per factor (pocketfft.c:1843-1881): tw[(j-1)*(ido-1) + 2i-2 / 2i-1] = cos/sin(2 pi j l1 i / n); a factor above 5 also gets
    // its own roots of unity (cos, sin)(2 pi i / ip), the upper half by conjugation
    const std::vector<double> circle = half_circle((size_t)padded);
    t.tw.assign(t.factors.size(), {});
    t.tws.assign(t.factors.size(), {});
    size_t l1 = 1;
    for (size_t k = 0; k < t.factors.size(); ++k) {
        const size_t ip = (size_t)t.factors[k], ido = (size_t)padded / (l1 * ip);
        if (k + 1 < t.factors.size()) {
            t.tw[k].assign((ip - 1) * (ido - 1), 0.0);
            for (size_t j = 1; j < ip; ++j)
                for (size_t i = 1; i <= (ido - 1) / 2; ++i) {
                    t.tw[k][(j - 1) * (ido - 1) + 2 * i - 2] = circle[2 * (j * l1 * i)];
                    t.tw[k][(j - 1) * (ido - 1) + 2 * i - 1] = circle[2 * (j * l1 * i) + 1];
                }
        }
        if (ip > 5) {
            std::vector<double> &r = t.tws[k];
            r.assign(2 * ip, 0.0);
            r[0] = 1.0;
            const size_t step = (size_t)padded / ip;
            for (size_t i = 1; i <= ip / 2; ++i) {
                r[2 * i] = r[2 * (ip - i)] = circle[2 * (i * step)];
                r[2 * i + 1] = circle[2 * (i * step) + 1];
                r[2 * (ip - i) + 1] = -circle[2 * (i * step) + 1];
            }
        }
        l1 *= ip;
    }
    t.pad_value = (float)std::log((double)1.1920928955078125e-07f);
    return true;
}

}  // namespace aprilx
