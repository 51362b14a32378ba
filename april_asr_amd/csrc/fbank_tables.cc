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
    // FFT lengths: multiples of 4 (the twiddle table below is pocketfft's n % 4 == 0 branch) whose prime factors are 2, 3 and 5 (the
    // radix 4 / 2 / 3 / 5 passes of kernels_fbank.hip); round_pow2 = 0 models have the frame length itself, e.g. 400 at 16 kHz / 25 ms
    if (padded < 8 || (padded & 3) != 0 || padded > 8192) return false;
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

    // factors: all 4s, then at most one 2 moved to the front, then the odd divisors in rising order (pocketfft.c:1798-1827)
    t.factors.clear();
    size_t len = (size_t)padded;
    while (len % 4 == 0) { t.factors.push_back(4); len >>= 2; }
    if (len % 2 == 0) { len >>= 1; t.factors.push_back(2); std::swap(t.factors.front(), t.factors.back()); }
    for (size_t divisor = 3; len > 1 && divisor <= 5; divisor += 2)
        while (len % divisor == 0) { t.factors.push_back((int)divisor); len /= divisor; }
    if (len != 1 || t.factors.size() > 16) return false;      // a prime factor above 5 (pocketfft's generic pass / Bluestein): not built

    // twiddles per factor (pocketfft.c:1843-1863): tw[(j-1)*(ido-1) + 2i-2 / 2i-1] = cos/sin(2 pi j l1 i / n)
    Circle circle((size_t)padded);
    t.tw.assign(t.factors.size(), {});
    size_t l1 = 1;
    for (size_t k = 0; k < t.factors.size(); ++k) {
        const size_t ip = (size_t)t.factors[k], ido = (size_t)padded / (l1 * ip);
        if (k + 1 < t.factors.size()) {
            t.tw[k].assign((ip - 1) * (ido - 1), 0.0);
            for (size_t j = 1; j < ip; ++j)
                for (size_t i = 1; i <= (ido - 1) / 2; ++i) {
                    double c, s;
                    circle.point(j * l1 * i, c, s);
                    t.tw[k][(j - 1) * (ido - 1) + 2 * i - 2] = c;
                    t.tw[k][(j - 1) * (ido - 1) + 2 * i - 1] = s;
                }
        }
        l1 *= ip;
    }
    t.pad_value = (float)std::log((double)1.1920928955078125e-07f);
    return true;
}

}  // namespace aprilx
