// Online log-mel filterbank on gfx950: one wavefront per 10 ms frame, batched over
// sessions x new frames.  Bit-compatible restatement of the reference's per-frame
// arithmetic (src/fbank.c:228-296) including its FFT (src/fft/pocketfft.c radf4 /
// radf2 / radf3 / radf5 passes and the generic radfg pass for any other factor,
// :1111-1409,1730-1764): fp64 butterflies in the same
// operation order, fp32 power and mel accumulation in the same order, compiled
// with -ffp-contract=off (the reference is built without FMA contraction).
//
// Data flow per frame: the frame's `padded` PCM16 samples are read from the staged
// HBM buffer (coalesced 2-byte loads), windowed into LDS as fp64, the radix passes
// ping-pong between two LDS buffers (one butterfly task per lane per step), power
// goes back to LDS as fp32, each of the first `nbins` lanes walks its triangular
// mel filter sequentially, and the log row is written to the session's feature ring
// in HBM.  Padding rows (flush) are rows of log(kEps) (src/fbank.c:308-325).
//
// DC removal: the reference keeps a *float* running sum of the frame (fbank.c:241-246).
// Samples are k/32768 with integer k, so for padded <= 512 every partial sum is an
// exact multiple of 2^-15 below 2^9 and the float sum is exact; it equals the integer
// sum of the PCM samples / 32768, which is what the wave reduction computes.  For
// larger frames lane 0 replays the sequential float chain.
//
// Third-party notice.  The radix-4 / 2 / 3 / 5 / generic real-FFT pass structure and the twiddle-factor polynomial
// coefficients restated here follow pocketfft (the FFT the reference links, src/fft/pocketfft.c):
//   Copyright (C) 2010-2019 Max-Planck-Society.  All rights reserved.  BSD 3-Clause License
//   (https://gitlab.mpcdf.mpg.de/mtr/pocketfft/-/blob/81d171a6/LICENSE.md); the full text, including the
//   disclaimer, is in THIRD_PARTY_NOTICES.md at the repository root.
#include "kernels.h"
#include <atomic>

namespace aprilx {

__device__ __forceinline__ float pcm_to_float(int16_t s) { return (float)s / 32768.0f; }   // april_session.c:520-522

// radix-4 pass: cc = in[(a) + ido*((b) + l1*(c))], ch = out[(a) + ido*((b) + 4*(c))]
__device__ void fft_pass4(int ido, int l1, const double *in, double *out, const double *w, int lane)
{
    const double hsqt2 = 0.70710678118654752440;
#define CC(a, b, c) in[(a) + ido * ((b) + l1 * (c))]
#define CH(a, b, c) out[(a) + ido * ((b) + 4 * (c))]
#define WA(x, i) w[(i) + (x) * (ido - 1)]
    for (int k = lane; k < l1; k += 64) {
        const double tr1 = CC(0, k, 3) + CC(0, k, 1);
        CH(0, 2, k) = CC(0, k, 3) - CC(0, k, 1);
        const double tr2 = CC(0, k, 0) + CC(0, k, 2);
        CH(ido - 1, 1, k) = CC(0, k, 0) - CC(0, k, 2);
        CH(0, 0, k) = tr2 + tr1;
        CH(ido - 1, 3, k) = tr2 - tr1;
    }
    if ((ido & 1) == 0) {
        for (int k = lane; k < l1; k += 64) {
            const double ti1 = -hsqt2 * (CC(ido - 1, k, 1) + CC(ido - 1, k, 3));
            const double tr1 = hsqt2 * (CC(ido - 1, k, 1) - CC(ido - 1, k, 3));
            CH(ido - 1, 0, k) = CC(ido - 1, k, 0) + tr1;
            CH(ido - 1, 2, k) = CC(ido - 1, k, 0) - tr1;
            CH(0, 3, k) = ti1 + CC(ido - 1, k, 2);
            CH(0, 1, k) = ti1 - CC(ido - 1, k, 2);
        }
    }
    if (ido > 2) {
        const int nq = (ido - 1) / 2;                 // i = 2, 4, ... < ido
        for (int t = lane; t < l1 * nq; t += 64) {
            const int k = t / nq, i = 2 + 2 * (t % nq), ic = ido - i;
            const double cr2 = WA(0, i - 2) * CC(i - 1, k, 1) + WA(0, i - 1) * CC(i, k, 1);
            const double ci2 = WA(0, i - 2) * CC(i, k, 1) - WA(0, i - 1) * CC(i - 1, k, 1);
            const double cr3 = WA(1, i - 2) * CC(i - 1, k, 2) + WA(1, i - 1) * CC(i, k, 2);
            const double ci3 = WA(1, i - 2) * CC(i, k, 2) - WA(1, i - 1) * CC(i - 1, k, 2);
            const double cr4 = WA(2, i - 2) * CC(i - 1, k, 3) + WA(2, i - 1) * CC(i, k, 3);
            const double ci4 = WA(2, i - 2) * CC(i, k, 3) - WA(2, i - 1) * CC(i - 1, k, 3);
            const double tr1 = cr4 + cr2, tr4 = cr4 - cr2;
            const double ti1 = ci2 + ci4, ti4 = ci2 - ci4;
            const double tr2 = CC(i - 1, k, 0) + cr3, tr3 = CC(i - 1, k, 0) - cr3;
            const double ti2 = CC(i, k, 0) + ci3, ti3 = CC(i, k, 0) - ci3;
            CH(i - 1, 0, k) = tr2 + tr1;  CH(ic - 1, 3, k) = tr2 - tr1;
            CH(i, 0, k) = ti1 + ti2;      CH(ic, 3, k) = ti1 - ti2;
            CH(i - 1, 2, k) = tr3 + ti4;  CH(ic - 1, 1, k) = tr3 - ti4;
            CH(i, 2, k) = tr4 + ti3;      CH(ic, 1, k) = tr4 - ti3;
        }
    }
#undef CC
#undef CH
#undef WA
}

__device__ void fft_pass2(int ido, int l1, const double *in, double *out, const double *w, int lane)
{
#define CC(a, b, c) in[(a) + ido * ((b) + l1 * (c))]
#define CH(a, b, c) out[(a) + ido * ((b) + 2 * (c))]
    for (int k = lane; k < l1; k += 64) {
        CH(0, 0, k) = CC(0, k, 0) + CC(0, k, 1);
        CH(ido - 1, 1, k) = CC(0, k, 0) - CC(0, k, 1);
    }
    if ((ido & 1) == 0) {
        for (int k = lane; k < l1; k += 64) {
            CH(0, 1, k) = -CC(ido - 1, k, 1);
            CH(ido - 1, 0, k) = CC(ido - 1, k, 0);
        }
    }
    if (ido > 2) {
        const int nq = (ido - 1) / 2;
        for (int t = lane; t < l1 * nq; t += 64) {
            const int k = t / nq, i = 2 + 2 * (t % nq), ic = ido - i;
            const double tr2 = w[i - 2] * CC(i - 1, k, 1) + w[i - 1] * CC(i, k, 1);
            const double ti2 = w[i - 2] * CC(i, k, 1) - w[i - 1] * CC(i - 1, k, 1);
            CH(i - 1, 0, k) = CC(i - 1, k, 0) + tr2;
            CH(ic - 1, 1, k) = CC(i - 1, k, 0) - tr2;
            CH(i, 0, k) = ti2 + CC(i, k, 0);
            CH(ic, 1, k) = ti2 - CC(i, k, 0);
        }
    }
#undef CC
#undef CH
}

// radix-3 pass (pocketfft.c:1136-1168): FFT lengths of models with round_pow2 = 0 (e.g. 480 = 2 4 4 3 5)
__device__ void fft_pass3(int ido, int l1, const double *in, double *out, const double *w, int lane)
{
    const double taur = -0.5, taui = 0.86602540378443864676;
#define CC(a, b, c) in[(a) + ido * ((b) + l1 * (c))]
#define CH(a, b, c) out[(a) + ido * ((b) + 3 * (c))]
#define WA(x, i) w[(i) + (x) * (ido - 1)]
    for (int k = lane; k < l1; k += 64) {
        const double cr2 = CC(0, k, 1) + CC(0, k, 2);
        CH(0, 0, k) = CC(0, k, 0) + cr2;
        CH(0, 2, k) = taui * (CC(0, k, 2) - CC(0, k, 1));
        CH(ido - 1, 1, k) = CC(0, k, 0) + taur * cr2;
    }
    if (ido > 2) {
        const int nq = (ido - 1) / 2;                 // i = 2, 4, ... < ido
        for (int t = lane; t < l1 * nq; t += 64) {
            const int k = t / nq, i = 2 + 2 * (t % nq), ic = ido - i;
            const double dr2 = WA(0, i - 2) * CC(i - 1, k, 1) + WA(0, i - 1) * CC(i, k, 1);
            const double di2 = WA(0, i - 2) * CC(i, k, 1) - WA(0, i - 1) * CC(i - 1, k, 1);
            const double dr3 = WA(1, i - 2) * CC(i - 1, k, 2) + WA(1, i - 1) * CC(i, k, 2);
            const double di3 = WA(1, i - 2) * CC(i, k, 2) - WA(1, i - 1) * CC(i - 1, k, 2);
            const double cr2 = dr2 + dr3, ci2 = di2 + di3;
            CH(i - 1, 0, k) = CC(i - 1, k, 0) + cr2;
            CH(i, 0, k) = CC(i, k, 0) + ci2;
            const double tr2 = CC(i - 1, k, 0) + taur * cr2;
            const double ti2 = CC(i, k, 0) + taur * ci2;
            const double tr3 = taui * (di2 - di3);
            const double ti3 = taui * (dr3 - dr2);
            CH(i - 1, 2, k) = tr2 + tr3;  CH(ic - 1, 1, k) = tr2 - tr3;
            CH(i, 2, k) = ti3 + ti2;      CH(ic, 1, k) = ti3 - ti2;
        }
    }
#undef CC
#undef CH
#undef WA
}

// radix-5 pass (pocketfft.c:1211-1260): 400 = 4 4 5 5 (16 kHz, 25 ms, round_pow2 = 0)
__device__ void fft_pass5(int ido, int l1, const double *in, double *out, const double *w, int lane)
{
    const double tr11 = 0.3090169943749474241, ti11 = 0.95105651629515357212, tr12 = -0.8090169943749474241, ti12 = 0.58778525229247312917;
#define CC(a, b, c) in[(a) + ido * ((b) + l1 * (c))]
#define CH(a, b, c) out[(a) + ido * ((b) + 5 * (c))]
#define WA(x, i) w[(i) + (x) * (ido - 1)]
    for (int k = lane; k < l1; k += 64) {
        const double cr2 = CC(0, k, 4) + CC(0, k, 1), ci5 = CC(0, k, 4) - CC(0, k, 1);
        const double cr3 = CC(0, k, 3) + CC(0, k, 2), ci4 = CC(0, k, 3) - CC(0, k, 2);
        CH(0, 0, k) = CC(0, k, 0) + cr2 + cr3;
        CH(ido - 1, 1, k) = CC(0, k, 0) + tr11 * cr2 + tr12 * cr3;
        CH(0, 2, k) = ti11 * ci5 + ti12 * ci4;
        CH(ido - 1, 3, k) = CC(0, k, 0) + tr12 * cr2 + tr11 * cr3;
        CH(0, 4, k) = ti12 * ci5 - ti11 * ci4;
    }
    if (ido > 2) {
        const int nq = (ido - 1) / 2;
        for (int t = lane; t < l1 * nq; t += 64) {
            const int k = t / nq, i = 2 + 2 * (t % nq), ic = ido - i;
            const double dr2 = WA(0, i - 2) * CC(i - 1, k, 1) + WA(0, i - 1) * CC(i, k, 1);
            const double di2 = WA(0, i - 2) * CC(i, k, 1) - WA(0, i - 1) * CC(i - 1, k, 1);
            const double dr3 = WA(1, i - 2) * CC(i - 1, k, 2) + WA(1, i - 1) * CC(i, k, 2);
            const double di3 = WA(1, i - 2) * CC(i, k, 2) - WA(1, i - 1) * CC(i - 1, k, 2);
            const double dr4 = WA(2, i - 2) * CC(i - 1, k, 3) + WA(2, i - 1) * CC(i, k, 3);
            const double di4 = WA(2, i - 2) * CC(i, k, 3) - WA(2, i - 1) * CC(i - 1, k, 3);
            const double dr5 = WA(3, i - 2) * CC(i - 1, k, 4) + WA(3, i - 1) * CC(i, k, 4);
            const double di5 = WA(3, i - 2) * CC(i, k, 4) - WA(3, i - 1) * CC(i - 1, k, 4);
            const double cr2 = dr5 + dr2, ci5 = dr5 - dr2;
            const double ci2 = di2 + di5, cr5 = di2 - di5;
            const double cr3 = dr4 + dr3, ci4 = dr4 - dr3;
            const double ci3 = di3 + di4, cr4 = di3 - di4;
            CH(i - 1, 0, k) = CC(i - 1, k, 0) + cr2 + cr3;
            CH(i, 0, k) = CC(i, k, 0) + ci2 + ci3;
            const double tr2 = CC(i - 1, k, 0) + tr11 * cr2 + tr12 * cr3;
            const double ti2 = CC(i, k, 0) + tr11 * ci2 + tr12 * ci3;
            const double tr3 = CC(i - 1, k, 0) + tr12 * cr2 + tr11 * cr3;
            const double ti3 = CC(i, k, 0) + tr12 * ci2 + tr11 * ci3;
            const double tr5 = cr5 * ti11 + cr4 * ti12, tr4 = cr5 * ti12 - cr4 * ti11;
            const double ti5 = ci5 * ti11 + ci4 * ti12, ti4 = ci5 * ti12 - ci4 * ti11;
            CH(i - 1, 2, k) = tr2 + tr5;  CH(ic - 1, 1, k) = tr2 - tr5;
            CH(i, 2, k) = ti5 + ti2;      CH(ic, 1, k) = ti5 - ti2;
            CH(i - 1, 4, k) = tr3 + tr4;  CH(ic - 1, 3, k) = tr3 - tr4;
            CH(i, 4, k) = ti4 + ti3;      CH(ic, 3, k) = ti4 - ti3;
        }
    }
#undef CC
#undef CH
#undef WA
}

// generic pass for any odd factor ip > 5 (pocketfft.c:1266-1409, radfg): three sweeps with a barrier between them, every element of a
// sweep one lane task with the reference's operation order (oracle/orc_fbank.c passg states the same tasks sequentially):
//   1. in place on x: twiddle products of the columns j / ip - j and their sum / difference pairs (the i = 0 column without twiddles);
//   2. x -> y: row l of the ip x ip real DFT over the columns -- cosine sums into y[.][l], sine sums into y[.][ip - l]; the terms are
//      added three at first, then in groups of four, two, one (the grouping is part of the bit pattern) -- and the plain sum into y[.][0];
//   3. y -> x: the half-complex interleave.  The result is in x (the pass's INPUT buffer).
__device__ void fft_pass_generic(int ido, int ip, int l1, double *x, double *y, const double *w, const double *cs, int lane)
{
    const int half = (ip + 1) / 2, idl1 = ido * l1, nq = (ido - 1) / 2;
#define X1(a, b, c) x[(a) + ido * ((b) + l1 * (c))]
#define X2(a, b) x[(a) + idl1 * (b)]
#define Y2(a, b) y[(a) + idl1 * (b)]
#define Y1(a, b, c) y[(a) + ido * ((b) + l1 * (c))]
#define XO(a, b, c) x[(a) + ido * ((b) + ip * (c))]
    // sweep 1: tasks (j, k, q) with q = nq standing for the i = 0 column
    for (int t = lane; t < (half - 1) * l1 * (nq + 1); t += 64) {
        const int q = t % (nq + 1), k = (t / (nq + 1)) % l1, j = 1 + t / ((nq + 1) * l1), jc = ip - j;
        if (q == nq) {
            const double a = X1(0, k, j), b = X1(0, k, jc);
            X1(0, k, j) = a + b;
            X1(0, k, jc) = b - a;
        } else {
            const int i = 1 + 2 * q;
            const double *wj = w + (j - 1) * (ido - 1) + 2 * q, *wc = w + (jc - 1) * (ido - 1) + 2 * q;
            const double t1 = X1(i, k, j), t2 = X1(i + 1, k, j), t3 = X1(i, k, jc), t4 = X1(i + 1, k, jc);
            const double x1 = wj[0] * t1 + wj[1] * t2, x2 = wj[0] * t2 - wj[1] * t1;
            const double x3 = wc[0] * t3 + wc[1] * t4, x4 = wc[0] * t4 - wc[1] * t3;
            X1(i, k, j) = x1 + x3;  X1(i, k, jc) = x2 - x4;
            X1(i + 1, k, j) = x2 + x4;  X1(i + 1, k, jc) = x3 - x1;
        }
    }
    __syncthreads();
    // sweep 2: tasks (l, ik), l = 0 standing for the plain column sum
    for (int t = lane; t < half * idl1; t += 64) {
        const int ik = t % idl1, l = t / idl1;
        if (l == 0) {
            double s = X2(ik, 0);
            for (int j = 1; j < half; ++j) s += X2(ik, j);
            Y2(ik, 0) = s;
            continue;
        }
        double re = X2(ik, 0) + cs[2 * l] * X2(ik, 1) + cs[4 * l] * X2(ik, 2);
        double im = cs[2 * l + 1] * X2(ik, ip - 1) + cs[4 * l + 1] * X2(ik, ip - 2);
        int ang = 2 * l, j = 3, jc = ip - 3;
        for (; j + 3 < half; j += 4, jc -= 4) {
            int a1 = ang + l; if (a1 >= ip) a1 -= ip;
            int a2 = a1 + l; if (a2 >= ip) a2 -= ip;
            int a3 = a2 + l; if (a3 >= ip) a3 -= ip;
            int a4 = a3 + l; if (a4 >= ip) a4 -= ip;
            ang = a4;
            re += cs[2 * a1] * X2(ik, j) + cs[2 * a2] * X2(ik, j + 1) + cs[2 * a3] * X2(ik, j + 2) + cs[2 * a4] * X2(ik, j + 3);
            im += cs[2 * a1 + 1] * X2(ik, jc) + cs[2 * a2 + 1] * X2(ik, jc - 1) + cs[2 * a3 + 1] * X2(ik, jc - 2) + cs[2 * a4 + 1] * X2(ik, jc - 3);
        }
        for (; j + 1 < half; j += 2, jc -= 2) {
            int a1 = ang + l; if (a1 >= ip) a1 -= ip;
            int a2 = a1 + l; if (a2 >= ip) a2 -= ip;
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
        Y2(ik, ip - l) = im;
    }
    __syncthreads();
    // sweep 3: tasks (j, k, q); j = 0: the copy of column 0 (q runs over all ido elements there)
    for (int t = lane; t < l1 * ido; t += 64) { const int i = t % ido, k = t / ido; XO(i, 0, k) = Y1(i, k, 0); }
    for (int t = lane; t < (half - 1) * l1 * (nq + 1); t += 64) {
        const int q = t % (nq + 1), k = (t / (nq + 1)) % l1, j = 1 + t / ((nq + 1) * l1), jc = ip - j, j2 = 2 * j - 1;
        if (q == nq) {
            XO(ido - 1, j2, k) = Y1(0, k, j);
            XO(0, j2 + 1, k) = Y1(0, k, jc);
        } else {
            const int i = 1 + 2 * q, ic = ido - i - 2;
            XO(i, j2 + 1, k) = Y1(i, k, j) + Y1(i, k, jc);
            XO(ic, j2, k) = Y1(i, k, j) - Y1(i, k, jc);
            XO(i + 1, j2 + 1, k) = Y1(i + 1, k, j) + Y1(i + 1, k, jc);
            XO(ic + 1, j2, k) = Y1(i + 1, k, jc) - Y1(i + 1, k, j);
        }
    }
#undef X1
#undef X2
#undef Y2
#undef Y1
#undef XO
}

__global__ __launch_bounds__(64) void fbank_kernel(FbankArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int lane = threadIdx.x;
    const FbankFrameDesc d = a.desc[blockIdx.x];
    const int n = a.t.padded, nbins = a.t.nbins, nfft = n >> 1;
    float *out = a.ring + ((size_t)d.slot * a.ring_frames + d.ring_row) * nbins;
    if (d.pcm_off < 0) {
        for (int m = lane; m < nbins; m += 64) out[m] = a.pad_value;
        return;
    }
    const int16_t *pcm = a.pcm + d.pcm_off;
    double *A = sh, *B = sh + n;

    // DC offset (fbank.c:241-246)
    float mean;
    if (n <= 512) {
        int part = 0;
        for (int j = lane; j < n; j += 64) part += (int)pcm[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        const float sum = (float)part / 32768.0f;
        mean = sum / (float)n;
    } else {
        float sum = 0.0f;
        if (lane == 0) for (int j = 0; j < n; ++j) sum = (float)((double)sum + (double)pcm_to_float(pcm[j]));
        sum = __shfl(sum, 0);
        mean = sum / (float)n;
    }
    // subtract mean, pre-emphasis (back-to-front form of fbank.c:249-253 is order independent), window
    const double pe = (double)0.97f;
    for (int j = lane; j < n; j += 64) {
        const double dj = (double)pcm_to_float(pcm[j]) - (double)mean;
        const double dp = j > 0 ? (double)pcm_to_float(pcm[j - 1]) - (double)mean : dj;
        double v = dj - pe * dp;
        v *= (double)a.t.window[j];
        A[j] = v;
    }
    __syncthreads();

    // forward real FFT: factors are consumed last to first (pocketfft.c:1741-1760)
    double *src = A, *dst = B;
    int l1 = n;
    for (int k1 = 0; k1 < a.t.nfct; ++k1) {
        const int k = a.t.nfct - k1 - 1;
        const int ip = a.t.fct[k];
        const int ido = n / l1;
        l1 /= ip;
        if (ip == 4) fft_pass4(ido, l1, src, dst, a.t.tw[k], lane);
        else if (ip == 2) fft_pass2(ido, l1, src, dst, a.t.tw[k], lane);
        else if (ip == 3) fft_pass3(ido, l1, src, dst, a.t.tw[k], lane);
        else if (ip == 5) fft_pass5(ido, l1, src, dst, a.t.tw[k], lane);
        else { fft_pass_generic(ido, ip, l1, src, dst, a.t.tw[k], a.t.tws[k], lane); __syncthreads(); continue; }      // (its result is in src)
        __syncthreads();
        double *t = src; src = dst; dst = t;
    }
    // src = half-complex spectrum r0, r1,i1, ..., r_{n/2}; power of bins 0..n/2-1 (fbank.c:259-280)
    float *pw = reinterpret_cast<float *>(dst);
    for (int k = lane; k < nfft; k += 64) {
        const float re = (float)(k == 0 ? src[0] : src[2 * k - 1]);
        const float im = (float)(k == 0 ? 0.0 : src[2 * k]);
        pw[k] = re * re + im * im;
    }
    __syncthreads();
    const float kFloor = 1.1920928955078125e-07f;
    for (int m = lane; m < nbins; m += 64) {
        const float *w = a.t.mel + (size_t)m * nfft;
        float val = 0.0f;
        const int hi = a.t.mel_hi[m];
        // eight taps' weights (global memory) and powers (LDS) are fetched together, then added in bin order: the sum is the same sequence
        // of products, but a filter of thirty taps waits for four round trips instead of thirty (the loop with one load per tap was
        // half of a frame's latency, and the frames in flight per CU are bounded by LDS)
        for (int k0 = a.t.mel_lo[m]; k0 < hi; k0 += 8) {
            float wv[8], pv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int k = k0 + i < hi ? k0 + i : hi - 1; wv[i] = w[k]; pv[i] = pw[k]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) if (k0 + i < hi) val += pv[i] * wv[i];
        }
        const float v = kFloor > val ? kFloor : val;
        out[m] = (float)log((double)v);
    }
}

void launch_fbank(const FbankArgs &a, hipStream_t s)
{
    if (a.n_frames <= 0) return;
    const size_t lds = sizeof(double) * 2 * (size_t)a.t.padded;
    if (lds > 64 * 1024) {          // frames above 4096 samples (the tables go up to 8192): dynamic LDS beyond 64 KB has to be announced, per device
        static std::atomic<uint64_t> attr_devs{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fbank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_devs.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)a.n_frames), dim3(64), lds, s, a);
}

}  // namespace aprilx
