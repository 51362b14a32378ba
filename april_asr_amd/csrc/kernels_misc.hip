// Row-wise epilogues, the encoder's convolutional front end and the decoder's
// embedding/conv front end (gfx950).  All bandwidth/latency-type work: one
// workgroup per session row, coalesced row accesses, LDS for the conv stack's
// intermediates, wavefront shuffles for the row reductions.  Reduction orders are
// fixed (independent of the batch).
#include "kernels.h"
#include "device_utils.h"
#include <algorithm>

namespace aprilx {

__device__ __forceinline__ float sigmoid_dev(float x) { return fast_sigmoid(x); }
__device__ __forceinline__ float dswish_dev(float y) { return y * sigmoid_dev(y - 1.0f); }

// ---------------------------------------------------------------- row kernel
// Finishes the split-K GEMMs: s[n] = ((ws[0]+ws[1])+...)+ws[kz-1] in slab order, then the
// mode-specific tail.  Covers: LSTM projection + residual (ROW_HR), FFN-down / embed-linear +
// bias + residual + BasicNorm (ROW_NORM), encoder_proj / decoder_proj (ROW_BIAS_STORE) and the
// joiner's masked arg-max (ROW_ARGMAX, reference src/april_session.c:311-320,329).
template <int MODE>
__global__ __launch_bounds__(256) void row_kernel(RowArgs r)
{
    __shared__ float scratch[4];
    __shared__ float s_best[4];
    __shared__ int s_idx[4];
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    const int slot = r.slot_idx ? r.slot_idx[m] : m;

    // r.kz partial planes (1, 2, 4 or 8; each already a balanced-tree sum of consecutive K slabs): finish the tree
    auto slab_sum = [&](int n) { return tree_sum(r.ws, r.kz, r.m_stride, r.N, m, n); };

    if (MODE == ROW_HR) {
        for (int n = tid; n < r.N; n += 256) {
            const float s = slab_sum(n);
            r.state[(size_t)slot * r.ld_state + n] = s;
            r.out[(size_t)m * r.ldo + n] = r.resid[(size_t)m * r.ldr + n] + s;
        }
    } else if (MODE == ROW_NORM) {
        // N <= 8 * 256 (checked on the host): keep the row in registers between the two passes
        float y[8];
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = tid + i * 256;
            y[i] = 0.0f;
            if (n < r.N) {
                float v = slab_sum(n) + r.bias[n];
                if (r.resid) v = r.resid[(size_t)m * r.ldr + n] + v;
                y[i] = v;
                sq += v * v;
            }
        }
        const float total = block_sum_256(sq, scratch);
        const float scale = powf(total / (float)r.N + r.eps, -0.5f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = tid + i * 256;
            if (n < r.N) r.out[(size_t)m * r.ldo + n] = y[i] * scale;
        }
    } else if (MODE == ROW_BIAS_STORE) {
        for (int n = tid; n < r.N; n += 256) r.out[(size_t)slot * r.ldo + n] = slab_sum(n) + r.bias[n];
    } else {   // ROW_ARGMAX
        float best = -9999999999.0f;
        int best_i = -1;
        float blank_v = 0.0f;
        for (int n = tid; n < r.n_valid; n += 256) {
            const float v = slab_sum(n) + r.bias[n];
            if (r.logits_dump) r.logits_dump[(size_t)m * r.n_valid + n] = v;
            if (n == r.blank) blank_v = v;
            else if (v > best) { best = v; best_i = n; }
        }
        // lowest index wins on ties, as a sequential scan with '>' would
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(best_i, off);
            const bool take = (oi >= 0) && (best_i < 0 || ov > best || (ov == best && oi < best_i));
            if (take) { best = ov; best_i = oi; }
            blank_v += __shfl_xor(blank_v, off);     // exactly one thread holds a non-zero term
        }
        const int wave = tid >> 6;
        if ((tid & 63) == 0) { s_best[wave] = best; s_idx[wave] = best_i; scratch[wave] = blank_v; }
        __syncthreads();
        if (tid == 0) {
            float b = s_best[0]; int bi = s_idx[0];
            for (int w = 1; w < 4; ++w) {
                const float ov = s_best[w]; const int oi = s_idx[w];
                if (oi >= 0 && (bi < 0 || ov > b || (ov == b && oi < bi))) { b = ov; bi = oi; }
            }
            // the blank logit sits in exactly one wave; the others contributed +0.0f
            float bl = 0.0f;
            const int bw = (r.blank & 255) >> 6;
            bl = scratch[bw];
            r.joint[m].idx = bi;
            r.joint[m].max_val = b;
            r.joint[m].blank_val = bl;
        }
    }
}

void launch_row(const RowArgs &r, hipStream_t s)
{
    dim3 grid((unsigned)r.M), block(256);
    switch (r.mode) {
    case ROW_HR: hipLaunchKernelGGL(row_kernel<ROW_HR>, grid, block, 0, s, r); break;
    case ROW_NORM: hipLaunchKernelGGL(row_kernel<ROW_NORM>, grid, block, 0, s, r); break;
    case ROW_BIAS_STORE: hipLaunchKernelGGL(row_kernel<ROW_BIAS_STORE>, grid, block, 0, s, r); break;
    default: hipLaunchKernelGGL(row_kernel<ROW_ARGMAX>, grid, block, 0, s, r); break;
    }
}

// ---------------------------------------------------------------- conv front end
// Conv2d(1->c0,k3,s0) -> DoubleSwish -> Conv2d(c0->c1,k3,s1) -> DoubleSwish, then the receptive
// fields of the third convolution written out as GEMM rows (im2col): the third conv (72 % of the
// front end's FLOPs) runs on the MFMA GEMM with a fused bias+DoubleSwish epilogue.
//   grid = (sessions, channel groups of conv 2); each workgroup gathers the 9 x mel chunk from the
//   session's feature ring in HBM (coalesced 320-byte rows) into LDS, recomputes the cheap first
//   conv, computes its group of second-conv channels into LDS and writes, for every output
//   position ow of conv 3, the slice k = ci*9 + i*3 + j of row (session*W3 + ow):
//       A3[row][k] = act2[ci][i][ow*s2 + j]           (H2 == 3 == kernel height, so H3 == 1)
// Accumulation order per conv output: input channel, kernel row, kernel column; bias last.
__global__ __launch_bounds__(256) void conv12_kernel(ConvEmbedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m = blockIdx.x, grp = blockIdx.y;
    const int tid = threadIdx.x;
    const int H0 = a.seg, W0 = a.mel;
    const int s0 = a.stride[0], s1 = a.stride[1], s2 = a.stride[2];
    const int H1 = (H0 - 3) / s0 + 1, W1 = (W0 - 3) / s0 + 1;
    const int H2 = (H1 - 3) / s1 + 1, W2 = (W1 - 3) / s1 + 1;
    const int W3 = (W2 - 3) / s2 + 1;
    const int c0 = a.ch[0], cg = a.ch1_per_group;
    float *x = lds;                              // H0*W0
    float *a1 = x + H0 * W0;                     // c0*H1*W1
    float *a2 = a1 + c0 * H1 * W1;               // cg*H2*W2

    if (a.x_direct) {
        for (int i = tid; i < H0 * W0; i += 256) x[i] = a.x_direct[(size_t)m * H0 * W0 + i];
    } else {
        const int slot = a.slot_idx[m];
        const int tail = a.ring_tail[m];
        const float *ring = a.ring + (size_t)slot * a.ring_frames * W0;
        for (int i = tid; i < H0 * W0; i += 256) {
            const int row = i / W0, col = i % W0;
            x[i] = ring[(size_t)((tail + row) % a.ring_frames) * W0 + col];
        }
    }
    __syncthreads();
    for (int o = tid; o < c0 * H1 * W1; o += 256) {
        const int c = o / (H1 * W1), oh = (o / W1) % H1, ow = o % W1;
        const float *w = a.w[0] + c * 9;
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc += x[(oh * s0 + i) * W0 + ow * s0 + j] * w[i * 3 + j];
        acc += a.b[0][c];
        a1[o] = dswish_dev(acc);
    }
    __syncthreads();
    for (int o = tid; o < cg * H2 * W2; o += 256) {
        const int cl = o / (H2 * W2), oh = (o / W2) % H2, ow = o % W2;
        const int c = grp * cg + cl;
        const float *w = a.w[1] + (size_t)c * c0 * 9;
        float acc = 0.0f;
        for (int ci = 0; ci < c0; ++ci) {
            const float *src = a1 + ci * H1 * W1 + (oh * s1) * W1 + ow * s1;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc += src[i * W1 + j] * w[ci * 9 + i * 3 + j];
        }
        acc += a.b[1][c];
        a2[o] = dswish_dev(acc);
    }
    __syncthreads();
    // im2col slice of this channel group: consecutive threads write consecutive k (coalesced runs of cg*9 floats)
    const int kslice = cg * 9;
    for (int e = tid; e < W3 * kslice; e += 256) {
        const int ow = e / kslice, kl = e % kslice;
        const int cl = kl / 9, i = (kl % 9) / 3, j = kl % 3;
        a.out[((size_t)m * W3 + ow) * a.ldo + grp * kslice + kl] = a2[cl * H2 * W2 + i * W2 + ow * s2 + j];
    }
}

void launch_conv_embed(const ConvEmbedArgs &a, hipStream_t s)
{
    const int H1 = (a.seg - 3) / a.stride[0] + 1, W1 = (a.mel - 3) / a.stride[0] + 1;
    const int H2 = (H1 - 3) / a.stride[1] + 1, W2 = (W1 - 3) / a.stride[1] + 1;
    const size_t lds = sizeof(float) * ((size_t)a.seg * a.mel + (size_t)a.ch[0] * H1 * W1 + (size_t)a.ch1_per_group * H2 * W2);
    hipLaunchKernelGGL(conv12_kernel, dim3((unsigned)a.M, (unsigned)(a.ch[1] / a.ch1_per_group)), dim3(256), lds, s, a);
}

// ---------------------------------------------------------------- decoder front end
// Embedding gather of the `context` previous tokens, grouped Conv1d over the context axis
// (kernel = context, so one output position), ReLU.  Pure function of the token context
// (reference src/april_session.c:151-163,181-196).
__global__ __launch_bounds__(256) void dec_embed_kernel(DecEmbedArgs a)
{
    const int m = blockIdx.x;
    const int cg = a.d / a.groups;                 // input channels per group == output channels per group
    for (int o = threadIdx.x; o < a.d; o += 256) {
        const int g0 = (o / cg) * cg;
        const float *w = a.conv_w + (size_t)o * cg * a.context;
        float acc = 0.0f;
        for (int ci = 0; ci < cg; ++ci)
            for (int t = 0; t < a.context; ++t) {
                const int tok = a.ctx[m * a.context + t];
                acc += a.emb[(size_t)tok * a.d + g0 + ci] * w[ci * a.context + t];
            }
        if (a.conv_b) acc += a.conv_b[o];
        a.out[(size_t)m * a.ldo + o] = acc > 0.0f ? acc : 0.0f;
    }
}

void launch_dec_embed(const DecEmbedArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(dec_embed_kernel, dim3((unsigned)a.M), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------- fp16 weight copies
__global__ __launch_bounds__(256) void cvt_f16_kernel(const float *src, _Float16 *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (_Float16)src[i];
}

void launch_cvt_f16(const float *src, void *dst, size_t n, hipStream_t s)
{
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(cvt_f16_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, src, reinterpret_cast<_Float16 *>(dst), n);
}

}  // namespace aprilx
