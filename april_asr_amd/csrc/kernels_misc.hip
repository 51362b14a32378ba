// Row-wise epilogues, the encoder's convolutional front end and the decoder's
// embedding/conv front end (gfx950).  All bandwidth/latency-type work: one
// workgroup per session row, coalesced row accesses, LDS for the conv stack's
// intermediates, wavefront shuffles for the row reductions.  Reduction orders are
// fixed (independent of the batch).
#include "kernels.h"
#include "device_utils.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace aprilx {

using h4 = __attribute__((ext_vector_type(4))) _Float16;
__device__ __forceinline__ float sigmoid_dev(float x) { return fast_sigmoid(x); }
__device__ __forceinline__ float dswish_dev(float y) { return y * sigmoid_dev(y - 1.0f); }

// ---------------------------------------------------------------- row kernels
// Small-batch path: finish the split-K GEMMs.  s[n] = balanced tree over the partial planes, then the mode-specific
// tail -- the same arithmetic, in the same order, as the GEMM's fused row epilogues (EPI_HR / EPI_RESID_SSQ /
// EPI_SLOT_STORE), so a session computes the same bits whichever schedule its batch size selects.
// Thread t owns the 4-column quad t, t + 256, ... of the row.
template <int MODE>
__device__ __forceinline__ void row_body(const RowArgs &r)
{
    if (r.run_flag && *r.run_flag != r.run_gen) return;
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    const int slot = r.slot_idx ? r.slot_idx[m] : m;
    if (MODE == ROW_SLOT_STORE) { if (r.row_mask && !r.row_mask[m]) return; }
    float rs = 1.0f;
    const bool scaled = MODE == ROW_HR || (MODE == ROW_SLOT_STORE && r.r_scale.ssq != nullptr);
    if (scaled) rs = row_scale(r.r_scale, m);
    const int nq = r.N >> 2;                            // N is a multiple of 64
    for (int q0 = 0; q0 < nq; q0 += 256) {
        const int q = q0 + tid;
        const bool ok = q < nq;
        const int n = q * 4;
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) s = tree_sum4(r.ws, r.parts, r.m_stride, r.N, m, n);
        if (MODE == ROW_HR) {
            if (ok) {
                const f32x4 y = *reinterpret_cast<const f32x4 *>(r.resid + (size_t)m * r.ldr + n);
                const f32x4 o = y * rs + s;
                *reinterpret_cast<f32x4 *>(r.state + (size_t)slot * r.ld_state + n) = s;
                *reinterpret_cast<f32x4 *>(r.out + (size_t)m * r.ldo + n) = o;
                if (r.state16) *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(r.state16) + (size_t)slot * r.ld_state + n) = h4{(_Float16)s.x, (_Float16)s.y, (_Float16)s.z, (_Float16)s.w};
                if (r.out16) *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(r.out16) + (size_t)m * r.ldo + n) = h4{(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
            }
        } else if (MODE == ROW_RESID_SSQ) {
            f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                y = s + *reinterpret_cast<const f32x4 *>(r.bias + n);
                if (r.resid) y = *reinterpret_cast<const f32x4 *>(r.resid + (size_t)m * r.ldr + n) + y;
                *reinterpret_cast<f32x4 *>(r.out + (size_t)m * r.ldo + n) = y;
                if (r.out16) *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(r.out16) + (size_t)m * r.ldo + n) = h4{(_Float16)y.x, (_Float16)y.y, (_Float16)y.z, (_Float16)y.w};
            }
            const float ss = granule_ssq(y);           // every lane takes part in the shuffles
            if (ok && (q & 7) == 0) r.ssq_out[(size_t)m * (r.N / SSQ_COLS) + n / SSQ_COLS] = ss;
        } else {   // ROW_SLOT_STORE
            if (ok) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(r.bias + n);
                *reinterpret_cast<f32x4 *>(r.out + (size_t)slot * r.ldo + n) = scaled ? s * rs + b : s + b;
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void row_kernel(RowArgs r) { row_body<MODE>(r); }

// the same rows for several problems of one shape in one launch (the layers active in a feed wavefront): blockIdx.y picks the
// argument block, read with scalar loads through the read-only kernel argument
template <int MODE>
__global__ __launch_bounds__(256) void row_zkernel(const RowArgs *__restrict__ zargs)
{
    const RowArgs r = zargs[blockIdx.y];
    row_body<MODE>(r);
}

void launch_row(const RowArgs &r, hipStream_t s)
{
    dim3 grid((unsigned)r.M), block(256);
    switch (r.mode) {
    case ROW_HR: hipLaunchKernelGGL(row_kernel<ROW_HR>, grid, block, 0, s, r); break;
    case ROW_RESID_SSQ: hipLaunchKernelGGL(row_kernel<ROW_RESID_SSQ>, grid, block, 0, s, r); break;
    default: hipLaunchKernelGGL(row_kernel<ROW_SLOT_STORE>, grid, block, 0, s, r); break;
    }
}

void launch_row_z(const RowArgs *host_args, int n, const RowArgs *dev_args, hipStream_t s)
{
    if (n <= 0) return;
    const RowArgs &r = host_args[0];
    for (int i = 1; i < n; ++i)
        if (host_args[i].mode != r.mode || host_args[i].M != r.M || host_args[i].N != r.N || host_args[i].parts != r.parts) {
            fprintf(stderr, "libapril(mi355x): launch_row_z: the problems of one launch must have one shape\n"); abort();
        }
    dim3 grid((unsigned)r.M, (unsigned)n), block(256);
    switch (r.mode) {
    case ROW_HR: hipLaunchKernelGGL(row_zkernel<ROW_HR>, grid, block, 0, s, dev_args); break;
    case ROW_RESID_SSQ: hipLaunchKernelGGL(row_zkernel<ROW_RESID_SSQ>, grid, block, 0, s, dev_args); break;
    default: hipLaunchKernelGGL(row_zkernel<ROW_SLOT_STORE>, grid, block, 0, s, dev_args); break;
    }
}

// ---------------------------------------------------------------- decoder front end (device function)
// Embedding gather of the `context` previous tokens, grouped Conv1d over the context axis
// (kernel = context, so one output position), ReLU.  Pure function of the token context
// (reference src/april_session.c:151-163,181-196).  Called by all 256 threads of a workgroup for one row.
__device__ __forceinline__ void dec_embed_row(const DecEmbedParams &p, int tok0, int tok1, float *out)
{
    const int cg = p.d / p.groups;                 // input channels per group == output channels per group
    for (int o = threadIdx.x; o < p.d; o += 256) {
        const int g0 = (o / cg) * cg;
        const float *w = p.conv_w + (size_t)o * cg * p.context;
        float acc = 0.0f;
        for (int ci = 0; ci < cg; ++ci) {
            acc += p.emb[(size_t)tok0 * p.d + g0 + ci] * w[ci * p.context + 0];
            acc += p.emb[(size_t)tok1 * p.d + g0 + ci] * w[ci * p.context + 1];
        }
        if (p.conv_b) acc += p.conv_b[o];
        out[o] = acc > 0.0f ? acc : 0.0f;
    }
}

// ---------------------------------------------------------------- joiner decision
// logits = tree(ws) + bias; masked arg-max (reference src/april_session.c:311-320: first maximum wins, blank excluded);
// then the part of aas_process_logits (:322-429) that the NEXT network call depends on: blank or not, context push,
// the >= 2200 ms silence reset of the context, and the class of the last active token (digit-dot rule).  Everything the
// callbacks need (token text, active list, de-duplication) stays on the host, which replays the same decisions from the
// 16-byte record written here.
__global__ __launch_bounds__(256) void decide_kernel(DecideArgs a)
{
    __shared__ float s_best[4], s_blank[4];
    __shared__ int s_idx[4];
    __shared__ int s_ctx[3];                        // [0..1] context for the decoder front end, [2] re-run flag
    if (a.round > 0 && a.run_flags && a.run_flags[a.round] != a.gen) return;      // every row resolved in an earlier round
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    StepRecord *rec = a.rec ? a.rec + m : a.rec_ring + (size_t)a.rec_off[0] + (size_t)a.rec_slot * a.M + m;
    if (a.round > 0 && a.active[m] != a.gen) {      // uniform per workgroup
        if (tid == 0) { rec->idx = -1; rec->max_val = 0.0f; rec->blank_val = 0.0f; rec->flags = 0; a.dirty[m] = 0; }
        return;
    }
    float best = -9999999999.0f;
    int best_i = -1;
    float blank_v = 0.0f;
    for (int n = tid; n < a.n_valid; n += 256) {
        const float v = tree_sum(a.ws, a.parts, a.m_stride, a.N, m, n) + a.bias[n];
        if (a.logits_dump) a.logits_dump[(size_t)m * a.n_valid + n] = v;
        if (n == a.blank) blank_v = v;
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
    if ((tid & 63) == 0) { s_best[wave] = best; s_idx[wave] = best_i; s_blank[wave] = blank_v; }
    __syncthreads();
    if (tid == 0) {
        float b = s_best[0]; int bi = s_idx[0];
        for (int w = 1; w < 4; ++w) {
            const float ov = s_best[w]; const int oi = s_idx[w];
            if (oi >= 0 && (bi < 0 || ov > b || (ov == b && oi < bi))) { b = ov; bi = oi; }
        }
        // the blank logit sits in exactly one wave; the others contributed +0.0f
        const float bl = s_blank[(a.blank & 255) >> 6];
        rec->idx = bi; rec->max_val = b; rec->blank_val = bl;

        // ---- decision (Greedy::on_joint on the host replays exactly this from the record)
        const int slot = a.slot_idx[m];
        GreedyState st = a.state[slot];
        int tok = bi; float tv = b;
        if (tok < 0) { tok = a.blank == 0 ? 1 : 0; tv = -9999999999.0f; }        // no logit beat the initial value (NaNs)
        const bool cleared = st.ctx1 == a.blank;                                  // :322
        const bool same = st.ctx1 == tok;                                         // :326
        const float ee = same ? 0.0f : a.early_emit;
        bool is_blank = (bl - ee) > tv;                                           // :329-330
        const unsigned tc = a.tok_class[tok];
        bool punct = (tc & (TKC_SENT_END | TKC_COMMA)) != 0;
        if (punct && st.last_tok >= 0 && (a.tok_class[st.last_tok] & TKC_DIGIT_START) && (tc & TKC_DOT)) punct = false;   // :345-351
        if (!cleared && punct && !same && tv > (bl - 3.5f)) is_blank = false;     // :356-358
        const unsigned now = (unsigned)a.now_ms[m];
        unsigned flags = REC_VALID;
        bool rerun = false;
        if (!is_blank) {                                                          // :361-400
            st.last_emit_ms = now;
            st.ctx0 = st.ctx1; st.ctx1 = tok;
            st.last_tok = tok;
            rerun = true;
        } else {                                                                  // :401-426
            flags |= REC_BLANK;
            if (now - st.last_emit_ms >= 2200u) {                                 // FINAL, clear context, SILENCE
                st.last_tok = -1;
                if (st.ctx0 != a.blank) { st.ctx0 = a.blank; st.ctx1 = a.blank; rerun = true; }   // :296-301
            }
        }
        a.active[m] = is_blank ? 0 : a.gen;
        if (rerun) flags |= REC_CTX;
        rec->flags = flags;
        a.state[slot] = st;
        a.dirty[m] = rerun ? 1 : 0;
        if (a.run_flags) {
            if (!is_blank && a.round < 2) a.run_flags[a.round + 1] = a.gen;   // plain stores of the same value: no atomics needed
            if (rerun) a.rerun_flags[a.round] = a.gen;
        }
        s_ctx[0] = st.ctx0; s_ctx[1] = st.ctx1; s_ctx[2] = rerun ? 1 : 0;
    }
    __syncthreads();
    if (s_ctx[2] && a.de_out) dec_embed_row(a.dec, s_ctx[0], s_ctx[1], a.de_out + (size_t)m * a.ld_de);     // (no table: the decoder runs for this row)
}

void launch_decide(const DecideArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(decide_kernel, dim3((unsigned)a.M), dim3(256), 0, s, a);
}

// decoder front end for listed slots with the context held on the device (first use of a session, end of a flush)
__global__ __launch_bounds__(256) void dec_rows_kernel(DecRowsArgs a)
{
    __shared__ int s_ctx[2];
    const int m = blockIdx.x;
    if (threadIdx.x == 0) {
        const int slot = a.slot_idx[m];
        GreedyState st = a.state[slot];
        if (a.op == 1) {                            // src/april_session.c:561-563 after the host's FINAL: forget, clear
            st.last_tok = -1;
            if (st.ctx0 != a.blank) { st.ctx0 = a.blank; st.ctx1 = a.blank; }
            a.state[slot] = st;
        }
        s_ctx[0] = st.ctx0; s_ctx[1] = st.ctx1;
    }
    __syncthreads();
    if (a.de_out) dec_embed_row(a.dec, s_ctx[0], s_ctx[1], a.de_out + (size_t)m * a.ld_de);
}

void launch_dec_rows(const DecRowsArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(dec_rows_kernel, dim3((unsigned)a.M), dim3(256), 0, s, a);
}

__global__ __launch_bounds__(256) void dec_embed_kernel(DecEmbedArgs a)
{
    const int m = blockIdx.x;
    dec_embed_row(a.dec, a.ctx[m * a.dec.context + 0], a.ctx[m * a.dec.context + 1], a.out + (size_t)m * a.ldo);
}

void launch_dec_embed(const DecEmbedArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(dec_embed_kernel, dim3((unsigned)a.M), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------- step bookkeeping
__global__ __launch_bounds__(1024) void advance_kernel(AdvanceArgs a)
{
    __shared__ int s_k;
    if (threadIdx.x == 0) s_k = a.counter[0];
    __syncthreads();
    const int kc = s_k, k = kc & a.index_mask;
    const int *src = a.host_ring + a.host_step_off[k];          // pinned host memory, read once per step
    for (int arr = 0; arr < a.n_arrays; ++arr) {
        for (int i = threadIdx.x; i < a.len[arr]; i += 1024) a.dst[arr * a.dst_stride + i] = src[i];
        src += a.len[arr];
    }
    if (threadIdx.x < a.n_flags) a.flags[threadIdx.x] = 0;
    if (threadIdx.x == 0) { a.rec_off[0] = a.host_rec_off[k]; a.counter[0] = (kc + 1) & 0x7fffffff; }
}

void launch_advance(const AdvanceArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1024), 0, s, a);
}

__global__ __launch_bounds__(256) void zero_slot_kernel(ZeroSlotArgs a)
{
    const int l = blockIdx.x;                        // one workgroup per layer, the last one also resets the per-slot rows
    const size_t s = (size_t)(a.n_list > 0 ? a.list[blockIdx.y] : a.slot);
    if (l < a.n_layers) {
        float *h = a.h + ((size_t)l * a.slots + s) * a.d_model, *c = a.c + ((size_t)l * a.slots + s) * a.hidden;
        for (int i = threadIdx.x; i < a.d_model; i += 256) h[i] = 0.0f;
        if (a.h16) { _Float16 *hh = reinterpret_cast<_Float16 *>(a.h16) + ((size_t)l * a.slots + s) * a.d_model; for (int i = threadIdx.x; i < a.d_model; i += 256) hh[i] = (_Float16)0.0f; }
        for (int i = threadIdx.x; i < a.hidden; i += 256) c[i] = 0.0f;
    } else {
        for (int i = threadIdx.x; i < a.joiner; i += 256) { a.eout[s * a.joiner + i] = 0.0f; a.dout[s * a.joiner + i] = 0.0f; }
        if (threadIdx.x == 0) { GreedyState st; st.ctx0 = a.blank; st.ctx1 = a.blank; st.last_tok = -1; st.last_emit_ms = 0; a.state[s] = st; }
    }
}

void launch_zero_slot(const ZeroSlotArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(zero_slot_kernel, dim3((unsigned)a.n_layers + 1, (unsigned)std::max(1, a.n_list)), dim3(256), 0, s, a);
}

__global__ __launch_bounds__(256) void block_setup_kernel(BlockSetupArgs a)
{
    for (int i = threadIdx.x; i < a.count; i += 256) { a.now_dst[i] = a.now_src[i]; a.rows_dst[i] = a.row0 + i; }
    if ((int)threadIdx.x < a.n_flags) a.flags[threadIdx.x] = 0;
    if (threadIdx.x == 0) a.rec_off_block[0] = a.rec_off_step[0] + a.rec_add;
}

void launch_block_setup(const BlockSetupArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(block_setup_kernel, dim3(1), dim3(256), 0, s, a);
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
    // Register blocking: a thread owns an output POSITION and computes several output channels there from one set of
    // input-patch registers; the weights of a channel are wave-uniform addresses (scalar loads), so the only LDS traffic
    // is the patch itself (the previous one-output-per-thread form was bound by LDS reads: 2 per multiply-add).
    // Accumulation order per output: input channel, kernel row, kernel column; bias last.
    for (int p = tid; p < H1 * W1; p += 256) {
        const int oh = p / W1, ow = p % W1;
        float patch[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) patch[i * 3 + j] = x[(oh * s0 + i) * W0 + ow * s0 + j];
        for (int c = 0; c < c0; ++c) {
            const float *w = a.w[0] + c * 9;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc += patch[k] * w[k];
            acc += a.b[0][c];
            a1[c * H1 * W1 + p] = dswish_dev(acc);
        }
    }
    __syncthreads();
    {
        const int NP = H2 * W2;
        const int per = ((NP + 63) / 64) * 64;                     // positions rounded up to whole waves
        int nsub = per <= 256 ? 256 / per : 1;                     // channel subsets handled side by side
        while (nsub > 1 && (cg % nsub) != 0) --nsub;
        const int cps = cg / nsub;                                 // channels per thread (<= 8 by construction of the launch)
        const int stride = per <= 256 ? per : 256;
        const int sub = per <= 256 ? __builtin_amdgcn_readfirstlane(tid / per) : 0;     // uniform: `per` is a multiple of the wave size
        for (int base = 0; base < NP; base += stride) {
            const int p = base + (per <= 256 ? tid % per : tid);
            if (sub < nsub && p < NP) {
                const int oh = p / W2, ow = p % W2;
                float acc[8];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) acc[cc] = 0.0f;
                for (int ci = 0; ci < c0; ++ci) {
                    const float *src = a1 + ci * H1 * W1 + (oh * s1) * W1 + ow * s1;
                    float patch[9];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) patch[i * 3 + j] = src[i * W1 + j];
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc)
                        if (cc < cps) {
                            const float *w = a.w[1] + ((size_t)(grp * cg + sub * cps + cc) * c0 + ci) * 9;
#pragma unroll
                            for (int k = 0; k < 9; ++k) acc[cc] += patch[k] * w[k];
                        }
                }
#pragma unroll
                for (int cc = 0; cc < 8; ++cc)
                    if (cc < cps) {
                        const int cl = sub * cps + cc;
                        a2[cl * NP + p] = dswish_dev(acc[cc] + a.b[1][grp * cg + cl]);
                    }
            }
        }
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

// The same front end for MANY chunks (round 6): one workgroup per chunk computes ALL second-conv channels -- the first conv once instead
// of once per channel group (a quarter of the old form's multiplies were that recomputation), CPS channels of a position per thread with
// the weights transposed once at load ([ci][k][channel]: a position's channels sit side by side at wave-uniform addresses, i.e. scalar
// loads, no LDS traffic for weights) so that two channels share one
// packed multiply and one packed add (v_pk_mul_f32 / v_pk_add_f32: the build keeps multiply and add apart, -ffp-contract=off, as the
// oracle's C does).  Every output is the same chain as in conv12_kernel -- input channel, kernel row, kernel column, bias last -- so the
// two forms agree bit for bit (tests/test_gpu_golden.py runs both).  At 2048 sessions the old form took 239 us per 100 ms step.
using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr int pad4(int n) { return (n + 3) & ~3; }
// (channel counts and the subset count are compile-time: the index arithmetic of the im2col rows and the channel loops is then
// multiplications and shifts -- with run-time divisors it was a third of the kernel's instructions)
template <int C0, int C1, int NSUB>
__global__ __launch_bounds__(256) void conv12_wide_kernel(ConvEmbedArgs a)
{
    constexpr int CPS = C1 / NSUB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    const int H0 = a.seg, W0 = a.mel;
    const int s0 = a.stride[0], s1 = a.stride[1], s2 = a.stride[2];
    const int H1 = (H0 - 3) / s0 + 1, W1 = (W0 - 3) / s0 + 1;
    const int H2 = (H1 - 3) / s1 + 1, W2 = (W1 - 3) / s1 + 1;
    const int W3 = (W2 - 3) / s2 + 1;
    constexpr int c0 = C0, c1 = C1, WS = C1;
    const int NP = H2 * W2;
    float *x = lds;                              // H0*W0
    float *a1 = x + pad4(H0 * W0);               // c0*H1*W1
    float *a2 = a1 + pad4(c0 * H1 * W1);         // c1*NP
    // the weights come transposed from the engine (launch_conv_weight_transpose): w1t[ci * 9 + k][c1], w0t[k][c0] -- a position's channels
    // are consecutive, their addresses uniform over the wave: scalar loads, the products take them straight from SGPR pairs
    const float *__restrict__ wl = a.w1t, *__restrict__ w0l = a.w0t;

    if (a.x_direct) {
        for (int i = tid; i < H0 * W0; i += 256) x[i] = a.x_direct[(size_t)m * H0 * W0 + i];
    } else {
        // eight rows x 32-column strides per pass: no divisions (the ring row wraps at most once: tail < ring_frames, H0 <= ring_frames)
        const int slot = a.slot_idx[m];
        const int tail = a.ring_tail[m];
        const float *ring = a.ring + (size_t)slot * a.ring_frames * W0;
        for (int row = tid >> 5; row < H0; row += 8) {
            int rr = tail + row;
            if (rr >= a.ring_frames) rr -= a.ring_frames;
            for (int col = tid & 31; col < W0; col += 32) x[row * W0 + col] = ring[(size_t)rr * W0 + col];
        }
    }
    __syncthreads();
    for (int p = tid; p < H1 * W1; p += 256) {
        const int oh = p / W1, ow = p % W1;
        float patch[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) patch[i * 3 + j] = x[(oh * s0 + i) * W0 + ow * s0 + j];
        for (int cq = 0; cq < c0; cq += 4) {                               // four channels at a time (c0 % 4 == 0, checked on the host)
            f32x2 lo = f32x2{0.0f, 0.0f}, hi = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(w0l + k * c0 + cq);
                const f32x2 pk = f32x2{patch[k], patch[k]};
                lo = lo + pk * f32x2{w.x, w.y};
                hi = hi + pk * f32x2{w.z, w.w};
            }
            a1[(cq + 0) * H1 * W1 + p] = dswish_dev(lo.x + a.b[0][cq + 0]);
            a1[(cq + 1) * H1 * W1 + p] = dswish_dev(lo.y + a.b[0][cq + 1]);
            a1[(cq + 2) * H1 * W1 + p] = dswish_dev(hi.x + a.b[0][cq + 2]);
            a1[(cq + 3) * H1 * W1 + p] = dswish_dev(hi.y + a.b[0][cq + 3]);
        }
    }
    __syncthreads();
    for (int kl = tid; kl < c1 * 9; kl += 256) reinterpret_cast<int *>(x)[kl] = (kl / 9) * NP + ((kl % 9) / 3) * W2 + kl % 3;      // (x is free now; read behind the next barrier)
    {
        // threads = (channel subset, position): `per` positions (whole waves) x 256 / per subsets of CPS channels (checked on the host)
        constexpr int per = 256 / NSUB;                                    // (>= NP, whole waves: checked on the host)
        const int sub = __builtin_amdgcn_readfirstlane(tid / per);
        const int pq = tid % per, pc = pq < NP ? pq : NP - 1;              // (idle lanes recompute the last position; never stored)
        const int oh = pc / W2, ow = pc % W2;
        const int cbase = sub * CPS;
        f32x2 acc[CPS / 2];
#pragma unroll
        for (int q = 0; q < CPS / 2; ++q) acc[q] = f32x2{0.0f, 0.0f};
        for (int ci = 0; ci < c0; ++ci) {
            const float *src = a1 + ci * H1 * W1 + (oh * s1) * W1 + ow * s1;
            float patch[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) patch[i * 3 + j] = src[i * W1 + j];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const f32x4 *wq = reinterpret_cast<const f32x4 *>(wl + (ci * 9 + k) * WS + cbase);
                const f32x2 pk = f32x2{patch[k], patch[k]};
#pragma unroll
                for (int q = 0; q < CPS / 4; ++q) {
                    const f32x4 w = wq[q];
                    acc[2 * q] = acc[2 * q] + pk * f32x2{w.x, w.y};
                    acc[2 * q + 1] = acc[2 * q + 1] + pk * f32x2{w.z, w.w};
                }
            }
        }
        if (pq < NP) {
#pragma unroll
            for (int q = 0; q < CPS / 2; ++q) {
                const int cl = cbase + 2 * q;
                a2[cl * NP + pq] = dswish_dev(acc[q].x + a.b[1][cl]);
                a2[(cl + 1) * NP + pq] = dswish_dev(acc[q].y + a.b[1][cl + 1]);
            }
        }
    }
    __syncthreads();
    // im2col rows of the third conv, four consecutive k per thread and store (rows of c1 * 9 floats, 16-byte aligned: ldo % 4 == 0 is
    // checked on the host): where k = cl * 9 + i * 3 + j reads from is looked up in a table built once per workgroup (x's space, free
    // since the first conv) -- decoding k per element was a third of this kernel's instructions
    constexpr int krow = c1 * 9, qrow = krow / 4;
    const int *koff = reinterpret_cast<const int *>(x);
    for (int e = tid; e < W3 * qrow; e += 256) {
        const int ow = e / qrow, q = e - ow * qrow;
        const int4 o = *reinterpret_cast<const int4 *>(koff + 4 * q);
        const int sh = ow * s2;
        const f32x4 v = {a2[o.x + sh], a2[o.y + sh], a2[o.z + sh], a2[o.w + sh]};
        *reinterpret_cast<f32x4 *>(a.out + (size_t)((unsigned)(m * W3 + ow) * (unsigned)a.ldo + 4u * (unsigned)q)) = v;
    }
}

// w1 [c1][c0][9] -> w1t [c0 * 9][c1], w0 [c0][9] -> w0t [9][c0] (once, at load)
__global__ __launch_bounds__(256) void conv_weight_transpose_kernel(const float *w0, const float *w1, int c0, int c1, float *w0t, float *w1t)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < c1 * c0 * 9; i += gridDim.x * 256) w1t[(i % (c0 * 9)) * c1 + i / (c0 * 9)] = w1[i];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < c0 * 9; i += gridDim.x * 256) w0t[(i % 9) * c0 + i / 9] = w0[i];
}
void launch_conv_weight_transpose(const float *w0, const float *w1, int c0, int c1, float *w0t, float *w1t, hipStream_t s)
{
    hipLaunchKernelGGL(conv_weight_transpose_kernel, dim3(8), dim3(256), 0, s, w0, w1, c0, c1, w0t, w1t);
}

// the wide form where it exists, from APRIL_CONV_WIDE_MIN (default 48; -1 = never) chunks per launch (below that the channel groups of
// conv12_kernel are what fills the chip)
static bool conv_wide_ok(const ConvEmbedArgs &a)
{
    static const int min_chunks = [] { const char *e = getenv("APRIL_CONV_WIDE_MIN"); return e && *e ? atoi(e) : 48; }();
    if (min_chunks < 0 || a.M < min_chunks || !a.w1t || !a.w0t) return false;
    const int H1 = (a.seg - 3) / a.stride[0] + 1, W1 = (a.mel - 3) / a.stride[0] + 1;
    const int H2 = (H1 - 3) / a.stride[1] + 1, W2 = (W1 - 3) / a.stride[1] + 1;
    const int NP = H2 * W2;
    const size_t lds = sizeof(float) * ((size_t)pad4(a.seg * a.mel) + pad4(a.ch[0] * H1 * W1) + pad4(a.ch[1] * NP));
    // the instantiated shape: 8 -> 32 channels, the second conv's positions in two waves (aprilv0 and the larger encoder: 80 mel bins, 9 frames)
    return H2 == 3 && a.ch[0] == 8 && a.ch[1] == 32 && NP > 64 && NP <= 128 && a.seg <= a.ring_frames && a.ldo % 4 == 0 && a.seg * a.mel >= a.ch[1] * 9 && lds <= 64 * 1024;
}

void launch_conv_embed(const ConvEmbedArgs &a, hipStream_t s)
{
    if (conv_wide_ok(a)) {
        const int H1 = (a.seg - 3) / a.stride[0] + 1, W1 = (a.mel - 3) / a.stride[0] + 1;
        const int H2 = (H1 - 3) / a.stride[1] + 1, W2 = (W1 - 3) / a.stride[1] + 1;
        const size_t lds = sizeof(float) * ((size_t)pad4(a.seg * a.mel) + pad4(a.ch[0] * H1 * W1) + pad4(a.ch[1] * H2 * W2));
        hipLaunchKernelGGL((conv12_wide_kernel<8, 32, 2>), dim3((unsigned)a.M), dim3(256), lds, s, a);
        return;
    }
    const int H1 = (a.seg - 3) / a.stride[0] + 1, W1 = (a.mel - 3) / a.stride[0] + 1;
    const int H2 = (H1 - 3) / a.stride[1] + 1, W2 = (W1 - 3) / a.stride[1] + 1;
    const size_t lds = sizeof(float) * ((size_t)a.seg * a.mel + (size_t)a.ch[0] * H1 * W1 + (size_t)a.ch1_per_group * H2 * W2);
    if (a.ch1_per_group > 8 || a.ch[1] % a.ch1_per_group) { fprintf(stderr, "libapril(mi355x): conv front end: channels per workgroup must divide the second conv and be <= 8\n"); abort(); }
    hipLaunchKernelGGL(conv12_kernel, dim3((unsigned)a.M, (unsigned)(a.ch[1] / a.ch1_per_group)), dim3(256), lds, s, a);
}

// ---------------------------------------------------------------- fp16 weight copies
__global__ __launch_bounds__(256) void cvt_f16_kernel(const float *src, _Float16 *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (_Float16)src[i];
}

__global__ __launch_bounds__(256) void prefetch_kernel(const PrefetchItem *__restrict__ items)
{
    const PrefetchItem it = items[blockIdx.y];
    const unsigned long long lines = it.bytes >> 7;
    const char *p = reinterpret_cast<const char *>(it.ptr);
    unsigned acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < lines; i += (unsigned long long)gridDim.x * 256)
        acc ^= *reinterpret_cast<const volatile unsigned *>(p + (i << 7));
    asm volatile("" :: "v"(acc));
}

void launch_prefetch(const PrefetchItem *dev_items, int n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(prefetch_kernel, dim3(48, (unsigned)n), dim3(256), 0, s, dev_items);
}

void launch_cvt_f16(const float *src, void *dst, size_t n, hipStream_t s)
{
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(cvt_f16_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, src, reinterpret_cast<_Float16 *>(dst), n);
}

// one thread per output element: (ntile, 32-k block, lane, j) of the x32 B-fragment order <- the fp32 16-k-block pack
__global__ __launch_bounds__(256) void repack_x32_kernel(const float *src, _Float16 *dst, int K, int N)
{
    const size_t total = (size_t)K * N;
    const int KB32 = K / 32, KB16 = K / 16;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int j = (int)(o & 7), lane = (int)((o >> 3) & 63);
        const size_t blk = o >> 9;                           // ntile * KB32 + kb
        const int kb = (int)(blk % KB32), ntile = (int)(blk / KB32);
        const int k = kb * 32 + (lane >> 4) * 8 + j, n16 = lane & 15;
        // fp32 pack: Wp[((ntile * KB16 + k / 16) * 64 + ((k % 16) / 4) * 16 + n16) * 4 + k % 4]
        dst[o] = (_Float16)src[(((size_t)ntile * KB16 + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + n16) * 4 + (k & 3)];
    }
}

void launch_repack_x32(const float *src_packed, void *dst, int K, int N, hipStream_t s)
{
    const size_t n = (size_t)K * N;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 65535);
    hipLaunchKernelGGL(repack_x32_kernel, dim3(blocks), dim3(256), 0, s, src_packed, reinterpret_cast<_Float16 *>(dst), K, N);
}

}  // namespace aprilx
