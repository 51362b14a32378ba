// Device kernel interface (gfx950 only).  Launch wrappers are implemented in the
// kernels_*.hip files; engine.cc calls them on its stream.
//
// Every floating-point reduction below has a FIXED order that does not depend
// on the batch size or on the position of a row inside the batch, so a session
// stepped alone produces bit-identical state and logits to the same session
// stepped inside a batch of thousands (SURVEY.md section 7 "determinism rule").
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace aprilx {

// ---------------------------------------------------------------- GEMM
// out[M,N] = A[M,K] x W[K,N]  with W pre-packed for v_mfma_f32_16x16x4_f32:
//   Wp[((ntile*KB + kb)*64 + lane)*4 + j] = W[kb*16 + (lane>>4)*4 + j][ntile*16 + (lane&15)]
// so one 16-byte load per lane feeds four MFMA k-steps.  Canonical summation per
// output element: K is cut into `kz` slabs (grid.z); inside a slab the four waves of
// the workgroup each own a contiguous quarter and run one in-order fp32 FMA chain
// (that is what the MFMA does); the four chains are added ((p0+p1)+p2)+p3.  With
// kz > 1 slab results are combined PAIRWISE in slab order (balanced tree): a workgroup may
// own 1, 2, 4 or all slabs of its tile (more as the batch grows and tiles alone fill the
// chip); what is left goes to a workspace and the row kernel that follows finishes the same tree.
enum GemmEpilogue {
    EPI_PARTIAL = 0,      // ws[z][m][n] = slab sum                         (consumer: row kernels)
    EPI_LSTM = 1,         // columns are unit-major (unit*4 + gate i,f,g,o): cell update, c in place, u out
    EPI_BIAS_DSWISH = 2   // out = y * sigmoid(y - 1), y = acc + bias
};
enum GemmAOp { AOP_NONE = 0, AOP_TANH_ADD = 1 };   // A element = tanh(a0[row] + a0b[row])  (joiner)

struct GemmArgs {
    // A operand as up to two K segments with optional row indirection (slot ids)
    const float *a0 = nullptr; int lda0 = 0; const int *aidx0 = nullptr; int K0 = 0;
    const float *a1 = nullptr; int lda1 = 0; const int *aidx1 = nullptr; int K1 = 0;
    const float *a0b = nullptr;            // AOP_TANH_ADD second addend (same ld/idx as a0)
    int a_op = AOP_NONE;
    const void *wp = nullptr;              // packed weights (fp32, or fp16 in the same element order when wt == 1)
    int wt = 0;                            // 0: fp32 operands, v_mfma_f32_16x16x4_f32; 1: fp16 weights, A rounded to fp16 on load,
                                           //    v_mfma_f32_16x16x16_f16 with fp32 accumulation (BASELINE configs[4])
    int M = 0, N = 0, K = 0;               // N multiple of 16, K multiple of 16
    int kz = 1;                            // K slabs (power of two); canonical summation unit
    int zs = 1;                            // slabs handled per workgroup (set by launch_gemm)
    int epi = EPI_PARTIAL;
    float *out = nullptr; int ldo = 0;     // EPI_PARTIAL: workspace [kz][m_stride][N]; others: [M][ldo]
    int m_stride = 0;
    const float *bias = nullptr;
    float *c_state = nullptr;              // EPI_LSTM: [slots][hidden] for this layer
    const int *slot_idx = nullptr;         // EPI_LSTM: row -> slot
    int hidden = 0;
    int skew = 0;                          // start delay (x 4096 cycles) for every second generation of workgroups
    int asm_loop = 0;                      // != 0: hand-scheduled K loop (gemm_mainloop_asm.inc) in the fused-epilogue 64x64 fp32 tiles
    unsigned long long *trace = nullptr;   // measurement only: per-workgroup s_memtime stamps [wg][8] (wave 0, lane 0)
    int debug = 0;                         // measurement only: 1 = skip the MFMA main loop, 2 = skip the epilogue math, 3 = all k blocks read block 0
};
void launch_gemm(const GemmArgs &g, hipStream_t s);
// number of partial planes launch_gemm will write for an EPI_PARTIAL GEMM of this shape (kz / slabs-per-workgroup)
int gemm_partials(int M, int N, int kz);

// ---------------------------------------------------------------- row kernels (one workgroup per row)
enum RowMode {
    ROW_HR = 0,          // s = sum_z ws; h_state[slot] = s; out = resid + s
    ROW_NORM = 1,        // y = resid + (s + bias); out = y * (mean(y^2) + eps)^-0.5   (resid optional)
    ROW_BIAS_STORE = 2,  // out[slot] = s + bias
    ROW_ARGMAX = 3       // logits = s + bias; arg-max over n != blank (lowest index wins), blank logit
};
struct JointResult { int32_t idx; float max_val; float blank_val; };
struct RowArgs {
    int mode = ROW_HR;
    const float *ws = nullptr; int kz = 1; int m_stride = 0; int N = 0; int M = 0;
    int n_valid = 0;                       // ROW_ARGMAX: vocabulary size (<= N)
    const float *bias = nullptr;
    const float *resid = nullptr; int ldr = 0;
    float *out = nullptr; int ldo = 0;
    const int *slot_idx = nullptr;         // row -> slot for state / slot-indexed outputs
    float *state = nullptr; int ld_state = 0;   // ROW_HR: h state of this layer
    float eps = 0.0f;
    int blank = 0;
    JointResult *joint = nullptr;
    float *logits_dump = nullptr;          // optional [M][n_valid]
};
void launch_row(const RowArgs &r, hipStream_t s);

// ---------------------------------------------------------------- encoder front (conv 1+2, im2col for conv 3)
struct ConvEmbedArgs {
    const float *ring = nullptr;           // [slots][ring_frames][mel]
    int ring_frames = 0, mel = 0, seg = 0;
    const int *slot_idx = nullptr;         // row -> slot
    const int *ring_tail = nullptr;        // row -> first ring row of the chunk
    const float *w[3] = {nullptr, nullptr, nullptr};
    const float *b[3] = {nullptr, nullptr, nullptr};
    int ch[3] = {0, 0, 0};
    int ch1_per_group = 0;                 // second-conv channels per workgroup (grid.y = ch[1] / this)
    int stride[3] = {1, 2, 2};
    float *out = nullptr; int ldo = 0;     // im2col rows of the third conv: [M * f_out][ldo], k = ci*9 + i*3 + j
    int M = 0;
    const float *x_direct = nullptr;       // debug path: x given as [M][seg][mel] instead of the ring
};
void launch_conv_embed(const ConvEmbedArgs &a, hipStream_t s);

// ---------------------------------------------------------------- decoder front
struct DecEmbedArgs {
    const float *emb = nullptr;            // [vocab][d]
    const float *conv_w = nullptr;         // [d][d/groups][context]
    const float *conv_b = nullptr;         // optional [d]
    const int *ctx = nullptr;              // [M][context]
    int d = 0, groups = 0, context = 0, vocab = 0, M = 0;
    float *out = nullptr; int ldo = 0;     // relu(conv(emb)) [M][d]
};
void launch_dec_embed(const DecEmbedArgs &a, hipStream_t s);

// fp32 -> fp16 (round to nearest even), elementwise; used once at load for the fp16 weight copies
void launch_cvt_f16(const float *src, void *dst, size_t n, hipStream_t s);

// ---------------------------------------------------------------- fbank
struct FbankTables {                       // device pointers
    const float *window = nullptr;         // [padded]
    const double *tw[16] = {nullptr};      // per factor twiddles (pocketfft layout)
    int fct[16] = {0}; int nfct = 0;
    const float *mel = nullptr;            // [nbins][padded/2]
    const int *mel_lo = nullptr;           // [nbins] first non-zero fft bin
    const int *mel_hi = nullptr;           // [nbins] one past the last non-zero fft bin
    int padded = 0, nbins = 0;
};
struct FbankFrameDesc { int32_t slot; int32_t ring_row; int32_t pcm_off; };   // pcm_off < 0: padding row log(kEps)
struct FbankArgs {
    FbankTables t;
    const int16_t *pcm = nullptr;          // staged samples
    const FbankFrameDesc *desc = nullptr;
    int n_frames = 0;
    float *ring = nullptr; int ring_frames = 0;
    float pad_value = 0.0f;
};
void launch_fbank(const FbankArgs &a, hipStream_t s);

}  // namespace aprilx
