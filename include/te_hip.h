/*
 * te_hip.h — C ABI of libte_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * TransEditor generator hot path.
 *
 * Conventions (SURVEY §8b "What a C-ABI replacement must export"):
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated; the CALLER owns every
 *     buffer (the Python host allocates through the torch caching allocator);
 *   - functions only ENQUEUE work on `stream` (a hipStream_t passed as void*): no allocation,
 *     no synchronisation; safe to call from several host threads.  No global mutable state takes part in any RESULT; the only
 *     process-wide state are five debug / A-B switches that select between kernels with the same results (bit-identical for
 *     te_conv_wino6_form, te_conv_s2s6_form, te_conv_t2s6_form and te_wgrad_t2_wide, fp32-equivalent for te_wgrad_split_bf16; all
 *     atomics, initialised from the environment, never written by the product's own code paths) and the per-thread last-error
 *     string;
 *   - return 0 on success, a negative TE_ERR_* for argument validation failures, or a positive
 *     hipError_t if the launch failed; nothing throws across the ABI.  te_last_error_string()
 *     describes the calling thread's most recent failure.
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * BillyXYB/TransEditor repository root).
 *
 * Precision: the model path is fp32 (suffix _f32).  The reference's two CUDA ops dispatch over half / float / double
 * (AT_DISPATCH_FLOATING_TYPES_AND_HALF, fused_bias_act_kernel.cu:79, upfirdn2d_kernel.cu:196); the TransEditor scripts
 * only ever run them in fp32 (no autocast / .half() anywhere on the path), so the tuned kernels are fp32, and K1 / K2 —
 * the two ops that ARE the reference's native boundary — also exist as te_bias_act_f16 / _f64 and te_upfirdn2d_f16 / _f64
 * (plain kernels, same semantics).  Every other entry point is fp32 only: the Python wrappers raise on any other dtype
 * instead of silently casting (tests/test_gpu_generator.py::test_non_contiguous_and_wrong_dtype_inputs).
 *
 * Run-to-run reproducibility.  EVERY result of this library is bit-reproducible from run to run (same inputs, same
 * library, same device): there is no atomic add on any path.  Reductions that span thread blocks go through per-block
 * partials in a caller-owned workspace and a fixed-order second pass: the split-K convolution (te_conv_ws_f32 /
 * te_conv_res_f32; te_conv_f32, which has no workspace argument, does not split), te_small_gemm_splitk_f32, the
 * per-(sample, chunk) correlation slabs of te_wgrad_f32 / te_rgb_wgrad_f32, the bias gradient of te_bias_act_bwd_f32 /
 * te_bias_act_bwd_rgb_f32 (workspace: te_bias_act_bwd_ws_floats), the per-tile bias-gradient partials of te_blur_actgrad_f32 /
 * te_blur_gradact_f32, all three outputs of te_wgrad_reduce_f32 (workspace: te_wgrad_reduce_ws_floats), te_chan_dot_f32, the
 * layer / pixel norm and minibatch-stddev kernels.  The library is built WITHOUT -munsafe-fp-atomics.
 * (tests/test_gpu_determinism.py runs the 256-px generator and discriminator backward twice and compares bit for bit.)
 */
#ifndef TE_HIP_H
#define TE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TE_ABI_VERSION 3

#define TE_ERR_NULL -1      /* required pointer is NULL              */
#define TE_ERR_SHAPE -2     /* non-positive / inconsistent dimension */
#define TE_ERR_UNSUPPORTED -3
#define TE_ERR_WORKSPACE -4 /* workspace too small                   */

typedef void* te_stream_t;

int te_version(void);
const char* te_last_error_string(void);
/* name of the code-object architecture this library was built for ("gfx950") */
const char* te_arch(void);

/* ---------------------------------------------------------------------------------------------
 * K1  fused bias + activation.  Replaces pybind `fused.fused_bias_act(input, bias, refer, act,
 * grad, alpha, scale)` — utils/op/fused_bias_act.cpp:11-21, kernel fused_bias_act_kernel.cu:18-49.
 *   x' = x + b[(i / step_b) % size_b]     (b may be NULL = no bias; integer index math bit-exact)
 *   act*10+grad: 10/11 linear, 12 -> 0, 30 lrelu(x'), 31 x' * (ref>0 ? 1 : alpha), 32 -> 0
 *   out = y * scale.   `ref` may be NULL (treated as 0).  In-place (out == x) is allowed.
 */
int te_bias_act_f32(float* out, const float* x, const float* b, const float* ref, int act, int grad,
                    float alpha, float scale, int64_t size_x, int64_t step_b, int64_t size_b,
                    te_stream_t stream);
/* The reference dispatches this op over half / float / double (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * fused_bias_act_kernel.cu:79) and converts its float `alpha` / `scale` to scalar_t; same here.  Half buffers are IEEE
 * binary16 (`__half` / torch.float16) passed as void*; arithmetic in fp32 with one rounding at the store. */
int te_bias_act_f16(void* out, const void* x, const void* b, const void* ref, int act, int grad,
                    float alpha, float scale, int64_t size_x, int64_t step_b, int64_t size_b,
                    te_stream_t stream);
int te_bias_act_f64(double* out, const double* x, const double* b, const double* ref, int act, int grad,
                    float alpha, float scale, int64_t size_x, int64_t step_b, int64_t size_b,
                    te_stream_t stream);

/* Backward of the fused lrelu in ONE pass — replaces FusedLeakyReLUFunctionBackward.forward,
 * utils/op/fused_act.py:18-38 (kernel call + grad_input.sum(dim)):
 *   gi[n,c,i] = g[n,c,i] * (ref[n,c,i] > 0 ? 1 : alpha) * scale ;  gb[c] = sum_{n,i} gi[n,c,i]
 * Layout [outer][C][inner].  gb (may be NULL) is WRITTEN (no zero fill needed).  The streaming form combines per-block partial
 * sums through the caller's workspace `ws` of te_bias_act_bwd_ws_floats(outer, C, inner) floats (0: none needed, ws may be
 * NULL) and a fixed-order second pass: bit-reproducible, no atomics.  */
int64_t te_bias_act_bwd_ws_floats(int64_t outer, int64_t C, int64_t inner);
int te_bias_act_bwd_f32(float* gi, float* gb, float* ws, const float* g, const float* ref, float alpha, float scale,
                        int64_t outer, int64_t C, int64_t inner, te_stream_t stream);
/* te_bias_act_bwd_f32 with the data gradient of a ToRGB layer (1x1 modulated convolution to 3 channels, ToRGB.forward,
 * model_spatial_query.py:416-425) folded in — te_rgb_dgrad_f32, the gradient-accumulation add and the activation gradient
 * in one pass over the activation-sized tensors:
 *     gi = ( g + wscale * srgb[n,c] * sum_o wrgb[o,c] * grgb[n,o,:] ) * (ref > 0 ? 1 : alpha) * scale ,   gb[c] = sum gi
 * g [outer,C,inner] may be NULL (nothing but ToRGB consumes the activation), grgb [outer,3,inner], wrgb [3,C], srgb [outer,C] or
 * NULL; gb written or NULL (needs ws, same size as above).  te_bias_act_bwd_rgb_supported: inner % 4 == 0 and inner >= 1024. */
int te_bias_act_bwd_rgb_supported(int64_t outer, int64_t C, int64_t inner);
int te_bias_act_bwd_rgb_f32(float* gi, float* gb, float* ws, const float* g, const float* ref, const float* grgb, const float* wrgb,
                            const float* srgb, float wscale, float alpha, float scale, int64_t outer, int64_t C, int64_t inner,
                            te_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2  upfirdn2d.  Replaces pybind `upfirdn2d_op.upfirdn2d(input[major,H,W,minor], kernel, up_x,
 * up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)` — utils/op/upfirdn2d.cpp:12-23, kernel
 * upfirdn2d_kernel.cu:52-137.  Output dims follow upfirdn2d.py:101-102:
 *   out_h = (in_h*up_y + pad_y0 + pad_y1 - kh) / down_y + 1 (same for w); caller allocates
 *   out[major, out_h, out_w, minor].  True convolution (taps flipped), zero fill outside the
 *   input.  Unlike the reference (6 template modes, garbage otherwise) every (up, down, k) works.
 * Optional fused epilogue (b != NULL or act != 0), minor must be 1: channel = major_index % size_b,
 *   out = act(out + b[channel]) * scale with act 0 = linear, 3 = lrelu(alpha).
 */
int te_upfirdn2d_f32(float* out, const float* x, const float* k, int64_t major, int in_h, int in_w,
                     int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                     int pad_x1, int pad_y0, int pad_y1, const float* b, int64_t size_b, int act,
                     float alpha, float scale, te_stream_t stream);
/* half / double forms of the same op (upfirdn2d_kernel.cu:57-58 dispatches over both): every (up, down, taps), no fused
 * epilogue (the reference op has none); taps in the tensors' own type; accumulation in fp32 / double. */
int te_upfirdn2d_f16(void* out, const void* x, const void* k, int64_t major, int in_h, int in_w,
                     int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                     int pad_x1, int pad_y0, int pad_y1, te_stream_t stream);
int te_upfirdn2d_f64(double* out, const double* x, const double* k, int64_t major, int in_h, int in_w,
                     int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                     int pad_x1, int pad_y0, int pad_y1, te_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * F1  modulated convolution family (reference: ModulatedConv2d.forward, model_spatial_query.py:
 * 296-337, which the reference runs as stock grouped F.conv2d / F.conv_transpose2d with B
 * materialised weight copies).  Here ONE shared weight tensor is used and the per-sample style
 * modulation / demodulation are row/column scalings fused into the kernel:
 *     out[b,m,:,:] = act( osc[b,m] * conv(isc[b,k] * in[b,k,:,:], W)[m] + bias[m] )
 *
 * Weights are consumed in a packed layout  Wp[tap][Kp][Mp]  (Kp = K rounded up to 16, Mp = M
 * rounded up to 128, zero padded) produced by te_conv_pack_weights_f32.
 */

/* kind of convolution executed by te_conv_f32 */
#define TE_CONV_3X3 0    /* 3x3, stride 1, pad 1           in [B,K,H,W]       -> out [B,M,H,W]        */
#define TE_CONV_T2 1     /* 3x3 transposed, stride 2, pad 0 in [B,K,H,W]       -> out [B,M,2H+1,2W+1]  */
#define TE_CONV_S2 2     /* 3x3, stride 2, pad 0           in [B,K,2H+1,2W+1] -> out [B,M,H,W]        */
#define TE_CONV_1X1 3    /* 1x1                            in [B,K,H,W]       -> out [B,M,H,W]        */
#define TE_CONV_3X3W 4   /* TE_CONV_3X3 through the 1-D Winograd F(2,3) kernel (2/3 of the MFMAs; same result to fp32 round-off).
                            Shapes: te_conv_wino_supported; weights packed TE_PACK_WFWD / TE_PACK_WDGRAD; never split */
#define TE_CONV_3X3W6 5  /* the same Winograd form with its products on the bf16 matrix pipe: every fp32 operand split into three bf16
                            pieces, six exact piece products accumulated in fp32 - fp32-equivalent results (the dropped terms are below
                            2^-24 of the product; measured deviation from double not larger than the fp32 MFMA chain's).
                            Shapes: te_conv_wino6_supported; weights packed TE_PACK_W6FWD / TE_PACK_W6DGRAD; never split.
                            Range: finite operands of any fp32 magnitude up to ~1.7e38 (the transform adds two neighbours) behave as
                            in TE_CONV_3X3 (tests: scale sweep 1e-30 ... 1e+30 at the 5e-6 bar); below |x| ~ 1e-33 the third, then
                            the second piece of an element underflows and that element's products lose bits (relative error 2^-15
                            at 1e-36: contributions that small are below the rounding unit of any O(1e-30)+ sum anyway).
                            Non-finite operands: an Inf / NaN input element makes exactly the outputs whose 3x3 window contains it
                            non-finite - the same set as TE_CONV_3X3 - but as NaN where the direct kernel gives +-Inf
                            (inf = h, inf - h = NaN is the second piece).  TE_SPLIT_BF16=0 (host) selects TE_CONV_3X3W instead. */

#define TE_CONV_S2S6 6   /* TE_CONV_S2 with its products on the bf16 matrix pipe (three-piece split, six exact piece products per multiply-add,
                            fp32 accumulation: fp32-equivalent like TE_CONV_3X3W6, same range / non-finite behaviour).
                            Shapes: te_conv_s2s6_supported; weights packed TE_PACK_S6FWD / TE_PACK_S6SWAP; never split.
                            Reference: F.conv2d(stride 2) of ConvLayer(downsample=True), model_spatial_query.py:765-779, and the adjoint
                            of conv_transpose2d(stride 2), :318 */

#define TE_CONV_T2S6 7   /* TE_CONV_T2 with its products on the bf16 matrix pipe (three-piece split, fp32-equivalent like TE_CONV_3X3W6): the body
                            cells [0,H) x [0,W) in csrc/t2s6.hip, the last output row and column through the fp32 kernel.
                            Shapes: te_conv_t2s6_supported; weights packed TE_PACK_T6FWD / TE_PACK_T6SWAP; never split; no residual / mask.
                            Reference: conv_transpose2d(stride 2) of ModulatedConv2d.forward, model_spatial_query.py:310-321 */

#define TE_CONV_1X1S6 8   /* TE_CONV_1X1 with its products on the bf16 matrix pipe (round 6; three-piece split, six exact piece products per
                            multiply-add, fp32 accumulation: fp32-equivalent like TE_CONV_3X3W6, same range / non-finite behaviour): the
                            skip branch of the discriminator's ResBlocks and its data gradient.  Plain product + optional residual only
                            (isc, osc, bias, mask_ref must be NULL, act 0).  Shapes: te_conv_p1s6_supported; weights packed
                            TE_PACK_P6FWD / TE_PACK_P6DGRAD; never split.
                            Reference: F.conv2d of EqualConv2d (1x1) in ResBlock.skip, model_spatial_query.py:173-181, :780-798 */

/* how te_conv_pack_weights_f32 reads the source weight w[Co][Ci][kh][kw] (model layout,
 * ModulatedConv2d.weight[0]) */
#define TE_PACK_FWD 0    /* M = Co, K = Ci, taps as stored         (forward 3x3 / 1x1 / T2, and S2) */
#define TE_PACK_DGRAD 1  /* M = Ci, K = Co, taps flipped           (data gradient of 3x3 / 1x1)     */
#define TE_PACK_SWAP 2   /* M = Ci, K = Co, taps as stored         (S2 as data gradient of T2)       */
#define TE_PACK_WFWD 3   /* TE_CONV_3X3W forward:       U[K/8][ky][c][8][M], U[ky][c] = G w[.., ky, :]  (12 K M floats)  */
#define TE_PACK_WDGRAD 4 /* TE_CONV_3X3W data gradient: the same transform of the flipped, transposed taps (M = Ci)     */
#define TE_PACK_W6FWD 5  /* TE_CONV_3X3W6 forward: U = G w split into bf16 pieces, MFMA fragment order
                            U6[K/16][piece][ky][c][M/32][64 lanes][8 bf16]  (36 K M bf16 = 18 K M floats).  The packing accepts
                            Co % 32 == Ci % 32 == 0; the kernel (te_conv_wino6_supported) needs K % 32 == 0, M % 64 == 0                */
#define TE_PACK_W6DGRAD 6 /* TE_CONV_3X3W6 data gradient (flipped, transposed taps; M = Ci)                               */
#define TE_PACK_S6FWD 7  /* TE_CONV_S2S6, M = Co, K = Ci, taps as stored: three bf16 pieces per weight, MFMA fragment order
                            S6[K/16][piece][tap][M/32][64 lanes][8 bf16]  (27 K M bf16).  The packing accepts M % 32 == 0, K % 16 == 0;
                            the kernel (te_conv_s2s6_supported) needs M % 64 == 0, K % 16 == 0, K >= 32                                 */
#define TE_PACK_S6SWAP 8 /* TE_CONV_S2S6 as data gradient of the transposed kind: M = Ci, K = Co, taps as stored                  */
#define TE_PACK_T6FWD 9  /* TE_CONV_T2S6, M = Co, K = Ci: the TE_PACK_S6FWD layout followed (16-byte aligned) by the TE_PACK_FWD layout  */
#define TE_PACK_T6SWAP 10 /* TE_CONV_T2S6 as data gradient of the strided kind: TE_PACK_S6SWAP followed by TE_PACK_SWAP                   */
#define TE_PACK_P6FWD 11  /* TE_CONV_1X1S6, M = Co, K = Ci (1x1 weights): P6[K/16][piece][M/32][64 lanes][8 bf16] (3 K M bf16; Co % 32 == 0,
                             Ci % 16 == 0)                                                                                                */
#define TE_PACK_P6DGRAD 12 /* TE_CONV_1X1S6 as data gradient: M = Ci, K = Co (Ci % 32 == 0, Co % 16 == 0)                                  */

int64_t te_conv_packed_numel(int kind_pack, int Co, int Ci, int ksize);
int te_conv_pack_weights_f32(float* wp, const float* w, float wscale, int kind_pack, int Co, int Ci,
                             int ksize, te_stream_t stream);
/* two layouts of the same weight in one launch (a training forward packs the data-gradient layout along) */
int te_conv_pack_weights2_f32(float* wp_a, int kind_a, float* wp_b, int kind_b, const float* w, float wscale, int Co, int Ci,
                              int ksize, te_stream_t stream);

/* n (weight, layout) jobs in one launch per 64: job e packs w[e] [Co[e]][Ci[e]][ksize[e]]^2 into wp[e] with layout
 * kind_pack[e] and constant wscale[e] (host arrays).  The training step refreshes every packed layout of a model with it right
 * after the optimiser step (torch.optim.Adam's step, train_spatial_query.py:207 / :224) instead of ~60 tiny launches per iteration. */
int te_conv_pack_weights_multi_f32(int n, float* const* wp, const float* const* w, const float* wscale, const int* kind_pack,
                                   const int* Co, const int* Ci, const int* ksize, te_stream_t stream);

/* `H`,`W` are ALWAYS the low-resolution size (the H,W of the table above).  isc [B,K], osc [B,M],
 * bias [M] may be NULL.  act: 0 linear, 3 lrelu(0.2)*sqrt(2), 4 lrelu(0.2) with gain 1 (a residual branch that folds the
 * 1/sqrt(2) of `(out + skip) / sqrt(2)`, model_spatial_query.py:796) — applied after osc and bias. */
int te_conv_f32(float* out, const float* in, const float* wp, const float* isc, const float* osc,
                const float* bias, int act, int kind, int B, int K, int M, int H, int W,
                te_stream_t stream);
/* Small images (4x4 ... 16x16 layers) split the input-channel loop over the grid so every CU gets work.
 * te_conv_splitk_count returns the number of splits S of a problem (1 = no split).  te_conv_ws_f32 is te_conv_f32 with
 * a caller-owned workspace ws[S][B][M][Ho][Wo] (ignored / may be NULL when S == 1): each split writes its own slab and a
 * second kernel sums them in a fixed order (DETERMINISTIC, no atomics, no memset).  te_conv_f32 itself (no workspace)
 * never splits (same result up to summation order, slower on 4x4 ... 16x16 images). */
int te_conv_splitk_count(int kind, int B, int K, int M, int H, int W);
/* 1 if TE_CONV_3X3W covers the problem: K % 8 == 0, W % 32 == 0, and M % 128 == 0 with H % 4 == 0, or M % 64 == 0 with
 * H % 8 == 0, or M % 32 == 0 with H % 16 == 0 (block tile = 128 / 64 / 32 output channels x 4 / 8 / 16 rows x 32 columns):
 * every 3x3 stride-1 layer of the generator and discriminator from 32x32 up incl. the 32 / 64-channel tail of FFHQ-1024; the
 * 513-channel final convolution of the discriminator and images narrower than 32 stay on TE_CONV_3X3.  Reference: the grouped F.conv2d of ModulatedConv2d.forward, model_spatial_query.py:331-333,
 * and EqualConv2d.forward :173-181. */
int te_conv_wino_supported(int B, int K, int M, int H, int W);
/* 1 if TE_CONV_3X3W6 covers the problem: K % 32 == 0, M % 64 == 0, H % 8 == 0, and W % 32 == 0 or (round 6) W == 16 with an even batch
 * (two samples side by side in a 32-column tile row) */
int te_conv_wino6_supported(int B, int K, int M, int H, int W);
/* 1 if TE_CONV_S2S6 covers the problem (H, W = OUTPUT size): K % 16 == 0 and K >= 32, M % 64 == 0, H % 8 == 0, W % 16 == 0 */
int te_conv_s2s6_supported(int B, int K, int M, int H, int W);
/* 1 if TE_CONV_T2S6 covers the problem (H, W = INPUT size): K % 16 == 0 and K >= 32, M % 64 == 0, H % 8 == 0, W % 16 == 0 */
int te_conv_t2s6_supported(int B, int K, int M, int H, int W);
/* 1 if TE_CONV_1X1S6 covers the problem: K % 64 == 0, M % 128 == 0, H * W % 256 == 0 and a grid of at least half the CUs */
int te_conv_p1s6_supported(int B, int K, int M, int H, int W);
/* Kernel form of TE_CONV_3X3W6 (a DEBUG / A-B switch, process-wide - the one piece of mutable state behind this ABI besides
 * te_wgrad_split_bf16; the results do not depend on it; returns the previous value; anything but 0 .. 3 only queries):
 *   2 = two-image (round 6, default): as 1, but a block owns 128 output channels - every staged half tile is multiplied by two
 *       64-channel weight images, so the style scale / B^T d / three-piece split of an input element is done once per 128 output
 *       channels instead of once per 64; launches with M % 128 != 0, or whose grid would leave CUs without a block, run form 1
 *       (3 = the two-image form wherever M % 128 == 0, whatever the grid: tests);
 *   1 = ping-pong (round 5): the two waves of every SIMD work half a stage apart - one feeds the matrix pipe from its
 *       half tile while the other transforms / splits / writes the next half tile and renews half of the weight image;
 *   0 = block-phase (round 4): all eight waves multiply, barrier, all eight waves stage, barrier.
 * All forms issue the same products in the same order per output element: results are bit-identical.  TE_W6_FORM in the
 * environment sets the initial value (A/B measurements). */
int te_conv_wino6_form(int form);
/* Kernel form of TE_CONV_S2S6 (returns the previous value; form < 0 only queries), a process-wide A/B switch like the one above:
 *   1 = (round 6, default) the two-image form: a block owns 128 output channels and multiplies every staged half tile by two
 *       64-channel weight images (half the fetches, split arithmetic and LDS writes per MFMA) where M % 128 == 0 and the grid
 *       still gives every CU a block, the ping-pong form elsewhere (2 = the two-image form wherever M % 128 == 0: tests);
 *   0 = ping-pong (round 5).
 * Same products in the same order per output element: results are bit-identical.  TE_S2S6_FORM in the environment sets the
 * initial value. */
int te_conv_s2s6_form(int form);
/* the same switch for TE_CONV_T2S6 (t2s6q_kernel / t2s6_kernel; TE_T2S6_FORM) */
int te_conv_t2s6_form(int form);
/* TE_CONV_T2S6 only: `ws` of te_conv_ws_f32 / te_conv_res_f32 is an OPTIONAL scratch of te_conv_t2s6_ws_floats(B, K, H) = B * K * H
 * floats through which the body kernel hands the (style-scaled) last input column to the kernel that computes the last output
 * row / column (round 6); NULL is legal - that kernel then gathers the column from `in` itself, one cache line per element. */
int64_t te_conv_t2s6_ws_floats(int B, int K, int H);
int te_conv_ws_f32(float* out, float* ws, const float* in, const float* wp, const float* isc, const float* osc,
                   const float* bias, int act, int kind, int B, int K, int M, int H, int W, te_stream_t stream);
/* te_conv_ws_f32 with two more epilogue stages (not for TE_CONV_T2; a split launch, S > 1, needs the workspace):
 *     out = ( act(osc * conv + bias) + res ) * (mask_ref > 0 ? mask_gain : 0.2 * mask_gain)
 * res (shaped like out, may be NULL) carries the sum of a ResBlock's two branches (model_spatial_query.py:796, forward) and
 * the sum of the two gradient branches that meet at the block's input (backward) without an extra elementwise pass;
 * mask_ref (shaped like out, may be NULL) is the saved output of a fused bias + leaky-ReLU(0.2) * mask_gain layer whose
 * OUTPUT this data gradient lands on: its activation gradient (fused_bias_act_kernel.cu:26-47, grad pass) in this epilogue. */
int te_conv_res_f32(float* out, float* ws, const float* in, const float* wp, const float* isc, const float* osc,
                    const float* bias, const float* res, const float* mask_ref, float mask_gain, int act, int kind, int B,
                    int K, int M, int H, int W, te_stream_t stream);

/* Weight-gradient correlation, per sample and per pixel chunk ("slabs"), NO modulation applied:
 *   slab[b][s][co][ci][tap] = sum_{pixels of chunk s} g[b,co,p (+) tap] * x[b,ci,p]
 * kind TE_CONV_3X3 / TE_CONV_1X1: g [B,Co,H,W], x [B,Ci,H,W];
 * kind TE_CONV_T2: g [B,Co,2H+1,2W+1], x [B,Ci,H,W]  (weight gradient of the transposed conv).
 * te_wgrad_slab_count returns S (chunks per sample) for the given problem; the caller allocates
 * slabs[B][S][Co][Ci][taps] and reduces them with te_wgrad_reduce_f32. */
int te_wgrad_slab_count(int kind, int B, int Co, int Ci, int H, int W);
/* 1 when te_wgrad_f32 / te_wgrad_group_f32 run this problem in the PAIR form (3x3 only: horizontal cell pairs, the three taps of
 * a kernel row from four products per pair - 1-D Winograd F(3,2) - i.e. 2/3 of the direct form's multiply-adds on the matrix
 * pipe for the same slabs), else 0.  Only a report for FLOP accounting (bench.py): results and slab layout do not depend on it. */
int te_wgrad_pair_form(int kind, int Co, int Ci, int H, int W);
/* Round 5: the 3x3 correlation on the bf16 matrix pipe (csrc/wgrad6.hip: pair form, every fp32 operand split into three bf16 pieces,
 * six exact piece products per multiply-add, fp32 accumulation - fp32-equivalent slabs, same layout, same reducers).
 * te_wgrad_split_supported: 1 where it applies (Co % 64 == 0, Ci % 64 == 0; kind TE_CONV_3X3: W % 32 == 0; kind TE_CONV_T2 - direct
 * form, nine taps x six products - W % 16 == 0; round 6: kind TE_CONV_1X1 with Co % 128 == 0, Ci % 128 == 0, W % 16 == 0).
 * te_wgrad_split_bf16(0 | 1): process-wide switch (environment TE_SPLIT_WGRAD at load time), returns the previous value; any other
 * argument only queries.  With the switch on, te_wgrad_f32 / te_wgrad_group_f32 take the kernel where it applies and the fp32
 * kernel elsewhere. */
int te_wgrad_split_supported(int kind, int Co, int Ci, int H, int W);
int te_wgrad_split_bf16(int on);
/* Round 6: form of the split transposed-kind kernel - 1 (default): a block owns 64 channels of the (2H+1) x (2W+1) tensor x 128 of the
 * H x W one where Ci % 128 == 0 (half the staging work and re-reads of the big tensor per MFMA), 0: 64 x 64 everywhere.  Bit-identical
 * slabs; returns the previous value, any other argument only queries; TE_WGRAD_T2_WIDE in the environment sets the initial value. */
int te_wgrad_t2_wide(int on);
int te_wgrad_f32(float* slabs, const float* g, const float* x, int kind, int B, int Co, int Ci, int H,
                 int W, int S, te_stream_t stream);
/* GROUPED form for the PLAIN (unmodulated) weight gradient of small images: NB consecutive samples share one slab
 * (slabs [B / NB][S][Co][Ci][taps]), a block walks the cell tiles of its NB samples.  te_wgrad_group_plan picks NB (a divisor
 * of B; 1 when grouping does not apply) and S.  Not for layers whose reducer needs per-sample slabs (style / demodulation
 * gradients of ModulatedConv2d): those keep te_wgrad_f32.  Replaces torch's conv2d weight gradient of the discriminator's
 * 4x4 ... 32x32 layers (model_spatial_query.py:731-798), where B slabs of 9.4 MB were the traffic. */
int te_wgrad_group_plan(int kind, int B, int Co, int Ci, int H, int W, int* NB, int* S);
int te_wgrad_group_f32(float* slabs, const float* g, const float* x, int kind, int B, int Co, int Ci, int H, int W, int S, int NB,
                       te_stream_t stream);

/* Combine slabs (SURVEY §7 step 6 "reductions for ds, dd"):
 *   gw[co,ci,t]  = wscale * sum_{b,s} osc[b,co]*isc[b,ci] * slab          (gw   may be NULL)
 *   gisc[b,ci]   = sum_{co,t,s} wscale*w[co,ci,t] * osc[b,co] * slab      (gisc may be NULL)
 *   gosc[b,co]   = sum_{ci,t,s} wscale*w[co,ci,t] * isc[b,ci] * slab      (gosc may be NULL)
 * isc / osc NULL = all ones.  All three outputs are WRITTEN.  Shares of different thread blocks (channel tiles, slab chunks,
 * samples) meet through the caller's workspace `ws` of te_wgrad_reduce_ws_floats(...) floats and a fixed-order second pass:
 * bit-reproducible, no atomics, no memset. */
int64_t te_wgrad_reduce_ws_floats(int B, int S, int Co, int Ci, int taps, int want_w, int want_isc, int want_osc);
int te_wgrad_reduce_f32(float* gw, float* gisc, float* gosc, float* ws, const float* slabs, const float* w,
                        float wscale, const float* isc, const float* osc, int B, int S, int Co, int Ci,
                        int taps, te_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * M3  ToRGB: 1x1 modulated convolution to 3 channels (reference: ToRGB.forward, model_spatial_query.py:
 * 416-425 — grouped 1x1 F.conv2d, demodulate=False).  HBM-bound streaming kernels (x [B,K,HW] touched once):
 *   fwd   out[b,o,p] = sum_k wscale w[o,k] isc[b,k] x[b,k,p] + bias[o]   w [3,K], isc/bias may be NULL
 *   dgrad gx[b,k,p]  = wscale isc[b,k] sum_o w[o,k] g[b,o,p]
 *   wgrad slabs[b][c][o][k] = sum_{p in chunk c} g[b,o,p] x[b,k,p]       (finish with te_wgrad_reduce_f32, taps = 1)
 * te_rgb_supported: 1 if (M == 3, K <= 512, HW % 4 == 0), else use te_conv_f32(TE_CONV_1X1).
 */
int te_rgb_supported(int M, int K, int HW);
int te_rgb_fwd_f32(float* out, const float* x, const float* w, const float* isc, const float* bias, float wscale, int B,
                   int K, int HW, te_stream_t stream);
int te_rgb_dgrad_f32(float* gx, const float* g, const float* w, const float* isc, float wscale, int B, int K, int HW,
                     te_stream_t stream);
/* The same streaming form as the FORWARD of a 1x1 convolution from 3 channels (the discriminator's from-RGB stem,
 * ConvLayer(3, C, 1), model_spatial_query.py:815): out[b,k,p] = act(wscale * sum_o w[o,k] x3[b,o,p] + bias[k]),
 * w [3,K] (the [K,3] model weight transposed), bias [K] or NULL, act as te_conv_f32.  Its data gradient is te_rgb_fwd_f32,
 * its weight gradient te_rgb_wgrad_f32 with the two operands exchanged. */
int te_rgb_expand_f32(float* out, const float* x3, const float* w, const float* bias, int act, float wscale, int B, int K,
                      int HW, te_stream_t stream);
int te_rgb_wgrad_slab_count(int B, int K, int HW);
int te_rgb_wgrad_f32(float* slabs, const float* g, const float* x, int B, int K, int HW, int S, te_stream_t stream);
/* te_rgb_wgrad_f32 with a 4th slab row: slabs[b][c][3][k] = sum_{p in chunk c} x[b,k,p]  (slabs [B][S][4][K]).  With the
 * operands exchanged (g = the 3-channel image, x = the gradient of the from-RGB stem's pre-activation) rows 0-2 are the stem's
 * weight gradient and row 3 its bias gradient: one pass over the 128-channel gradient instead of two. */
int te_rgb_wgrad_sum_f32(float* slabs, const float* g, const float* x, int B, int K, int HW, int S, te_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * F2  attention core of the dual-space cross-attention block (reference: Attention.forward,
 * model_spatial_query.py:888-894): per sample and head,  sim = softmax(scale * q k^T),
 * o = sim v, with QK^T and sim.V on v_mfma_f32_16x16x4_f32.
 *   q [N, M, G*D], k,v [N, L, G*D] (token-major, as produced by the q/k/v linears), o [N, M, G*D],
 *   sim [N, G, M, L].  Supported: M == L == 16, D == 32 (the only shape the model produces).
 */
int te_attn_fwd_f32(float* o, float* sim, const float* q, const float* k, const float* v, float scale,
                    int N, int G, int M, int L, int D, te_stream_t stream);
int te_attn_bwd_f32(float* gq, float* gk, float* gv, const float* go, const float* gsim_or_null,
                    const float* q, const float* k, const float* v, const float* sim, float scale, int N,
                    int G, int M, int L, int D, te_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2b  backward of "blur -> + bias -> leaky-ReLU * sqrt(2)" (the upsampling StyledConv tail, model_spatial_query.py:321 +
 * :401; reference backward = fused_bias_act_kernel.cu grad pass, then upfirdn2d with flipped taps) in ONE pass:
 *   gpre = g * (ref > 0 ? scale : alpha * scale)      ref = saved forward output, g / ref: [major, in_h, in_w]
 *   gx   = upfirdn2d(gpre, k, up = down = 1, pads)    k = the already flipped 4x4 taps, pads >= 0
 *   partial[plane][tile] = sum of gpre over the part of the plane the tile owns (bias gradient = sum over planes of the
 *   same channel and over tiles); te_blur_actgrad_tiles gives the tile count per plane.
 */
int te_blur_actgrad_tiles(int in_h, int in_w, int kh, int kw, int pad_x0, int pad_x1, int pad_y0, int pad_y1);
int te_blur_actgrad_f32(float* gx, float* partial, const float* g, const float* ref, const float* k, int64_t major, int in_h,
                        int in_w, int kh, int kw, int pad_x0, int pad_x1, int pad_y0, int pad_y1, float alpha, float scale,
                        te_stream_t stream);
/* K2c  the other order, backward of "conv + bias -> leaky-ReLU * scale -> blur" (first half of the discriminator's ResBlock,
 * model_spatial_query.py:744-768): the adjoint FIR first, the activation gradient in its epilogue —
 *   gx = upfirdn2d(g, k, up = down = 1, pads) * (ref > 0 ? scale : alpha * scale)    g [major, in_h, in_w],
 *   ref = the saved activation output, shaped like gx;  partial[plane][tile] = sum of the tile's gx (bias gradient).
 * Same tile count as te_blur_actgrad_tiles; in_w >= 4. */
int te_blur_gradact_f32(float* gx, float* partial, const float* g, const float* ref, const float* k, int64_t major, int in_h,
                        int in_w, int kh, int kw, int pad_x0, int pad_x1, int pad_y0, int pad_y1, float alpha, float scale,
                        te_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * G2/A2  small dense layers (reference: EqualLinear.forward, model_spatial_query.py:213-221 — F.linear on
 * weight * scale with bias * lr_mul, optional activation).  One fused launch on fp32 MFMA:
 *     C[i,j] = act( alpha * sum_k A(i,k) * B(k,j) + beta * bias[j] ) + residual[i,j]      C, residual, pre: [I,J] row-major
 * A(i,k) = a[i*sai + k*sak],  B(k,j) = b[k*sbk + j*sbj]  (element strides, so y = x W^T, dx = g W and dW = g^T x all map
 * onto it).  bias / residual / pre (pre-activation copy) may be NULL.  act: 0 none, 1 GELU(erf), 3 lrelu(0.2)*sqrt(2).
 * arowsum (may be NULL): arowsum[i] = rs_scale * sum_k A(i,k) — the bias gradient, for free, in the dW = g^T x call.
 */
int te_small_gemm_f32(float* c, float* pre, const float* a, const float* b, const float* bias, const float* residual,
                      float* arowsum, float rs_scale, int I, int J, int K, int64_t sai, int64_t sak, int64_t sbk,
                      int64_t sbj, float alpha, float beta, int act, te_stream_t stream);

/* the same GEMM for WIDE reductions (reference: the discriminator's EqualLinear(8192, 512, 'fused_lrelu'),
 * model_spatial_query.py:831-834): K is split into S chunks (K % S == 0, (K / S) % 8 == 0) that run as S x tiles blocks,
 * partial tiles go to the caller's workspace ws[S][I][J], a second kernel sums them in a fixed order and applies the
 * epilogue (deterministic, no atomics).  C / pre / residual are dense [I, J]. */
int te_small_gemm_splitk_f32(float* c, float* pre, float* ws, int S, const float* a, const float* b, const float* bias,
                             const float* residual, int I, int J, int K, int64_t sai, int64_t sak, int64_t sbk, int64_t sbj,
                             float alpha, float beta, int act, te_stream_t stream);

/* D1  minibatch standard deviation + channel concat of the discriminator (reference: Discriminator.forward,
 * model_spatial_query.py:844-852) as one launch: with n = B / group, sample b = g * n + j belongs to set j;
 *   s_j = mean over the C*HW positions of sqrt(var_g x[g*n + j] + eps)      (biased variance over the `group` samples)
 *   y[b, :C] = x[b];  y[b, C, :] = s_{b mod n}                              x [B, C, HW], y [B, C + 1, HW]
 * backward: gx = gy[:, :C] + d s / d x * (sum of gy[:, C] over the set).  group <= 4, B % group == 0. */
int te_minibatch_stddev_fwd_f32(float* y, const float* x, int B, int group, int C, int HW, float eps, te_stream_t stream);
int te_minibatch_stddev_bwd_f32(float* gx, const float* gy, const float* x, int B, int group, int C, int HW, float eps,
                                te_stream_t stream);

/* A2  parameter-free layer norm over whole samples (reference: AttentionBlock.forward, model_spatial_query.py:924 / 931,
 * F.layer_norm(x, x.size()[1:]), eps 1e-5): x / y / g / gx are [R, N] rows; stats [R, 2] = (mean, rstd) saved for the
 * backward  gx = rstd * (g - mean(g) - y * mean(g * y)).  N % 4 == 0, N <= 16384, 16-byte aligned rows. */
int te_layer_norm_supported(int64_t R, int N);
int te_layer_norm_fwd_f32(float* y, float* stats, const float* x, int64_t R, int N, float eps, te_stream_t stream);
int te_layer_norm_bwd_f32(float* gx, const float* g, const float* y, const float* stats, int64_t R, int N, te_stream_t stream);

/* G1  PixelNorm over the channel axis of the [B, D, C] latent codes (reference: PixelNorm.forward, model_spatial_query.py:
 * 80-81, pixel_norm_op_dim = 1): y = x * rsqrt(mean_d x^2 + eps); r [B, C] = the factor, saved for the backward
 * gx = r * (g - y * mean_d(g * y)).  C must divide 256. */
int te_pixel_norm_supported(int64_t B, int D, int C);
int te_pixel_norm_fwd_f32(float* y, float* r, const float* x, int64_t B, int D, int C, float eps, te_stream_t stream);
int te_pixel_norm_bwd_f32(float* gx, const float* g, const float* y, const float* r, int64_t B, int D, int C, te_stream_t stream);

/* G2  the token-wise mapping loops (reference: Generator.forward, model_spatial_query.py:626-646 — for each of the 16
 * tokens its own EqualLinear + fused leaky-ReLU, 64 launches + 32 slice copies) as ONE launch: the same kernel batched
 * over blockIdx.z.  Operand z uses a + z*za, c + z*zc (uniform element strides) and either b + z*zb / bias + z*zbias or,
 * when the per-token parameters are separate allocations, b + b_tab[z] / bias + bias_tab[z] (host arrays of nz <= 16
 * element offsets, NULL for the uniform form).  C(i,j) = c[i*sci + j*scj], so the result can be written straight into the
 * [B, tokens, D] or [B, D, tokens] layout the next stage reads.  No pre / residual / row sums in this form. */
int te_small_gemm_batched_f32(float* c, const float* a, const float* b, const float* bias, int nz, int64_t za, int64_t zc,
                              int64_t zb, int64_t zbias, const int64_t* b_tab, const int64_t* bias_tab, int I, int J, int K,
                              int64_t sai, int64_t sak, int64_t sbk, int64_t sbj, int64_t sci, int64_t scj, float alpha,
                              float beta, int act, te_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * M1  demodulation coefficients (reference: ModulatedConv2d.forward, model_spatial_query.py:300-304 —
 * rsqrt(sum (scale * W * style)^2 + 1e-8) over B materialised weight copies).  Shared-weight form:
 *   wsq[co,ci] = wscale^2 sum_t w[co,ci,t]^2            (output, kept for the backward)
 *   d[b,co]    = rsqrt(sum_ci s[b,ci]^2 wsq[co,ci] + eps)
 * backward, u = -gd d^3 / 2:  gw[co,ci,t] = 2 wscale^2 w sum_b u[b,co] s[b,ci]^2,  gs[b,ci] = 2 s sum_co u[b,co] wsq[co,ci]
 * T == 0 in the forward: `w` already IS wsq[Co,Ci] (kept by the host for frozen weights), wsq output unused (may be NULL).
 * (gw / gs may be NULL; B <= 64 in the backward).  w [Co,Ci,T], s [B,Ci], d / gd [B,Co].  accumulate != 0: the results are
 * ADDED to gw / gs (which then hold the convolution's own dW / d style from te_wgrad_reduce_f32).
 */
int te_demod_fwd_f32(float* d, float* wsq, const float* w, const float* s, float wscale, float eps, int B, int Co, int Ci,
                     int T, te_stream_t stream);
int te_demod_bwd_f32(float* gw, float* gs, const float* gd, const float* d, const float* w, const float* wsq,
                     const float* s, float wscale, int B, int Co, int Ci, int T, int accumulate, te_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * T2  multi-tensor optimiser kernels: ONE launch over every parameter tensor of a module (reference:
 * train_spatial_query.py:458-473 — two torch.optim.Adam over 257 / 38 tensors — and accumulate(), :56-61, the
 * g_ema update; hundreds of tiny launches per iteration there).
 * The tensors are described by DEVICE tables:
 *   te_mt_adam_f32: table[5][n] (int64) = param ptr, grad ptr (0 = no gradient: tensor skipped), exp_avg ptr,
 *                   exp_avg_sq ptr, numel;  te_mt_ema_f32: table[3][n] = dst ptr, src ptr, numel;
 *   chunks[2][n_chunks] (int32) = tensor index, chunk index within the tensor; chunk c covers elements
 *   [chunk_index * chunk_elems, +chunk_elems) of its tensor (chunk_elems % 4 == 0).
 * Adam (no weight decay, no amsgrad), same operation order as torch.optim.Adam:
 *   m = lerp(m, g, 1 - beta1);  v = beta2 v + (1 - beta2) g^2;  p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 *   (beta1 == 0, the reference's setting: m is written (= g) but never read).  Hyper-parameters arrive as doubles (the
 *   bias corrections are formed in double on the host, as torch does) and are applied in fp32.
 * EMA: dst = dst * decay + (1 - decay) * src.
 */
int te_mt_adam_f32(const int64_t* table, const int32_t* chunks, int n_tensors, int n_chunks, int chunk_elems, double lr,
                   double beta1, double beta2, double eps, int step, te_stream_t stream);
int te_mt_ema_f32(const int64_t* table, const int32_t* chunks, int n_tensors, int n_chunks, int chunk_elems, double decay,
                  te_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-(sample, channel) scaling of an activation tensor and its adjoint: the elementwise pieces of the any-order composite
 * of the modulated convolution (style scale on the input, demodulation on the output: model_spatial_query.py:299-304 moved
 * from B weight copies onto the activations), used when a backward pass is recorded (path-length regulariser).
 *   te_chan_scale_f32: out[r, j] = x[r, j] * s[r]           rows r = (sample, channel), hw pixels each
 *   te_chan_dot_f32  : out[r]    = sum_j a[r, j] * b[r, j]   (fixed summation order)
 */
int te_chan_scale_f32(float* out, const float* x, const float* s, int64_t rows, int64_t hw, te_stream_t stream);
int te_chan_dot_f32(float* out, const float* a, const float* b, int64_t rows, int64_t hw, te_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TE_HIP_H */
