/*
 * sfhip.h - C ABI of libsfhip.so: hand-written gfx950 (MI355X) kernels for the streamflow
 * model's training / inference hot path.
 *
 * The reference project is pure PyTorch and has no FFI; each entry point below replaces the
 * stock ATen op(s) that the cited reference lines invoke (paths relative to the reference
 * checkout).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller unless stated otherwise;
 *   - the library never allocates or frees device memory and keeps no pointer after return;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*), nothing synchronises;
 *   - return 0 on success, a negative errno-style code otherwise; sf_last_error() returns the
 *     message of the calling thread's last failure;
 *   - activations inside the library are NHWC ("channels last") with the channel count padded
 *     to a multiple of 8; dtype codes: SF_F32 = 0, SF_BF16 = 1.
 */
#ifndef SFHIP_H
#define SFHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_DT_F32 0
#define SF_DT_BF16 1

#define SF_OUT_NHWC 0 /* out[pixel(j)][i]: conv output, NHWC, placed through sf_conv_geom */
#define SF_OUT_JI 1   /* out[j*ldo + i] */
#define SF_OUT_IJ 2   /* out[i*ldo + j] */
#define SF_OUT_OIHW 3 /* conv weight gradient, f32 OIHW [ldo][cin_real][RS], j = tap*C + c; rows i >= ldo dropped */

int sf_version(void);
const char* sf_last_error(void);

/* Geometry of a conv-style gather (B operand) and of the NHWC output placement. */
typedef struct sf_conv_geom {
    int N, IH, IW, C;       /* gathered tensor [N,IH,IW,C] (C = stored, padded channel count) */
    int MH, MW;             /* per-image grid of GEMM rows handled by this launch */
    int sh, sw;             /* gathered coordinate = grid coordinate * stride + tap offset */
    int OH, OW;             /* output tensor spatial size (SF_OUT_NHWC) */
    int osh, osw, ooy, oox; /* output coordinate = grid coordinate * ostride + offset */
    int ntaps;
    int dy[16], dx[16];     /* tap offsets in gathered coordinates */
    int tap_rs[16];         /* position of tap t inside the full R*S filter (SF_OUT_OIHW) */
    int cin_real, RS;       /* real input channels and R*S of the OIHW tensor (SF_OUT_OIHW) */
} sf_conv_geom;

/*
 * D[i][j] = sum_r A(i,r) * B(j,r) on the MFMA units (bf16 inputs: v_mfma_f32_32x32x16_bf16;
 * f32 inputs: v_mfma_f32_32x32x2_f32, i.e. exact f32 FMA chains), f32 accumulation.
 * One kernel family serves:
 *   - Conv2d forward and data-gradient as implicit GEMM (A = packed weights [K][taps*C],
 *     B = conv-gathered NHWC pixels): nn.Conv2d at src/model/cnn_encoder.py:45-51,
 *     src/model/decoder.py:148-152, src/model/gan.py:26-51;
 *   - Conv2d weight gradient (A = dOut [pixel][K], B = gathered input [pixel][tap*C], both
 *     reduction-major);
 *   - the two 262144-wide nn.Linear layers and their gradients
 *     (src/model/transformer_module.py:73, src/model/lstm_module.py:36,
 *     src/model/__init__.py:92-95).
 */
typedef struct sf_gemm_desc {
    int dtype;              /* element type of A and B */
    int a_kmajor, b_kmajor; /* 0: stored [row][r]; 1: stored [r][row] */
    int b_gather;           /* 1: B rows (b_kmajor=0) / lines (b_kmajor=1) are gathered pixels */
    long long I, J, R;
    long long lda, ldb, ldo;
    int nsplit;             /* >1: reduction split over blocks, f32 atomic accumulation into out */
    int out_mode;
    int out_dtype;
    int act;                /* 0 none, 1 ReLU, 2 LeakyReLU(0.2); after bias; nsplit == 1 only */
    int aux_mode;           /* 0 none; 1: out *= (aux > 0); 2: out *= (aux > 0 ? 1 : 0.2) */
    sf_conv_geom g;
    long long split_stride; /* with nsplit > 1: 0 = atomic accumulation into out; > 0 = split s stores its
                             * partial result at out + s*split_stride floats with plain stores (all nsplit
                             * slices are defined on return; the caller sums them, e.g. sf_colsum_f32) */
} sf_gemm_desc;

int sf_gemm(const sf_gemm_desc* d, const void* A, const void* B, const float* bias,
            const void* aux, void* out, void* stream);

/*
 * n <= 4 independent sf_gemm problems in one launch (more workgroups in flight than four small
 * launches): used for the four output-parity classes of the PatchGAN's stride-2 data gradients
 * (transposed nn.Conv2d(k=4,s=2,p=1), src/model/gan.py:20-33).  d: array of n descriptors; A, B,
 * out: arrays of n pointers; bias, aux: arrays of n pointers or NULL.  All problems: row-major
 * operands, nsplit 1, one dtype, I on the same side of 64.
 */
int sf_gemm_batch(int n, const sf_gemm_desc* d, const void* const* A, const void* const* B,
                  const float* const* bias, const void* const* aux, void* const* out, void* stream);

/* sf_conv3x3 that also writes dot_out[pixel] = dot_b[0] + sum_k dot_w[k] * out[pixel][k] (f32, one
 * value per pixel, from the values as stored in bf16): the decoder's ReLU -> 1x1 output conv
 * (src/model/decoder.py:152,175) folded into conv2's epilogue, so that conv2's output is written
 * for the backward pass but not read back.  Only for shapes that run on the kernel with resident
 * weights (sf_conv3x3_dot_ok: C = 64, K <= 64 in 8-channel groups, bf16 output); dot_w holds K floats. */
int sf_conv3x3_dot_ok(int C, int K, int ldo, int out_dtype);
/* Diagnostic (host only, no GPU work): the pixel tile th x tw the persistent bf16 3x3 kernels use on
 * H x W images with LDS rows of row_bytes (32: the 128-output-channel shape, 64: the 64-channel
 * shapes), and, when read_cycles is not NULL, the LDS cycles per patch-fragment read that tile
 * costs (4.0 = free of bank conflicts).  No counterpart in the reference. */
int sf_conv3x3_tile_shape(int H, int W, int row_bytes, int* th, int* tw, double* read_cycles);
int sf_conv3x3_dot(const void* in, const void* wp, const float* bias, const void* aux, void* out, int N,
                   int H, int W, int C, int K, int ldo, const int* dy, const int* dx, int act, int aux_mode,
                   int out_dtype, const float* dot_w, const float* dot_b, float* dot_out, void* stream);

/*
 * 3x3 stride-1 convolution, bf16 NHWC, forward or data gradient (nn.Conv2d(k=3,p=1) at
 * src/model/cnn_encoder.py:45-51 and src/model/decoder.py:148-150): the (TH+2)x(TW+2) halo
 * patch of each input-channel chunk is staged in LDS once and shared by all nine taps.
 * in [N,H,W,C] (C % 16 == 0); wp packed [K][9][C]; tap t reads the input at offset
 * (dy[t], dx[t]) in [-1,1]^2 (host arrays of 9 ints); out NHWC [N,H,W,ldo] (out_dtype);
 * bias f32 [K] or NULL; act / aux / aux_mode as in sf_gemm_desc.
 * w_staged != 0: wp is the output of sf_conv3x3_stage_weights (the per-stage LDS images of the
 * weights, so that every 1 KiB copy piece reads contiguous memory); only shapes for which
 * sf_conv3x3_staged_weight_bytes(C, K) > 0 take it.
 */
int sf_conv3x3(const void* in, const void* wp, const float* bias, const void* aux, void* out, int N,
               int H, int W, int C, int K, int ldo, const int* dy, const int* dx, int act,
               int aux_mode, int out_dtype, int w_staged, void* stream);
size_t sf_conv3x3_staged_weight_bytes(int C, int K);
int sf_conv3x3_stage_weights(const void* wp, void* out, int C, int K, void* stream);

/*
 * 4x4 stride-2 pad-1 convolution, bf16 NHWC, forward or data gradient (the PatchGAN blocks,
 * nn.Conv2d(k=4,s=2,p=1) at src/model/gan.py:24-34), on the same persistent LDS-DMA scheme: the
 * filter splits into four 2x2 filters by pixel parity, each stage stages one parity plane's patch.
 * The weights go in as the stages' LDS images (so that every 1 KiB copy piece reads contiguous
 * memory): ws = sf_conv4x4s2_stage_weights(wp) of sf_conv4x4s2_staged_weight_bytes(C, K) bytes, from
 * dgrad = 0: in [N,IH,IW,C] -> out [N,OH,OW,ldo], OH = (IH-2)/2+1; wp packed [K][16][C] with tap
 *   index (py*2+px)*4 + a*2+b holding filter position (2a+py, 2b+px).
 * dgrad = 1: in = gradient of the conv output [N,IH,IW,C], out = gradient of the conv input
 *   [N,OH,OW,ldo], IH = (OH-2)/2+1; wp packed transposed [K = conv input channels][16][C = conv
 *   output channels], tap index (py*2+px)*4 + a*2+b holding filter position (3-py-2a, 3-px-2b).
 * C % 16 == 0, K % 8 == 0, K <= 1024; bias f32 [K] or NULL; act / aux / aux_mode as in
 * sf_gemm_desc (aux: bf16 tensor shaped like out).
 */
size_t sf_conv4x4s2_staged_weight_bytes(int C, int K);
int sf_conv4x4s2_stage_weights(const void* wp, void* ws, int C, int K, void* stream);
int sf_conv4x4s2(const void* in, const void* ws, const float* bias, const void* aux, void* out, int N,
                 int IH, int IW, int C, int OH, int OW, int K, int ldo, int dgrad, int act,
                 int aux_mode, void* stream);

/* Process-wide knob of sf_wgrad3x3 / sf_wgrad4x4s2.  dynamic = 0 (default): every workgroup streams
 * a fixed slice of the tiles and the result is bit-reproducible.  dynamic = 1: tiles are dealt at run
 * time, so that a workgroup which starts late or shares its CU with another kernel (RCCL's, in
 * data-parallel training) does less instead of stretching the launch -- 64 x 64 blocks (layers of at
 * most 64 output channels): the slice is the workgroup's own list and a workgroup that has finished
 * it takes tiles from the lists of the others; 128 x 64 blocks (all other layers): the workgroups of
 * one XCD and block draw from one queue.  The partition of the sums then differs from the fixed one
 * (last-bit differences).  dynamic = 2: as 1, and on 64 x 64 blocks every second workgroup leaves its
 * whole list to the others (a self-test of the stealing, for the test-suite).  No counterpart in
 * the reference (it has no data-parallel path). */
int sf_set_wgrad_scheduling(int dynamic);
/*
 * Convolution weight gradients, bf16 NHWC, f32 OIHW result (overwritten):
 *   sf_wgrad3x3:   3x3 stride 1 pad 1 (src/model/cnn_encoder.py:45-51, src/model/decoder.py:148-150)
 *                  dw[k][c][r*3+s] = sum_pixels dout[y][x][k] * x[y+r-1][x+s-1][c]
 *   sf_wgrad4x4s2: 4x4 stride 2 pad 1 (src/model/gan.py:26-31)
 *                  dw[k][c][r*4+s] = sum_pixels dout[oy][ox][k] * x[2*oy+r-1][2*ox+s-1][c]
 * x [N,H,W,Cx], dout [N,OH,OW,Kd] (Cx, Kd = channel strides, multiples of 8); K, Cin = real
 * channel counts of dw.  ws: caller-provided scratch of *_workspace_bytes(...) bytes holding
 * per-slice partial sums that are reduced in a fixed order (bitwise reproducible, no atomics).
 * One image of x and of dout must stay below 2 GiB (32-bit offsets inside an image).
 */
size_t sf_wgrad3x3_workspace_bytes(int N, int H, int W, int K, int Cin);
int sf_wgrad3x3(const void* x, const void* dout, float* dw, float* ws, size_t ws_bytes, int N, int H,
                int W, int Cx, int Kd, int K, int Cin, void* stream);
size_t sf_wgrad4x4s2_workspace_bytes(int N, int IH, int IW, int K, int Cin);
int sf_wgrad4x4s2(const void* x, const void* dout, float* dw, float* ws, size_t ws_bytes, int N, int IH,
                  int IW, int Cx, int Kd, int K, int Cin, void* stream);

/* ---- layout / packing -------------------------------------------------------------- */
/* NCHW f32 -> NHWC (dtype) with channels zero-padded to Cpad. (images.view at src/model/__init__.py:125) */
int sf_nchw_to_nhwc(const float* in, void* out, int dtype, int N, int C, int H, int W, int Cpad,
                    void* stream);
/* the same layout change for a bf16 NCHW source (the gradient that arrives at the encoder from the
 * bf16 Linear layer): bf16 [N,C,H,W] -> bf16 [N,H,W,Cpad], Cpad >= 64 */
int sf_nchw_bf16_to_nhwc(const void* in, void* out, int N, int C, int H, int W, int Cpad, void* stream);
/* sf_nchw_to_nhwc that also performs the input check of the training loop (src/train.py:193,
 * torch.isnan(images).any() or torch.isinf(images).any()): *flag |= 1 if any element of `in` is Inf
 * or NaN; the flag is never cleared here.  Few-channel bf16 inputs are checked inside the layout
 * kernel (no second read of the images), other shapes by a pass of sf_nan_inf_flag. */
int sf_nchw_to_nhwc_checked(const float* in, void* out, int dtype, int N, int C, int H, int W,
                            int Cpad, int* flag, void* stream);
/* NHWC (dtype, Cpad) -> NCHW f32 (first C channels). */
int sf_nhwc_to_nchw(const void* in, int dtype, float* out, int N, int C, int H, int W, int Cpad,
                    void* stream);
/* OIHW f32 weight -> [rows][ntaps][cols_pad] (dtype).  transpose=0: rows=O, cols=I;
 * transpose=1: rows=I, cols=O (data-gradient operand).  tap t reads filter position tap_rs[t]
 * (host array of ntaps ints). */
int sf_pack_conv_weight(const float* w, void* out, int dtype, int O, int I, int RS, int transpose,
                        int ntaps, const int* tap_rs, int rows_pad, int cols_pad, void* stream);
/* f32 [n] -> dtype [n] */
int sf_cast(const float* in, void* out, int dtype, long long n, void* stream);

/* ---- BatchNorm / activation / pooling (src/model/cnn_encoder.py:54-61,69,88-91; src/model/gan.py:32-33) */
/* Reduction workspaces of this group: SF_BN_WS_DOUBLES(C) doubles.  Every workgroup writes its
 * partial sums to a row of its own and a fold kernel adds the rows in a fixed order into the first
 * 2*C doubles, which is what the consumers (sf_bn_finalize, sf_bn_bwd_apply) read: no atomics, no
 * zeroing, bit-reproducible sums. */
#define SF_BN_WS_PARTS 1024
#define SF_BN_WS_DOUBLES(C) (2 * (size_t)(C) * (1 + SF_BN_WS_PARTS))
/* Per-channel sum and sum of squares of x[M][C] into ws[0..2C). */
int sf_bn_stats(const void* x, int dtype, long long M, int C, double* ws, void* stream);
/* training=1: batch statistics from ws (biased var for normalisation; running stats updated
 * with momentum and the unbiased var).  training=0: running statistics.
 * Outputs scale = gamma*invstd, shift = beta - mean*scale, and mean / invstd (each [C]). */
int sf_bn_finalize(const double* ws, long long M, int C, int Creal, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum,
                   float eps, int training, float* scale, float* shift, float* mean,
                   float* invstd, void* stream);
/* y = maxpool2x2(relu(x*scale+shift)); x [N,H,W,C] -> y [N,H/2,W/2,C] */
int sf_bn_relu_maxpool_fwd(const void* x, int dtype, const float* scale, const float* shift,
                           void* y, int N, int H, int W, int C, void* stream);
/* y = adaptive_avg_pool(relu(x*scale+shift)) -> NCHW (out_dtype) [N,Creal,OH,OW]; OH==H: no pooling */
int sf_bn_relu_avgpool_fwd(const void* x, int dtype, const float* scale, const float* shift,
                           void* y, int out_dtype, int N, int H, int W, int C, int Creal, int OH,
                           int OW, void* stream);
/* y = leaky_relu(x*scale+shift, 0.2), NHWC -> NHWC */
int sf_bn_lrelu_fwd(const void* x, int dtype, const float* scale, const float* shift, void* y,
                    long long M, int C, void* stream);
/* y = relu(x*scale+shift), NHWC -> NHWC (the Conv-BN-ReLU blocks of LightweightCNNReducer,
 * src/model/baseline_models.py:37-50) */
int sf_bn_relu_fwd(const void* x, int dtype, const float* scale, const float* shift, void* y,
                   long long M, int C, void* stream);
/*
 * BatchNorm backward fused with the backward of what follows it.  mode 0: ReLU+maxpool2x2
 * (dy = gradient wrt the pooled map, NHWC dtype); mode 1: ReLU+adaptive avgpool (dy = NHWC
 * dtype [N,OH,OW,C]); mode 2: LeakyReLU(0.2) (dy NHWC dtype); mode 3: ReLU, no pooling (dy NHWC dtype).
 * Pass 1 (sf_bn_bwd_reduce) leaves sum(dz) and sum(dz*xhat) per channel in ws[0..2C)
 * (ws: SF_BN_WS_DOUBLES(C) doubles); pass 2 (sf_bn_bwd_apply) writes
 * dx = scale*(dz - mean(dz) - xhat*mean(dz*xhat)) (training) or scale*dz (eval) and
 * dgamma = sum(dz*xhat), dbeta = sum(dz).
 */
int sf_bn_bwd_reduce(int mode, const void* x, int dtype, const void* dy, int dy_dtype,
                     const float* scale, const float* shift, const float* mean,
                     const float* invstd, double* ws, int N, int H, int W, int C, int Creal,
                     int OH, int OW, void* stream);
int sf_bn_bwd_apply(int mode, const void* x, int dtype, const void* dy, int dy_dtype,
                    const float* scale, const float* shift, const float* mean,
                    const float* invstd, const float* gamma, const double* ws, int training,
                    void* dx, float* dgamma, float* dbeta, int N, int H, int W, int C, int Creal,
                    int OH, int OW, void* stream);
/*
 * The last encoder layer's BatchNorm -> ReLU -> AdaptiveAvgPool2d in training
 * (src/model/cnn_encoder.py:54-56,69,88-91) with the reduction pass of its backward taken off the
 * pre-BN map.  Both channel sums are linear in the pooled gradient g:
 *   sum_p dz      = sum_bin g * pos / |bin|,   pos = number of pixels of the bin with x*scale+shift > 0
 *   sum_p dz*xhat = sum_bin g * xh,            xh  = (1/|bin|) * sum over those pixels of (x-mean)*invstd
 * sf_bn_relu_avgpool_fwd_moments is sf_bn_relu_avgpool_fwd that also writes pos (uint16) and xh
 * (dtype), both NHWC [N,OH,OW,C]; OW <= 63, bins of at most 65535 pixels.
 * sf_bn_bwd_reduce_pooled leaves the two sums in ws[0..2C) from dy [N,OH,OW,C] (dtype), pos and
 * xh, exactly where sf_bn_bwd_reduce(mode 1) leaves them; sf_bn_bwd_apply(mode 1) follows.
 */
int sf_bn_relu_avgpool_fwd_moments(const void* x, int dtype, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, void* y, int out_dtype,
                                   void* pos, void* xh, int N, int H, int W, int C, int Creal,
                                   int OH, int OW, void* stream);
int sf_bn_bwd_reduce_pooled(const void* dy, int dtype, const void* pos, const void* xh, double* ws,
                            int N, int H, int W, int C, int OH, int OW, void* stream);
/* out[c] = sum_m x[m][c] (bias gradients); out f32 [Creal]; ws: SF_BN_WS_DOUBLES(C) doubles. */
int sf_channel_sum(const void* x, int dtype, long long M, int C, int Creal, float* out, double* ws,
                   void* stream);

/* ---- first encoder layer without its pre-activation tensor --------------------------
 * Conv2d(C<=16,K,3,padding=1,bias=False) -> BatchNorm2d -> ReLU -> MaxPool2d(2)
 * (src/model/cnn_encoder.py:42-50, layer 0), bf16.  x NHWC [N,H,W,16] (channels zero-padded to
 * 16); w packed [K][9][16] (sf_pack_conv_weight); K = 128 or 256.  The conv output is
 * recomputed in each pass and never stored:
 *   sf_enc1_stats: ws[0..K) = sum, ws[K..2K) = sum of squares of the bf16-rounded conv output
 *                  (ws zeroed here; feed sf_bn_finalize with M = N*H*W)
 *   sf_enc1_fwd:   y[N,H/2,W/2,K] = maxpool2x2(relu(conv*scale + shift)); when `codes` is not
 *                  NULL (sf_enc1_codes_bytes bytes) it also leaves, per pooling window and
 *                  channel, which of the four positions held the maximum (the first on ties,
 *                  as torch's max-pool backward picks it) and whether it passed the ReLU */
int sf_enc1_stats(const void* x, const void* w, double* ws, int N, int H, int W, int K, void* stream);
size_t sf_enc1_codes_bytes(int N, int H, int W, int K);
int sf_enc1_fwd(const void* x, const void* w, const float* scale, const float* shift, void* y,
                void* codes, int N, int H, int W, int K, void* stream);
/*   sf_enc1_bwd:   from dy (gradient wrt the pooled map, NHWC bf16 [N,H/2,W/2,K]) the gradients
 *                  dw f32 [K,Creal,3,3], dgamma, dbeta f32 [K] of conv weight and BatchNorm
 *                  (training != 0: batch statistics took part in the forward); the input gets
 *                  no gradient.  Creal <= 14 real input channels; ws: 8-byte aligned workspace of
 *                  sf_enc1_bwd_workspace_bytes bytes (per-workgroup partial results, summed in
 *                  a fixed order: reproducible).  codes: what sf_enc1_fwd left for the same x, w,
 *                  scale and shift, or NULL.  With codes (and Creal <= 10) the pooled gradient is
 *                  routed by them and everything that involves the conv output comes from two
 *                  channel-independent sums (the correlation of the routed gradient with the
 *                  input patches, and the patches' Gram matrix), so neither the conv nor the
 *                  arg-max is computed again; without, both are recomputed from x.  gram: the
 *                  record of sf_enc1_gram for the same x, or NULL (only read with codes). */
size_t sf_enc1_bwd_workspace_bytes(int N, int H, int W, int K, int Creal);
int sf_enc1_bwd(const void* x, const void* w, const void* dy, const void* codes, const double* gram,
                const float* scale, const float* shift, const float* mean, const float* invstd,
                int training, float* dw, float* dgamma, float* dbeta, float* ws, size_t ws_bytes,
                int N, int H, int W, int K, int Creal, void* stream);
/* The Gram record of the layer's input: G = sum over the output pixels of xp xp^T, xp the 9 * Creal
 * values of the pixel's 3x3 input patch (upper-triangle 32 x 32 tiles), followed by the sums of the
 * input over the image, its border rows / columns and corners (which give sum xp).  Both are
 * independent of the weights and of the output channel, and everything the layer needs of its conv
 * output z = W xp beyond single pixels is made of them:
 *   sf_enc1_stats_gram: BatchNorm statistics, sum z = W . sum xp, sum z^2 = W G W^T (the record
 *       sf_enc1_stats writes, but of the unrounded z) -- no pass over the conv output at all;
 *   sf_enc1_bwd(gram = rec): the conv-output terms of the BatchNorm backward (otherwise that call
 *       accumulates the record itself).
 * Creal <= 10.  rec: SF_ENC1_GRAM_DOUBLES doubles; ws: sf_enc1_gram_workspace_bytes bytes. */
#define SF_ENC1_GRAM_DOUBLES 6400
size_t sf_enc1_gram_workspace_bytes(int N, int H, int W, int Creal);
int sf_enc1_gram(const void* x, int N, int H, int W, int Creal, double* rec, float* ws,
                 size_t ws_bytes, void* stream);
int sf_enc1_stats_gram(const void* w, const double* rec, int K, int Creal, double* ws, void* stream);

/* ---- bilinear resize, align_corners=False (src/model/decoder.py:165-170, src/loss.py:49-54) */
/* src NHWC f32 [B,h,w,C] -> dst NHWC (dtype) [B,OH,OW,C]; C == Cpad */
int sf_bilinear_fwd(const void* src, int src_dtype, void* dst, int dtype, int B, int C, int h,
                    int w, int OH, int OW, int Cpad, void* stream);
/* ddst NHWC [B,OH,OW,C] -> dsrc NHWC f32 [B,h,w,C]; tmp: B*OH*w*C floats (separable:
 * reduce along x into tmp, then along y) */
int sf_bilinear_bwd(const void* ddst, int dtype, float* dsrc, float* tmp, int B, int C, int h,
                    int w, int OH, int OW, int Cpad, void* stream);

/* ---- 3x3 conv of a bilinearly resized map, resized map never materialised ------------
 * (src/model/decoder.py:165-173: F.interpolate(x, (OH,OW), mode='bilinear',
 * align_corners=False) followed by Conv2d(C,K,3,padding=1) [+ReLU]).  The channel mixing is
 * done first, on the source pixels, by the caller (a GEMM):
 *   y9 f32 [B,h,w,9,K], y9[b][s][t][k] = sum_c W[k][c][t] x[b][s][c]   (t = 3*r + s of the filter)
 * sf_upconv3_fwd: out[b][p][k] = act(bias[k] + sum_t resize(y9[..][t][..])[p + tap_t][k]), zero
 * outside the resized image; out NHWC dtype [B,OH,OW,K]; act SF_ACT_NONE or SF_ACT_RELU.
 * sf_upconv3_bwd: the adjoint, g[b][s][t][k] = sum_p lambda(p + tap_t -> s) dz[b][p][k]
 * (g NHWC g_dtype [B,h,w,9,K]; dz NHWC dtype [B,OH,OW,K]; tmp: B*OH*3*w*K floats).  From g the
 * caller gets dW = g^T x, dx = g W and db = column sums of g's centre tap with GEMMs. */
int sf_upconv3_fwd(const float* y9, const float* bias, void* out, int dtype, int B, int h, int w,
                   int OH, int OW, int K, int act, void* stream);
int sf_upconv3_bwd(const void* dz, int dtype, float* tmp, void* g, int g_dtype, int B, int h,
                   int w, int OH, int OW, int K, void* stream);

/* ---- small convolutions that do not map onto MFMA tiles ---------------------------- */
/* First PatchGAN layer: Conv2d(1,K,4,s2,p1)+bias+LeakyReLU (src/model/gan.py:37-41).
 * x f32 [N,1,H,W]; w f32 [K,1,4,4]; y NHWC dtype [N,OH,OW,K]. */
int sf_dconv1_fwd(const float* x, const float* w, const float* b, void* y, int dtype, int N,
                  int H, int W, int K, void* stream);
/* dz NHWC dtype [N,OH,OW,K] (gradient wrt the pre-activation) -> dx f32 [N,1,H,W] */
int sf_dconv1_bwd_data(const void* dz, int dtype, const float* w, float* dx, int N, int H, int W,
                       int K, void* stream);
/* dw f32 [K,1,4,4], db f32 [K] (both zeroed here, accumulated with atomics) */
int sf_dconv1_bwd_weight(const float* x, const void* dz, int dtype, float* dw, float* db, int N,
                         int H, int W, int K, void* stream);
/* The same gradients (src/model/gan.py:37-41 under loss.backward()) with a caller-provided
 * workspace of sf_dconv1_bwd_weight_workspace_bytes(N, H, W, K) bytes (0 when the shape has no
 * use for one): per-workgroup partial sums stored there and added in a fixed order instead of
 * atomics (bit-reproducible, and the atomics' contention was most of the pass).  ws may be null:
 * then this is sf_dconv1_bwd_weight. */
size_t sf_dconv1_bwd_weight_workspace_bytes(int N, int H, int W, int K);
int sf_dconv1_bwd_weight_ws(const float* x, const void* dz, int dtype, float* dw, float* db,
                            float* ws, size_t ws_bytes, int N, int H, int W, int K, void* stream);
/* Single-output-channel conv: Conv2d(C,1,R,s1,pad)+bias (src/model/gan.py:48-51 with R=4,pad=1;
 * src/model/decoder.py:152 with R=1,pad=0).  x NHWC dtype [N,H,W,C]; w f32 [1,Creal,R,R];
 * y f32 [N,1,OH,OW]. */
int sf_conv_k1_fwd(const void* x, int dtype, const float* w, const float* b, float* y, int N,
                   int H, int W, int C, int Creal, int R, int pad, void* stream);
/* sf_conv_k1_fwd with a caller-provided workspace (sf_conv_k1_fwd_workspace_bytes; 0 = the shape has
 * no use for one and the call forwards to sf_conv_k1_fwd): for the bf16 4x4 PatchGAN head
 * (src/model/gan.py:33) the workspace takes the per-input-pixel tap products [N*H*W][16] of an MFMA
 * pass that reads the input once, and a second launch sums the 16 shifted products per output. */
size_t sf_conv_k1_fwd_workspace_bytes(int dtype, int N, int H, int W, int C, int R);
int sf_conv_k1_fwd_ws(const void* x, int dtype, const float* w, const float* b, float* y, void* ws,
                      size_t ws_bytes, int N, int H, int W, int C, int Creal, int R, int pad, void* stream);
/* dx NHWC dtype [N,H,W,C] = sum_taps dy * w, optionally multiplied by (aux>0) (ReLU mask) */
int sf_conv_k1_bwd_data(const float* dy, const float* w, const void* aux, void* dx, int dtype,
                        int N, int H, int W, int C, int Creal, int R, int pad, void* stream);
/* dw f32 [1,Creal,R,R], db f32 [1] (zeroed here) */
int sf_conv_k1_bwd_weight(const void* x, int dtype, const float* dy, float* dw, float* db, int N,
                          int H, int W, int C, int Creal, int R, int pad, void* stream);
/* Backward of ReLU -> Conv2d(C,1,1)+bias (src/model/decoder.py:151-152) in one pass over x (the
 * ReLU output, NHWC dtype [M,C]) and dy (f32 [M]): dx[p][c] = dy[p]*w[c]*(x[p][c] > 0) (dtype),
 * dw[c] = sum_p dy[p]*x[p][c], db = sum_p dy[p], dsum[c] = sum_p dx[p][c] (the bias gradient of
 * the conv under the ReLU).  dw, dsum f32 [Creal], db f32 [1]; C / (16-byte chunk) a power of two. */
int sf_conv_k1_bwd_fused(const void* x, int dtype, const float* dy, const float* w, void* dx,
                         float* dw, float* db, float* dsum, long long M, int C, int Creal,
                         void* stream);

/* ---- temporal modules ---------------------------------------------------------------- */
/* One post-LN nn.TransformerEncoderLayer (relu FFN, batch_first) forward / backward for
 * sequences of T <= 8 steps, d_model <= 128, nhead heads (src/model/transformer_module.py:81-93).
 * w: the 12 parameter tensors in state_dict order (in_proj_weight, in_proj_bias,
 * out_proj.weight, out_proj.bias, linear1.weight, linear1.bias, linear2.weight, linear2.bias,
 * norm1.weight, norm1.bias, norm2.weight, norm2.bias), host array of device pointers.
 * save: per-sample activations for backward (sf_transformer_save_floats(T,D,F) floats each).
 * p_drop / seed: dropout probability (training) and Philox-free counter hash seed. */
size_t sf_transformer_save_floats(int T, int D, int F);
int sf_transformer_layer_fwd(const float* x, float* y, const float* const* w, float* save, int B,
                             int T, int D, int F, int nhead, float p_drop,
                             unsigned long long seed, void* stream);
/* Offset (floats per sample) of saved array `which` (0 qkv [T,3D], 1 probs [8,T,T], 2 ctx [T,D],
 * 3 xhat1, 4 rstd1 [8], 5 x1 [T,D], 6 hidden [T,F], 7 xhat2, 8 rstd2): array a of sample b starts at
 * save + B*offset_a + b*size_a, so each array is one contiguous [B*T, width] matrix. */
int sf_transformer_save_offset(int T, int D, int F, int which);
/* x: the layer's forward input.  dw: 12 f32 gradient tensors; only the VECTOR entries (biases,
 * LayerNorm weights: indices 1,3,5,7,8,9,10,11) are accumulated here (atomics; caller zeroes).
 * gtok (B*T*(5D+F) floats) receives the per-token gradients g_qkv [B*T,3D], g_sa [B*T,D],
 * g_z [B*T,F], g_ff [B*T,D], which the caller contracts with x, ctx, x1 and hidden over the
 * B*T tokens (sf_gemm, both operands reduction-major) to obtain the four weight-matrix gradients. */
int sf_transformer_layer_bwd(const float* x, const float* dy, float* dx, const float* const* w, float* const* dw,
                             const float* save, float* gtok, int B, int T, int D, int F, int nhead,
                             float p_drop, unsigned long long seed, void* stream);
/* x[B,T,D] + pe[T,D] (fwd) ; mean over T -> [B,D] (src/model/transformer_module.py:112,118) */
int sf_add_pe(const float* x, const float* pe, float* y, int B, int T, int D, void* stream);
int sf_mean_t_fwd(const float* x, float* y, int B, int T, int D, void* stream);
int sf_mean_t_bwd(const float* dy, float* dx, int B, int T, int D, void* stream);
/* nn.LSTM(H,H,L,batch_first), gate order i,f,g,o, zero initial state, returns h_n[-1]
 * (src/model/lstm_module.py:40-47,66-75).  w: per layer W_ih, W_hh, b_ih, b_hh (4*L pointers).
 * save: sf_lstm_save_floats(T,H,L) floats per sample. */
size_t sf_lstm_save_floats(int T, int H, int L);
int sf_lstm_fwd(const float* x, float* y, const float* const* w, float* save, int B, int T, int H,
                int L, float p_drop, unsigned long long seed, void* stream);
int sf_lstm_bwd(const float* x, const float* dy, float* dx, const float* const* w, float* const* dw,
                const float* save, int B, int T, int H, int L, float p_drop,
                unsigned long long seed, void* stream);

/* ---- losses ---------------------------------------------------------------------------- */
/* StreamflowDataset._lonlat_to_pixel (src/dataset.py:352-373): float64 arithmetic in the
 * reference's order, round-half-to-even, clamp.  lon/lat device f64 [n]; out int32 [n] each. */
int sf_lonlat_to_pixel(const double* lon, const double* lat, int n, double lon_min, double lon_max,
                       double lat_min, double lat_max, int W, int H, int* px, int* py,
                       void* stream);
/*
 * ImageLoss + StationLoss + their weighted sum (src/loss.py:47-70, 116-160, 227-230).
 * pred f32 [B,1,H,W]; target f32 [B,1,TH,TW] (bilinearly resized to HxW on the fly when the
 * sizes differ); stations packed: pos int32 [S][2] (x=column, y=row), runoff f32 [S],
 * sample offsets int32 [B+1].  metric 0 = MSE, 1 = MAE (image_metric / station_metric).
 *
 * sf_image_station_loss_fwd -- one launch, one read of pred and target.
 *   losses f32 [5]: total, image, station, 1 / (sum of the image loss's pixel weights),
 *     1 / (number of stations with a non-empty window) or 0 when there is none.
 *   station_diff f32 [S], nullable: window mean - runoff of every station (0 for an empty window),
 *     kept for the backward pass.
 *   nan_flag int32 [1], nullable: |= 1 when pred holds Inf / NaN (the prediction check of the
 *     training loop, src/train.py:198, which then needs no pass of its own); never cleared here.
 *   ws: SF_LOSS_WS_DOUBLES doubles of device scratch: per-block partial sums (folded in a fixed
 *     order by the block that finishes last: bit-reproducible losses) and, in its last
 *     SF_LOSS_WS_COUNTER_DOUBLES doubles, arrival counters that MUST BE ZERO before the first call
 *     using this workspace; every call leaves them zero.  Calls sharing a workspace must be
 *     ordered on one stream.
 * sf_image_station_loss_bwd -- one launch, one read of pred and target, one write of dpred.
 *   dpred f32 [B,1,H,W] = upstream * d total / d pred, complete (image term and the station
 *     windows' terms summed in registers before the one store: no atomics, bit-reproducible).
 *   upstream: device f32 [1], the gradient arriving at the total loss; NULL = 1.
 *   station_diff: what the forward call wrote, or NULL (the window means are gathered again).
 *   fwd_losses: the losses [5] the forward call on the same inputs wrote.
 * sf_image_station_loss -- both, back to back, upstream 1: losses f32 [4] = total, image,
 *   station, 1 (dpred is already divided by the weight sum; callers that multiply dpred by
 *   losses[3] stay correct).  Its ws may hold anything (the counters are cleared on the stream).
 */
#define SF_LOSS_WS_DOUBLES 4096
#define SF_LOSS_WS_COUNTER_DOUBLES 496
int sf_image_station_loss_fwd(const float* pred, const float* target, const int* pos,
                              const float* runoff, const int* offsets, int S, int B, int H, int W,
                              int TH, int TW, int image_metric, float nonzero_weight,
                              float zero_threshold, int ksize, int station_metric,
                              float image_weight, float station_weight, float* losses,
                              float* station_diff, int* nan_flag, double* ws, void* stream);
int sf_image_station_loss_bwd(const float* pred, const float* target, const int* pos,
                              const float* runoff, const int* offsets, int S, int B, int H, int W,
                              int TH, int TW, int image_metric, float nonzero_weight,
                              float zero_threshold, int ksize, int station_metric,
                              float image_weight, float station_weight, const float* upstream,
                              const float* station_diff, const float* fwd_losses, float* dpred,
                              void* stream);
int sf_image_station_loss(const float* pred, const float* target, const int* pos,
                          const float* runoff, const int* offsets, int S, int B, int H, int W,
                          int TH, int TW, int image_metric, float nonzero_weight,
                          float zero_threshold, int ksize, int station_metric, float image_weight,
                          float station_weight, float* losses, float* dpred, double* ws,
                          void* stream);
/*
 * The same computation for callers that cannot allocate device memory or pass a workspace (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart
 * in the reference.  The partial sums live in a 32 KB file-scope __device__ array of the library
 * (one copy per GPU, held until the process ends) whose per-device address is taken with
 * hipGetSymbolAddress on the first call on that device; the call itself allocates and frees
 * nothing.  Limits (SF_EINVAL beyond them): B*H*W <= SF_LOSS_STATIC_MAX_PIXELS and
 * S <= SF_LOSS_STATIC_MAX_STATIONS.  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (different streams or host threads) race; calls ordered on one stream do not
 * (every call rewrites all the slots it reads, and the arrival counter, zero when the library is
 * loaded, is put back to zero by the call itself).  `stream` must belong to the current device.
 * Make the first call on a device outside graph capture.  Results are those of
 * sf_image_station_loss.
 */
#define SF_LOSS_STATIC_MAX_PIXELS (1LL << 28)
#define SF_LOSS_STATIC_MAX_STATIONS (1 << 20)
int sf_image_station_loss_static(const float* pred, const float* target, const int* pos,
                                 const float* runoff, const int* offsets, int S, int B, int H,
                                 int W, int TH, int TW, int image_metric, float nonzero_weight,
                                 float zero_threshold, int ksize, int station_metric,
                                 float image_weight, float station_weight, float* losses,
                                 float* dpred, void* stream);
/* mean BCE-with-logits (src/model/gan.py:82,102-108) against `labels` (f32 [n]) or, when
 * labels is NULL, the constant `label`:
 * loss[0] = mean(max(x,0) - x*y + log1p(exp(-|x|))); dx (nullable) = (sigmoid(x) - y)/n. ws: 1 double. */
int sf_bce_logits(const float* x, long long n, float label, const float* labels, float* loss,
                  float* dx, double* ws, void* stream);
/* out[c] = sum_m x[m][c] for a short, very wide f32 matrix (bias gradient of the 262144-wide Linear) */
int sf_colsum_f32(const float* x, long long M, long long C, float* out, void* stream);

/* ---- optimiser --------------------------------------------------------------------------- */
/* torch.optim.Adam step (src/train.py:80-91; eps 1e-8, no weight decay), one fused pass over
 * a flat parameter range.  bias corrections are passed in (1-beta^t).  The betas are doubles: the
 * 1 - beta factors are formed in double and then rounded to f32, as torch does (1 - 0.999 formed
 * in f32 is off by 5e-5 of its value). */
int sf_adam(float* p, const float* g, float* m, float* v, long long n, float lr, double beta1,
            double beta2, float eps, float bc1, float bc2, float grad_scale, void* stream);
/* The same update for optimiser steps that must be skippable without a host synchronisation.  The
 * reference steps through GradScaler (src/train.py:255-260), which skips optimizer.step() when the
 * gradients hold Inf/NaN, so the step count of torch.optim.Adam does not advance either.  Here the
 * count lives on the device: state int32 [2] = {steps taken, steps skipped}, zero before the
 * first step.  sf_adam_begin, once per optimiser step: found_inf (device int32, nullable) != 0 ->
 * state[1] += 1, else state[0] += 1.  sf_adam_guarded, once per flat range of that step: nothing
 * is touched when found_inf[0] != 0; otherwise the Adam update with bias corrections
 * 1 - beta^t formed in double, t = state[0] + step_offset, and, when shadow_bf16 is not NULL, a
 * bf16 copy of the updated parameters (the operand the bf16 kernels read) written in the same
 * pass.  step_offset <= 0: minus the number of optimiser steps this range sat out for want of a
 * gradient (torch.optim.Adam counts steps per parameter); 0 for a range updated at every step. */
int sf_adam_begin(int* state, const int* found_inf, void* stream);
int sf_adam_guarded(float* p, const float* g, float* m, float* v, void* shadow_bf16, long long n,
                    float lr, double beta1, double beta2, float eps, float grad_scale,
                    const int* state, const int* found_inf, int step_offset, void* stream);
/* flag[0] |= 1 if any element of x is NaN/Inf (src/train.py:193,198,221) */
int sf_nan_inf_flag(const float* x, long long n, int* flag, void* stream);

/* ---- device-resident dataset ---------------------------------------------------------------- */
/* Per (date, channel) statistics of frames f32 [D,C,HW] held on the device, the per-date loop
 * of StreamflowDataset._compute_statistics (src/dataset.py:219-264): out f64 [D,C,4] = sum, sum
 * of float32 squares, min, max over the selected pixels (mask > 0.5 where mask_row[d] >= 0
 * names a row of masks f32 [n_masks,HW]; every pixel otherwise), NaN -> 0 and +-Inf -> +-FLT_MAX
 * first (numpy.nan_to_num defaults, :252-254).  count i64 [D] = selected pixels; bad0 i32 [D] =
 * non-finite values in channel 0.  masks and mask_row are both NULL when no date has a mask. */
int sf_dataset_plane_stats(const float* frames, const float* masks, const int* mask_row, int D,
                           int C, long long HW, double* out, long long* count, int* bad0,
                           void* stream);
/* The per-frame arithmetic of StreamflowDataset.__getitem__ (src/dataset.py:412-435,474-491)
 * applied once to every resident frame, in place: multiply by the date's mask, non-finite -> 0,
 * |channel 0| < flow_threshold -> 0, then (normalize != 0) (v - a[c]) / b[c] evaluated in
 * float64 and rounded to float32 (a, b device f64 [C]: mean/std, or min/(max-min)).  target
 * (nullable, f32 [D,HW]) receives channel 0 as the reference prepares its output image
 * (:441-472,487,491): mask and threshold, no NaN replacement, same normalisation. */
int sf_dataset_transform(float* frames, float* target, const float* masks, const int* mask_row,
                         const double* a, const double* b, int normalize, float flow_threshold,
                         int D, int C, long long HW, void* stream);
/* dst frame i (n_elems floats, contiguous) = src frame rows[i] (frames src_stride floats apart,
 * n_src of them; rows device i64 [n]; a row outside [0,n_src) leaves its frame untouched): the
 * stacking of np.stack / the collate function (src/dataset.py:438, src/train.py:381-382) for
 * windows served from the resident frames.  16-byte accesses when sizes and bases allow. */
int sf_gather_frames(const float* src, long long src_stride, long long n_src,
                     const long long* rows, int n, long long n_elems, float* dst, void* stream);

/* ---- evaluation metrics ---------------------------------------------------------------------- */
/* Per-date image metrics of Evaluator.evaluate_dataset (src/evaluate.py:205-238) for a batch of
 * predictions and targets, f32 [B,H,W] each, without a host sync.  out f64 [B,13] per date:
 * sum (p-t)^2, sum |p-t|, sum p, sum t, sum p^2, sum t^2, sum p*t, min p, max p, min t, max t,
 * the sum over all pixels of the SSIM map of calculate_ssim (src/evaluate.py:17-60: 11 x 11
 * uniform window, scipy 'reflect' border, K1 = 0.01, K2 = 0.03, data range
 * max(t.max - t.min, p.max - p.min) in float32), and that data range.  float64 sums in a fixed
 * order.  ws: sf_date_image_metrics_workspace_bytes(B, H, W) bytes of device memory. */
size_t sf_date_image_metrics_workspace_bytes(int B, int H, int W);
int sf_date_image_metrics(const float* pred, const float* target, int B, int H, int W, double* out,
                          void* ws, size_t ws_bytes, void* stream);

/* ---- baseline models ------------------------------------------------------------------------- */
/* y = dropout(relu(x), p) over n f32 elements (the MLP of PureBPModel,
 * src/model/baseline_models.py:181-190): kept elements scaled by 1/(1-p), the mask drawn from
 * `seed` per element index; p = 0 is a plain ReLU.  Backward: dx = dy/(1-p) where y > 0. */
int sf_relu_dropout_fwd(const float* x, float* y, long long n, float p, unsigned long long seed,
                        void* stream);
int sf_relu_dropout_bwd(const float* y, const float* dy, float* dx, long long n, float p,
                        void* stream);
/* nn.GRU(H, H, L, batch_first=True) of AttentionGRUModel (src/model/baseline_models.py:289-296,316):
 * x f32 [B,T,H] -> y f32 [B,T,H], the output sequence of the last layer (gate order r, z, n;
 * inter-layer dropout p_drop drawn from seed).  w: 4*L device pointers weight_ih_l, weight_hh_l
 * [3H,H], bias_ih_l, bias_hh_l [3H].  save (nullable in inference): B * sf_gru_save_floats floats.
 * Backward: dy f32 [B,T,H] -> dx; dw (same order as w) accumulated with atomics, zeroed by the
 * caller.  T <= 8, H % 4 == 0, L <= 4. */
size_t sf_gru_save_floats(int T, int H, int L);
int sf_gru_fwd(const float* x, float* y, const float* const* w, float* save, int B, int T, int H,
               int L, float p_drop, unsigned long long seed, void* stream);
int sf_gru_bwd(const float* x, const float* dy, float* dx, const float* const* w, float* const* dw,
               const float* save, int B, int T, int H, int L, float p_drop,
               unsigned long long seed, void* stream);
/* TemporalAttention.forward (src/model/baseline_models.py:221-247): Q, K, V = Linear(g),
 * softmax(Q K^T / sqrt(H)) V averaged over time: g f32 [B,T,H] -> out f32 [B,H].  w = {Wq, bq, Wk,
 * bk, Wv, bv}; save: B * sf_attn_save_floats floats; dw accumulated with atomics. */
size_t sf_attn_save_floats(int T, int H);
int sf_attn_fwd(const float* g, float* out, const float* const* w, float* save, int B, int T, int H,
                void* stream);
int sf_attn_bwd(const float* g, const float* dout, float* dg, const float* const* w,
                float* const* dw, const float* save, int B, int T, int H, void* stream);
/* Four weight-matrix gradients over n tokens in one launch: out[q] f32 [I[q]][J[q]] =
 * sum_t g[q][t][:]^T a[q][t][:] with g[q] f32 [n][I[q]] (per-token gradients) and a[q] f32 [n][J[q]]
 * (saved activations): the in_proj, out_proj, linear1 and linear2 weight gradients of a
 * TransformerEncoderLayer (src/model/transformer_module.py:83-91).  Sizes multiples of 64. */
int sf_token_outer4(const float* const* g, const float* const* a, float* const* out, const int* I,
                    const int* J, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
