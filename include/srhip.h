/*
 * srhip.h -- C ABI of libsrhip.so, the MI355X (gfx950) implementation of the
 * convolutional super-resolution hot path.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless a parameter is documented as host.
 *   - The caller owns every buffer (inputs, outputs, packed weights); the library
 *     never transfers ownership and never synchronises the device.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Every function returns 0 on success and a negative srhip_status on error;
 *     srhip_last_error() returns a thread-local, human-readable message.
 *   - Activations travel between kernels as NHWC bf16 with a channel count that
 *     is a multiple of 16 (zero padded); the reference's fp32 NCHW tensors enter
 *     and leave through the layout kernels below.
 *
 * The reference has no native interface for this path (it calls torch.nn
 * operators); each entry point cites the reference operator call it replaces.
 */
#ifndef SRHIP_H
#define SRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRHIP_API __attribute__((visibility("default")))

typedef enum {
    SRHIP_OK = 0,
    SRHIP_ERR_INVALID = -1,   /* bad argument / unsupported shape */
    SRHIP_ERR_LAUNCH = -2,    /* HIP runtime error */
    SRHIP_ERR_DEVICE = -3     /* not a gfx950 device / no device */
} srhip_status;

typedef enum { SRHIP_ACT_NONE = 0, SRHIP_ACT_RELU = 1, SRHIP_ACT_LRELU = 2 } srhip_act;
typedef enum { SRHIP_BILINEAR = 0, SRHIP_NEAREST = 1 } srhip_interp;
typedef enum { SRHIP_RES_ADD = 0, SRHIP_RES_MASK = 1, SRHIP_RES_PRE = 2 } srhip_res_mode;
typedef enum { SRHIP_AUX_NONE = 0, SRHIP_AUX_EMIT = 1, SRHIP_AUX_MASKBITS = 2, SRHIP_AUX_DUALBITS = 3 } srhip_aux_mode;

SRHIP_API int srhip_version(void);
SRHIP_API const char* srhip_last_error(void);

/* ---------------------------------------------------------------------------
 * Layout kernels (boundary between the reference's fp32 NCHW tensors and the
 * internal NHWC bf16 activations).  HBM-bound.
 * ------------------------------------------------------------------------- */
/* src (N,C,H,W) fp32 -> dst (N,H,W,cpad) bf16, channels [C,cpad) zero filled.
 * Replaces the implicit layout of the tensors handed to nn.Conv2d at
 * core/models/model.py:161,167. */
SRHIP_API int srhip_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int n, int c, int h, int w,
                                          int cpad, void* stream);
/* src (N,H,W,cstride) bf16 -> dst (N,C,H,W) fp32 (first C channels). */
SRHIP_API int srhip_nhwc_bf16_to_nchw_f32(const void* src, float* dst, int n, int c, int h, int w,
                                          int cstride, void* stream);

/* ---------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolution on CDNA4 MFMA (implicit GEMM, bf16 in,
 * fp32 accumulate).  Replaces nn.Conv2d(+ReLU/LeakyReLU, + residual) at
 * core/models/model.py:10-14 (ConvActivation), :17-19 (Residual), :22-33
 * (SRResidualBlock: 0.2*f(x)+x), :111-115 (final_conv) and the 3x3 layers of
 * core/arch/RepRFN.py:158-161,252,315,322 (deploy form).
 *
 *   dst = alpha * act(conv(concat(src_a, src_b)) + bias) + res          (res_mode ADD)
 *   dst = X * (res > 0 ? 1 : m),  X = alpha * act(conv(...) + bias)     (res_mode MASK; m = 0, or with act NONE
 *                                                                        m = the args' slope: LeakyReLU backward)
 *   dst = act(alpha * (conv(...) + bias) + res)                         (res_mode PRE)
 *
 * The MASK form is the backward of conv -> ReLU chains: with transpose_flip
 * weights the convolution yields dL/dx of the NEXT layer and `res` is the saved
 * (Leaky)ReLU output of THIS layer, so the product is dL/d(pre-activation) in one pass.
 *
 * The channel concat of core/models/model.py:169-170 is folded into the K loop:
 * src_b may be NULL (cb = 0).
 * ------------------------------------------------------------------------- */
typedef struct {
    const void* src_a;   /* (N,H,W,ca) bf16, ca % 16 == 0 */
    const void* src_b;   /* (N,H,W,cb) bf16 or NULL, cb % 16 == 0 */
    const void* wpack;   /* packed weights, see srhip_conv3x3_pack_weights */
    const float* bias;   /* fp32[cout] or NULL */
    const void* res;     /* optional residual, (N,H,W,cout_stride) bf16; may equal dst (read before written) */
    void* dst;           /* (N,H,W,cout_stride) bf16, or (N,cout,H,W) fp32 if dst_nchw_f32 */
    int n, h, w;
    int ca, cb;          /* padded channel counts of src_a / src_b */
    int cout;            /* real output channels (<= 64) */
    int cout_stride;     /* channels of dst/res in NHWC form (multiple of 8, >= cout) */
    int act;             /* srhip_act */
    float slope;         /* LeakyReLU negative slope; with act NONE and res_mode MASK: the mask's negative-side factor */
    float alpha;         /* scale applied after the activation (1.0 for none) */
    int dst_nchw_f32;    /* 1: write fp32 NCHW (network output / fp32-accumulate checks); no residual */
    int res_mode;        /* srhip_res_mode: how `res` enters the epilogue */
    int ca_stride;       /* 0, or channels per pixel of the tensor src_a points INTO (>= ca): the launch reads a
                            channel slice [src_a, src_a + ca) of every pixel of a wider NHWC tensor (single source) */
    int cout_w;          /* 0, or channels written per pixel (cout <= cout_w <= min(64, cout_stride), multiple of 8):
                            dst / res point at a channel slice of a wider NHWC tensor with cout_stride channels */
    /* Sign bits of a ReLU output, 1 bit per element instead of a bf16 tensor, for the backward pass of
     * core/models/model.py:17-19,61-79 under core/trainer.py:160 (64-channel layers; res_mode ADD or no residual):
     *   aux: (N, H, ceil(W/32), 64) uint32, srhip_conv3x3_signbits_bytes() bytes.  For pixel x and channel c the bit is
     *        word (x/32)*64 + (x%8)*8 + c/8, bit ((x%32)/8)*8 + c%8.
     *   SRHIP_AUX_EMIT:     also WRITE the bits of X = alpha*act(conv+bias) > 0 (before `res` is added)
     *   SRHIP_AUX_MASKBITS: dst = (X + res) where the bit is set, 0 elsewhere          (ReLU backward on a residual sum)
     *   SRHIP_AUX_DUALBITS: dst = X; dst2 = X where the bit is set, 0 elsewhere        (a gradient and its masked copy) */
    void* aux;
    int aux_mode;        /* srhip_aux_mode */
    void* dst2;          /* SRHIP_AUX_DUALBITS: (N,H,W,64) bf16 */
} srhip_conv3x3_args;

/* Bytes of the sign-bit tensor of an (n, h, w, 64) map. */
SRHIP_API size_t srhip_conv3x3_signbits_bytes(int n, int h, int w);

/* Bytes of the packed-weight buffer for a (ca+cb) -> cout 3x3 layer. */
SRHIP_API size_t srhip_conv3x3_packed_bytes(int cin_padded, int cout);

/* Packs fp32 OIHW master weights (device) into MFMA fragment order (bf16).
 *   w:        (w_o, w_i, 3, 3) fp32 -- nn.Conv2d.weight layout
 *   cout, cout_off: this launch computes output channels [cout_off, cout_off+cout)
 *   cin_*_real / c*_pad: real and padded channel counts of the two sources
 *   transpose_flip = 0: forward.  Uses W[cout_off+co][ci][ky][kx]; needs
 *                       w_i == cin_a_real + cin_b_real.
 *   transpose_flip = 1: data-gradient form W'[co][ci][ky][kx] = W[ci][cout_off+co][2-ky][2-kx]
 *                       (the "sources" carry dL/dy with w_o channels, the output is dL/dx for
 *                       forward input channels [cout_off, cout_off+cout)).
 */
SRHIP_API int srhip_conv3x3_pack_weights(const float* w, void* wpack, int w_o, int w_i, int cout,
                                         int cout_off, int cin_a_real, int ca_pad, int cin_b_real,
                                         int cb_pad, int transpose_flip, void* stream);

SRHIP_API int srhip_conv3x3_fwd(const srhip_conv3x3_args* args, void* stream);

/* The two chained 64 -> 64 3x3 convolutions of a residual block as ONE launch:
 *   mid = ReLU(conv1(src) + bias1);   dst = alpha * act2(conv2(mid) + bias2) + src
 * SRResidualBlock (core/models/model.py:22-33: act2 none, alpha 0.2) and Residual(ConvActivation, ConvActivation)
 * (core/models/model.py:17-19,70-73: act2 ReLU, alpha 1).  `mid` stays in LDS: the pair moves 256 bytes of HBM traffic
 * per pixel instead of the 640 of two srhip_conv3x3_fwd launches, and gives the same values (the same bf16 rounding of
 * `mid`).  If `mid` is not NULL the kernel also writes it (training keeps it for the backward pass): 384 bytes per pixel.
 * All tensors (N,H,W,64) bf16 NHWC, pairwise distinct; wpack1 / wpack2 from srhip_conv3x3_pack_weights(64 -> 64). */
typedef struct {
    const void* src;
    const void* wpack1;
    const float* bias1;   /* fp32[64] or NULL */
    const void* wpack2;
    const float* bias2;   /* fp32[64] or NULL */
    void* mid;            /* NULL, or receives ReLU(conv1(src) + bias1) */
    void* dst;
    int n, h, w;
    int act2;             /* srhip_act of the second layer */
    float slope2;         /* its LeakyReLU slope */
    float alpha;          /* > 0 */
    int cout2;            /* 0 or 64: the form above.  1..32: HEAD form for the network's last two layers
                             (core/models/model.py:111-115): wpack2 / bias2 hold a 64 -> cout2 layer, there is no residual,
                             dst = alpha * act2(conv2(relu(conv1(src) + bias1)) + bias2) as fp32 NCHW (n, cout2, h, w) */
    /* The ReLU pattern of `mid` as 1 bit per element (srhip_conv3x3_pair_bits_bytes() bytes, a layout private to this
     * entry point), for the backward pass of SRResidualBlock (core/models/model.py:22-33 under core/trainer.py:160; forms
     * with act2 none that also write `mid`):
     *   mid_bits_mode 1: also WRITE the bits of mid > 0 (the forward launch of a training step);
     *   mid_bits_mode 2: instead of the ReLU, zero `mid` where the bit is clear.  With transpose_flip packs of conv2 / conv1
     *                    as wpack1 / wpack2, src = dL/d(out), bias1 = bias2 = NULL and alpha = the block's scale this launch
     *                    is the block's data gradient:  mid = conv2^T(src) * [mid_fwd > 0]  (written: the first layer's
     *                    weight gradient needs it, scaled by alpha),  dst = alpha * conv1^T(mid) + src = dL/d(in). */
    void* mid_bits;
    int mid_bits_mode;
} srhip_conv3x3_pair_args;
SRHIP_API size_t srhip_conv3x3_pair_bits_bytes(int n, int h, int w);
SRHIP_API int srhip_conv3x3_pair_fwd(const srhip_conv3x3_pair_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * General convolution as one implicit GEMM on the matrix cores: nn.Conv2d with kernel 1 or 3, stride 1 or 2,
 * padding 0 or 1, any channel counts, for the layers outside the shape of srhip_conv3x3_fwd:
 * core/models/model_final.py:237-246 (160 -> 160 1x1), :272-276 and :329-363 (up to 384 -> 256 channels, 3x3 stride 2
 * down-sampling, torch.concat of up to three maps in front of a layer, :375-379), core/arch/RepRFN.py:285-291.
 *
 *   dst = alpha * act(conv(concat(src[0..nsrc-1])) + bias) + res        fp32 accumulation over the whole reduction
 *
 *   src[k]:     NHWC bf16 map; c[k] channels (multiple of 16) are read from each pixel, whose tensor holds cstride[k]
 *               channels per pixel (so a channel slice of a wider tensor can be a source)
 *   wpack:      srhip_conv_gemm_pack_weights(weight (cout, cin, k, k) fp32, ..., cin_padded = sum of c[k]); input channel
 *               positions >= cin carry zero weights (channel padding of the last source)
 *   bias:       cout fp32 or NULL;  res: NULL or a tensor of dst's layout, added after activation and alpha (res_mode 0)
 *               or used as a ReLU mask (res_mode 1)
 *   dst:        (n, ho, wo, cout_stride) bf16, ho = (h + 2 pad - k) / stride + 1; channels [0, pad16(cout)) are written
 *   alpha > 0;  act / slope as in srhip_conv3x3_args */
typedef struct {
    const void* src[3];
    int c[3];
    int cstride[3];
    int nsrc;
    const void* wpack;
    const float* bias;
    const void* res;
    void* dst;
    int n, h, w;
    int cout, cout_stride;
    int ksize, stride, pad;
    int act;
    float slope, alpha;
    int res_mode;        /* SRHIP_RES_ADD (0): + res;  SRHIP_RES_MASK (1): the result is zeroed where res <= 0 (the ReLU
                            backward of the layer in front, whose output res is, on a data-gradient launch) */
} srhip_conv_gemm_args;
SRHIP_API size_t srhip_conv_gemm_packed_bytes(int cin_padded, int cout, int ksize);
SRHIP_API int srhip_conv_gemm_pack_weights(const float* weight, void* packed, int cout, int cin, int ksize, int cin_padded,
                                           void* stream);
/* The same for the layer's data gradient (autograd of nn.Conv2d under loss.backward(), core/trainer.py:160): packs, from
 * the stored fp32 (w_o, w_i, k, k) weight, the layer whose output channel co is the stored input channel row0 + co
 * (co < rows: the channel range of one source of a concatenated input), whose input channels are the stored output
 * channels (padded to cin_padded) and whose kernel is the stored one flipped in both directions -- what
 * weight.transpose(0, 1).flip(2, 3)[row0 : row0 + rows] would hold, without materialising it.
 * packed: srhip_conv_gemm_packed_bytes(cin_padded, rows, ksize) bytes. */
SRHIP_API int srhip_conv_gemm_pack_weights_transposed(const float* weight, void* packed, int w_o, int w_i, int ksize, int row0,
                                                      int rows, int cin_padded, void* stream);
SRHIP_API int srhip_conv_gemm_fwd(const srhip_conv_gemm_args* args, void* stream);

/* A chain of three 1x1 convolutions, each + bias + activation, as ONE pass: the maps between the layers never leave the
 * CU.  Replaces three launches of srhip_conv_gemm_fwd with ksize 1 on the tails of the temporal network's G-buffer encoder
 * (core/models/model_final.py:237-246: conv_main[1:4] 160 -> 160 -> 160 -> 64 and conv_enhance[1:4] 112 -> 64 -> 64 -> 64,
 * the second scaled by enhance_param and added to the first, :256-258):
 *   dst = alpha * act2(W2 act1(W1 act0(W0 src + b0) + b1) + b2) + res
 * src (npix, c[0]), dst / res (npix, c[3]) NHWC bf16, channels contiguous per pixel; wpack[l]: layer l packed by
 * srhip_conv_gemm_pack_weights(ksize 1, cin_padded = c[l]); bias[l] fp32 (c[l+1]) or NULL; the intermediate maps are
 * rounded to bf16 as the layer-by-layer form stores them, sums in fp32 in the same order (results are bit-equal).
 * Kernels exist for (160,160,160,64) without residual and (112,64,64,64) with one: srhip_conv1x1_chain_supported says. */
typedef struct {
    const void* src;
    const void* res;      /* NULL, or the map added behind alpha * act(.) */
    void* dst;
    const void* wpack[3];
    const float* bias[3];
    long long npix;       /* N * H * W */
    int c[4];             /* channels: input, after layer 0, 1, 2 (multiples of 16) */
    int nlayers;          /* 3 */
    int act[3];           /* SRHIP_ACT_* per layer */
    float slope;          /* LeakyReLU slope */
    float alpha;          /* > 0 */
} srhip_conv1x1_chain_args;
SRHIP_API int srhip_conv1x1_chain_fwd(const srhip_conv1x1_chain_args* a, void* stream);
SRHIP_API int srhip_conv1x1_chain_supported(const int* c, int nlayers, int has_res);

/* Weight and bias gradient of the same layer (autograd backward of nn.Conv2d,
 * core/models/model.py:12,26-28,112-114 via loss.backward() at core/trainer.py:160):
 *   dw[co][ci_off+ci][ky][kx] (=|+=) scale * sum_{n,y,x} dz[n,y,x,co] * x[n,y+ky-1,x+kx-1,ci]
 *   db[co]                    (=|+=) scale * sum_{n,y,x} dz[n,y,x,co]      (db may be NULL)
 * (`scale` carries a constant factor of the layer's output, e.g. the 0.2 of
 * core/models/model.py:33, so that dz need not be rescaled in memory.)
 * x (N,H,W,ci_pad) and dz (N,H,W,co_pad) are NHWC bf16 with padded channel counts in
 * {16,32,64}; dw is the fp32 OIHW gradient of a (co_real.., w_i, 3, 3) weight, of which this
 * call fills input channels [ci_off, ci_off+ci_real) (a concat layer takes one call per source).
 * workspace: srhip_conv3x3_wgrad_workspace_bytes() bytes of device memory, caller-owned.
 * Deterministic (fixed-order two-stage reduction, no atomics). */
SRHIP_API size_t srhip_conv3x3_wgrad_workspace_bytes(void);
SRHIP_API int srhip_conv3x3_wgrad(const void* x, const void* dz, float* dw, float* db, void* workspace,
                                  int n, int h, int w, int ci_pad, int ci_real, int ci_off, int w_i,
                                  int co_pad, int co_real, int accumulate, float scale, void* stream);
/* The same with x and / or dz given as channel windows of wider NHWC bf16 maps (the layers of
 * core/models/model_final.py:329-379 with more than 64 input or output channels, and their concatenated inputs, take one
 * call per (<= 64 input, <= 64 output) channel pair without copying the slices out): the window of x is channels
 * [x_coff, x_coff + ci_pad) of a map with x_cstride channels per pixel, likewise dz.  Offsets and strides are multiples
 * of 8 channels.  A window may reach past its map's last channel (ci_real, co_real say how many channels count): what
 * is read there only meets gradient entries that are not written. */
SRHIP_API int srhip_conv3x3_wgrad_windows(const void* x, const void* dz, float* dw, float* db, void* workspace,
                                          int n, int h, int w, int ci_pad, int ci_real, int ci_off, int w_i,
                                          int co_pad, int co_real, int accumulate, float scale,
                                          int x_cstride, int x_coff, int dz_cstride, int dz_coff, void* stream);

/* Several (input window, gradient window) pairs of ONE layer in one launch pair (gradient + reduction): the channel-slice
 * pairs of a layer with more than 64 input or output channels (core/models/model_final.py:329-379 under loss.backward(),
 * core/trainer.py:160) share image size, tile shape (ci_pad, co_pad) and the weight's input-channel count w_i.  Job j
 * fills dw_j[co][ci_off + ci][ky][kx] for co < co_real, ci < ci_real, where dw_j points at the first output row of the
 * job's output-channel slice inside the fp32 (O, w_i, 3, 3) gradient; db_j (or NULL) likewise.  1 <= njobs <= 32; the
 * jobs' regions of dw must not overlap unless accumulate is set and the overlap is intended.  Same workspace as above. */
typedef struct {
    const void* x;        /* NHWC bf16 map that holds the input window */
    const void* dz;       /* NHWC bf16 map that holds the gradient window */
    float* dw;
    float* db;            /* or NULL */
    int ci_real, ci_off, co_real;
    int x_cstride, x_coff, dz_cstride, dz_coff;   /* channels per pixel of the maps, first channel of the windows */
    int reserved;         /* 0 */
} srhip_wgrad_job;
SRHIP_API int srhip_conv3x3_wgrad_multi(const srhip_wgrad_job* jobs, int njobs, void* workspace, int n, int h, int w,
                                        int ci_pad, int co_pad, int w_i, int accumulate, float scale, void* stream);
/* The same for 1x1 layers (core/models/model_final.py:237-246): dw_j[co][ci_off + ci] of an fp32 (O, w_i, 1, 1) gradient --
 * the centre tap alone (one accumulator tile per wave instead of nine). */
SRHIP_API int srhip_conv1x1_wgrad_multi(const srhip_wgrad_job* jobs, int njobs, void* workspace, int n, int h, int w,
                                        int ci_pad, int co_pad, int w_i, int accumulate, float scale, void* stream);

/* ---------------------------------------------------------------------------
 * Generic direct convolution (any kernel size / stride / padding, NHWC bf16,
 * fp32 accumulate on the vector ALU).  For the small memory-bound layers:
 * ESA 1x1 / strided 3x3 (core/arch/RepRFN.py:285-291), 3x1 / 1x3 RepBlock
 * branches (:165-168), stride-2 encoders (core/models/model_final.py:272-276).
 * w is fp32 OIHW (device).  dst = alpha*act(conv+bias) + res.
 * ------------------------------------------------------------------------- */
SRHIP_API int srhip_conv2d_direct(const void* src, const float* w, const float* bias, const void* res,
                                  void* dst, int n, int h, int wd, int cin, int cin_stride, int cout,
                                  int cout_stride, int kh, int kw, int sh, int sw, int ph, int pw,
                                  int act, float slope, float alpha, void* stream);

/* nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2) (+ReLU), TransConvActivation of
 * core/models/model_final.py:198-203.  src (N,h,w,cin_stride) bf16 -> dst (N,2h,2w,cout_stride) bf16;
 * w is the fp32 (cin, cout, 2, 2) weight of the module (device); act: none or ReLU. */
SRHIP_API int srhip_conv_transpose2x2s2(const void* src, const float* w, const float* bias, void* dst, int n,
                                        int h, int wd, int cin, int cin_stride, int cout, int cout_stride,
                                        int act, void* stream);

/* GradientMap of core/models/model_final.py:160-187 on fp32 (planes, h, w):
 * sqrt(gv^2 + gh^2 + eps) of the fixed vertical / horizontal 3x3 difference stencils, zero padding. */
SRHIP_API int srhip_gradient_map_f32(const float* src, float* dst, int planes, int h, int w, float eps,
                                     void* stream);

/* ---------------------------------------------------------------------------
 * Memory-bound operators.
 * ------------------------------------------------------------------------- */
/* Fixed-ratio bilinear up-sampling with the reference's exact border rule:
 * F.grid_sample(fmap, make_coord((th,tw)), "bilinear", zeros, align_corners=False)
 * (core/models/model.py:153,162; make_coord core/tools/math.py:107-130).
 * src (N,sh,sw,c) bf16 -> dst (N,th,tw,c) bf16, c % 8 == 0. */
SRHIP_API int srhip_upsample_grid_bilinear_nhwc(const void* src, void* dst, int n, int c, int sh,
                                                int sw, int th, int tw, void* stream);
/* Same operator on the reference's fp32 NCHW tensors. */
SRHIP_API int srhip_upsample_grid_bilinear_f32(const float* src, float* dst, int n, int c, int sh,
                                               int sw, int th, int tw, void* stream);
/* Its backward: dsrc (N,C,sh,sw) fp32 is ZEROED then accumulated (scatter-add). */
SRHIP_API int srhip_upsample_grid_bilinear_bwd_f32(const float* ddst, float* dsrc, int n, int c,
                                                   int sh, int sw, int th, int tw, void* stream);

/* Backward of the NHWC form as a gather (no atomics): ddst (N,th,tw,c) bf16 ->
 * dsrc (N,sh,sw,c) bf16.  If relu_out is not NULL it is the saved (N,sh,sw,c) ReLU output that
 * produced src, and dsrc is zeroed where relu_out <= 0 (ReLU backward fused). */
SRHIP_API int srhip_upsample_grid_bilinear_bwd_nhwc(const void* ddst, const void* relu_out, void* dsrc,
                                                    int n, int c, int sh, int sw, int th, int tw,
                                                    void* stream);

/* Element-wise helpers of the training graph on bf16 tensors of `count` elements (count % 8 == 0):
 *   relu_mask: dst = y > 0 ? dy : 0     (backward of nn.ReLU, core/models/model.py:13)
 *   add:       dst = a + b              (Residual, core/models/model.py:17-19, when the
 *                                        un-added branch must be kept for the backward)
 * dst may alias an input. */
SRHIP_API int srhip_relu_mask_bf16(const void* dy, const void* y, void* dst, size_t count, void* stream);
/* Sign bits (value > 0) of an (n, h, w, 64) bf16 map in the layout of srhip_conv3x3_args.aux, for a ReLU output whose
 * producing launch cannot emit them itself (the two-source layer torch.concat + conv, core/models/model.py:169):
 * the backward then applies the ReLU mask inside the data-gradient launch (SRHIP_AUX_MASKBITS) instead of a pass that
 * reads two maps and writes one.  bits: srhip_conv3x3_signbits_bytes(n, h, w) bytes. */
SRHIP_API int srhip_signbits_from_map_bf16(const void* map, void* bits, int n, int h, int w, void* stream);
SRHIP_API int srhip_add_bf16(const void* a, const void* b, void* dst, size_t count, void* stream);
/* dst = bf16(a + *scale * y) on bf16 maps of `count` elements, the scalar read from device memory: the G-buffer encoder's
 * fmap + enhance_param * enhance (core/models/model_final.py:256-258) in the training graph, where enhance_param is a
 * trained parameter (inference passes its value as the epilogue factor of the last convolution).  fp32 arithmetic, one
 * rounding.  The backward, one pass over dout and y: dy = bf16(*scale * dout), partials[g] = work-group g's share of
 * d scale = sum(dout * y) in fp32 (npartials work-groups are launched, each writes its entry; the caller adds them up);
 * the gradient of a is dout itself. */
SRHIP_API int srhip_scale_add_bf16(const void* a, const void* y, const float* scale, void* dst, size_t count, void* stream);
SRHIP_API int srhip_scale_add_bwd_bf16(const void* dout, const void* y, const float* scale, void* dy, float* partials,
                                       int npartials, size_t count, void* stream);

/* ESA attention helpers of RepRFN (core/arch/RepRFN.py:295-308) on NHWC bf16, c % 8 == 0:
 *   max_pool2d:      F.max_pool2d(x, kernel_size=k, stride=s)  (no padding)       :298
 *   resize_bilinear: F.interpolate(x, (ho,wo), mode="bilinear", align_corners) + addend
 *                    (addend (N,ho,wo,c) or NULL; the reference adds conv_f(c1_))  :303,305
 *                    align_corners = 1 is nn.Upsample(x2, bilinear, align_corners=True) of
 *                    core/models/model_final.py:281-289
 *   sigmoid_gate:    dst = x * sigmoid(gate)                                       :306-308 */
SRHIP_API int srhip_max_pool2d_nhwc(const void* src, void* dst, int n, int c, int h, int w, int k, int s,
                                    void* stream);
SRHIP_API int srhip_resize_bilinear_nhwc(const void* src, const void* addend, void* dst, int n, int c,
                                         int h, int w, int ho, int wo, int align_corners, void* stream);
SRHIP_API int srhip_sigmoid_gate_bf16(const void* x, const void* gate, void* dst, size_t count, void* stream);
/* The tail of the ESA block as ONE pass (core/arch/RepRFN.py:303-308, replaces the last four launches of the
 * layer-by-layer form):  dst = x * sigmoid(conv4(F.interpolate(c3, (h,w), "bilinear", align_corners=False) + conv_f(c1)))
 *   x, dst (N,h,w,cstride) bf16, cstride = cfeat or 64, the first cfeat channels gated (cfeat in {16,32,48,64}), the others
 *   copied;  c1 (N,h,w,16) bf16 = conv1(x);
 *   c3 (N,h3,w3,16) bf16 = conv3(max_pool(conv2(c1)));  wf (16,16), bf (16), w4 (cfeat,16), b4 (cfeat): fp32 1x1 kernels
 *   (output-major) and biases; the kernels are rounded to bf16 (matrix cores, fp32 accumulation).  conv_f's output, the
 *   sum and conv4's output are rounded to bf16 where the layer-by-layer form stores them. */
SRHIP_API int srhip_esa_tail_bf16(const void* x, const void* c1, const void* c3, const float* wf, const float* bf,
                                  const float* w4, const float* b4, void* dst, int n, int h, int w, int h3, int w3,
                                  int cfeat, int cstride, void* stream);

/* Fused loss of the default trainer configuration, forward AND backward in one call:
 *   xc   = clamp(sr, clip_min, clip_max)                       (core/trainer.py:154-156)
 *   loss = lambda_l1 * mean|xc - y| + lambda_ssim * (1 - SSIM(xc, y))
 * (LossGroup core/models/model.py:183-207; SSIMLoss core/modules/loss.py:12-29; _ssim
 * core/modules/functional.py:169-189: 11x11 Gaussian window, sigma 1.5, zero padding.)
 * sr, target, dsr: (planes, h, w) fp32 (planes = batch * channels); dsr = dloss/dsr;
 * out3 (device) = {l1, ssim_loss, total}.  Pass -INFINITY/+INFINITY to disable the clip.
 * workspace: srhip_l1_ssim_workspace_bytes(planes, h, w) bytes, caller-owned. */
SRHIP_API size_t srhip_l1_ssim_workspace_bytes(int planes, int h, int w);
SRHIP_API int srhip_l1_ssim_loss(const float* sr, const float* target, float* dsr, float* out3,
                                 void* workspace, int planes, int h, int w, float lambda_l1,
                                 float lambda_ssim, float clip_min, float clip_max, void* stream);

/* Pixel-weighted L1 of the temporal variant, forward AND backward in one call: the network's second output is a per-pixel
 * loss weight at the input resolution (enhance_weight, core/models/model_final.py:499-505); broadcast over the channels
 * and, nearest, over the (h / wh) x (w / ww) output pixels each of its pixels covers,
 *   loss = mean over (n, c, h, w) of (1 + weight[n, 0, y / (h/wh), x / (w/ww)]) * |sr - target|,
 * out1 (device) = loss, dsr = dloss/dsr ((1 + weight) * sign(sr - target) / count; 0 where sr == target).
 * sr, target, dsr: (n, c, h, w) fp32, 16-byte aligned, w a multiple of 4; weight: (n, 1, wh, ww) fp32 (no gradient: the
 * weight is computed from the inputs alone).  workspace: srhip_weighted_l1_workspace_bytes() bytes, caller-owned.
 * Deterministic (fixed-order two-stage sum). */
SRHIP_API size_t srhip_weighted_l1_workspace_bytes(void);
SRHIP_API int srhip_weighted_l1_loss(const float* sr, const float* target, const float* weight, float* dsr, float* out1,
                                     void* workspace, int n, int c, int h, int w, int wh, int ww, void* stream);

/* Motion-vector backward warp, core/tools/math.py:10-17 (grid (x,y)-last as the
 * function evidently intends): frame (N,C,h,w), motion (N,2,h,w) in pixels,
 * zero padding.  fp32 NCHW in/out. */
SRHIP_API int srhip_backward_warp_f32(const float* frame, const float* motion, float* dst, int n,
                                      int c, int h, int w, int mode, int align_corners, void* stream);
/* bf16 NHWC frame, fp32 (N,2,h,w) motion. c % 8 == 0. */
SRHIP_API int srhip_backward_warp_nhwc(const void* frame, const float* motion, void* dst, int n,
                                       int c, int h, int w, int mode, int align_corners, void* stream);
/* Trunk input with the previous output frame riding along (the slot core/trainer.py:130-137 fills with `velocity`
 * and the previous `sr_frames`; warp hook core/models/model.py:142-145, warp core/tools/math.py:10-17): buffers
 * (N,cb,h,w) and prev (N,cf,h,w) fp32 NCHW, motion (N,2,h,w) in pixels -> dst (N,h,w,16) bf16 NHWC with channels
 * [0,cb) = buffers, [cb,cb+cf) = prev warped (zero padding), rest 0.  cb + cf <= 16.  Bit-equal to
 * srhip_nchw_f32_to_nhwc_bf16 of the channel concatenation of buffers and srhip_backward_warp_f32(prev, motion).
 * prev may be NULL (no previous frame: those channels are zero; motion is then unused). */
SRHIP_API int srhip_warp_concat_nchw_f32_to_nhwc_bf16(const float* buffers, int cb, const float* prev, int cf,
                                                      const float* motion, void* dst, int n, int h, int w, int mode,
                                                      int align_corners, void* stream);
/* Backward of the bilinear warp: dframe (zeroed, then scatter-add) and dmotion. */
SRHIP_API int srhip_backward_warp_bwd_f32(const float* frame, const float* motion, const float* ddst,
                                          float* dframe, float* dmotion, int n, int c, int h, int w,
                                          int align_corners, void* stream);

/* nn.PixelShuffle / nn.PixelUnshuffle (core/arch/RepRFN.py:272,276;
 * core/models/model_final.py:234,408,449).  Pure permutation on NCHW tensors of
 * elem_size 2 or 4 bytes: bit-exact.
 *   shuffle:   src (N,C*r*r,H,W) -> dst (N,C,H*r,W*r)
 *   unshuffle: src (N,C,H*r,W*r) -> dst (N,C*r*r,H,W)          */
SRHIP_API int srhip_pixel_shuffle(const void* src, void* dst, int n, int c, int h, int w, int r,
                                  int elem_size, void* stream);
/* nn.PixelUnshuffle(4) (core/models/model_final.py:234,408) fused with the layout conversion in front of the
 * convolutions that consume it: src (N,c,4h,4w) fp32 NCHW -> dst (N,h,w,16c) bf16 NHWC, channel c*16 + i*4 + j.
 * Equal bit for bit to srhip_pixel_unshuffle followed by srhip_nchw_f32_to_nhwc_bf16.  c <= 23. */
SRHIP_API int srhip_pixel_unshuffle4_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int n, int c, int h, int w, void* stream);
/* The same pass, which also yields the largest element of src for the check at core/models/model_final.py:463
 * (`assert hr_buffer.max() < 10`) without reading the tensor a second time: on return (in stream order)
 * max(block_max[0 .. max_cap)) == max(src), NaN if src holds a NaN.  block_max: max_cap floats of device memory, all of
 * them written.  The launch uses at most max_cap work-groups (4096 keeps the full grid). */
SRHIP_API int srhip_pixel_unshuffle4_nchw_f32_to_nhwc_bf16_max(const float* src, void* dst, int n, int c, int h, int w,
                                                               float* block_max, int max_cap, void* stream);
SRHIP_API int srhip_pixel_unshuffle(const void* src, void* dst, int n, int c, int h, int w, int r,
                                    int elem_size, void* stream);
/* GradientMap + PixelUnshuffle(4) (core/models/model_final.py:160-187,406-409) + the layout conversion in front of the
 * convolution that consumes them, one pass: src (N,src_planes,4h,4w) fp32 NCHW, planes[0..nplanes) (HOST array) the
 * plane indices taken in order -> dst (N,h,w,16*nplanes) bf16 NHWC, channel k*16 + i*4 + j = gradient magnitude of plane
 * planes[k] at (4y+i, 4x+j).  Equal bit for bit to srhip_gradient_map_f32 on those planes followed by
 * srhip_pixel_unshuffle4_nchw_f32_to_nhwc_bf16.  nplanes <= 23. */
SRHIP_API int srhip_gradient_map_unshuffle4_nhwc_bf16(const float* src, int src_planes, const int* planes, int nplanes, void* dst, int n,
                                                      int h, int w, float eps, void* stream);
/* nn.PixelShuffle(4) between two convolutions (core/models/model_final.py:446-448): src (N,h,w,cstride >= 16c) bf16 NHWC,
 * channel k*16 + i*4 + j -> dst (N,4h,4w,16) bf16 NHWC with channels k < c real and the rest zero (the padded input of
 * the next layer).  c <= 4.  A permutation: bit-exact. */
SRHIP_API int srhip_pixel_shuffle4_nhwc_bf16_to_nhwc16(const void* src, void* dst, int n, int c, int h, int w, int cstride, void* stream);
/* Its backward (autograd of nn.PixelShuffle, core/models/model_final.py:446-448 under loss.backward(), core/trainer.py:160):
 * d (N,4h,4w,16) bf16 = gradient of the 16-channel image -> dt (N,h,w,cstride) bf16 = gradient of the feature map,
 * dt[..., k*16 + i*4 + j] = d[4y+i, 4x+j, k] for k < c, zero in the other channels.  Bit-exact (a permutation). */
SRHIP_API int srhip_pixel_shuffle4_nhwc16_bwd_bf16(const void* d, void* dt, int n, int c, int h, int w, int cstride, void* stream);

/* (N,ho,wo,C) bf16 NHWC -> (N,h,w,C) with the input at the even positions and zeros elsewhere (ho = ceil(h/2), wo =
 * ceil(w/2), C a multiple of 8): the output gradient of a 3x3 / stride 2 / pad 1 layer (core/models/model_final.py:272-276
 * under loss.backward(), core/trainer.py:160) as the gradient of the stride-1 layer it is a 2 x 2 pick of.  Bit-exact. */
SRHIP_API int srhip_zero_insert2_nhwc_bf16(const void* src, void* dst, int n, int ho, int wo, int h, int w, int c, void* stream);
/* The whole tail nn.PixelShuffle(4) -> nn.Conv2d(3, 3, 3, 1, 1) -> activation of core/models/model_final.py:446-449 in
 * one pass: src (N,h,w,cstride >= 48) bf16 NHWC, channel k*16 + i*4 + j; weight fp32 (3,3,3,3) OIHW and bias fp32[3]
 * (or NULL) in device memory, used as they are (fp32 arithmetic); dst (N,3,4h,4w) fp32 NCHW.
 * act 0 none, 1 ReLU, 2 LeakyReLU(slope). */
SRHIP_API int srhip_pixel_shuffle4_conv3x3_c3_nchw_f32(const void* src, const float* weight, const float* bias, float* dst, int n,
                                                       int h, int w, int cstride, int act, float slope, void* stream);
/* Fused network head: NHWC bf16 (N,H,W,cstride>=c*r*r) -> NCHW fp32 (N,c,H*r,W*r). */
SRHIP_API int srhip_pixel_shuffle_nhwc_to_nchw_f32(const void* src, float* dst, int n, int c, int h,
                                                   int w, int r, int cstride, void* stream);

/* F.interpolate(x, scale_factor=1/s, mode="nearest") (core/trainer.py:92-95):
 * src (NC,H,W) -> dst (NC,ho,wo), source index min(floor(dst*s'), size-1), s' = fp32(1/scale_factor). */
SRHIP_API int srhip_nearest_down(const void* src, void* dst, int nc, int h, int w, int ho, int wo,
                                 double scale_factor, int elem_size, void* stream);

/* Relative-coordinate features (code_test.py:193-202, core/models/lte.py:164-171):
 * dst (2,th,tw) fp32, bit-exact replay of the fp32 operator sequence. */
SRHIP_API int srhip_rel_coord(float* dst, int th, int tw, int sh, int sw, void* stream);
/* make_coord (core/tools/math.py:107-130): dst (2,h,w) fp32, ch0 = x, ch1 = y. */
SRHIP_API int srhip_make_coord(float* dst, int h, int w, void* stream);

/* torch.concat along channels (core/models/model.py:169-170) for NHWC bf16:
 * dst (P, ca+cb) <- a (P,ca), b (P,cb); ca, cb % 8 == 0.  (The conv kernel
 * folds the concat; this standalone form exists for callers that need the tensor.) */
SRHIP_API int srhip_concat_nhwc(const void* a, const void* b, void* dst, size_t pixels, int ca, int cb,
                                void* stream);

/* Channels-first LayerNorm (core/arch/CAMixerSR.py:32-37): x (N,C,HW) fp32. */
SRHIP_API int srhip_layernorm_cf_f32(const float* x, const float* weight, const float* bias, float* y,
                                     int n, int c, int hw, float eps, void* stream);

/* Disocclusion logic of the temporal variant in one pass (core/models/model_final.py:472-486 and :501-504 over
 * compute_mask_frame/_basecolor/_depth/_normal/_roughness/_metallic at :50-139; colour distance
 * core/tools/color.py:22-31).  fp32 NCHW; G-buffer channels: normal 0:3, metallic 3, specular 4, roughness 5,
 * base colour 6:9, depth 9.
 *   cur_frame (B,3,h,w), cur_buffer (B,10,h,w), hist_frames (B,F,3,h,w), hist_buffer (B,F,10,h,w) (warped history)
 *   mask (B,F,h,w) bytes or NULL; marked_frames (B*F,3,h,w); aux (B*F,14,h,w) = cur[0:3], cur[6:10],
 *   marked hist[0:3], marked hist[6:10]; enhance_weight (B,h,w). */
SRHIP_API int srhip_disocclusion_f32(const float* cur_frame, const float* cur_buffer, const float* hist_frames,
                                     const float* hist_buffer, unsigned char* mask, float* marked_frames, float* aux,
                                     float* enhance_weight, int b, int f, int h, int w, void* stream);

/* Backward pieces of RepRFN under loss.backward() (core/trainer.py:160): LeakyReLU / ReLU from the activation's output
 * (core/arch/RepRFN.py:17-33), and the ESA block's sigmoid gate, max_pool2d(7, 3) and bilinear resize (:295-308).
 * NHWC bf16; dx / dgate are overwritten; all deterministic (gather forms, no atomics).
 *   lrelu_mask:        dst = dy * (y > 0 ? 1 : slope)
 *   sigmoid_gate_bwd:  y = x * sigmoid(gate)  ->  dx = dy * sig, dgate = dy * x * sig * (1 - sig)
 *   max_pool2d_bwd:    src (N,h,w,c) the pooled input, dy (N,ho,wo,c) -> dx (N,h,w,c); first maximum wins a tie
 *   max_pool2d_arg / max_pool2d_bwd_arg: the same pair for a training step.  The forward also writes, per output
 *                      element, the window position ky*k+kx of its maximum into arg (N,ho,wo,c) uint8 (k <= 15), and the
 *                      backward gathers from arg and dy alone (what autograd's max_pool2d_with_indices keeps, :298)
 *   resize_bilinear_bwd: dy (N,ho,wo,c) -> dx (N,h,w,c), adjoint of srhip_resize_bilinear_nhwc */
SRHIP_API int srhip_lrelu_mask_bf16(const void* dy, const void* y, void* dst, size_t count, float slope, void* stream);
SRHIP_API int srhip_sigmoid_gate_bwd_bf16(const void* x, const void* gate, const void* dy, void* dx, void* dgate, size_t count,
                                          void* stream);
SRHIP_API int srhip_max_pool2d_bwd_nhwc(const void* src, const void* dy, void* dx, int n, int c, int h, int w, int k, int s,
                                        void* stream);
SRHIP_API int srhip_max_pool2d_arg_nhwc(const void* src, void* dst, void* arg, int n, int c, int h, int w, int k, int s,
                                        void* stream);
SRHIP_API int srhip_max_pool2d_bwd_arg_nhwc(const void* arg, const void* dy, void* dx, int n, int c, int h, int w, int k, int s,
                                            void* stream);
SRHIP_API int srhip_resize_bilinear_bwd_nhwc(const void* dy, void* dx, int n, int c, int h, int w, int ho, int wo,
                                             int align_corners, void* stream);

/* dilate_velocity (core/tools/math.py:80-100): motion (B,H,W,2) fp32, depth (B,1,H,W) fp32 -> dst (B,H,W,2): every
 * pixel takes the motion vector of the pixel with the smallest depth in its ksize x ksize neighbourhood.  As in the
 * reference the neighbourhood is zero-padded (F.unfold), the first minimum in row-major window order wins and the
 * chosen position is clamped to the frame.  Bit-exact (index arithmetic and a copy). */
SRHIP_API int srhip_dilate_velocity_f32(const float* motion, const float* depth, float* dst, int b, int h, int w, int ksize,
                                        void* stream);

/* Gated sum of the temporal network's main stage (core/models/model_final.py:422):
 *   dst = cur + sigmoid(f_self * g_cross) * fmap + sigmoid(g_self * f_cross) * grad
 * cur / fmap / grad / dst (pixels, c) bf16 NHWC, the four gates one bf16 value per pixel read at pixel * gate_stride
 * elements (the 1-channel attention outputs live in 16-channel padded maps); fp32 arithmetic, one rounding at the end. */
SRHIP_API int srhip_attention_fuse_bf16(const void* cur, const void* fmap, const void* grad, const void* f_self, const void* g_cross,
                                        const void* g_self, const void* f_cross, int gate_stride, void* dst, size_t pixels, int c,
                                        void* stream);

/* Backward of that sum (autograd of core/models/model_final.py:422 under loss.backward(), core/trainer.py:160) from
 * d = the gradient of dst (which is also the gradient of cur): dfmap = sigmoid(f_self * g_cross) * d, dgrad likewise, and
 * the four gate gradients (channel sums of d * fmap / d * grad times the sigmoid's derivative times the partner gate).
 * Each gate gradient is written as rows of dgate_stride channels per pixel, channel 0 the value and the rest zero: the
 * gradient of the whole padded map whose channel 0 the gate is.  c in {8, 16, 32, 64}. */
SRHIP_API int srhip_attention_fuse_bwd_bf16(const void* d, const void* fmap, const void* grad, const void* f_self, const void* g_cross,
                                            const void* g_self, const void* f_cross, int gate_stride, void* dfmap, void* dgrad,
                                            void* d_f_self, void* d_g_cross, void* d_g_self, void* d_f_cross, int dgate_stride,
                                            size_t pixels, int c, void* stream);

/* Second half of nn.ConvTranspose2d(k=2, s=2) (core/models/model_final.py:198-203) written as a 1x1 convolution to
 * 4*C channels: src (N,h,w,4C) bf16 with channel (i*2+j)*C + c  ->  dst (N,2h,2w,C), dst[2y+i][2x+j][c]. */
SRHIP_API int srhip_depth_to_space2_nhwc(const void* src, void* dst, int n, int h, int w, int c, void* stream);

/* Sample decode of the data loader (core/data/supersampling_sequence.py:153-200, _load_and_augment with
 * _fn_transform / _fn_transform_velocity): src fp16 (C,H,W) device array (the `arr_0` of a frame file) ->
 * dst fp32 (C,crop_h,crop_w) = src[:, y1:y1+crop_h, x1:x1+crop_w], rows reversed if flip_rows (the reference's
 * "hflip", np.flip(img, 1)), columns reversed if flip_cols ("vflip"); channel k negated when bit k of neg_mask
 * is set (motion vectors: flip_rows -> bit 1, flip_cols -> bit 0). */
SRHIP_API int srhip_crop_flip_f16_f32(const void* src, float* dst, int c, int h, int w, int x1, int y1, int crop_h, int crop_w,
                                      int flip_rows, int flip_cols, unsigned neg_mask, void* stream);

/* Depth range compression (core/models/model_final.py:40-46); src may equal dst. */
SRHIP_API int srhip_clamp_depth_f32(const float* src, float* dst, long long n, float thres1, float thres2, void* stream);

/* Backward of the channels-first LayerNorm (autograd of core/arch/CAMixerSR.py:32-37 under loss.backward(),
 * core/trainer.py:160): dx (N,C,HW), dweight[C], dbias[C] (overwritten; fixed summation order).
 * workspace: srhip_layernorm_cf_bwd_workspace_bytes(n, c, hw) bytes of device memory. */
SRHIP_API size_t srhip_layernorm_cf_bwd_workspace_bytes(int n, int c, int hw);
SRHIP_API int srhip_layernorm_cf_bwd_f32(const float* x, const float* weight, const float* dy, float* dx,
                                         float* dweight, float* dbias, void* workspace, int n, int c, int hw,
                                         float eps, void* stream);

/* srhip_layernorm_cf_bwd_f32 for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same kernels, same results;
 * the workspace is a fixed array of device memory inside the library (8.5 MB per GPU the library runs on, held
 * until the process ends), so the call takes no workspace and allocates nothing.  Limits: n * hw <= 1,048,576
 * pixels and c <= 256 channels; larger shapes are rejected with SRHIP_ERR_INVALID.  Calls that overlap on one GPU
 * (different streams or threads) share the array and race; the first call on a device looks the array's address up
 * (hipGetSymbolAddress), so make one call outside a stream capture before capturing. */
SRHIP_API int srhip_layernorm_cf_bwd_f32_static(const float* x, const float* weight, const float* dy, float* dx,
                                                float* dweight, float* dbias, int n, int c, int hw, float eps,
                                                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRHIP_H */
