/*
 * oracle.c -- CPU restatement of the reference super-resolution hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import, link or
 * call this file; it exists so that tests, the smoke check and the benchmark's
 * CPU-baseline leg can check / time the HIP path against an independent,
 * plainly written implementation of the same arithmetic.
 *
 * The reference path is a composition of PyTorch ATen CPU operators (the
 * reference pins no torch version; the goldens under tests/golden/ were
 * produced with torch 2.10.0 CPU kernels by importing the reference itself).
 * Every function below cites the reference call site whose semantics it
 * restates.  All tensors are dense fp32, NCHW, exactly as the reference holds
 * them.  Parity pinning: tests/test_oracle_golden.py checks every function
 * here against those goldens (bit-exact for index/permutation ops, <=1e-5
 * relative for convolutions and bilinear sampling, whose summation order
 * differs from ATen's).
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* conv2d forward: nn.Conv2d as used by ConvActivation / SRResidualBlock      */
/* (core/models/model.py:10-14,26-28,112-114) and by RepRFN                   */
/* (core/arch/RepRFN.py:9-14,120,129,158-178,285-291).  Cross-correlation,    */
/* zero padding, optional groups (SeqConv3x3 depthwise, RepRFN.py:129).       */
/* act: 0 none, 1 ReLU, 2 LeakyReLU(slope).                                   */
/* ------------------------------------------------------------------------- */
ORC_API void orc_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                            int N, int Cin, int H, int W, int Cout, int kh, int kw,
                            int sh, int sw, int ph, int pw, int groups, int act, float slope)
{
    const int Ho = (H + 2 * ph - kh) / sh + 1;
    const int Wo = (W + 2 * pw - kw) / sw + 1;
    const int cing = Cin / groups, coutg = Cout / groups;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co) {
            const int g = co / coutg;
            float* yp = y + ((size_t)n * Cout + co) * Ho * Wo;
            const float b = bias ? bias[co] : 0.0f;
            for (int i = 0; i < Ho * Wo; ++i) yp[i] = b;
            for (int ci = 0; ci < cing; ++ci) {
                const float* xp = x + ((size_t)n * Cin + g * cing + ci) * H * W;
                const float* wp = w + ((size_t)co * cing + ci) * kh * kw;
                for (int ky = 0; ky < kh; ++ky)
                    for (int kx = 0; kx < kw; ++kx) {
                        const float wv = wp[ky * kw + kx];
                        for (int oy = 0; oy < Ho; ++oy) {
                            const int iy = oy * sh - ph + ky;
                            if (iy < 0 || iy >= H) continue;
                            /* valid ox range: 0 <= ox*sw - pw + kx < W */
                            int ox0 = 0, ox1 = Wo;
                            while (ox0 < Wo && ox0 * sw - pw + kx < 0) ++ox0;
                            while (ox1 > ox0 && (ox1 - 1) * sw - pw + kx >= W) --ox1;
                            const float* xr = xp + (size_t)iy * W - pw + kx;
                            float* yr = yp + (size_t)oy * Wo;
                            if (sw == 1) {
                                for (int ox = ox0; ox < ox1; ++ox) yr[ox] += wv * xr[ox];
                            } else {
                                for (int ox = ox0; ox < ox1; ++ox) yr[ox] += wv * xr[ox * sw];
                            }
                        }
                    }
            }
            if (act == 1) {
                for (int i = 0; i < Ho * Wo; ++i) yp[i] = yp[i] > 0.0f ? yp[i] : 0.0f;
            } else if (act == 2) {
                for (int i = 0; i < Ho * Wo; ++i) yp[i] = yp[i] > 0.0f ? yp[i] : yp[i] * slope;
            }
        }
}

/* conv2d backward w.r.t. input (autograd of nn.Conv2d, reached through        */
/* loss.backward() at core/trainer.py:160).  dx[n,ci,iy,ix] = sum dy*w.        */
ORC_API void orc_conv2d_bwd_input(const float* dy, const float* w, float* dx,
                                  int N, int Cin, int H, int W, int Cout, int kh, int kw,
                                  int sh, int sw, int ph, int pw, int groups)
{
    const int Ho = (H + 2 * ph - kh) / sh + 1;
    const int Wo = (W + 2 * pw - kw) / sw + 1;
    const int cing = Cin / groups, coutg = Cout / groups;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int cin = 0; cin < Cin; ++cin) {
            const int g = cin / cing, ci = cin % cing;
            float* dxp = dx + ((size_t)n * Cin + cin) * H * W;
            for (int i = 0; i < H * W; ++i) dxp[i] = 0.0f;
            for (int cog = 0; cog < coutg; ++cog) {
                const int co = g * coutg + cog;
                const float* dyp = dy + ((size_t)n * Cout + co) * Ho * Wo;
                const float* wp = w + ((size_t)co * cing + ci) * kh * kw;
                for (int ky = 0; ky < kh; ++ky)
                    for (int kx = 0; kx < kw; ++kx) {
                        const float wv = wp[ky * kw + kx];
                        for (int oy = 0; oy < Ho; ++oy) {
                            const int iy = oy * sh - ph + ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int ox = 0; ox < Wo; ++ox) {
                                const int ix = ox * sw - pw + kx;
                                if (ix < 0 || ix >= W) continue;
                                dxp[(size_t)iy * W + ix] += wv * dyp[(size_t)oy * Wo + ox];
                            }
                        }
                    }
            }
        }
}

/* conv2d backward w.r.t. weight and bias (same autograd node).  dbias may be NULL. */
ORC_API void orc_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* dbias,
                                   int N, int Cin, int H, int W, int Cout, int kh, int kw,
                                   int sh, int sw, int ph, int pw, int groups)
{
    const int Ho = (H + 2 * ph - kh) / sh + 1;
    const int Wo = (W + 2 * pw - kw) / sw + 1;
    const int cing = Cin / groups, coutg = Cout / groups;
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; ++co) {
        const int g = co / coutg;
        if (dbias) {
            double s = 0.0;
            for (int n = 0; n < N; ++n) {
                const float* dyp = dy + ((size_t)n * Cout + co) * Ho * Wo;
                for (int i = 0; i < Ho * Wo; ++i) s += dyp[i];
            }
            dbias[co] = (float)s;
        }
        for (int ci = 0; ci < cing; ++ci)
            for (int ky = 0; ky < kh; ++ky)
                for (int kx = 0; kx < kw; ++kx) {
                    double s = 0.0;
                    for (int n = 0; n < N; ++n) {
                        const float* xp = x + ((size_t)n * Cin + g * cing + ci) * H * W;
                        const float* dyp = dy + ((size_t)n * Cout + co) * Ho * Wo;
                        for (int oy = 0; oy < Ho; ++oy) {
                            const int iy = oy * sh - ph + ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int ox = 0; ox < Wo; ++ox) {
                                const int ix = ox * sw - pw + kx;
                                if (ix < 0 || ix >= W) continue;
                                s += (double)xp[(size_t)iy * W + ix] * dyp[(size_t)oy * Wo + ox];
                            }
                        }
                    }
                    dw[(((size_t)co * cing + ci) * kh + ky) * kw + kx] = (float)s;
                }
    }
}

/* ------------------------------------------------------------------------- */
/* make_coord (core/tools/math.py:107-130): pixel-centre coordinates in       */
/* [-1,1]; out layout (2,h,w), channel 0 = x, channel 1 = y.  The reference    */
/* evaluates  (x_min + rx) + (2*rx) * arange(w)  with the two Python scalars   */
/* formed in double and then applied to an fp32 tensor: the product and the    */
/* sum are each one fp32 operation on the scalar rounded to fp32.              */
/* ------------------------------------------------------------------------- */
static inline float coord_value(int n, int j)
{
    const double r = (1.0 - (-1.0)) / (2.0 * (double)n); /* math.py:121-122 */
    const float step = (float)(2.0 * r);
    const float base = (float)(-1.0 + r);
    const float prod = step * (float)j;                  /* fp32 mul, math.py:124 */
    return base + prod;                                  /* fp32 add */
}

ORC_API void orc_make_coord(float* out, int h, int w)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            out[(size_t)y * w + x] = coord_value(w, x);
            out[(size_t)h * w + (size_t)y * w + x] = coord_value(h, y);
        }
}

/* ------------------------------------------------------------------------- */
/* grid_sample, padding_mode="zeros" (ATen grid_sampler_2d semantics), as      */
/* called at core/models/model.py:162 (bilinear, align_corners=False) and      */
/* core/tools/math.py:17 (mode/align_corners from the caller).                 */
/* grid layout (N,Ho,Wo,2), last dim (x,y).  mode: 0 bilinear, 1 nearest.      */
/* ------------------------------------------------------------------------- */
static inline float unnormalize(float c, int size, int align)
{
    if (align) return ((c + 1.0f) / 2.0f) * (float)(size - 1);
    return ((c + 1.0f) * (float)size - 1.0f) / 2.0f;
}

ORC_API void orc_grid_sample(const float* in, const float* grid, float* out,
                             int N, int C, int H, int W, int Ho, int Wo, int mode, int align)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const float* gp = grid + (((size_t)n * Ho + oy) * Wo + ox) * 2;
                const float ix = unnormalize(gp[0], W, align);
                const float iy = unnormalize(gp[1], H, align);
                if (mode == 0) {
                    const float fx = floorf(ix), fy = floorf(iy);
                    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
                    const float nw = ((float)x1 - ix) * ((float)y1 - iy);
                    const float ne = (ix - (float)x0) * ((float)y1 - iy);
                    const float sw_ = ((float)x1 - ix) * (iy - (float)y0);
                    const float se = (ix - (float)x0) * (iy - (float)y0);
                    for (int c = 0; c < C; ++c) {
                        const float* ip = in + ((size_t)n * C + c) * H * W;
                        float v = 0.0f;
                        if (y0 >= 0 && y0 < H && x0 >= 0 && x0 < W) v += ip[(size_t)y0 * W + x0] * nw;
                        if (y0 >= 0 && y0 < H && x1 >= 0 && x1 < W) v += ip[(size_t)y0 * W + x1] * ne;
                        if (y1 >= 0 && y1 < H && x0 >= 0 && x0 < W) v += ip[(size_t)y1 * W + x0] * sw_;
                        if (y1 >= 0 && y1 < H && x1 >= 0 && x1 < W) v += ip[(size_t)y1 * W + x1] * se;
                        out[(((size_t)n * C + c) * Ho + oy) * Wo + ox] = v;
                    }
                } else {
                    const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
                    for (int c = 0; c < C; ++c) {
                        const float* ip = in + ((size_t)n * C + c) * H * W;
                        float v = 0.0f;
                        if (yn >= 0 && yn < H && xn >= 0 && xn < W) v = ip[(size_t)yn * W + xn];
                        out[(((size_t)n * C + c) * Ho + oy) * Wo + ox] = v;
                    }
                }
            }
}

/* grid_sample backward (bilinear, zeros): gradient to the sampled input and,   */
/* when dgrid != NULL, to the grid.  din must be zero-filled by the caller.     */
ORC_API void orc_grid_sample_bwd(const float* in, const float* grid, const float* dout,
                                 float* din, float* dgrid,
                                 int N, int C, int H, int W, int Ho, int Wo, int align)
{
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const float* gp = grid + (((size_t)n * Ho + oy) * Wo + ox) * 2;
                const float ix = unnormalize(gp[0], W, align);
                const float iy = unnormalize(gp[1], H, align);
                const float gx_mult = align ? (float)(W - 1) / 2.0f : (float)W / 2.0f;
                const float gy_mult = align ? (float)(H - 1) / 2.0f : (float)H / 2.0f;
                const float fx = floorf(ix), fy = floorf(iy);
                const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
                const float wx1 = ix - (float)x0, wx0 = (float)x1 - ix;
                const float wy1 = iy - (float)y0, wy0 = (float)y1 - iy;
                float gix = 0.0f, giy = 0.0f;
                for (int c = 0; c < C; ++c) {
                    const float* ip = in + ((size_t)n * C + c) * H * W;
                    float* dp = din + ((size_t)n * C + c) * H * W;
                    const float g = dout[(((size_t)n * C + c) * Ho + oy) * Wo + ox];
                    const int vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
                    const int vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
                    if (vy0 && vx0) { dp[(size_t)y0 * W + x0] += g * wx0 * wy0; const float v = ip[(size_t)y0 * W + x0]; gix -= v * wy0 * g; giy -= v * wx0 * g; }
                    if (vy0 && vx1) { dp[(size_t)y0 * W + x1] += g * wx1 * wy0; const float v = ip[(size_t)y0 * W + x1]; gix += v * wy0 * g; giy -= v * wx1 * g; }
                    if (vy1 && vx0) { dp[(size_t)y1 * W + x0] += g * wx0 * wy1; const float v = ip[(size_t)y1 * W + x0]; gix -= v * wy1 * g; giy += v * wx0 * g; }
                    if (vy1 && vx1) { dp[(size_t)y1 * W + x1] += g * wx1 * wy1; const float v = ip[(size_t)y1 * W + x1]; gix += v * wy1 * g; giy += v * wx1 * g; }
                }
                if (dgrid) {
                    float* dg = dgrid + (((size_t)n * Ho + oy) * Wo + ox) * 2;
                    dg[0] = gx_mult * gix;
                    dg[1] = gy_mult * giy;
                }
            }
}

/* ------------------------------------------------------------------------- */
/* Fixed-ratio bilinear up-sampling of LR features to the HR grid exactly as   */
/* core/models/model.py:153,162 does it: grid_sample(fmap, make_coord(th,tw)), */
/* bilinear, zero padding, align_corners=False (border taps blend towards 0).  */
/* ------------------------------------------------------------------------- */
ORC_API void orc_upsample_grid_bilinear(const float* in, float* out, int N, int C,
                                        int sh, int sw, int th, int tw)
{
    float* grid = (float*)malloc((size_t)th * tw * 2 * sizeof(float));
    for (int y = 0; y < th; ++y)
        for (int x = 0; x < tw; ++x) {
            grid[((size_t)y * tw + x) * 2 + 0] = coord_value(tw, x);
            grid[((size_t)y * tw + x) * 2 + 1] = coord_value(th, y);
        }
    for (int n = 0; n < N; ++n)
        orc_grid_sample(in + (size_t)n * C * sh * sw, grid, out + (size_t)n * C * th * tw,
                        1, C, sh, sw, th, tw, 0, 0);
    free(grid);
}

/* ------------------------------------------------------------------------- */
/* backward_warping (core/tools/math.py:10-17): pixel-unit motion -> grid ->   */
/* grid_sample(zeros).  The committed `.permute(0,2,3,1)` after a dim=-1 stack */
/* is a layout bug (it only runs when H==2); the evident intent, used here, is  */
/* the (N,H,W,2) grid with (x,y) last.  frame (N,C,h,w), motion (N,2,h,w),      */
/* motion[:,0]=x (right), motion[:,1]=y (down).  mode: 0 bilinear, 1 nearest.   */
/* ------------------------------------------------------------------------- */
ORC_API void orc_warp_grid(const float* motion, float* grid, int N, int h, int w)
{
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const float mx = motion[(((size_t)n * 2 + 0) * h + y) * w + x];
                const float my = motion[(((size_t)n * 2 + 1) * h + y) * w + x];
                /* math.py:14-15: ((g + m) / (size-1)) * 2.0 - 1.0, all fp32 */
                const float gx = (((float)x + mx) / (float)(w - 1)) * 2.0f - 1.0f;
                const float gy = (((float)y + my) / (float)(h - 1)) * 2.0f - 1.0f;
                float* gp = grid + (((size_t)n * h + y) * w + x) * 2;
                gp[0] = gx;
                gp[1] = gy;
            }
}

ORC_API void orc_backward_warp(const float* frame, const float* motion, float* out,
                               int N, int C, int h, int w, int mode, int align)
{
    float* grid = (float*)malloc((size_t)N * h * w * 2 * sizeof(float));
    orc_warp_grid(motion, grid, N, h, w);
    orc_grid_sample(frame, grid, out, N, C, h, w, h, w, mode, align);
    free(grid);
}

/* ------------------------------------------------------------------------- */
/* Relative-coordinate features (code_test.py:193-202, core/models/lte.py:     */
/* 164-171, commented in core/models/model.py:153-159):                        */
/*   rel = hr_coord - grid_sample(lr_coord, hr_coord, nearest, False);         */
/*   rel[0] *= sw; rel[1] *= sh.   out (2,th,tw).                               */
/* ------------------------------------------------------------------------- */
ORC_API void orc_rel_coord(float* out, int th, int tw, int sh, int sw)
{
    for (int y = 0; y < th; ++y)
        for (int x = 0; x < tw; ++x) {
            const float cx = coord_value(tw, x), cy = coord_value(th, y);
            const float ix = unnormalize(cx, sw, 0), iy = unnormalize(cy, sh, 0);
            const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
            float qx = 0.0f, qy = 0.0f;
            if (xn >= 0 && xn < sw && yn >= 0 && yn < sh) {
                qx = coord_value(sw, xn);
                qy = coord_value(sh, yn);
            }
            out[(size_t)y * tw + x] = (cx - qx) * (float)sw;
            out[(size_t)th * tw + (size_t)y * tw + x] = (cy - qy) * (float)sh;
        }
}

/* ------------------------------------------------------------------------- */
/* PixelShuffle / PixelUnshuffle (core/arch/RepRFN.py:272,276;                  */
/* core/models/model_final.py:234,408,449; core/utils.py:76).                   */
/* shuffle:   out[n,c,h*r+i,w*r+j] = in[n,c*r*r+i*r+j,h,w]                       */
/* ------------------------------------------------------------------------- */
ORC_API void orc_pixel_shuffle(const float* in, float* out, int N, int C, int H, int W, int r)
{
    /* in (N, C*r*r, H, W) -> out (N, C, H*r, W*r) */
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int h = 0; h < H; ++h)
                for (int i = 0; i < r; ++i)
                    for (int w = 0; w < W; ++w)
                        for (int j = 0; j < r; ++j)
                            out[(((size_t)n * C + c) * H * r + h * r + i) * W * r + w * r + j] =
                                in[(((size_t)n * C * r * r + c * r * r + i * r + j) * H + h) * W + w];
}

ORC_API void orc_pixel_unshuffle(const float* in, float* out, int N, int C, int H, int W, int r)
{
    /* in (N, C, H*r, W*r) -> out (N, C*r*r, H, W) */
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int h = 0; h < H; ++h)
                for (int i = 0; i < r; ++i)
                    for (int w = 0; w < W; ++w)
                        for (int j = 0; j < r; ++j)
                            out[(((size_t)n * C * r * r + c * r * r + i * r + j) * H + h) * W + w] =
                                in[(((size_t)n * C + c) * H * r + h * r + i) * W * r + w * r + j];
}

/* ------------------------------------------------------------------------- */
/* Nearest down-sampling used to synthesise LR inputs (core/trainer.py:92-95): */
/* F.interpolate(x, scale_factor=1/s, mode="nearest"); output size             */
/* floor(H*(1/s)); source index min(floor(dst * s'), H-1) with s' = 1/(1/s)    */
/* evaluated in fp32 as ATen does.                                              */
/* ------------------------------------------------------------------------- */
ORC_API void orc_nearest_down(const float* in, float* out, int NC, int H, int W, double scale_factor,
                              int Ho, int Wo)
{
    const float inv = (float)(1.0 / scale_factor);
    for (int p = 0; p < NC; ++p)
        for (int oy = 0; oy < Ho; ++oy) {
            int iy = (int)floorf((float)oy * inv);
            if (iy > H - 1) iy = H - 1;
            for (int ox = 0; ox < Wo; ++ox) {
                int ix = (int)floorf((float)ox * inv);
                if (ix > W - 1) ix = W - 1;
                out[((size_t)p * Ho + oy) * Wo + ox] = in[((size_t)p * H + iy) * W + ix];
            }
        }
}

/* ------------------------------------------------------------------------- */
/* Channels-first LayerNorm (core/arch/CAMixerSR.py:32-37): per-pixel mean and  */
/* biased variance over C, eps inside the sqrt, affine.                         */
/* ------------------------------------------------------------------------- */
ORC_API void orc_layernorm_cf(const float* x, const float* weight, const float* bias, float* y,
                              int N, int C, int HW, float eps)
{
    for (int n = 0; n < N; ++n)
        for (int p = 0; p < HW; ++p) {
            const float* xp = x + (size_t)n * C * HW + p;
            float* yp = y + (size_t)n * C * HW + p;
            float u = 0.0f;
            for (int c = 0; c < C; ++c) u += xp[(size_t)c * HW];
            u /= (float)C;
            float s = 0.0f;
            for (int c = 0; c < C; ++c) { const float d = xp[(size_t)c * HW] - u; s += d * d; }
            s /= (float)C;
            const float den = sqrtf(s + eps);
            for (int c = 0; c < C; ++c)
                yp[(size_t)c * HW] = weight[c] * ((xp[(size_t)c * HW] - u) / den) + bias[c];
        }
}

/* Backward of the channels-first LayerNorm (autograd of CAMixerSR.py:32-37):     */
/*   xh = (x - u) / den, g = dy * w                                              */
/*   dx = (g - mean_c(g) - xh * mean_c(g * xh)) / den;  dw = sum dy * xh;  db = sum dy */
ORC_API void orc_layernorm_cf_bwd(const float* x, const float* weight, const float* dy, float* dx,
                                  float* dweight, float* dbias, int N, int C, int HW, float eps)
{
    for (int c = 0; c < C; ++c) { dweight[c] = 0.0f; dbias[c] = 0.0f; }
    for (int n = 0; n < N; ++n)
        for (int p = 0; p < HW; ++p) {
            const float* xp = x + (size_t)n * C * HW + p;
            const float* gp = dy + (size_t)n * C * HW + p;
            float* dp = dx + (size_t)n * C * HW + p;
            double u = 0.0, s = 0.0;
            for (int c = 0; c < C; ++c) u += xp[(size_t)c * HW];
            u /= C;
            for (int c = 0; c < C; ++c) { const double d = xp[(size_t)c * HW] - u; s += d * d; }
            s /= C;
            const double den = sqrt(s + (double)eps);
            double mg = 0.0, mgx = 0.0;
            for (int c = 0; c < C; ++c) {
                const double xh = (xp[(size_t)c * HW] - u) / den, g = (double)gp[(size_t)c * HW] * weight[c];
                mg += g; mgx += g * xh;
                dweight[c] += (float)(gp[(size_t)c * HW] * xh);
                dbias[c] += gp[(size_t)c * HW];
            }
            mg /= C; mgx /= C;
            for (int c = 0; c < C; ++c) {
                const double xh = (xp[(size_t)c * HW] - u) / den, g = (double)gp[(size_t)c * HW] * weight[c];
                dp[(size_t)c * HW] = (float)((g - mg - xh * mgx) / den);
            }
        }
}

/* ------------------------------------------------------------------------- */
/* Disocclusion logic of the temporal variant, composed as its forward does        */
/* (core/models/model_final.py:472-486 and :501-504 over the functions at :50-139, */
/* colour distance core/tools/color.py:22-31).  G-buffer channels: normal 0:3,     */
/* metallic 3, specular 4, roughness 5, base colour 6:9, depth 9.                  */
/*   mask  = frame(5x5 reflect min/max, all 3 channels outside) | colour(3x3       */
/*           reflect, distance > 0.5 for more than 0.7 of the 9) | depth           */
/*   marked_frames = mask ? 0 : hist_frames;  aux = cur[0:3], cur[6:10],           */
/*           marked hist[0:3], marked hist[6:10];  enhance = mean of the normal,   */
/*           roughness (< 0.02) and metallic (> 0.95) masks over history and kind  */
/* ------------------------------------------------------------------------- */
static int orc_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

ORC_API void orc_disocclusion(const float* cur_frame, const float* cur_buffer, const float* hist_frames,
                              const float* hist_buffer, unsigned char* mask, float* marked_frames, float* aux,
                              float* enhance, int B, int Fh, int H, int W)
{
    const size_t hw = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t pix = (size_t)y * W + x;
                const float* cf = cur_frame + (size_t)b * 3 * hw;
                const float* cb = cur_buffer + (size_t)b * 10 * hw;
                float lo[3], hi[3];
                for (int c = 0; c < 3; ++c) {
                    lo[c] = INFINITY; hi[c] = -INFINITY;
                    for (int dy = -2; dy <= 2; ++dy)
                        for (int dx = -2; dx <= 2; ++dx) {
                            const float v = cf[c * hw + (size_t)orc_reflect(y + dy, H) * W + orc_reflect(x + dx, W)];
                            lo[c] = v < lo[c] ? v : lo[c]; hi[c] = v > hi[c] ? v : hi[c];
                        }
                }
                int count = 0;
                for (int f = 0; f < Fh; ++f) {
                    const float* hf = hist_frames + ((size_t)b * Fh + f) * 3 * hw;
                    const float* hb = hist_buffer + ((size_t)b * Fh + f) * 10 * hw;
                    int m_frame = 1;
                    for (int c = 0; c < 3; ++c) { const float v = hf[c * hw + pix]; m_frame &= (v < lo[c]) | (v > hi[c]); }
                    int far = 0;
                    const float r2 = hb[6 * hw + pix], b2 = hb[7 * hw + pix], g2 = hb[8 * hw + pix];
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            const size_t q = (size_t)orc_reflect(y + dy, H) * W + orc_reflect(x + dx, W);
                            const float r1 = cb[6 * hw + q], b1 = cb[7 * hw + q], g1 = cb[8 * hw + q];
                            const float r = (r1 + r2) / 2.0f, dr = r1 - r2, dg = g1 - g2, db = b1 - b2;
                            const float d = sqrtf(((2.0f + r) * (dr * dr) + 4.0f * (dg * dg)) + (2.0f + (1.0f - r)) * (db * db));
                            far += d > 0.5f;
                        }
                    const int m_color = ((float)far / 9.0f) > 0.7f;
                    const float cd = cb[9 * hw + pix], hd = hb[9 * hw + pix];
                    const int m_depth = (cd - hd) > (0.1f * cd);
                    const int m = m_frame | m_color | m_depth;
                    const size_t bf = (size_t)b * Fh + f;
                    if (mask) mask[bf * hw + pix] = (unsigned char)m;
                    for (int c = 0; c < 3; ++c) marked_frames[(bf * 3 + c) * hw + pix] = m ? 0.0f : hf[c * hw + pix];
                    float* a = aux + bf * 14 * hw + pix;
                    for (int c = 0; c < 3; ++c) a[c * hw] = cb[c * hw + pix];
                    for (int c = 0; c < 4; ++c) a[(3 + c) * hw] = cb[(6 + c) * hw + pix];
                    for (int c = 0; c < 3; ++c) a[(7 + c) * hw] = m ? 0.0f : hb[c * hw + pix];
                    for (int c = 0; c < 4; ++c) a[(10 + c) * hw] = m ? 0.0f : hb[(6 + c) * hw + pix];
                    float dot = 0.0f;
                    for (int c = 0; c < 3; ++c) dot += ((cb[c * hw + pix] - 0.5f) * 2.0f) * ((hb[c * hw + pix] - 0.5f) * 2.0f);
                    count += !(fabsf(dot) < 1e-3f) && (dot < 0.707f);
                    const float cr = cb[5 * hw + pix], cm = cb[3 * hw + pix];
                    count += (cr < 0.02f) | (fabsf(cr - hb[5 * hw + pix]) > 0.25f);
                    count += (cm > 0.95f) | (fabsf(cm - hb[3 * hw + pix]) > 0.25f);
                }
                enhance[(size_t)b * hw + pix] = (float)count / (float)(3 * Fh);
            }
}

/* model_final.py:40-46 (the formula is the identity below thres1, so the host-side max() test is dropped) */
ORC_API void orc_clamp_depth(const float* src, float* dst, long long n, float thres1, float thres2)
{
    for (long long i = 0; i < n; ++i) {
        const float d = src[i];
        dst[i] = d >= thres2 ? thres1 + log2f(thres2 - thres1) + log10f((d - thres2) + 1.0f)
                             : (d >= thres1 ? thres1 + log2f((d - thres1) + 1.0f) : d);
    }
}

/* core/tools/math.py:80-100: every pixel takes the motion vector of the pixel with the smallest depth in its
 * k x k neighbourhood.  The neighbourhood is read through F.unfold's ZERO padding (a border pixel sees depth 0 outside
 * the frame), the first minimum in row-major window order wins, the chosen position is clamped to the frame.
 * motion (B,H,W,2), depth (B,1,H,W) -> out (B,H,W,2). */
ORC_API void orc_dilate_velocity(const float* motion, const float* depth, float* out, int B, int H, int W, int k)
{
    const int r = k / 2;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float best = 0.0f;
                int pos = -1;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const int yy = y + ky - r, xx = x + kx - r;
                        const float v = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0.0f : depth[((size_t)b * H + yy) * W + xx];
                        if (pos < 0 || v < best || (v != v && best == best)) { best = v; pos = ky * k + kx; }
                    }
                int sx = x + pos % k - r, sy = y + pos / k - r;
                sx = sx < 0 ? 0 : (sx > W - 1 ? W - 1 : sx);
                sy = sy < 0 ? 0 : (sy > H - 1 ? H - 1 : sy);
                const float* m = motion + (((size_t)b * H + sy) * W + sx) * 2;
                float* o = out + (((size_t)b * H + y) * W + x) * 2;
                o[0] = m[0];
                o[1] = m[1];
            }
}

/* ------------------------------------------------------------------------- */
/* Pieces used only by the RepRFN ESA block (core/arch/RepRFN.py:295-308):      */
/* max_pool2d(k,s) without padding, bilinear resize align_corners=False.        */
/* ------------------------------------------------------------------------- */
ORC_API void orc_max_pool2d(const float* in, float* out, int NC, int H, int W, int k, int s)
{
    const int Ho = (H - k) / s + 1, Wo = (W - k) / s + 1;
    for (int p = 0; p < NC; ++p)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                float m = -INFINITY;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const float v = in[((size_t)p * H + oy * s + ky) * W + ox * s + kx];
                        if (v > m || v != v) m = v;
                    }
                out[((size_t)p * Ho + oy) * Wo + ox] = m;
            }
}

ORC_API void orc_resize_bilinear(const float* in, float* out, int NC, int H, int W, int Ho, int Wo)
{
    /* F.interpolate(mode="bilinear", align_corners=False, size=(Ho,Wo)): ATen
       area_pixel_compute_source_index with scale = in/out, clamped at 0. */
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    for (int p = 0; p < NC; ++p)
        for (int oy = 0; oy < Ho; ++oy) {
            float fy = sy * ((float)oy + 0.5f) - 0.5f;
            if (fy < 0.0f) fy = 0.0f;
            const int y0 = (int)fy;
            const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
            const float ly = fy - (float)y0, hy = 1.0f - ly;
            for (int ox = 0; ox < Wo; ++ox) {
                float fx = sx * ((float)ox + 0.5f) - 0.5f;
                if (fx < 0.0f) fx = 0.0f;
                const int x0 = (int)fx;
                const int x1 = x0 + (x0 < W - 1 ? 1 : 0);
                const float lx = fx - (float)x0, hx = 1.0f - lx;
                const float* ip = in + (size_t)p * H * W;
                out[((size_t)p * Ho + oy) * Wo + ox] =
                    hy * (hx * ip[(size_t)y0 * W + x0] + lx * ip[(size_t)y0 * W + x1]) +
                    ly * (hx * ip[(size_t)y1 * W + x0] + lx * ip[(size_t)y1 * W + x1]);
            }
        }
}

/* nn.ConvTranspose2d(kernel 2, stride 2, padding 0), core/models/model_final.py:198-203:
   out[n][co][2y+i][2x+j] = b[co] + sum_ci in[n][ci][y][x] * w[ci][co][i][j]   (w: (Cin, Cout, 2, 2)) */
ORC_API void orc_conv_transpose2x2s2(const float* in, const float* w, const float* b, float* out, int N, int Cin,
                                     int H, int W, int Cout, int relu)
{
    const int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    const int y = oy >> 1, i = oy & 1, x = ox >> 1, j = ox & 1;
                    float acc = b ? b[co] : 0.0f;
                    for (int ci = 0; ci < Cin; ++ci)
                        acc += in[(((size_t)n * Cin + ci) * H + y) * W + x] * w[(((size_t)ci * Cout + co) * 2 + i) * 2 + j];
                    if (relu && acc < 0.0f) acc = 0.0f;
                    out[(((size_t)n * Cout + co) * Ho + oy) * Wo + ox] = acc;
                }
}

/* F.interpolate / nn.Upsample(mode="bilinear", align_corners=True): src = dst * (in-1)/(out-1)
   (core/models/model_final.py:281-289). */
ORC_API void orc_resize_bilinear_ac(const float* in, float* out, int NC, int H, int W, int Ho, int Wo)
{
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.0f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.0f;
    for (int p = 0; p < NC; ++p)
        for (int oy = 0; oy < Ho; ++oy) {
            const float fy = sy * (float)oy;
            const int y0 = (int)fy;
            const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
            const float ly = fy - (float)y0, hy = 1.0f - ly;
            for (int ox = 0; ox < Wo; ++ox) {
                const float fx = sx * (float)ox;
                const int x0 = (int)fx;
                const int x1 = x0 + (x0 < W - 1 ? 1 : 0);
                const float lx = fx - (float)x0, hx = 1.0f - lx;
                const float* ip = in + (size_t)p * H * W;
                out[((size_t)p * Ho + oy) * Wo + ox] =
                    hy * (hx * ip[(size_t)y0 * W + x0] + lx * ip[(size_t)y0 * W + x1]) +
                    ly * (hx * ip[(size_t)y1 * W + x0] + lx * ip[(size_t)y1 * W + x1]);
            }
        }
}

/* GradientMap, core/models/model_final.py:160-187: sqrt(gv^2 + gh^2 + eps) of the fixed 3x3 stencils
   kernel_v = [[-1,-1,-1],[0,0,0],[1,1,1]], kernel_h = [[-1,0,1],[-1,0,1],[-1,0,1]], zero padding. */
ORC_API void orc_gradient_map(const float* in, float* out, int NC, int H, int W, float eps)
{
    for (int p = 0; p < NC; ++p)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float gv = 0.0f, gh = 0.0f;
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const int yy = y + ky - 1, xx = x + kx - 1;
                        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                        const float v = in[((size_t)p * H + yy) * W + xx];
                        gv += (float)(ky - 1) * v;
                        gh += (float)(kx - 1) * v;
                    }
                out[((size_t)p * H + y) * W + x] = sqrtf(gv * gv + gh * gh + eps);
            }
}

ORC_API int orc_version(void) { return 1; }
