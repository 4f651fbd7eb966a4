// Generic direct convolution on the vector ALU (NHWC bf16 activations, fp32 OIHW weights, fp32
// accumulate).  Serves the small, memory-bound layers that do not fit the 3x3 MFMA kernel: 1x1,
// strided, 3x1 / 1x3 kernels and tiny channel counts.  blockIdx.y selects a group of 8 output
// channels so weight addresses are wave-uniform (scalar loads).
#include "srhip_common.h"

namespace {

__global__ __launch_bounds__(256) void conv_direct_kernel(
    const unsigned short* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bias,
    const unsigned short* __restrict__ res, unsigned short* __restrict__ dst, int n, int h, int wd, int cin,
    int cin_stride, int cout, int cout_stride, int kh, int kw, int sh, int sw, int ph, int pw, int ho, int wo, int act,
    float slope, float alpha)
{
    const int co0 = blockIdx.y * 8;
    const long long total = (long long)n * ho * wo;
    const int khw = kh * kw;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
         p += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(p % wo);
        long long t = p / wo;
        const int oy = (int)(t % ho);
        const int img = (int)(t / ho);
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = (bias != nullptr && co0 + o < cout) ? bias[co0 + o] : 0.0f;
        for (int ky = 0; ky < kh; ++ky) {
            const int iy = oy * sh - ph + ky;
            if (iy < 0 || iy >= h) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int ix = ox * sw - pw + kx;
                if (ix < 0 || ix >= wd) continue;
                const unsigned short* sp = src + (((long long)img * h + iy) * wd + ix) * cin_stride;
                for (int ci0 = 0; ci0 < cin; ci0 += 8) {
                    const uint4 v = *(const uint4*)(sp + ci0);
                    const float xin[8] = {bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y),
                                          bf16_lo(v.z), bf16_hi(v.z), bf16_lo(v.w), bf16_hi(v.w)};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (ci0 + j < cin) {
#pragma unroll
                            for (int o = 0; o < 8; ++o) {
                                if (co0 + o < cout)
                                    acc[o] += xin[j] * w[((long long)(co0 + o) * cin + ci0 + j) * khw + ky * kw + kx];
                            }
                        }
                    }
                }
            }
        }
        float v[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float a = acc[o];
            if (act == SRHIP_ACT_RELU) a = fmaxf(a, 0.0f);
            else if (act == SRHIP_ACT_LRELU) a = a > 0.0f ? a : a * slope;
            v[o] = (co0 + o < cout) ? a * alpha : 0.0f;
        }
        if (res != nullptr) {
            const uint4 r = *(const uint4*)(res + p * cout_stride + co0);
            v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
            v[4] += bf16_lo(r.z); v[5] += bf16_hi(r.z); v[6] += bf16_lo(r.w); v[7] += bf16_hi(r.w);
        }
        *(uint4*)(dst + p * cout_stride + co0) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                                            pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
}

// nn.ConvTranspose2d(kernel 2, stride 2, padding 0) (+ReLU): every output pixel (2y+i, 2x+j) sees exactly one
// input pixel (y, x) through the 1x1 slice w[:, :, i, j] of the (Cin, Cout, 2, 2) weight.
__global__ __launch_bounds__(256) void conv_transpose2x2s2_kernel(
    const unsigned short* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bias,
    unsigned short* __restrict__ dst, int n, int h, int wd, int cin, int cin_stride, int cout, int cout_stride, int act)
{
    const int co0 = blockIdx.y * 8;
    const int ho = 2 * h, wo = 2 * wd;
    const long long total = (long long)n * ho * wo;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
         p += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(p % wo);
        long long t = p / wo;
        const int oy = (int)(t % ho);
        const int img = (int)(t / ho);
        const int tap = (oy & 1) * 2 + (ox & 1);
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = (bias != nullptr && co0 + o < cout) ? bias[co0 + o] : 0.0f;
        const unsigned short* sp = src + (((long long)img * h + (oy >> 1)) * wd + (ox >> 1)) * cin_stride;
        for (int ci0 = 0; ci0 < cin; ci0 += 8) {
            const uint4 v = *(const uint4*)(sp + ci0);
            const float xin[8] = {bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y),
                                  bf16_lo(v.z), bf16_hi(v.z), bf16_lo(v.w), bf16_hi(v.w)};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (ci0 + j < cin) {
#pragma unroll
                    for (int o = 0; o < 8; ++o)
                        if (co0 + o < cout) acc[o] += xin[j] * w[((long long)(ci0 + j) * cout + co0 + o) * 4 + tap];
                }
            }
        }
        float v[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float a = acc[o];
            if (act == SRHIP_ACT_RELU) a = fmaxf(a, 0.0f);
            v[o] = (co0 + o < cout) ? a : 0.0f;
        }
        *(uint4*)(dst + p * cout_stride + co0) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                                            pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
}

}  // namespace

extern "C" SRHIP_API int srhip_conv2d_direct(const void* src, const float* w, const float* bias, const void* res,
                                             void* dst, int n, int h, int wd, int cin, int cin_stride, int cout,
                                             int cout_stride, int kh, int kw, int sh, int sw, int ph, int pw, int act,
                                             float slope, float alpha, void* stream)
{
    SRHIP_REQUIRE(src && w && dst, "conv2d_direct: null pointer");
    SRHIP_REQUIRE(n >= 1 && h >= 1 && wd >= 1 && cin >= 1 && cout >= 1, "conv2d_direct: bad shape");
    SRHIP_REQUIRE(cin_stride % 8 == 0 && cin_stride >= cin, "conv2d_direct: cin_stride must be a multiple of 8 and >= cin");
    SRHIP_REQUIRE(cout_stride % 8 == 0 && cout_stride >= cout, "conv2d_direct: cout_stride must be a multiple of 8 and >= cout");
    SRHIP_REQUIRE(kh >= 1 && kw >= 1 && sh >= 1 && sw >= 1 && ph >= 0 && pw >= 0, "conv2d_direct: bad kernel geometry");
    const int ho = (h + 2 * ph - kh) / sh + 1, wo = (wd + 2 * pw - kw) / sw + 1;
    SRHIP_REQUIRE(ho >= 1 && wo >= 1, "conv2d_direct: empty output");
    const long long total = (long long)n * ho * wo;
    long long gx = (total + 255) / 256;
    const long long cap = (long long)srhip_num_cus() * 8;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)gx, (unsigned)(cout_stride / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)src, w, bias, (const unsigned short*)res, (unsigned short*)dst, n, h, wd, cin,
                       cin_stride, cout, cout_stride, kh, kw, sh, sw, ph, pw, ho, wo, act, slope, alpha);
    return srhip_check_launch("conv_direct_kernel");
}

extern "C" SRHIP_API int srhip_conv_transpose2x2s2(const void* src, const float* w, const float* bias, void* dst, int n,
                                                   int h, int wd, int cin, int cin_stride, int cout, int cout_stride,
                                                   int act, void* stream)
{
    SRHIP_REQUIRE(src && w && dst, "conv_transpose2x2s2: null pointer");
    SRHIP_REQUIRE(n >= 1 && h >= 1 && wd >= 1 && cin >= 1 && cout >= 1, "conv_transpose2x2s2: bad shape");
    SRHIP_REQUIRE(cin_stride % 8 == 0 && cin_stride >= cin, "conv_transpose2x2s2: cin_stride must be a multiple of 8 and >= cin");
    SRHIP_REQUIRE(cout_stride % 8 == 0 && cout_stride >= cout, "conv_transpose2x2s2: cout_stride must be a multiple of 8 and >= cout");
    SRHIP_REQUIRE(act == SRHIP_ACT_NONE || act == SRHIP_ACT_RELU, "conv_transpose2x2s2: activation must be none or ReLU");
    const long long total = (long long)n * h * wd * 4;
    long long gx = (total + 255) / 256;
    const long long cap = (long long)srhip_num_cus() * 8;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(conv_transpose2x2s2_kernel, dim3((unsigned)gx, (unsigned)(cout_stride / 8)), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned short*)src, w, bias, (unsigned short*)dst, n, h, wd, cin,
                       cin_stride, cout, cout_stride, act);
    return srhip_check_launch("conv_transpose2x2s2_kernel");
}
