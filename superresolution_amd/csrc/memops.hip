// HBM-bound operators of the super-resolution path: layout changes, grid-sample style
// resampling (fixed-ratio up-sampling and motion-vector warp), pixel (un)shuffle, nearest
// down-sampling, coordinate features, channel concat and channels-first LayerNorm.
// One thread moves 16 bytes wherever the layout allows it; grids are capped and grid-strided.
#include <hip/hip_fp16.h>

#include <mutex>

#include <cstdlib>

#include "srhip_common.h"

// The coordinate / index arithmetic below replays the reference's fp32 operator sequence one
// rounding at a time.  HIP's __fmul_rn / __fadd_rn are plain operators that the default
// -ffp-contract=fast fuses into FMAs, so this file is compiled with -ffp-contract=off (Makefile)
// and the pragma below covers the code written here.
#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;
inline int grid_for(long long work, int ncu) {
    long long g = (work + kThreads - 1) / kThreads;
    const long long cap = (long long)ncu * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------------
// make_coord replay (fp32 operator sequence of the reference): coord(n, j) = base + step * j
// with base = fp32(-1 + 1/n), step = fp32(2 * (1/n)) formed in double on the host.
// ------------------------------------------------------------------------------------------
struct CoordAxis { float base, step; };
inline CoordAxis coord_axis(int n) {
    const double r = 2.0 / (2.0 * (double)n);
    CoordAxis a; a.step = (float)(2.0 * r); a.base = (float)(-1.0 + r); return a;
}
__device__ __forceinline__ float coord_val(CoordAxis a, int j) {
    return __fadd_rn(a.base, __fmul_rn(a.step, (float)j));  // two separate fp32 roundings, no fma
}
// ATen grid_sampler unnormalize
__device__ __forceinline__ float unnorm(float c, int size, bool align) {
    // division by 2 is exact in binary floating point, so * 0.5f is the same rounding as ATen's / 2
    return align ? __fmul_rn(__fmul_rn(__fadd_rn(c, 1.0f), 0.5f), (float)(size - 1))
                 : __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(c, 1.0f), (float)size), 1.0f), 0.5f);
}

struct Taps {  // bilinear footprint with zero padding
    int x0, y0;
    float wnw, wne, wsw, wse;
    bool vx0, vx1, vy0, vy1;
};
__device__ __forceinline__ Taps make_taps(float ix, float iy, int W, int H) {
    Taps t;
    const float fx = floorf(ix), fy = floorf(iy);
    t.x0 = (int)fx; t.y0 = (int)fy;
    const float x1f = fx + 1.0f, y1f = fy + 1.0f;
    t.wnw = __fmul_rn(x1f - ix, y1f - iy);
    t.wne = __fmul_rn(ix - fx, y1f - iy);
    t.wsw = __fmul_rn(x1f - ix, iy - fy);
    t.wse = __fmul_rn(ix - fx, iy - fy);
    t.vx0 = t.x0 >= 0 && t.x0 < W; t.vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    t.vy0 = t.y0 >= 0 && t.y0 < H; t.vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
    return t;
}

// Work-group order for kernels that gather from a neighbourhood of their own rows.  The dispatcher deals consecutive
// work-groups round-robin over the 8 XCDs, each with an L2 of its own: with the natural order every source line is
// fetched by most of the 8 L2s.  Here work-group b takes unit (b % 8) * ceil(units / 8) + b / 8, so each XCD walks ONE
// contiguous band of rows and neighbouring rows' taps hit in its L2.  Launch 8 * ceil(units / 8) work-groups.
__device__ __forceinline__ bool xcd_band_unit(unsigned units, unsigned& unit)
{
    const unsigned per = (units + 7u) >> 3;
    unit = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    return unit < units;
}
inline unsigned xcd_band_grid(long long units) { return (unsigned)(((units + 7) / 8) * 8); }

struct __attribute__((packed, aligned(4))) FloatPair { float a, b; };   // two floats at a 4-byte-aligned address

// ------------------------------------------------------------------------------------------
// layout kernels
// ------------------------------------------------------------------------------------------
template <int CPAD>
__global__ void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                             int n, int c, long long hw)
{
    const long long total = (long long)n * hw;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
         p += (long long)gridDim.x * blockDim.x) {
        const long long img = p / hw, pix = p - img * hw;
        const float* s = src + img * c * hw + pix;
        unsigned out[CPAD / 2];
#pragma unroll
        for (int k = 0; k < CPAD / 2; ++k) {
            const float a = (2 * k < c) ? s[(long long)(2 * k) * hw] : 0.0f;
            const float b = (2 * k + 1 < c) ? s[(long long)(2 * k + 1) * hw] : 0.0f;
            out[k] = pack_bf16x2(a, b);
        }
        uint4* d = (uint4*)(dst + p * CPAD);
#pragma unroll
        for (int k = 0; k < CPAD / 8; ++k) d[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
    }
}

// any channel count: one thread = one pixel x 8 channels
__global__ void nchw_f32_to_nhwc_bf16_generic_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                     int n, int c, long long hw, int cpad)
{
    const int cg = cpad / 8;
    const long long total = (long long)n * hw * cg;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i % ((long long)n * hw);  // pixel fastest: coalesced reads of each channel plane
        const int g = (int)(i / ((long long)n * hw));
        const long long img = p / hw, pix = p - img * hw;
        const float* sp = src + img * c * hw + pix;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (g * 8 + k < c) ? sp[(long long)(g * 8 + k) * hw] : 0.0f;
        *(uint4*)(dst + p * cpad + g * 8) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                                       pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
}

// The same conversion staged through LDS: a block takes 64 consecutive pixels, reads every channel plane with the pixel
// index fastest (coalesced 256-byte runs) and writes the 64 * cpad * 2 contiguous output bytes in linear order, 16 bytes
// per lane.  Row stride in LDS is padded to an odd number of 16-byte entries.
constexpr int kLayoutTile = 64;
__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_bf16_tiled_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                                         int n, int c, long long hw, int cpad)
{
    extern __shared__ uint4 tile[];
    const int cg = cpad / 8, cgp = cg | 1;
    const long long total = (long long)n * hw;
    const long long ntiles = (total + kLayoutTile - 1) / kLayoutTile;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long p0 = t * kLayoutTile;
        const int npx = (int)(total - p0 < kLayoutTile ? total - p0 : kLayoutTile);
        for (int it = threadIdx.x; it < kLayoutTile * cg; it += blockDim.x) {
            const int px = it & (kLayoutTile - 1), g = it / kLayoutTile;
            if (px < npx) {
                const long long p = p0 + px;
                const long long img = p / hw, pix = p - img * hw;
                const float* sp = src + img * c * hw + pix;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (g * 8 + k < c) ? sp[(long long)(g * 8 + k) * hw] : 0.0f;
                tile[px * cgp + g] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                                pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
            }
        }
        __syncthreads();
        uint4* out = (uint4*)(dst + p0 * cpad);
        for (int it = threadIdx.x; it < npx * cg; it += blockDim.x) {
            const int px = it / cg, g = it - px * cg;
            out[it] = tile[px * cgp + g];
        }
        __syncthreads();
    }
}

// nn.PixelUnshuffle(4) of an fp32 NCHW tensor written straight as NHWC bf16 (core/models/model_final.py:234,408 feed
// convolutions): out[n][y][x][c*16 + i*4 + j] = in[n][c][4y+i][4x+j].  A block takes 64 output pixels; every lane reads
// 16 contiguous bytes (the four j of one (c, i)), pixel index fastest, and the 64 * 16c * 2 output bytes leave in linear
// order through LDS.  LDS rows are 8-byte entries of four channels, padded by two entries.
// WITH_MAX: block b also leaves the largest input value it read in block_max[b] (NaN if it read one, as torch.max
// does) and block 0 fills block_max[gridDim.x .. max_cap) with -inf: the maximum of the whole tensor
// (core/models/model_final.py:463 asserts on it) is the maximum of block_max[0 .. max_cap), with no second pass over
// the input and no atomics.
template <bool WITH_MAX>
__global__ __launch_bounds__(256) void pixel_unshuffle4_nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ src,
                                                                                    unsigned short* __restrict__ dst, int n, int c,
                                                                                    int h, int w, float* __restrict__ block_max,
                                                                                    int max_cap)
{
    float vmax = -INFINITY;
    bool vnan = false;
    extern __shared__ uint4 tile[];
    uint2* t2 = (uint2*)tile;
    const int nq = c * 4, q = nq + 2, cg = c * 2;          // (c, i) pairs per pixel; row stride; 16-byte groups per pixel
    const long long total = (long long)n * h * w;
    const long long ntiles = (total + kLayoutTile - 1) / kLayoutTile;
    const long long H = 4ll * h, W = 4ll * w;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long p0 = t * kLayoutTile;
        const int npx = (int)(total - p0 < kLayoutTile ? total - p0 : kLayoutTile);
        for (int it = threadIdx.x; it < kLayoutTile * nq; it += blockDim.x) {
            const int px = it & (kLayoutTile - 1), ci = it / kLayoutTile;
            if (px < npx) {
                const long long p = p0 + px;
                const int x = (int)(p % w);
                const long long r = p / w;
                const int y = (int)(r % h);
                const long long img = r / h;
                const float4 v = *(const float4*)(src + ((img * c + (ci >> 2)) * H + 4ll * y + (ci & 3)) * W + 4ll * x);
                t2[px * q + ci] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                if constexpr (WITH_MAX) {
                    vmax = fmaxf(fmaxf(vmax, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                    vnan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
                }
            }
        }
        __syncthreads();
        uint4* out = (uint4*)(dst + p0 * (16ll * c));
        for (int it = threadIdx.x; it < npx * cg; it += blockDim.x) {
            const int px = it / cg, g = it - px * cg;
            out[it] = *(const uint4*)(t2 + px * q + 2 * g);
        }
        __syncthreads();
    }
    if constexpr (WITH_MAX) {
        float* red = (float*)tile;           // the tile is free again (barrier at the end of the loop body)
        if (vnan) vmax = __int_as_float(0x7fc00000);
        red[threadIdx.x] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = -INFINITY;
            bool bad = false;
            for (int i = 0; i < (int)blockDim.x; ++i) { const float r = red[i]; bad |= (r != r); m = fmaxf(m, r); }
            block_max[blockIdx.x] = bad ? __int_as_float(0x7fc00000) : m;
        }
        if (blockIdx.x == 0)
            for (int i = (int)gridDim.x + (int)threadIdx.x; i < max_cap; i += (int)blockDim.x) block_max[i] = -INFINITY;
    }
}

__global__ void nhwc_bf16_to_nchw_f32_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst,
                                             int n, int c, long long hw, int cstride)
{
    const long long total = (long long)n * hw;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
         p += (long long)gridDim.x * blockDim.x) {
        const long long img = p / hw, pix = p - img * hw;
        const unsigned short* s = src + p * cstride;
        float* d = dst + img * c * hw + pix;
        for (int k = 0; k < c; ++k) d[(long long)k * hw] = bf16_bits_to_f32(s[k]);
    }
}

// ------------------------------------------------------------------------------------------
// grid-sample family.  NHWC bf16: one thread = one output pixel x 8 channels (16 B).
// ------------------------------------------------------------------------------------------
// Branch-free: every tap is loaded from a clamped (always valid) address and contributes +0 when it lies outside the
// image, so the four 16-byte loads of a pixel are in flight together (behind a branch per tap each load waited for the
// one before it).  acc starts at +0 and a sum that started from +0 is never -0, so adding +0 leaves it as skipping the tap
// would: bit-identical to the branching form.
__device__ __forceinline__ void blend8(const unsigned short* __restrict__ src, long long img_base, int W, int H, int C,
                                       int c0, const Taps& t, unsigned short* __restrict__ out)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int xa = min(max(t.x0, 0), W - 1), xb = min(max(t.x0 + 1, 0), W - 1);
    const int ya = min(max(t.y0, 0), H - 1), yb = min(max(t.y0 + 1, 0), H - 1);
    const uint4 v00 = *(const uint4*)(src + (img_base + (long long)ya * W + xa) * C + c0);
    const uint4 v01 = *(const uint4*)(src + (img_base + (long long)ya * W + xb) * C + c0);
    const uint4 v10 = *(const uint4*)(src + (img_base + (long long)yb * W + xa) * C + c0);
    const uint4 v11 = *(const uint4*)(src + (img_base + (long long)yb * W + xb) * C + c0);
    auto tap = [&](const uint4& v, bool ok, float wgt) {
        const float e[8] = {bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y), bf16_lo(v.z), bf16_hi(v.z), bf16_lo(v.w), bf16_hi(v.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += ok ? e[j] * wgt : 0.0f;
    };
    tap(v00, t.vy0 && t.vx0, t.wnw);
    tap(v01, t.vy0 && t.vx1, t.wne);
    tap(v10, t.vy1 && t.vx0, t.wsw);
    tap(v11, t.vy1 && t.vx1, t.wse);
    *(uint4*)out = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]),
                              pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
}

__global__ void upsample_nhwc_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                     int c, int sh, int sw, int th, int tw, CoordAxis ax, CoordAxis ay)
{
    // grid: (chunks of a target row, target row, image): no 64-bit index arithmetic per thread
    const unsigned cg = (unsigned)c / 8u;
    const unsigned xi = blockIdx.x * blockDim.x + threadIdx.x;
    if (xi >= (unsigned)tw * cg) return;
    const int x = (int)(xi / cg), g = (int)(xi - (unsigned)x * cg);
    const int y = blockIdx.y, img = blockIdx.z;
    const float ix = unnorm(coord_val(ax, x), sw, false);
    const float iy = unnorm(coord_val(ay, y), sh, false);
    const Taps t = make_taps(ix, iy, sw, sh);
    blend8(src, (long long)img * sh * sw, sw, sh, c, g * 8, t, dst + (((long long)img * th + y) * tw + x) * c + g * 8);
}

// Integer ratio R = 2 or 4 (the 4x network): one thread = one SOURCE cell x 8 channels = R x R output pixels.  The
// general kernel above spends 183 vector instructions per 16-byte output, most of them index and weight arithmetic and
// the unpacking of four taps, and is bound by them (0.15 ms of pure issue time at 4K); here the 3 x 3 source
// neighbourhood is loaded and unpacked once (pixels outside the image as zeros: the sampler's zero padding) and the R
// column and R row weight pairs are evaluated once each, with the general kernel's own fp32 arithmetic, so the output
// is identical bit for bit.  Output column R*j + b reads source columns (j-1, j) for b < R/2 and (j, j+1) otherwise:
// the sampling positions of an even ratio lie at least 1/(2R) away from every source pixel centre, far beyond the
// rounding error of the coordinate arithmetic, so that choice is made at compile time.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int R>
__global__ __launch_bounds__(256) void upsample_int_nhwc_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                                                int c, int sh, int sw, CoordAxis ax, CoordAxis ay)
{
    // grid: (chunks of a source row x channel groups, source row, image)
    const unsigned cg = (unsigned)c / 8u;
    const unsigned xi = blockIdx.x * blockDim.x + threadIdx.x;
    if (xi >= (unsigned)sw * cg) return;
    const int j = (int)(xi / cg), g = (int)(xi - (unsigned)j * cg);
    const int i = blockIdx.y, img = blockIdx.z;
    const int th = sh * R, tw = sw * R;

    // 3 x 3 neighbourhood, 8 channels each, as 4 float pairs
    f32x2_t v[3][3][4];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int yy = i - 1 + dy, xx = j - 1 + dx;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (yy >= 0 && yy < sh && xx >= 0 && xx < sw)
                q = *(const uint4*)(src + (((long long)img * sh + yy) * sw + xx) * c + g * 8);
            v[dy][dx][0] = f32x2_t{bf16_lo(q.x), bf16_hi(q.x)};
            v[dy][dx][1] = f32x2_t{bf16_lo(q.y), bf16_hi(q.y)};
            v[dy][dx][2] = f32x2_t{bf16_lo(q.z), bf16_hi(q.z)};
            v[dy][dx][3] = f32x2_t{bf16_lo(q.w), bf16_hi(q.w)};
        }
    // weight pairs of the R output columns and rows (make_taps' arithmetic)
    float wx0[R], wx1[R], wy0[R], wy1[R];
#pragma unroll
    for (int b = 0; b < R; ++b) {
        const float ix = unnorm(coord_val(ax, R * j + b), sw, false), fx = floorf(ix);
        wx0[b] = (fx + 1.0f) - ix; wx1[b] = ix - fx;
        const float iy = unnorm(coord_val(ay, R * i + b), sh, false), fy = floorf(iy);
        wy0[b] = (fy + 1.0f) - iy; wy1[b] = iy - fy;
    }
    unsigned short* o = dst + ((((long long)img * th + (long long)R * i) * tw + (long long)R * j) * c + g * 8);
#pragma unroll
    for (int a = 0; a < R; ++a) {
        constexpr int H2 = R / 2;
        const int r0 = a < H2 ? 0 : 1;
#pragma unroll
        for (int b = 0; b < R; ++b) {
            const int c0 = b < H2 ? 0 : 1;
            const float wnw = __fmul_rn(wx0[b], wy0[a]), wne = __fmul_rn(wx1[b], wy0[a]);
            const float wsw = __fmul_rn(wx0[b], wy1[a]), wse = __fmul_rn(wx1[b], wy1[a]);
            unsigned pk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // 0 + nw, + ne, + sw, + se: the general kernel's order (its first addition to zero is exact)
                f32x2_t acc = v[r0][c0][k] * wnw;
                acc = acc + v[r0][c0 + 1][k] * wne;
                acc = acc + v[r0 + 1][c0][k] * wsw;
                acc = acc + v[r0 + 1][c0 + 1][k] * wse;
                pk[k] = pack_bf16x2(acc.x, acc.y);
            }
            *(uint4*)(o + ((long long)a * tw + b) * c) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
}

// Backward of upsample_nhwc_kernel as a gather: one thread = one SOURCE pixel x 8 channels.  Every
// target pixel whose bilinear footprint touches the source pixel is found by re-evaluating the
// forward's own fp32 index arithmetic over a conservative candidate range, so the weights are the
// forward's weights bit for bit and no atomics are needed.
__device__ __forceinline__ float tap_weight(CoordAxis a, int t, int ssize, int s)
{
    const float i = unnorm(coord_val(a, t), ssize, false);
    const float f = floorf(i);
    const int i0 = (int)f;
    if (i0 == s) return (f + 1.0f) - i;
    if (i0 + 1 == s) return i - f;
    return 0.0f;
}
__global__ void upsample_bwd_nhwc_kernel(const unsigned short* __restrict__ ddst, const unsigned short* __restrict__ relu_out,
                                         unsigned short* __restrict__ dsrc, int n, int c, int sh, int sw, int th,
                                         int tw, CoordAxis ax, CoordAxis ay)
{
    const int cg = c / 8;
    const long long total = (long long)n * sh * sw * cg;
    const float rx = (float)tw / (float)sw, ry = (float)th / (float)sh;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long long p = i / cg;
        const int sx = (int)(p % sw); p /= sw;
        const int sy = (int)(p % sh);
        const int img = (int)(p / sh);
        const int xlo = max(0, (int)floorf(((float)sx - 0.5f) * rx - 0.5f) - 1);
        const int xhi = min(tw - 1, (int)ceilf(((float)sx + 1.5f) * rx - 0.5f) + 1);
        const int ylo = max(0, (int)floorf(((float)sy - 0.5f) * ry - 0.5f) - 1);
        const int yhi = min(th - 1, (int)ceilf(((float)sy + 1.5f) * ry - 0.5f) + 1);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int y = ylo; y <= yhi; ++y) {
            const float wy = tap_weight(ay, y, sh, sy);
            if (wy == 0.0f) continue;
            const unsigned short* row = ddst + (((long long)img * th + y) * tw) * c + g * 8;
            for (int x = xlo; x <= xhi; ++x) {
                const float wx = tap_weight(ax, x, sw, sx);
                if (wx == 0.0f) continue;
                const float wgt = __fmul_rn(wx, wy);
                const uint4 v = *(const uint4*)(row + (long long)x * c);
                acc[0] += bf16_lo(v.x) * wgt; acc[1] += bf16_hi(v.x) * wgt;
                acc[2] += bf16_lo(v.y) * wgt; acc[3] += bf16_hi(v.y) * wgt;
                acc[4] += bf16_lo(v.z) * wgt; acc[5] += bf16_hi(v.z) * wgt;
                acc[6] += bf16_lo(v.w) * wgt; acc[7] += bf16_hi(v.w) * wgt;
            }
        }
        const long long o = (((long long)img * sh + sy) * sw + sx) * c + g * 8;
        if (relu_out != nullptr) {
            const uint4 m = *(const uint4*)(relu_out + o);
            if (!(bf16_lo(m.x) > 0.0f)) acc[0] = 0.0f;
            if (!(bf16_hi(m.x) > 0.0f)) acc[1] = 0.0f;
            if (!(bf16_lo(m.y) > 0.0f)) acc[2] = 0.0f;
            if (!(bf16_hi(m.y) > 0.0f)) acc[3] = 0.0f;
            if (!(bf16_lo(m.z) > 0.0f)) acc[4] = 0.0f;
            if (!(bf16_hi(m.z) > 0.0f)) acc[5] = 0.0f;
            if (!(bf16_lo(m.w) > 0.0f)) acc[6] = 0.0f;
            if (!(bf16_hi(m.w) > 0.0f)) acc[7] = 0.0f;
        }
        *(uint4*)(dsrc + o) = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]),
                                         pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
    }
}
// The same gather for the integer ratio R (th = R sh, tw = R sw): the taps with a non-zero weight are 2 R consecutive
// target columns / rows starting at R s - R / 2, so a thread evaluates 2 R + 2 R tap weights (the very same fp32 coordinate
// replay as above) instead of one per visited (row, column) pair, and its 2 R loads of a row are independent.  Same
// products, same order of the sum: bit-equal to the kernel above.
template <int R>
__global__ __launch_bounds__(kThreads) void upsample_bwd_int_nhwc_kernel(const unsigned short* __restrict__ ddst,
                                                                        const unsigned short* __restrict__ relu_out,
                                                                        unsigned short* __restrict__ dsrc, int n, int c, int sh, int sw,
                                                                        CoordAxis ax, CoordAxis ay)
{
    constexpr int NT = 2 * R;
    const int cg = c / 8, th = R * sh, tw = R * sw;
    const long long total = (long long)n * sh * sw * cg;
    // work-groups in XCD bands: neighbouring source rows share half of their target rows, and with the natural order those
    // were fetched by most of the eight L2s (2.25 GB fetched for a 1.07 GB gradient)
    unsigned unit;
    if (!xcd_band_unit((unsigned)((total + kThreads - 1) / kThreads), unit)) return;
    {
        const long long i = (long long)unit * kThreads + threadIdx.x;
        if (i >= total) return;
        const int g = (int)(i % cg);
        long long p = i / cg;
        const int sx = (int)(p % sw); p /= sw;
        const int sy = (int)(p % sh);
        const int img = (int)(p / sh);
        const int x0 = R * sx - R / 2, y0 = R * sy - R / 2;
        float wx[NT], wy[NT];
        int xi[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int x = x0 + t, y = y0 + t;
            wx[t] = (x >= 0 && x < tw) ? tap_weight(ax, x, sw, sx) : 0.0f;
            wy[t] = (y >= 0 && y < th) ? tap_weight(ay, y, sh, sy) : 0.0f;
            xi[t] = min(max(x, 0), tw - 1);
        }
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ty = 0; ty < NT; ++ty) {
            const int y = min(max(y0 + ty, 0), th - 1);
            const unsigned short* row = ddst + (((long long)img * th + y) * tw) * c + g * 8;
            uint4 v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = *(const uint4*)(row + (long long)xi[t] * c);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float wgt = __fmul_rn(wx[t], wy[ty]);
                if (wx[t] != 0.0f && wy[ty] != 0.0f) {   // taps with a zero weight are skipped, not added as 0 * v
                    acc[0] += bf16_lo(v[t].x) * wgt; acc[1] += bf16_hi(v[t].x) * wgt;
                    acc[2] += bf16_lo(v[t].y) * wgt; acc[3] += bf16_hi(v[t].y) * wgt;
                    acc[4] += bf16_lo(v[t].z) * wgt; acc[5] += bf16_hi(v[t].z) * wgt;
                    acc[6] += bf16_lo(v[t].w) * wgt; acc[7] += bf16_hi(v[t].w) * wgt;
                }
            }
        }
        const long long o = (((long long)img * sh + sy) * sw + sx) * c + g * 8;
        if (relu_out != nullptr) {
            const uint4 m = *(const uint4*)(relu_out + o);
            if (!(bf16_lo(m.x) > 0.0f)) acc[0] = 0.0f;
            if (!(bf16_hi(m.x) > 0.0f)) acc[1] = 0.0f;
            if (!(bf16_lo(m.y) > 0.0f)) acc[2] = 0.0f;
            if (!(bf16_hi(m.y) > 0.0f)) acc[3] = 0.0f;
            if (!(bf16_lo(m.z) > 0.0f)) acc[4] = 0.0f;
            if (!(bf16_hi(m.z) > 0.0f)) acc[5] = 0.0f;
            if (!(bf16_lo(m.w) > 0.0f)) acc[6] = 0.0f;
            if (!(bf16_hi(m.w) > 0.0f)) acc[7] = 0.0f;
        }
        *(uint4*)(dsrc + o) = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]),
                                         pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
    }
}


// element-wise training helpers: 8 bf16 per thread
__device__ __forceinline__ unsigned mask2(unsigned dy, unsigned y)
{
    const unsigned lo = (bf16_lo(y) > 0.0f) ? (dy & 0xffffu) : 0u;
    const unsigned hi = (bf16_hi(y) > 0.0f) ? (dy & 0xffff0000u) : 0u;
    return lo | hi;
}
__global__ void relu_mask_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ y, uint4* __restrict__ dst,
                                 long long nvec)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        const uint4 a = dy[i], b = y[i];
        dst[i] = make_uint4(mask2(a.x, b.x), mask2(a.y, b.y), mask2(a.z, b.z), mask2(a.w, b.w));
    }
}
__device__ __forceinline__ unsigned add2(unsigned a, unsigned b)
{
    return pack_bf16x2(bf16_lo(a) + bf16_lo(b), bf16_hi(a) + bf16_hi(b));
}
// Sign bits of a 64-channel bf16 map in the layout the convolution's sign-bit forms read (srhip_conv3x3_args.aux): word
// (x / 32) * 64 + (x % 8) * 8 + c / 8 of a row, bit ((x % 32) / 8) * 8 + c % 8.  One thread = one word: four 16-byte
// loads (pixels x0 + 8k, 8 channels), 8 lanes per pixel, so a wave reads 1 KiB of contiguous pixels per load.
__global__ void signbits_from_map_kernel(const unsigned short* __restrict__ map, unsigned* __restrict__ bits, int w, int groups,
                                         long long rows)
{
    // grid: (row groups of 4 words-rows per block ...): thread index -> (row, 32-pixel group, lane)
    const long long total = rows * groups * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long long rg = i >> 6;
        const long long row = rg / groups;
        const int g = (int)(rg - row * groups);
        const int xb = g * 32 + (lane >> 3);
        const unsigned short* rp = map + (row * w) * 64 + (lane & 7) * 8;
        unsigned word = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = xb + 8 * k;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (x < w) v = *(const uint4*)(rp + (long long)x * 64);
            const unsigned q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const short h = (short)((e & 1) ? (q[e >> 1] >> 16) : (q[e >> 1] & 0xffffu));   // bf16 > 0 <=> its bits as int16 > 0
                word |= (h > 0 ? 1u : 0u) << (8 * k + e);
            }
        }
        bits[i] = word;
    }
}

__global__ void add_bf16_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ dst,
                                long long nvec)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        const uint4 u = a[i], v = b[i];
        dst[i] = make_uint4(add2(u.x, v.x), add2(u.y, v.y), add2(u.z, v.z), add2(u.w, v.w));
    }
}

// dst = bf16(a + scale * y) with a trained scalar read from device memory (the G-buffer encoder's
// fmap + enhance_param * enhance, core/models/model_final.py:256-258, in the training graph): the product and the sum are
// formed in fp32 and rounded once, as the inference epilogue forms them.
__global__ void scale_add_bf16_kernel(const uint4* __restrict__ a, const uint4* __restrict__ y, const float* __restrict__ scale,
                                      uint4* __restrict__ dst, long long nvec)
{
    const float p = *scale;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        const uint4 u = a[i], v = y[i];
        // (one explicit fma per element, the instruction of the inference epilogues that form the same sum)
        dst[i] = make_uint4(pack_bf16x2(__fmaf_rn(p, bf16_lo(v.x), bf16_lo(u.x)), __fmaf_rn(p, bf16_hi(v.x), bf16_hi(u.x))),
                            pack_bf16x2(__fmaf_rn(p, bf16_lo(v.y), bf16_lo(u.y)), __fmaf_rn(p, bf16_hi(v.y), bf16_hi(u.y))),
                            pack_bf16x2(__fmaf_rn(p, bf16_lo(v.z), bf16_lo(u.z)), __fmaf_rn(p, bf16_hi(v.z), bf16_hi(u.z))),
                            pack_bf16x2(__fmaf_rn(p, bf16_lo(v.w), bf16_lo(u.w)), __fmaf_rn(p, bf16_hi(v.w), bf16_hi(u.w))));
    }
}
// Its backward in one pass over dout and y: dy = bf16(scale * dout), and the work-group's share of
// d scale = sum(dout * y) (fp32) in partials[blockIdx.x]; the gradient of a is dout itself.  The grid is fixed by the
// caller (one partial per work-group), so the sum has one order of additions.
__global__ __launch_bounds__(kThreads) void scale_add_bwd_bf16_kernel(const uint4* __restrict__ dout, const uint4* __restrict__ y,
                                                                     const float* __restrict__ scale, uint4* __restrict__ dy,
                                                                     float* __restrict__ partials, long long nvec)
{
    const float p = *scale;
    float acc = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        const uint4 g = dout[i], v = y[i];
        float gf[8], yf[8];
        gf[0] = bf16_lo(g.x); gf[1] = bf16_hi(g.x); gf[2] = bf16_lo(g.y); gf[3] = bf16_hi(g.y);
        gf[4] = bf16_lo(g.z); gf[5] = bf16_hi(g.z); gf[6] = bf16_lo(g.w); gf[7] = bf16_hi(g.w);
        yf[0] = bf16_lo(v.x); yf[1] = bf16_hi(v.x); yf[2] = bf16_lo(v.y); yf[3] = bf16_hi(v.y);
        yf[4] = bf16_lo(v.z); yf[5] = bf16_hi(v.z); yf[6] = bf16_lo(v.w); yf[7] = bf16_hi(v.w);
        float part = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) part += gf[e] * yf[e];
        acc += part;
        dy[i] = make_uint4(pack_bf16x2(p * gf[0], p * gf[1]), pack_bf16x2(p * gf[2], p * gf[3]), pack_bf16x2(p * gf[4], p * gf[5]),
                           pack_bf16x2(p * gf[6], p * gf[7]));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    __shared__ float wsum[kThreads / 64];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < kThreads / 64; ++k) t += wsum[k];
        partials[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------
// ESA helpers (core/arch/RepRFN.py:295-308) on NHWC bf16: one thread = one output pixel x 8 channels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void unpack8(const uint4 v, float* f)
{
    f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
    f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f)
{
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}
__global__ void max_pool_nhwc_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int n, int c,
                                     int h, int w, int ho, int wo, int k, int s)
{
    const int cg = c / 8;
    const long long total = (long long)n * ho * wo * cg;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long long p = i / cg;
        const int ox = (int)(p % wo); p /= wo;
        const int oy = (int)(p % ho);
        const int img = (int)(p / ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx) {
                float v[8];
                unpack8(*(const uint4*)(src + (((long long)img * h + oy * s + ky) * w + ox * s + kx) * c + g * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (v[e] > m[e] || v[e] != v[e]) m[e] = v[e];
            }
        *(uint4*)(dst + (((long long)img * ho + oy) * wo + ox) * c + g * 8) = pack8(m);
    }
}
// The same pooling for windows of K <= 8 rows with eight lanes per output: lane (i & 7) takes window row ky = i & 7 (its K
// loads are independent, so they are all in flight at once; the kernel above walks the k * k taps one after the other and
// a launch with few outputs -- the ESA branch pools 269 x 479 down to 88 x 158 -- waits k * k memory latencies), then
// three exchange steps over the eight lanes.  The winner is the same value: max is associative, and any NaN wins.
template <int K>
__global__ void max_pool_rows_nhwc_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int n, int c,
                                          int h, int w, int ho, int wo, int s)
{
    static_assert(K <= 8, "one lane per window row");
    const int cg = c / 8;
    const long long total = (long long)n * ho * wo * cg * 8;
    const long long stride = (long long)gridDim.x * blockDim.x;   // a multiple of 8: the eight lanes of an output stay together
    for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 - (threadIdx.x & 7) < total; i0 += stride) {
        const bool live = i0 < total;
        const long long i = live ? i0 : total - 1;
        const int ky = (int)(i & 7);
        long long p = i >> 3;
        const int g = (int)(p % cg); p /= cg;
        const int ox = (int)(p % wo); p /= wo;
        const int oy = (int)(p % ho);
        const int img = (int)(p / ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
        if (ky < K) {
            uint4 t[K];
            const unsigned short* rp = src + (((long long)img * h + oy * s + ky) * w + ox * s) * c + g * 8;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) t[kx] = *(const uint4*)(rp + (long long)kx * c);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                float v[8];
                unpack8(t[kx], v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (v[e] > m[e] || v[e] != v[e]) m[e] = v[e];
            }
        }
#pragma unroll
        for (int d = 1; d < 8; d <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float o = __shfl_xor(m[e], d, 8);
                if (o > m[e] || o != o) m[e] = o;
            }
        if (live && ky == 0) *(uint4*)(dst + (((long long)img * ho + oy) * wo + ox) * c + g * 8) = pack8(m);
    }
}
// F.interpolate / nn.Upsample(size=(ho,wo), mode="bilinear", align_corners=align) (+ optional addend of the output shape)
__global__ void resize_bilinear_nhwc_kernel(const unsigned short* __restrict__ src, const unsigned short* __restrict__ addend,
                                            unsigned short* __restrict__ dst, int n, int c, int h, int w, int ho, int wo, int align)
{
    const int cg = c / 8;
    const long long total = (long long)n * ho * wo * cg;
    const float sy = align ? (ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.0f) : (float)h / (float)ho;
    const float sx = align ? (wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.0f) : (float)w / (float)wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long long p = i / cg;
        const int ox = (int)(p % wo); p /= wo;
        const int oy = (int)(p % ho);
        const int img = (int)(p / ho);
        float fy = align ? __fmul_rn(sy, (float)oy) : __fsub_rn(__fmul_rn(sy, (float)oy + 0.5f), 0.5f);
        float fx = align ? __fmul_rn(sx, (float)ox) : __fsub_rn(__fmul_rn(sx, (float)ox + 0.5f), 0.5f);
        if (fy < 0.0f) fy = 0.0f;
        if (fx < 0.0f) fx = 0.0f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = fy - (float)y0, hy = 1.0f - ly, lx = fx - (float)x0, hx = 1.0f - lx;
        const unsigned short* ip = src + (long long)img * h * w * c + g * 8;
        float a[8], b[8], cc[8], d[8], o[8];
        unpack8(*(const uint4*)(ip + ((long long)y0 * w + x0) * c), a);
        unpack8(*(const uint4*)(ip + ((long long)y0 * w + x1) * c), b);
        unpack8(*(const uint4*)(ip + ((long long)y1 * w + x0) * c), cc);
        unpack8(*(const uint4*)(ip + ((long long)y1 * w + x1) * c), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = hy * (hx * a[e] + lx * b[e]) + ly * (hx * cc[e] + lx * d[e]);
        const long long off = (((long long)img * ho + oy) * wo + ox) * c + g * 8;
        if (addend != nullptr) {
            float t[8];
            unpack8(*(const uint4*)(addend + off), t);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += t[e];
        }
        *(uint4*)(dst + off) = pack8(o);
    }
}
// dst = x * sigmoid(gate)
__global__ void sigmoid_gate_kernel(const uint4* __restrict__ x, const uint4* __restrict__ gate, uint4* __restrict__ dst,
                                    long long nvec)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        float a[8], b[8];
        unpack8(x[i], a);
        unpack8(gate[i], b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = a[e] * (1.0f / (1.0f + expf(-b[e])));
        dst[i] = pack8(a);
    }
}
// The tail of RepRFN's ESA block in one pass (core/arch/RepRFN.py:303-308):
//   dst = x * sigmoid(conv4(resize_bilinear(c3 -> h x w, align_corners=False) + conv_f(c1)))
// conv_f: 1x1 16 -> 16 on the full-resolution map c1, c3: the 16-channel map of the pooled branch (a few thousand pixels:
// its taps stay in cache), conv4: 1x1 16 -> CF.  A wave takes 32 consecutive pixels per step; both small layers are one
// matrix instruction per 32 output channels with K = 16:
//   * lane (pixel l & 31, half hl = l >> 5) loads channels 8 hl .. 8 hl + 7 of c1: the wave's load is 1 KiB contiguous and
//     already the instruction's pixel operand;
//   * conv_f's result arrives as channels {4 hl .. 4 hl + 3, 8 + 4 hl .. 8 + 4 hl + 3} of the lane's pixel.  The sum with the
//     resized branch is formed for exactly those channels and fed to conv4 AS IS: the contraction index of a matrix
//     instruction may run over the channels in any order as long as both operands agree, so conv4's weight fragment is
//     gathered in that order and no lane exchange is needed;
//   * conv4's result: 4 consecutive channels per register quad -> gate, multiply, 8-byte stores.
// The maps the layer-by-layer composition stores between its launches (conv_f's output, the sum, conv4's output) are
// rounded to bf16 here at the same points, in registers; the weights are rounded to bf16 like every matrix-core layer's.
template <int CF, int CS>
__global__ __launch_bounds__(kThreads) void esa_tail_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ c1,
                                                           const unsigned short* __restrict__ c3, const float* __restrict__ wf,
                                                           const float* __restrict__ bf, const float* __restrict__ w4,
                                                           const float* __restrict__ b4, unsigned short* __restrict__ dst, int n, int h,
                                                           int w, int h3, int w3)
{
    // CS: channels per pixel of x and dst in memory (CF or 64)
    constexpr int F = 16, NT = (CF + 31) / 32, NPJ = CS * 4 / 64;   // NPJ: 16-byte pieces of x per lane and step
    static_assert(CS % 16 == 0 && CS >= CF && CS <= 64, "32 pixels of x are a whole number of 1-KiB wave loads");
    const int lane = threadIdx.x & 63, pl = lane & 31, hl = lane >> 5;
    const long long total = (long long)n * h * w;
    const float sy = (float)h3 / (float)h, sx = (float)w3 / (float)w;
    // channel of contraction index j in conv4's operands: the order conv_f's accumulators come in
    auto chan = [&](const int j) { return 8 * (j >> 2) + 4 * hl + (j & 3); };
    bf16x8 wf_frag, w4_frag[NT];
    float bfv[8], b4v[NT][4][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        wf_frag[j] = (short)f32_to_bf16_bits(pl < F ? wf[pl * F + 8 * hl + j] : 0.0f);
        bfv[j] = bf[chan(j)];
    }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        const int o = 32 * ct + pl;
#pragma unroll
        for (int j = 0; j < 8; ++j) w4_frag[ct][j] = (short)f32_to_bf16_bits(o < CF ? w4[o * F + chan(j)] : 0.0f);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 32 * ct + 8 * g + 4 * hl + e;
                b4v[ct][g][e] = c < CF ? b4[c] : 0.0f;
            }
    }
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    __shared__ __attribute__((aligned(16))) char stage[kThreads / 64][32 * 64 * 2];   // per wave: 32 pixels x up to 64 channels
    char* sw = stage[threadIdx.x >> 6];
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long base = wave0 * 32; base < total; base += nwaves * 32) {
        const long long i = base + pl;
        const bool live = i < total;
        const long long ic = live ? i : total - 1;   // lanes past the end repeat the last pixel and store nothing
        // (the host checks that the pixel count fits 31 bits: 32-bit divisions, a fraction of the 64-bit ones' instructions)
        const unsigned pu = (unsigned)ic, rowi = pu / (unsigned)w;
        const int ox = (int)(pu - rowi * (unsigned)w);
        const int img = (int)(rowi / (unsigned)h);
        const int oy = (int)(rowi - (unsigned)img * (unsigned)h);
        // every load of the step is issued before anything waits: x (the wave's 32 pixels are one contiguous piece of
        // memory, fetched 16 bytes per lane, whole cache lines per instruction), c1, the four taps of c3
        constexpr long long cbytes = CS * 2;
        const long long chunk0 = base * cbytes, endb = total * cbytes;
        uint4 xv4[NPJ];
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
            const long long off = chunk0 + (long long)(j * 64 + lane) * 16;
            xv4[j] = *(const uint4*)((const char*)x + (off < endb ? off : endb - 16));
        }
        const uint4 c1v = *(const uint4*)(c1 + ic * F + 8 * hl);
        bf16x8 c1_frag;
        __builtin_memcpy(&c1_frag, &c1v, 16);
        const f32x16 a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_frag, c1_frag, zero, 0, 0, 0);
        // the resized pooled branch at this pixel (same arithmetic as resize_bilinear_nhwc_kernel, align_corners = 0)
        float fy = __fsub_rn(__fmul_rn(sy, (float)oy + 0.5f), 0.5f);
        float fx = __fsub_rn(__fmul_rn(sx, (float)ox + 0.5f), 0.5f);
        if (fy < 0.0f) fy = 0.0f;
        if (fx < 0.0f) fx = 0.0f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h3 - 1 ? 1 : 0), x1 = x0 + (x0 < w3 - 1 ? 1 : 0);
        const float ly = fy - (float)y0, hy = 1.0f - ly, lx = fx - (float)x0, hx = 1.0f - lx;
        const unsigned short* ip = c3 + (long long)img * h3 * w3 * F + 4 * hl;
        float sv[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {   // channels 8 q + 4 hl .. + 3
            const uint2 t00 = *(const uint2*)(ip + ((long long)y0 * w3 + x0) * F + 8 * q);
            const uint2 t01 = *(const uint2*)(ip + ((long long)y0 * w3 + x1) * F + 8 * q);
            const uint2 t10 = *(const uint2*)(ip + ((long long)y1 * w3 + x0) * F + 8 * q);
            const uint2 t11 = *(const uint2*)(ip + ((long long)y1 * w3 + x1) * F + 8 * q);
            const float a[4] = {bf16_lo(t00.x), bf16_hi(t00.x), bf16_lo(t00.y), bf16_hi(t00.y)};
            const float b[4] = {bf16_lo(t01.x), bf16_hi(t01.x), bf16_lo(t01.y), bf16_hi(t01.y)};
            const float c[4] = {bf16_lo(t10.x), bf16_hi(t10.x), bf16_lo(t10.y), bf16_hi(t10.y)};
            const float d[4] = {bf16_lo(t11.x), bf16_hi(t11.x), bf16_lo(t11.y), bf16_hi(t11.y)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float cf = bf16_bits_to_f32(f32_to_bf16_bits(a1[4 * q + e] + bfv[4 * q + e]));
                sv[4 * q + e] = (hy * (hx * a[e] + lx * b[e]) + ly * (hx * c[e] + lx * d[e])) + cf;
            }
        }
        bf16x8 s_frag;
#pragma unroll
        for (int j = 0; j < 8; ++j) s_frag[j] = (short)f32_to_bf16_bits(sv[j]);
        // x goes to LDS as it came, is gated in place in the accumulators' (pixel, 4 channels) order, and is copied out the
        // way it came in.  Only the wave itself touches its piece, so its own in-order LDS queue is all the ordering it needs.
#pragma unroll
        for (int j = 0; j < NPJ; ++j) *(uint4*)(sw + (j * 64 + lane) * 16) = xv4[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        char* xl = sw + pl * (int)cbytes + 8 * hl;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x16 a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w4_frag[ct], s_frag, zero, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (32 * ct + 8 * g >= CF) continue;
                const uint2 xv = *(const uint2*)(xl + (32 * ct + 8 * g) * 2);
                const float xf[4] = {bf16_lo(xv.x), bf16_hi(xv.x), bf16_lo(xv.y), bf16_hi(xv.y)};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gate = bf16_bits_to_f32(f32_to_bf16_bits(a2[4 * g + e] + b4v[ct][g][e]));
                    o[e] = xf[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-gate));
                }
                *(uint2*)(xl + (32 * ct + 8 * g) * 2) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
            const long long off = chunk0 + (long long)(j * 64 + lane) * 16;
            if (off < endb) *(uint4*)((char*)dst + off) = *(const uint4*)(sw + (j * 64 + lane) * 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
// dst = cur + sigmoid(f_self * g_cross) * fmap + sigmoid(g_self * f_cross) * grad with one gate value per pixel
// (core/models/model_final.py:422): NHWC bf16 maps of c channels, gates read at pixel * gate_stride, fp32 arithmetic.
__global__ void attention_fuse_kernel(const uint4* __restrict__ cur, const uint4* __restrict__ fmap, const uint4* __restrict__ grad,
                                      const unsigned short* __restrict__ f_self, const unsigned short* __restrict__ g_cross,
                                      const unsigned short* __restrict__ g_self, const unsigned short* __restrict__ f_cross,
                                      int gate_stride, uint4* __restrict__ dst, long long nvec, int cg)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        const long long px = i / cg;
        const float wf = 1.0f / (1.0f + expf(-(bf16_bits_to_f32(f_self[px * gate_stride]) * bf16_bits_to_f32(g_cross[px * gate_stride]))));
        const float wg = 1.0f / (1.0f + expf(-(bf16_bits_to_f32(g_self[px * gate_stride]) * bf16_bits_to_f32(f_cross[px * gate_stride]))));
        float a[8], b[8], c[8];
        unpack8(cur[i], a);
        unpack8(fmap[i], b);
        unpack8(grad[i], c);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = __fadd_rn(__fadd_rn(a[e], __fmul_rn(wf, b[e])), __fmul_rn(wg, c[e]));
        dst[i] = pack8(a);
    }
}

// Backward of the gated sum above (autograd of core/models/model_final.py:422 under loss.backward(), core/trainer.py:160),
// d = gradient of dst:  dcur = d (the caller reuses d),  dfmap = wf * d,  dgrad = wg * d,
//   d f_self = S_f * wf (1 - wf) * g_cross,  d g_cross = S_f * wf (1 - wf) * f_self,  S_f = sum over channels of d * fmap,
//   d g_self = S_g * wg (1 - wg) * f_cross,  d f_cross = S_g * wg (1 - wg) * g_self,  S_g = sum over channels of d * grad.
// LPP = c / 8 lanes per pixel (1, 2, 4 or 8), each 8 channels; the channel sums cross the lanes of a pixel by shuffles.
// A gate gradient is written as a whole row of dgate_stride channels: channel 0 the value, the others zero (the gates
// are channel 0 of 16-channel padded maps, and the layer that produced them takes the gradient of the whole map).
template <int LPP>
__global__ void attention_fuse_bwd_kernel(const uint4* __restrict__ d, const uint4* __restrict__ fmap, const uint4* __restrict__ grad,
                                          const unsigned short* __restrict__ f_self, const unsigned short* __restrict__ g_cross,
                                          const unsigned short* __restrict__ g_self, const unsigned short* __restrict__ f_cross,
                                          int gate_stride, uint4* __restrict__ dfmap, uint4* __restrict__ dgrad,
                                          uint4* __restrict__ d_f_self, uint4* __restrict__ d_g_cross, uint4* __restrict__ d_g_self,
                                          uint4* __restrict__ d_f_cross, int dgate_vecs, long long nvec)
{
    // the grid covers nvec rounded up to whole waves: every lane of a wave takes part in the shuffles
    const long long nvec_up = (nvec + 63) / 64 * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec_up; i += (long long)gridDim.x * blockDim.x) {
        const bool live = i < nvec;
        const long long iv = live ? i : nvec - 1;
        const long long px = iv / LPP;
        const float fs = bf16_bits_to_f32(f_self[px * gate_stride]), gc = bf16_bits_to_f32(g_cross[px * gate_stride]);
        const float gs = bf16_bits_to_f32(g_self[px * gate_stride]), fc = bf16_bits_to_f32(f_cross[px * gate_stride]);
        const float wf = 1.0f / (1.0f + expf(-(fs * gc)));
        const float wg = 1.0f / (1.0f + expf(-(gs * fc)));
        float dv[8], fv[8], gv[8], o1[8], o2[8];
        unpack8(d[iv], dv);
        unpack8(fmap[iv], fv);
        unpack8(grad[iv], gv);
        float sf = 0.0f, sg = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o1[e] = __fmul_rn(wf, dv[e]);
            o2[e] = __fmul_rn(wg, dv[e]);
            sf = __builtin_fmaf(dv[e], fv[e], sf);
            sg = __builtin_fmaf(dv[e], gv[e], sg);
        }
        if (live) {
            dfmap[i] = pack8(o1);
            dgrad[i] = pack8(o2);
        }
#pragma unroll
        for (int m = 1; m < LPP; m <<= 1) {   // the lanes of a pixel are LPP consecutive lanes of one wave
            sf += __shfl_xor(sf, m);
            sg += __shfl_xor(sg, m);
        }
        if (live && (iv % LPP) == 0) {
            const float kf = sf * wf * (1.0f - wf), kg = sg * wg * (1.0f - wg);
            const unsigned vals[4] = {f32_to_bf16_bits(kf * gc), f32_to_bf16_bits(kf * fs), f32_to_bf16_bits(kg * fc), f32_to_bf16_bits(kg * gs)};
            uint4* const rows[4] = {d_f_self, d_g_cross, d_g_self, d_f_cross};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                rows[k][px * dgate_vecs] = make_uint4(vals[k], 0u, 0u, 0u);
                for (int v = 1; v < dgate_vecs; ++v) rows[k][px * dgate_vecs + v] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
}

// ---- backward kernels of the ESA block and of LeakyReLU (autograd of core/arch/RepRFN.py:295-308,17-33) ----
// dst = dy * (y > 0 ? 1 : slope): backward of (Leaky)ReLU from its OUTPUT (slope > 0 keeps the sign)
__global__ void lrelu_mask_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ y, uint4* __restrict__ dst,
                                  long long nvec, float slope)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        float a[8], b[8];
        unpack8(dy[i], a);
        unpack8(y[i], b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = b[e] > 0.0f ? a[e] : a[e] * slope;
        dst[i] = pack8(a);
    }
}
// y = x * sigmoid(g):  dx = dy * sig,  dg = dy * x * sig * (1 - sig)
__global__ void sigmoid_gate_bwd_kernel(const uint4* __restrict__ x, const uint4* __restrict__ gate, const uint4* __restrict__ dy,
                                        uint4* __restrict__ dx, uint4* __restrict__ dgate, long long nvec)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        float a[8], b[8], d[8], o1[8], o2[8];
        unpack8(x[i], a);
        unpack8(gate[i], b);
        unpack8(dy[i], d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sg = 1.0f / (1.0f + expf(-b[e]));
            o1[e] = d[e] * sg;
            o2[e] = d[e] * a[e] * sg * (1.0f - sg);
        }
        dx[i] = pack8(o1);
        dgate[i] = pack8(o2);
    }
}
// max_pool2d(k, s) backward, gather form: an input pixel receives dy of every window in which it is the FIRST
// maximum in row-major order (the forward kernel's and ATen's tie rule).  Deterministic, no atomics.
__global__ void max_pool_bwd_nhwc_kernel(const unsigned short* __restrict__ src, const unsigned short* __restrict__ dy,
                                         unsigned short* __restrict__ dx, int n, int c, int h, int w, int ho, int wo, int k, int s)
{
    const int cg = c / 8;
    const long long total = (long long)n * h * w * cg;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long long p = i / cg;
        const int x = (int)(p % w); p /= w;
        const int y = (int)(p % h);
        const int img = (int)(p / h);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
        const int oy_lo = y - k + 1 > 0 ? (y - k + 1 + s - 1) / s : 0, oy_hi = y / s < ho - 1 ? y / s : ho - 1;
        const int ox_lo = x - k + 1 > 0 ? (x - k + 1 + s - 1) / s : 0, ox_hi = x / s < wo - 1 ? x / s : wo - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                float m[8];
                int arg[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; arg[e] = -1; }
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        float v[8];
                        unpack8(*(const uint4*)(src + (((long long)img * h + oy * s + ky) * w + ox * s + kx) * c + g * 8), v);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (v[e] > m[e] || v[e] != v[e]) { m[e] = v[e]; arg[e] = ky * k + kx; }
                    }
                const int mine = (y - oy * s) * k + (x - ox * s);
                float d[8];
                unpack8(*(const uint4*)(dy + (((long long)img * ho + oy) * wo + ox) * c + g * 8), d);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (arg[e] == mine) acc[e] += d[e];
            }
        *(uint4*)(dx + (((long long)img * h + y) * w + x) * c + g * 8) = pack8(acc);
    }
}
// max_pool2d(k, s) that also records, per output element, which window position (ky * k + kx) supplied the maximum
// (the first one in scan order; a NaN wins like in max_pool_nhwc_kernel).  One byte per element: k * k <= 255.
__global__ void max_pool_arg_nhwc_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                         unsigned char* __restrict__ arg, int n, int c, int h, int w, int ho, int wo, int k, int s)
{
    const int cg = c / 8;
    const long long total = (long long)n * ho * wo * cg;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long long p = i / cg;
        const int ox = (int)(p % wo); p /= wo;
        const int oy = (int)(p % ho);
        const int img = (int)(p / ho);
        float m[8];
        unsigned int a[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; a[e] = 0u; }
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx) {
                float v[8];
                unpack8(*(const uint4*)(src + (((long long)img * h + oy * s + ky) * w + ox * s + kx) * c + g * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (v[e] > m[e] || v[e] != v[e]) { m[e] = v[e]; a[e] = (unsigned int)(ky * k + kx); }
            }
        const long long o = (((long long)img * ho + oy) * wo + ox) * c + g * 8;
        *(uint4*)(dst + o) = pack8(m);
        *(uint2*)(arg + o) = make_uint2(a[0] | (a[1] << 8) | (a[2] << 16) | (a[3] << 24), a[4] | (a[5] << 8) | (a[6] << 16) | (a[7] << 24));
    }
}
// Backward of the above from the recorded positions, gather form (deterministic, no atomics): an input pixel sums dy
// of the windows that cover it and named it.
__global__ void max_pool_bwd_arg_nhwc_kernel(const unsigned char* __restrict__ arg, const unsigned short* __restrict__ dy,
                                             unsigned short* __restrict__ dx, int n, int c, int h, int w, int ho, int wo, int k, int s)
{
    const int cg = c / 8;
    const long long total = (long long)n * h * w * cg;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long long p = i / cg;
        const int x = (int)(p % w); p /= w;
        const int y = (int)(p % h);
        const int img = (int)(p / h);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
        const int oy_lo = y - k + 1 > 0 ? (y - k + 1 + s - 1) / s : 0, oy_hi = y / s < ho - 1 ? y / s : ho - 1;
        const int ox_lo = x - k + 1 > 0 ? (x - k + 1 + s - 1) / s : 0, ox_hi = x / s < wo - 1 ? x / s : wo - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const long long o = (((long long)img * ho + oy) * wo + ox) * c + g * 8;
                const uint2 a = *(const uint2*)(arg + o);
                const unsigned int mine = (unsigned int)((y - oy * s) * k + (x - ox * s));
                float d[8];
                unpack8(*(const uint4*)(dy + o), d);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned int ae = ((e < 4 ? a.x : a.y) >> (8 * (e & 3))) & 0xffu;
                    if (ae == mine) acc[e] += d[e];
                }
            }
        *(uint4*)(dx + (((long long)img * h + y) * w + x) * c + g * 8) = pack8(acc);
    }
}
// bilinear resize backward, gather form: source pixel (y, x) sums w_y * w_x * dy over the output pixels whose
// four taps include it.  The tap positions and weights replay the forward kernel's arithmetic.
__device__ __forceinline__ void resize_taps(int o, int n_in, float scale, int align, int& i0, int& i1, float& l)
{
    float f = align ? __fmul_rn(scale, (float)o) : __fsub_rn(__fmul_rn(scale, (float)o + 0.5f), 0.5f);
    if (f < 0.0f) f = 0.0f;
    i0 = (int)f;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l = f - (float)i0;
}
__global__ void resize_bilinear_bwd_nhwc_kernel(const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx, int n, int c,
                                                int h, int w, int ho, int wo, int align)
{
    const int cg = c / 8;
    const long long total = (long long)n * h * w * cg;
    const float sy = align ? (ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.0f) : (float)h / (float)ho;
    const float sx = align ? (wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.0f) : (float)w / (float)wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long long p = i / cg;
        const int x = (int)(p % w); p /= w;
        const int y = (int)(p % h);
        const int img = (int)(p / h);
        // conservative range of output rows / columns whose taps can touch (y, x)
        int oy_lo = sy > 0.0f ? (int)floorf(((float)y - 1.0f + 0.5f) / sy - 0.5f) - 1 : 0;
        int oy_hi = sy > 0.0f ? (int)ceilf(((float)y + 1.0f + 0.5f) / sy - 0.5f) + 1 : ho - 1;
        int ox_lo = sx > 0.0f ? (int)floorf(((float)x - 1.0f + 0.5f) / sx - 0.5f) - 1 : 0;
        int ox_hi = sx > 0.0f ? (int)ceilf(((float)x + 1.0f + 0.5f) / sx - 0.5f) + 1 : wo - 1;
        if (y == 0) oy_lo = 0;
        if (x == 0) ox_lo = 0;
        oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
        oy_hi = oy_hi > ho - 1 ? ho - 1 : oy_hi; ox_hi = ox_hi > wo - 1 ? wo - 1 : ox_hi;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1; float ly;
            resize_taps(oy, h, sy, align, y0, y1, ly);
            float wy = 0.0f;
            if (y0 == y) wy += 1.0f - ly;
            if (y1 == y) wy += ly;
            if (y0 == y1 && y0 == y) wy = 1.0f;  // clamped at the last row: both taps are this row, weights sum to 1
            if (wy == 0.0f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1; float lx;
                resize_taps(ox, w, sx, align, x0, x1, lx);
                float wx = 0.0f;
                if (x0 == x) wx += 1.0f - lx;
                if (x1 == x) wx += lx;
                if (x0 == x1 && x0 == x) wx = 1.0f;
                if (wx == 0.0f) continue;
                float d[8];
                unpack8(*(const uint4*)(dy + (((long long)img * ho + oy) * wo + ox) * c + g * 8), d);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += wy * wx * d[e];
            }
        }
        *(uint4*)(dx + (((long long)img * h + y) * w + x) * c + g * 8) = pack8(acc);
    }
}

// GradientMap (core/models/model_final.py:160-187) on fp32 planes: sqrt(gv^2 + gh^2 + eps) with the fixed
// stencils kernel_v = rows (-1, 0, +1), kernel_h = columns (-1, 0, +1), zero padding.
__global__ void gradient_map_kernel(const float* __restrict__ src, float* __restrict__ dst, long long planes, int h, int w,
                                    float eps)
{
    const long long total = planes * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        const int y = (int)((i / w) % h);
        const float* pl = src + (i / ((long long)h * w)) * h * w;
        float gv = 0.0f, gh = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + ky - 1, xx = x + kx - 1;
                if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                const float v = pl[(long long)yy * w + xx];
                gv += (float)(ky - 1) * v;
                gh += (float)(kx - 1) * v;
            }
        dst[i] = sqrtf(gv * gv + gh * gh + eps);
    }
}

// core/tools/math.py:80-100: each pixel takes the motion vector of the pixel with the smallest depth in its k x k
// neighbourhood (zero outside the frame, as F.unfold pads; first minimum in window order; position clamped to the frame).
__global__ void dilate_velocity_kernel(const float2* __restrict__ motion, const float* __restrict__ depth, float2* __restrict__ dst,
                                       int h, int w, int k)
{
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (x >= w) return;
    const int y = (int)blockIdx.y, r = k / 2;
    const float* dp = depth + (long long)blockIdx.z * h * w;
    float best = 0.0f;
    int pos = -1;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
            const int yy = y + ky - r, xx = x + kx - r;
            const float v = (yy < 0 || yy >= h || xx < 0 || xx >= w) ? 0.0f : dp[yy * w + xx];
            if (pos < 0 || v < best || (v != v && best == best)) { best = v; pos = ky * k + kx; }
        }
    const int sx = min(max(x + pos % k - r, 0), w - 1), sy = min(max(y + pos / k - r, 0), h - 1);
    const long long base = (long long)blockIdx.z * h * w;
    dst[base + y * w + x] = motion[base + sy * w + sx];
}

// The same map on a (column block, row, plane) grid: 32-bit index arithmetic, no division per element.
__global__ void gradient_map_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, float eps)
{
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (x >= w) return;
    const int y = (int)blockIdx.y;
    const float* pl = src + (long long)blockIdx.z * h * w;
    float gv = 0.0f, gh = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
            const float v = pl[yy * w + xx];
            gv += (float)(ky - 1) * v;
            gh += (float)(kx - 1) * v;
        }
    dst[(long long)blockIdx.z * h * w + y * w + x] = sqrtf(gv * gv + gh * gh + eps);
}

// Four adjacent pixels per lane (w % 4 == 0): three 16-byte row loads plus the two neighbours per row instead of 36 scalar
// loads.  Taps outside the plane enter as zeros, in the same order of additions as above.
__global__ void gradient_map_rows4_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, float eps)
{
    const int x0 = (int)(blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (x0 >= w) return;
    const int y = (int)blockIdx.y;
    const float* pl = src + (long long)blockIdx.z * h * w;
    float v[3][6];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if (yy < 0 || yy >= h) {
#pragma unroll
            for (int k = 0; k < 6; ++k) v[ky][k] = 0.0f;
        } else {
            const float* row = pl + yy * w;
            const float4 m = *(const float4*)(row + x0);
            v[ky][0] = x0 > 0 ? row[x0 - 1] : 0.0f;
            v[ky][1] = m.x; v[ky][2] = m.y; v[ky][3] = m.z; v[ky][4] = m.w;
            v[ky][5] = x0 + 4 < w ? row[x0 + 4] : 0.0f;
        }
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float gv = 0.0f, gh = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                gv += (float)(ky - 1) * v[ky][j + kx];
                gh += (float)(kx - 1) * v[ky][j + kx];
            }
        o[j] = sqrtf(gv * gv + gh * gh + eps);
    }
    *(float4*)(dst + (long long)blockIdx.z * h * w + y * w + x0) = make_float4(o[0], o[1], o[2], o[3]);
}

// GradientMap followed by PixelUnshuffle(4) and the NHWC bf16 conversion (core/models/model_final.py:406-409 feeds the
// result to a convolution): out[n][y][x][k*16 + i*4 + j] = bf16(gradient of plane planes[k] of image n at (4y+i, 4x+j)).
// The block / LDS organisation of pixel_unshuffle4_nchw_f32_to_nhwc_bf16_kernel; every item computes the sixteen
// gradients of one output pixel's 4 x 4 block of one plane from six 16-byte row loads and the two neighbours per row, in
// the order of additions of gradient_map_rows4_kernel, so the result equals the two passes in sequence bit for bit.
struct GradPlanes { int idx[24]; };
__global__ __launch_bounds__(256) void gradient_map_unshuffle4_nhwc_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                                               int n, int src_planes, GradPlanes planes, int c, int h,
                                                                               int w, float eps)
{
    extern __shared__ uint4 tile[];
    uint2* t2 = (uint2*)tile;
    const int nq = c * 4, q = nq + 2, cg = c * 2;
    const long long total = (long long)n * h * w;
    const long long ntiles = (total + kLayoutTile - 1) / kLayoutTile;
    const int H = 4 * h, W = 4 * w;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long p0 = t * kLayoutTile;
        const int npx = (int)(total - p0 < kLayoutTile ? total - p0 : kLayoutTile);
        // one item = (pixel, plane): the six input rows around the pixel's 4 x 4 block are loaded once for its four rows
        for (int it = threadIdx.x; it < kLayoutTile * c; it += blockDim.x) {
            const int px = it & (kLayoutTile - 1), k = it / kLayoutTile;
            if (px < npx) {
                const long long p = p0 + px;
                const int x = (int)(p % w);
                const long long r = p / w;
                const int y = (int)(r % h);
                const long long img = r / h;
                const float* pl = src + (img * src_planes + planes.idx[k]) * (long long)H * W;
                const int x0 = 4 * x;
                float v[6][6];
#pragma unroll
                for (int ry = 0; ry < 6; ++ry) {
                    const int yy = 4 * y + ry - 1;
                    if (yy < 0 || yy >= H) {
#pragma unroll
                        for (int e = 0; e < 6; ++e) v[ry][e] = 0.0f;
                    } else {
                        const float* row = pl + (long long)yy * W;
                        const float4 m = *(const float4*)(row + x0);
                        v[ry][0] = x0 > 0 ? row[x0 - 1] : 0.0f;
                        v[ry][1] = m.x; v[ry][2] = m.y; v[ry][3] = m.z; v[ry][4] = m.w;
                        v[ry][5] = x0 + 4 < W ? row[x0 + 4] : 0.0f;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float gv = 0.0f, gh = 0.0f;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                gv += (float)(ky - 1) * v[i + ky][j + kx];
                                gh += (float)(kx - 1) * v[i + ky][j + kx];
                            }
                        o[j] = sqrtf(gv * gv + gh * gh + eps);
                    }
                    t2[px * q + k * 4 + i] = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
                }
            }
        }
        __syncthreads();
        uint4* out = (uint4*)(dst + p0 * (16ll * c));
        for (int it = threadIdx.x; it < npx * cg; it += blockDim.x) {
            const int px = it / cg, g = it - px * cg;
            out[it] = *(const uint4*)(t2 + px * q + 2 * g);
        }
        __syncthreads();
    }
}

// PixelShuffle(4) of an NHWC bf16 map written as the NHWC bf16 input of the next convolution, c <= 4 real channels in a
// 16-channel pixel (core/models/model_final.py:446-448: conv -> PixelShuffle -> conv): out[n][4y+i][4x+j][k] =
// in[n][y][x][k*16 + i*4 + j] for k < c, zero for k >= c.  One lane = one 16-byte half of an output pixel (the stores of a
// wave are one contiguous kilobyte); the four lanes of an input pixel's row segment read neighbouring 2-byte elements.
__global__ void pixel_shuffle4_nhwc_bf16_to_nhwc16_kernel(const unsigned short* __restrict__ src, uint4* __restrict__ dst, int n, int c,
                                                          int h, int w, int cs)
{
    const long long total = (long long)n * h * 4 * w * 4 * 2;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if ((it & 1) == 0) {
            const long long pix = it >> 1;
            const int xo = (int)(pix % (4ll * w));
            const long long r = pix / (4ll * w);     // (image, y, i) = output row
            const unsigned short* s = src + ((r >> 2) * w + (xo >> 2)) * cs + (int)(r & 3) * 4 + (xo & 3);
            unsigned a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = k < c ? (unsigned)s[k * 16] : 0u;
            o.x = a[0] | (a[1] << 16);
            o.y = a[2] | (a[3] << 16);
        }
        dst[it] = o;
    }
}

// Its backward (autograd of nn.PixelShuffle under loss.backward(), core/trainer.py:160): the gradient of the 16-channel
// image gathered back into the feature map's layout, dt[n][y][x][k*16 + i*4 + j] = d[n][4y+i][4x+j][k] for k < c, zero in
// the feature map's other channels.  One lane = one 16-byte chunk of a feature pixel: channels 8q..8q+7 = colour q / 2,
// sub-rows 2 (q & 1) and 2 (q & 1) + 1, four columns each.
__global__ void pixel_shuffle4_nhwc16_bwd_kernel(const unsigned short* __restrict__ d, uint4* __restrict__ dt, int n, int c, int h, int w,
                                                 int cs)
{
    const int chunks = cs / 8;
    const long long total = (long long)n * h * w * chunks;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(it % chunks);
        const long long pix = it / chunks;
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        const int k = q >> 1;
        if (k < c) {
            const int x = (int)(pix % w);
            const long long r = pix / w;                 // (image, y)
            unsigned a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = 2 * (q & 1) + (e >> 2), j = e & 3;
                a[e] = (unsigned)d[((r * 4 + i) * (4ll * w) + 4 * x + j) * 16 + k];
            }
            o = make_uint4(a[0] | (a[1] << 16), a[2] | (a[3] << 16), a[4] | (a[5] << 16), a[6] | (a[7] << 16));
        }
        dt[it] = o;
    }
}

// (N,ho,wo,C) -> (N,h,w,C) with the input at the even positions and zeros elsewhere: the output gradient of a stride-2
// layer (core/models/model_final.py:272-276 under loss.backward()) seen from the stride-1 layer it is a 2 x 2 pick of.
__global__ void zero_insert2_nhwc_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int ho, int wo, int h, int w,
                                         int chunks, long long total)
{
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(it % chunks);
        long long t = it / chunks;
        const int x = (int)(t % w);
        t /= w;
        const int y = (int)(t % h);
        const long long img = t / h;
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (((x | y) & 1) == 0) o = src[((img * ho + (y >> 1)) * wo + (x >> 1)) * chunks + q];
        dst[it] = o;
    }
}

// The tail of the temporal network (core/models/model_final.py:446-449: PixelShuffle(4) -> Conv2d(3, 3, 3, 1, 1) -> ReLU)
// in one pass: src (N,h,w,cs >= 48) bf16 NHWC, channel k*16 + i*4 + j = colour k of output pixel (4y+i, 4x+j);
// dst (N,3,4h,4w) fp32 NCHW = act(conv3x3(shuffled) + bias).  The shuffled image never exists: a work-group stages an
// 8 x 32 tile of input pixels (plus one pixel of halo, zeros outside the image = the convolution's zero padding) in
// LDS; one item = one input pixel and sub-row i = four output pixels of one row x three colours, whose 3 x 6 x 3
// neighbourhood is nine 8-byte LDS reads plus the two neighbours per row; fp32 weights, fp32 arithmetic (81 fused
// multiply-adds per output pixel on the vector ALU: 150 MB of traffic instead of the 900 MB of the 16-channel padded
// detour through the MFMA kernel).
constexpr int kTailLX = 32, kTailLY = 8;
__global__ __launch_bounds__(256) void pixel_shuffle4_conv3x3_c3_kernel(const unsigned short* __restrict__ src, const float* __restrict__ wgt,
                                                                       const float* __restrict__ bias, float* __restrict__ dst, int n, int h,
                                                                       int w, int cs, float slope)
{
    __shared__ uint4 tile[(kTailLY + 2) * (kTailLX + 2) * 6];
    const unsigned short* ts = (const unsigned short*)tile;
    const int tiles_x = (w + kTailLX - 1) / kTailLX, tiles_y = (h + kTailLY - 1) / kTailLY;
    const long long ntiles = (long long)n * tiles_y * tiles_x;
    const long long HW = 16ll * h * w;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y);
        const long long img = t / ((long long)tiles_x * tiles_y);
        for (int it = threadIdx.x; it < (kTailLY + 2) * (kTailLX + 2) * 6; it += blockDim.x) {
            const int chunk = it % 6, px = it / 6;
            const int X = tx * kTailLX - 1 + px % (kTailLX + 2), Y = ty * kTailLY - 1 + px / (kTailLX + 2);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (X >= 0 && X < w && Y >= 0 && Y < h) v = *(const uint4*)(src + ((img * h + Y) * w + X) * cs + chunk * 8);
            tile[it] = v;
        }
        __syncthreads();
        for (int it = threadIdx.x; it < kTailLY * kTailLX * 4; it += blockDim.x) {
            const int lx = it % kTailLX, r = it / kTailLX, i = r & 3, ly = r >> 2;
            const int X = tx * kTailLX + lx, Y = ty * kTailLY + ly;
            if (X >= w || Y >= h) continue;
            float acc[3][4];
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                const float b = bias != nullptr ? bias[co] : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[co][j] = b;
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int ii = i + ky - 1;
                const int lyy = ly + 1 + (ii < 0 ? -1 : (ii > 3 ? 1 : 0));
                const int base = (lyy * (kTailLX + 2) + lx + 1) * 48 + (ii & 3) * 4;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const uint2 m = *(const uint2*)(ts + base + ci * 16);
                    float v[6];
                    v[0] = bf16_bits_to_f32(ts[base + ci * 16 - 48 + 3]);
                    v[1] = bf16_lo(m.x); v[2] = bf16_hi(m.x); v[3] = bf16_lo(m.y); v[4] = bf16_hi(m.y);
                    v[5] = bf16_bits_to_f32(ts[base + ci * 16 + 48]);
#pragma unroll
                    for (int co = 0; co < 3; ++co)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float wv = wgt[((co * 3 + ci) * 3 + ky) * 3 + kx];
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[co][j] = __builtin_fmaf(wv, v[j + kx], acc[co][j]);
                        }
                }
            }
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = fmaxf(acc[co][j], acc[co][j] * slope);   // act(t) = max(t, slope * t)
                *(float4*)(dst + (img * 3 + co) * HW + (4ll * Y + i) * (4ll * w) + 4ll * X) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();
    }
}

// GPT = channel groups (of 8) per thread: with 2 the motion-vector and tap arithmetic (two IEEE divisions, ~150 vector
// instructions) is paid once per 32 bytes of output instead of once per 16.
template <int GPT>
__global__ void warp_nhwc_kernel(const unsigned short* __restrict__ frame, const float* __restrict__ motion,
                                 unsigned short* __restrict__ dst, int c, int h, int w, int mode, int align,
                                 unsigned chunks, unsigned units)
{
    // units = (chunk of a row, row, image), chunk fastest, walked in XCD bands
    unsigned unit;
    if (!xcd_band_unit(units, unit)) return;
    const unsigned rowid = unit / chunks, chunk = unit - rowid * chunks;
    const unsigned cg = (unsigned)c / (8u * GPT);
    const unsigned xi = chunk * blockDim.x + threadIdx.x;
    if (xi >= (unsigned)w * cg) return;
    const int x = (int)(xi / cg), g = (int)(xi - (unsigned)x * cg) * GPT;
    const int img = (int)(rowid / (unsigned)h), y = (int)(rowid - (unsigned)img * (unsigned)h);
    const long long hw = (long long)h * w;
    const float mx = motion[((long long)img * 2 + 0) * hw + (long long)y * w + x];
    const float my = motion[((long long)img * 2 + 1) * hw + (long long)y * w + x];
    const float gx = __fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn((float)x, mx), (float)(w - 1)), 2.0f), 1.0f);
    const float gy = __fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn((float)y, my), (float)(h - 1)), 2.0f), 1.0f);
    const float ix = unnorm(gx, w, align), iy = unnorm(gy, h, align);
    unsigned short* o = dst + (((long long)img * h + y) * w + x) * c + g * 8;
    if (mode == SRHIP_BILINEAR) {
        const Taps t = make_taps(ix, iy, w, h);
#pragma unroll
        for (int k = 0; k < GPT; ++k) blend8(frame, (long long)img * hw, w, h, c, (g + k) * 8, t, o + k * 8);
    } else {
        const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
        const bool ok = xn >= 0 && xn < w && yn >= 0 && yn < h;
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *(const uint4*)(frame + ((long long)img * hw + (long long)yn * w + xn) * c + (g + k) * 8);
            *(uint4*)(o + k * 8) = v;
        }
    }
}

// fp32 NCHW grid-sample: one thread = one output pixel, loops over channels (coalesced along x).
// GRID_FROM: 0 = make_coord grid (fixed-ratio up-sampling), 1 = motion vectors.
template <int GRID_FROM>
__global__ void sample_f32_kernel(const float* __restrict__ src, const float* __restrict__ motion,
                                  float* __restrict__ dst, int n, int c, int sh, int sw, int th, int tw, int mode,
                                  int align, CoordAxis ax, CoordAxis ay, unsigned chunks, unsigned units)
{
    // units = (chunk of a target row, target row, image), chunk fastest, walked in XCD bands
    unsigned unit;
    if (!xcd_band_unit(units, unit)) return;
    const unsigned rowid = unit / chunks, chunk = unit - rowid * chunks;
    const long long thw = (long long)th * tw, shw = (long long)sh * sw;
    const int x = (int)(chunk * blockDim.x + threadIdx.x);
    if (x >= tw) return;
    const int img = (int)(rowid / (unsigned)th), y = (int)(rowid - (unsigned)img * (unsigned)th);
    (void)n;
    {
        const long long pix = (long long)y * tw + x;
        float gx, gy;
        if (GRID_FROM == 0) {
            gx = coord_val(ax, x); gy = coord_val(ay, y);
        } else {
            const float mx = motion[((long long)img * 2 + 0) * thw + pix];
            const float my = motion[((long long)img * 2 + 1) * thw + pix];
            gx = __fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn((float)x, mx), (float)(tw - 1)), 2.0f), 1.0f);
            gy = __fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn((float)y, my), (float)(th - 1)), 2.0f), 1.0f);
        }
        const float ix = unnorm(gx, sw, align), iy = unnorm(gy, sh, align);
        const float* s = src + (long long)img * c * shw;
        float* d = dst + (long long)img * c * thw + pix;
        if (mode == SRHIP_BILINEAR) {
            const Taps t = make_taps(ix, iy, sw, sh);
            // branch-free: every tap is loaded from a clamped (always valid) address and replaced by 0 when it lies
            // outside the image, so the 4*c loads of a pixel are all in flight together; adding +0 keeps the result
            // identical to skipping the tap
            const int ya = t.vy0 ? t.y0 : (t.y0 < 0 ? 0 : sh - 1), yb = t.vy1 ? t.y0 + 1 : (t.y0 + 1 < 0 ? 0 : sh - 1);
            const bool m00 = t.vy0 && t.vx0, m01 = t.vy0 && t.vx1, m10 = t.vy1 && t.vx0, m11 = t.vy1 && t.vx1;
            // the two taps of a row are neighbours in memory: ONE 8-byte load from a base clamped into the row
            // (xp, xp + 1 both valid); dx = x0 - xp says which half is which tap (0 inside the image, -1 / +1 when
            // the left / right tap falls off the edge; further out both taps are masked and the data is unused)
            const int xp = sw >= 2 ? min(max(t.x0, 0), sw - 2) : 0;
            const int dx = t.x0 - xp;
            const int oa = ya * sw + xp, ob = yb * sw + xp;
            for (int k = 0; k < c; ++k) {
                const float* sp = s + (long long)k * shw;
                float a00, a01, a10, a11;
                if (sw >= 2) {
                    const FloatPair pa = *(const FloatPair*)(sp + oa), pb = *(const FloatPair*)(sp + ob);
                    a00 = dx > 0 ? pa.b : pa.a; a01 = dx < 0 ? pa.a : pa.b;
                    a10 = dx > 0 ? pb.b : pb.a; a11 = dx < 0 ? pb.a : pb.b;
                } else {   // one-column source: both taps clamp to column 0
                    a00 = a01 = sp[ya * sw]; a10 = a11 = sp[yb * sw];
                }
                float v = 0.0f;
                v = __fadd_rn(v, m00 ? __fmul_rn(a00, t.wnw) : 0.0f);
                v = __fadd_rn(v, m01 ? __fmul_rn(a01, t.wne) : 0.0f);
                v = __fadd_rn(v, m10 ? __fmul_rn(a10, t.wsw) : 0.0f);
                v = __fadd_rn(v, m11 ? __fmul_rn(a11, t.wse) : 0.0f);
                d[(long long)k * thw] = v;
            }
        } else {
            const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
            const bool ok = xn >= 0 && xn < sw && yn >= 0 && yn < sh;
            for (int k = 0; k < c; ++k)
                d[(long long)k * thw] = ok ? s[(long long)k * shw + (long long)yn * sw + xn] : 0.0f;
        }
    }
}

// Input of the network's HR trunk when the previous output frame rides along (core/trainer.py:130-137 hands the model
// `velocity` and the previous `sr_frames`; core/models/model.py:142-145 is the warp hook): channels [0, cb) are the
// G-buffers, [cb, cb + cf) the previous frame warped by the motion vectors (core/tools/math.py:10-17 + grid_sample,
// zeros padding), the rest zero; fp32 NCHW sources, ONE 16-channel NHWC bf16 pixel (32 bytes) out per thread.  Bit-equal
// to nchw_f32_to_nhwc_bf16(cat(buffers, backward_warp_f32(prev, motion))): same operations in the same order, the fp32
// intermediate never stored.  prev == nullptr (no previous frame yet): the warped channels are zero.
__global__ void warp_concat_nhwc16_kernel(const float* __restrict__ buffers, const float* __restrict__ prev,
                                          const float* __restrict__ motion, unsigned short* __restrict__ dst, int cb, int cf,
                                          int h, int w, int mode, int align, unsigned chunks, unsigned units)
{
    unsigned unit;
    if (!xcd_band_unit(units, unit)) return;   // (chunk of a row, row, image), chunk fastest, walked in XCD bands
    const unsigned rowid = unit / chunks, chunk = unit - rowid * chunks;
    const int x = (int)(chunk * blockDim.x + threadIdx.x);
    if (x >= w) return;
    const int img = (int)(rowid / (unsigned)h), y = (int)(rowid - (unsigned)img * (unsigned)h);
    const long long hw = (long long)h * w, pix = (long long)y * w + x;
    float v[16];
    const float* bsrc = buffers + (long long)img * cb * hw + pix;
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = k < cb ? bsrc[(long long)k * hw] : 0.0f;
    if (prev != nullptr) {
        const float mx = motion[((long long)img * 2 + 0) * hw + pix];
        const float my = motion[((long long)img * 2 + 1) * hw + pix];
        const float gx = __fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn((float)x, mx), (float)(w - 1)), 2.0f), 1.0f);
        const float gy = __fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn((float)y, my), (float)(h - 1)), 2.0f), 1.0f);
        const float ix = unnorm(gx, w, align), iy = unnorm(gy, h, align);
        const float* s = prev + (long long)img * cf * hw;
        if (mode == SRHIP_BILINEAR) {
            // the fp32 warp's arithmetic (sample_f32_kernel): clamped addresses, masked taps, paired loads
            const Taps t = make_taps(ix, iy, w, h);
            const int ya = t.vy0 ? t.y0 : (t.y0 < 0 ? 0 : h - 1), yb = t.vy1 ? t.y0 + 1 : (t.y0 + 1 < 0 ? 0 : h - 1);
            const bool m00 = t.vy0 && t.vx0, m01 = t.vy0 && t.vx1, m10 = t.vy1 && t.vx0, m11 = t.vy1 && t.vx1;
            const int xp = min(max(t.x0, 0), w - 2);
            const int dx = t.x0 - xp;
            const int oa = ya * w + xp, ob = yb * w + xp;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k >= cb && k < cb + cf) {
                    const float* sp = s + (long long)(k - cb) * hw;
                    const FloatPair pa = *(const FloatPair*)(sp + oa), pb = *(const FloatPair*)(sp + ob);
                    const float a00 = dx > 0 ? pa.b : pa.a, a01 = dx < 0 ? pa.a : pa.b;
                    const float a10 = dx > 0 ? pb.b : pb.a, a11 = dx < 0 ? pb.a : pb.b;
                    float r = 0.0f;
                    r = __fadd_rn(r, m00 ? __fmul_rn(a00, t.wnw) : 0.0f);
                    r = __fadd_rn(r, m01 ? __fmul_rn(a01, t.wne) : 0.0f);
                    r = __fadd_rn(r, m10 ? __fmul_rn(a10, t.wsw) : 0.0f);
                    r = __fadd_rn(r, m11 ? __fmul_rn(a11, t.wse) : 0.0f);
                    v[k] = r;
                }
            }
        } else {
            const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
            const bool ok = xn >= 0 && xn < w && yn >= 0 && yn < h;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k >= cb && k < cb + cf) v[k] = ok ? s[(long long)(k - cb) * hw + (long long)yn * w + xn] : 0.0f;
        }
    }
    uint4* d = (uint4*)(dst + ((long long)img * hw + pix) * 16);
    d[0] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    d[1] = make_uint4(pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15]));
}

// backward of the bilinear sampler: scatter-add into dsrc (fp32 atomics), optional dmotion.
template <int GRID_FROM>
__global__ void sample_bwd_f32_kernel(const float* __restrict__ src, const float* __restrict__ motion,
                                      const float* __restrict__ ddst, float* __restrict__ dsrc,
                                      float* __restrict__ dmotion, int n, int c, int sh, int sw, int th, int tw,
                                      int align, CoordAxis ax, CoordAxis ay)
{
    const long long thw = (long long)th * tw, shw = (long long)sh * sw;
    const long long total = (long long)n * thw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int img = (int)(i / thw);
        const long long pix = i - (long long)img * thw;
        const int y = (int)(pix / tw), x = (int)(pix - (long long)y * tw);
        float gx, gy;
        if (GRID_FROM == 0) {
            gx = coord_val(ax, x); gy = coord_val(ay, y);
        } else {
            const float mx = motion[((long long)img * 2 + 0) * thw + pix];
            const float my = motion[((long long)img * 2 + 1) * thw + pix];
            gx = __fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn((float)x, mx), (float)(tw - 1)), 2.0f), 1.0f);
            gy = __fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn((float)y, my), (float)(th - 1)), 2.0f), 1.0f);
        }
        const float ix = unnorm(gx, sw, align), iy = unnorm(gy, sh, align);
        const Taps t = make_taps(ix, iy, sw, sh);
        const float wx1 = ix - floorf(ix), wx0 = 1.0f - wx1;
        const float wy1 = iy - floorf(iy), wy0 = 1.0f - wy1;
        const long long o00 = (long long)t.y0 * sw + t.x0;
        float gix = 0.0f, giy = 0.0f;
        for (int k = 0; k < c; ++k) {
            const float g = ddst[((long long)img * c + k) * thw + pix];
            float* dp = dsrc + ((long long)img * c + k) * shw;
            const float* sp = (GRID_FROM == 1) ? src + ((long long)img * c + k) * shw : nullptr;
            if (t.vy0 && t.vx0) { atomicAdd(dp + o00, g * t.wnw); if (GRID_FROM == 1) { const float v = sp[o00]; gix -= v * wy0 * g; giy -= v * wx0 * g; } }
            if (t.vy0 && t.vx1) { atomicAdd(dp + o00 + 1, g * t.wne); if (GRID_FROM == 1) { const float v = sp[o00 + 1]; gix += v * wy0 * g; giy -= v * wx1 * g; } }
            if (t.vy1 && t.vx0) { atomicAdd(dp + o00 + sw, g * t.wsw); if (GRID_FROM == 1) { const float v = sp[o00 + sw]; gix -= v * wy1 * g; giy += v * wx0 * g; } }
            if (t.vy1 && t.vx1) { atomicAdd(dp + o00 + sw + 1, g * t.wse); if (GRID_FROM == 1) { const float v = sp[o00 + sw + 1]; gix += v * wy1 * g; giy += v * wx1 * g; } }
        }
        if (GRID_FROM == 1 && dmotion) {
            // d ix / d gx = (W-1)/2 (align) or W/2; d gx / d mx = 2/(W-1)
            const float sx = (align ? (float)(sw - 1) * 0.5f : (float)sw * 0.5f) * (2.0f / (float)(tw - 1));
            const float sy = (align ? (float)(sh - 1) * 0.5f : (float)sh * 0.5f) * (2.0f / (float)(th - 1));
            dmotion[((long long)img * 2 + 0) * thw + pix] = gix * sx;
            dmotion[((long long)img * 2 + 1) * thw + pix] = giy * sy;
        }
    }
}

// ------------------------------------------------------------------------------------------
// pixel shuffle / unshuffle (pure permutations), nearest down-sampling
// ------------------------------------------------------------------------------------------
template <typename T, bool UNSHUFFLE>
__global__ void pixel_shuffle_kernel(const T* __restrict__ src, T* __restrict__ dst, int c, int h, int w, int r, int vec)
{
    // "big" tensor (N,C,H*r,W*r), "deep" tensor (N,C*r*r,H,W).  grid: (chunks of W, H*r, N*C); a thread owns
    // column xx of the deep tensor for one big row: r deep planes, each read/written coalesced across the
    // wave, against r contiguous elements of the big row.
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= w) return;
    const int Y = blockIdx.y, plane = blockIdx.z;          // plane = img * c + ch
    const int yy = Y / r, ii = Y - yy * r;
    const long long hw = (long long)h * w;
    const long long deep0 = ((long long)plane * r * r + (long long)ii * r) * hw + (long long)yy * w + xx;
    const long long big0 = ((long long)plane * h * r + Y) * ((long long)w * r) + (long long)xx * r;
    (void)c;
    if (vec && r == 4 && sizeof(T) == 4) {  // vec: the big tensor is 16-byte aligned
        if (UNSHUFFLE) {
            const uint4 v = *(const uint4*)(src + big0);
            dst[deep0] = (T)v.x; dst[deep0 + hw] = (T)v.y; dst[deep0 + 2 * hw] = (T)v.z; dst[deep0 + 3 * hw] = (T)v.w;
        } else {
            *(uint4*)(dst + big0) = make_uint4((unsigned)src[deep0], (unsigned)src[deep0 + hw], (unsigned)src[deep0 + 2 * hw],
                                               (unsigned)src[deep0 + 3 * hw]);
        }
    } else {
        for (int jj = 0; jj < r; ++jj) {
            if (UNSHUFFLE) dst[deep0 + jj * hw] = src[big0 + jj];
            else dst[big0 + jj] = src[deep0 + jj * hw];
        }
    }
}

__global__ void pixel_shuffle_nhwc_to_nchw_f32_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst,
                                                      int c, int h, int w, int r, int cstride, int vec)
{
    // grid: (chunks of W, H*r, N*C); a thread converts the r channels (ch*r*r + ii*r + 0..r-1) of source pixel
    // (yy, xx) into r contiguous fp32 of the output row
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= w) return;
    const int Y = blockIdx.y, plane = blockIdx.z;
    const int img = plane / c, ch = plane - img * c;
    const int yy = Y / r, ii = Y - yy * r;
    const unsigned short* sp = src + (((long long)img * h + yy) * w + xx) * cstride + ch * r * r + ii * r;
    float* dp = dst + ((long long)plane * h * r + Y) * ((long long)w * r) + (long long)xx * r;
    if (vec && r == 4) {
        const uint2 v = *(const uint2*)sp;   // 4 bf16, 8-byte aligned: cstride % 8 == 0 and r == 4
        *(float4*)dp = make_float4(bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y));
    } else {
        for (int jj = 0; jj < r; ++jj) dp[jj] = bf16_bits_to_f32(sp[jj]);
    }
}

// r == 4 form: a thread owns ONE source pixel and walks its channels, so every source line is fetched once (with one
// thread per (plane, output row) the 96-byte pixel of the 48-channel head was re-read by 12 work-groups far apart in
// dispatch order: 6x the source bytes came through the L2).  Per output channel it reads 16 bf16 (two 16-byte loads)
// and writes four float4, one per output row; consecutive lanes write consecutive 16 bytes of each row.
__global__ void pixel_shuffle4_nhwc_to_nchw_f32_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst,
                                                       int c, int h, int w, int cstride)
{
    // grid: (chunks of W, H, N)
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= w) return;
    const int yy = blockIdx.y, img = blockIdx.z;
    const uint4* sp = (const uint4*)(src + (((long long)img * h + yy) * w + xx) * cstride);
    const long long ow = (long long)w * 4, plane = (long long)h * 4 * ow;
    float* dp = dst + (long long)img * c * plane + (long long)yy * 4 * ow + (long long)xx * 4;
    for (int ch = 0; ch < c; ++ch) {
        const uint4 lo = sp[2 * ch], hi = sp[2 * ch + 1];   // channels 16ch .. 16ch+15 = rows ii = 0..3 x columns jj = 0..3
        float* o = dp + ch * plane;
        *(float4*)(o) = make_float4(bf16_lo(lo.x), bf16_hi(lo.x), bf16_lo(lo.y), bf16_hi(lo.y));
        *(float4*)(o + ow) = make_float4(bf16_lo(lo.z), bf16_hi(lo.z), bf16_lo(lo.w), bf16_hi(lo.w));
        *(float4*)(o + 2 * ow) = make_float4(bf16_lo(hi.x), bf16_hi(hi.x), bf16_lo(hi.y), bf16_hi(hi.y));
        *(float4*)(o + 3 * ow) = make_float4(bf16_lo(hi.z), bf16_hi(hi.z), bf16_lo(hi.w), bf16_hi(hi.w));
    }
}

template <typename T>
__global__ void nearest_down_kernel(const T* __restrict__ src, T* __restrict__ dst, int h, int w, int ho, int wo, float inv)
{
    // grid: (chunks of an output row, output row, plane)
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    if (ox >= wo) return;
    const int oy = blockIdx.y;
    const long long pl = blockIdx.z;
    int iy = (int)floorf(__fmul_rn((float)oy, inv)); if (iy > h - 1) iy = h - 1;
    int ix = (int)floorf(__fmul_rn((float)ox, inv)); if (ix > w - 1) ix = w - 1;
    dst[(pl * ho + oy) * wo + ox] = src[(pl * h + iy) * w + ix];
}

// ------------------------------------------------------------------------------------------
// coordinate features
// ------------------------------------------------------------------------------------------
__global__ void make_coord_kernel(float* __restrict__ dst, int h, int w, CoordAxis ax, CoordAxis ay)
{
    const long long hw = (long long)h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / w), x = (int)(i - (long long)y * w);
        dst[i] = coord_val(ax, x);
        dst[hw + i] = coord_val(ay, y);
    }
}

__global__ void rel_coord_kernel(float* __restrict__ dst, int th, int tw, int sh, int sw, CoordAxis hx, CoordAxis hy,
                                 CoordAxis lx, CoordAxis ly)
{
    const long long hw = (long long)th * tw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / tw), x = (int)(i - (long long)y * tw);
        const float cx = coord_val(hx, x), cy = coord_val(hy, y);
        const int xn = (int)nearbyintf(unnorm(cx, sw, false)), yn = (int)nearbyintf(unnorm(cy, sh, false));
        float qx = 0.0f, qy = 0.0f;
        if (xn >= 0 && xn < sw && yn >= 0 && yn < sh) { qx = coord_val(lx, xn); qy = coord_val(ly, yn); }
        dst[i] = __fmul_rn(__fsub_rn(cx, qx), (float)sw);
        dst[hw + i] = __fmul_rn(__fsub_rn(cy, qy), (float)sh);
    }
}

// ------------------------------------------------------------------------------------------
// concat (NHWC bf16, 16-B chunks) and channels-first LayerNorm (fp32 NCHW)
// ------------------------------------------------------------------------------------------
__global__ void concat_nhwc_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ dst,
                                   long long pixels, int qa, int qb)
{
    const int q = qa + qb;
    const long long total = pixels * q;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / q;
        const int k = (int)(i - p * q);
        dst[i] = k < qa ? a[p * qa + k] : b[p * qb + (k - qa)];
    }
}

__global__ void layernorm_cf_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                    const float* __restrict__ bias, float* __restrict__ y, int n, int c, long long hw,
                                    float eps)
{
    const long long total = (long long)n * hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / hw, pix = i - img * hw;
        const float* xp = x + img * c * hw + pix;
        float* yp = y + img * c * hw + pix;
        float u = 0.0f;
        for (int k = 0; k < c; ++k) u += xp[(long long)k * hw];
        u = u / (float)c;
        float s = 0.0f;
        for (int k = 0; k < c; ++k) { const float d = xp[(long long)k * hw] - u; s += d * d; }
        s = s / (float)c;
        const float den = sqrtf(s + eps);
        for (int k = 0; k < c; ++k) yp[(long long)k * hw] = weight[k] * ((xp[(long long)k * hw] - u) / den) + bias[k];
    }
}

// Disocclusion logic of the temporal variant in one pass over the LR frame (model_final.py:472-486,501-504 over
// the mask functions at :50-139; colour distance core/tools/color.py:22-31).  One thread per (image, pixel):
// the 5x5 range of the current frame and the 3x3 base-colour neighbourhood do not depend on the history index.
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

__global__ void disocclusion_kernel(const float* __restrict__ cur_frame, const float* __restrict__ cur_buffer,
                                    const float* __restrict__ hist_frames, const float* __restrict__ hist_buffer,
                                    unsigned char* __restrict__ mask, float* __restrict__ marked_frames,
                                    float* __restrict__ aux, float* __restrict__ enhance, int fh, int h, int w)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    const int y = blockIdx.y, b = blockIdx.z;
    const int hw = h * w, pix = y * w + x;
    const float* cf = cur_frame + (long long)b * 3 * hw;
    const float* cb = cur_buffer + (long long)b * 10 * hw;
    float lo[3], hi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = INFINITY; hi[c] = -INFINITY; }
    for (int dy = -2; dy <= 2; ++dy) {
        const int row = reflect_idx(y + dy, h) * w;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int q = row + reflect_idx(x + dx, w);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = cf[c * hw + q];
                lo[c] = v < lo[c] ? v : lo[c];
                hi[c] = v > hi[c] ? v : hi[c];
            }
        }
    }
    float nr[9], nb[9], ng[9];  // base colour of the 3x3 neighbourhood (channel 1 is named b, channel 2 g: color.py:23)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int q = reflect_idx(y + dy, h) * w + reflect_idx(x + dx, w);
            const int k = (dy + 1) * 3 + dx + 1;
            nr[k] = cb[6 * hw + q]; nb[k] = cb[7 * hw + q]; ng[k] = cb[8 * hw + q];
        }
    float cv[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) cv[c] = cb[c * hw + pix];
    int count = 0;
    for (int f = 0; f < fh; ++f) {
        const long long bf = (long long)b * fh + f;
        const float* hf = hist_frames + bf * 3 * hw + pix;
        const float* hb = hist_buffer + bf * 10 * hw + pix;
        float fv[3], hv[10];
#pragma unroll
        for (int c = 0; c < 3; ++c) fv[c] = hf[c * hw];
#pragma unroll
        for (int c = 0; c < 10; ++c) hv[c] = hb[c * hw];
        bool m_frame = true;
#pragma unroll
        for (int c = 0; c < 3; ++c) m_frame = m_frame && ((fv[c] < lo[c]) || (fv[c] > hi[c]));
        int far = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float r = (nr[k] + hv[6]) / 2.0f, dr = nr[k] - hv[6], dg = ng[k] - hv[8], db = nb[k] - hv[7];
            const float d = __fsqrt_rn(((2.0f + r) * (dr * dr) + 4.0f * (dg * dg)) + (2.0f + (1.0f - r)) * (db * db));
            far += d > 0.5f ? 1 : 0;
        }
        const bool m_color = __fdiv_rn((float)far, 9.0f) > 0.7f;
        const bool m_depth = (cv[9] - hv[9]) > (0.1f * cv[9]);
        const bool m = m_frame || m_color || m_depth;
        if (mask != nullptr) mask[bf * hw + pix] = m ? 1 : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) marked_frames[(bf * 3 + c) * hw + pix] = m ? 0.0f : fv[c];
        float* a = aux + bf * 14 * hw + pix;
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c * hw] = cv[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) a[(3 + c) * hw] = cv[6 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) a[(7 + c) * hw] = m ? 0.0f : hv[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) a[(10 + c) * hw] = m ? 0.0f : hv[6 + c];
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) dot += ((cv[c] - 0.5f) * 2.0f) * ((hv[c] - 0.5f) * 2.0f);
        count += (!(fabsf(dot) < 1e-3f) && (dot < 0.707f)) ? 1 : 0;
        count += ((cv[5] < 0.02f) || (fabsf(cv[5] - hv[5]) > 0.25f)) ? 1 : 0;
        count += ((cv[3] > 0.95f) || (fabsf(cv[3] - hv[3]) > 0.25f)) ? 1 : 0;
    }
    enhance[(long long)b * hw + pix] = __fdiv_rn((float)count, (float)(3 * fh));
}

// (N,h,w,4C) with channel = (i*2+j)*C + c  ->  (N,2h,2w,C): out[2y+i][2x+j][c].  16-byte chunks.
__global__ void depth_to_space2_nhwc_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int h, int w, int q)
{
    // grid: (chunks of an output row, output row, image); q = C/8 chunks per output pixel
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * w * q) return;
    const int Y = blockIdx.y, n = blockIdx.z;
    const int X = i / q, k = i - X * q;
    const long long in_px = ((long long)n * h + (Y >> 1)) * w + (X >> 1);
    const long long out_px = ((long long)n * 2 * h + Y) * (2 * w) + X;
    dst[out_px * q + k] = src[in_px * 4 * q + ((Y & 1) * 2 + (X & 1)) * q + k];
}

// Sample decode of the data loader (core/data/supersampling_sequence.py:153-200): fp16 (C,H,W) -> fp32 crop with the
// two flips; channel k is negated when bit k of neg_mask is set (the motion-vector sign fix of the flips).
__global__ void crop_flip_f16_f32_kernel(const __half* __restrict__ src, float* __restrict__ dst, int h, int w, int x1, int y1,
                                         int ch, int cw, int flip_rows, int flip_cols, unsigned neg_mask)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= cw) return;
    const int y = blockIdx.y, c = blockIdx.z;
    const int sy = y1 + (flip_rows ? ch - 1 - y : y), sx = x1 + (flip_cols ? cw - 1 - x : x);
    const float v = __half2float(src[((long long)c * h + sy) * w + sx]);
    dst[((long long)c * ch + y) * cw + x] = ((neg_mask >> c) & 1u) ? -v : v;
}

__global__ void clamp_depth_kernel(const float* src, float* dst, long long n, float t1, float t2, float log2_span)   // dst may be src
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = src[i];
        dst[i] = d >= t2 ? t1 + log2_span + log10f((d - t2) + 1.0f) : (d >= t1 ? t1 + log2f((d - t1) + 1.0f) : d);
    }
}

// LayerNorm backward, pass 1: dx, and per pixel the two numbers pass 2 needs (mean, 1/den).
__global__ void layernorm_cf_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                           const float* __restrict__ dy, float* __restrict__ dx,
                                           float* __restrict__ stats, int n, int c, long long hw, float eps)
{
    const long long total = (long long)n * hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / hw, pix = i - img * hw;
        const float* xp = x + img * c * hw + pix;
        const float* gp = dy + img * c * hw + pix;
        float* dp = dx + img * c * hw + pix;
        float u = 0.0f;
        for (int k = 0; k < c; ++k) u += xp[(long long)k * hw];
        u = u / (float)c;
        float s = 0.0f;
        for (int k = 0; k < c; ++k) { const float d = xp[(long long)k * hw] - u; s += d * d; }
        s = s / (float)c;
        const float rden = 1.0f / sqrtf(s + eps);
        float mg = 0.0f, mgx = 0.0f;
        for (int k = 0; k < c; ++k) {
            const float xh = (xp[(long long)k * hw] - u) * rden, g = gp[(long long)k * hw] * weight[k];
            mg += g; mgx += g * xh;
        }
        mg = mg / (float)c; mgx = mgx / (float)c;
        for (int k = 0; k < c; ++k) {
            const float xh = (xp[(long long)k * hw] - u) * rden, g = gp[(long long)k * hw] * weight[k];
            dp[(long long)k * hw] = (g - mg - xh * mgx) * rden;
        }
        stats[2 * i] = u; stats[2 * i + 1] = rden;
    }
}

// Register-resident forms for c <= 64 (the path's LayerNorms have 60-64 channels): a thread keeps its pixel's x (and dy)
// in registers, so every tensor crosses HBM once (the loops above re-read x four times and dy twice through the caches, and
// the parameter pass reads both again).  Same operation order per pixel as the loops above: bit-equal y and dx.
constexpr int kLnRegC = 64;
__global__ __launch_bounds__(256) void layernorm_cf_reg_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                                               const float* __restrict__ bias, float* __restrict__ y, int n, int c,
                                                               long long hw, float eps)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n * hw) return;
    const long long img = i / hw, pix = i - img * hw;
    const float* xp = x + img * c * hw + pix;
    float* yp = y + img * c * hw + pix;
    float xv[kLnRegC];
#pragma unroll
    for (int k = 0; k < kLnRegC; ++k) xv[k] = k < c ? xp[(long long)k * hw] : 0.0f;
    float u = 0.0f;
#pragma unroll
    for (int k = 0; k < kLnRegC; ++k) if (k < c) u += xv[k];
    u = u / (float)c;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < kLnRegC; ++k) if (k < c) { const float d = xv[k] - u; s += d * d; }
    s = s / (float)c;
    const float den = sqrtf(s + eps);
#pragma unroll
    for (int k = 0; k < kLnRegC; ++k) if (k < c) yp[(long long)k * hw] = weight[k] * ((xv[k] - u) / den) + bias[k];
}

// Backward: dx as above, and the parameter sums of the block's 256 pixels.  A wave adds its 64 lanes' 128 values
// (dy * xhat and dy per channel, interleaved) by a halving butterfly: at distance m a lane hands the half of its values
// that its partner keeps across and adds what it receives to the half it keeps (126 exchanges instead of 6 per value);
// afterwards lane k holds both sums of channel k.  The four waves' sums are added in order and go to
// partial[block][channel][2]; a second kernel adds the blocks in a fixed order: deterministic.
__global__ __launch_bounds__(256) void layernorm_cf_bwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                                                   const float* __restrict__ dy, float* __restrict__ dx,
                                                                   float* __restrict__ partial, int n, int c, long long hw, float eps)
{
    __shared__ float2 sred[4][kLnRegC];
    const long long total = (long long)n * hw;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i0 < total;
    const long long i = live ? i0 : total - 1;
    const long long img = i / hw, pix = i - img * hw;
    const float* xp = x + img * c * hw + pix;
    const float* gp = dy + img * c * hw + pix;
    float* dp = dx + img * c * hw + pix;
    float xv[kLnRegC], gv[kLnRegC];
#pragma unroll
    for (int k = 0; k < kLnRegC; ++k) { xv[k] = k < c ? xp[(long long)k * hw] : 0.0f; gv[k] = k < c ? gp[(long long)k * hw] : 0.0f; }
    float u = 0.0f;
#pragma unroll
    for (int k = 0; k < kLnRegC; ++k) if (k < c) u += xv[k];
    u = u / (float)c;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < kLnRegC; ++k) if (k < c) { const float d = xv[k] - u; s += d * d; }
    s = s / (float)c;
    const float rden = 1.0f / sqrtf(s + eps);
    float mg = 0.0f, mgx = 0.0f;
#pragma unroll
    for (int k = 0; k < kLnRegC; ++k) if (k < c) {
        const float xh = (xv[k] - u) * rden, g = gv[k] * weight[k];
        mg += g; mgx += g * xh;
    }
    mg = mg / (float)c; mgx = mgx / (float)c;
    float v[2 * kLnRegC];
#pragma unroll
    for (int k = 0; k < kLnRegC; ++k) {
        const float xh = (xv[k] - u) * rden, g = gv[k] * (k < c ? weight[k] : 0.0f);
        if (k < c && live) dp[(long long)k * hw] = (g - mg - xh * mgx) * rden;
        v[2 * k] = live ? gv[k] * xh : 0.0f;
        v[2 * k + 1] = live ? gv[k] : 0.0f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#define SRHIP_LN_STEP(M, N) \
    { \
        const bool up = (lane & M) != 0; \
        _Pragma("unroll") for (int j = 0; j < N; ++j) { \
            const float send = up ? v[j] : v[j + N]; \
            const float keep = up ? v[j + N] : v[j]; \
            v[j] = keep + __shfl_xor(send, M); \
        } \
    }
    SRHIP_LN_STEP(32, 64) SRHIP_LN_STEP(16, 32) SRHIP_LN_STEP(8, 16) SRHIP_LN_STEP(4, 8) SRHIP_LN_STEP(2, 4) SRHIP_LN_STEP(1, 2)
#undef SRHIP_LN_STEP
    sred[wave][lane] = make_float2(v[0], v[1]);   // lane k: channel k
    __syncthreads();
    if (threadIdx.x < kLnRegC) {
        const int k = threadIdx.x;
        const float2 a = sred[0][k], b = sred[1][k], d = sred[2][k], e = sred[3][k];
        float2 t;
        t.x = ((a.x + b.x) + d.x) + e.x;
        t.y = ((a.y + b.y) + d.y) + e.y;
        ((float2*)partial)[(size_t)blockIdx.x * kLnRegC + k] = t;
    }
}
__global__ __launch_bounds__(256) void layernorm_cf_bwd_reg_final_kernel(const float* __restrict__ partial, int nblocks,
                                                                         float* __restrict__ dweight, float* __restrict__ dbias)
{
    __shared__ float sw[256], sb[256];
    const int k = blockIdx.x;
    float aw = 0.0f, ab = 0.0f;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        const float2 t = ((const float2*)partial)[(size_t)b * kLnRegC + k];
        aw += t.x; ab += t.y;
    }
    sw[threadIdx.x] = aw; sb[threadIdx.x] = ab;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { sw[threadIdx.x] += sw[threadIdx.x + st]; sb[threadIdx.x] += sb[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { dweight[k] = sw[0]; dbias[k] = sb[0]; }
}

// pass 2: block (k, slice) sums dy*xh and dy of channel k over its slice of the pixels in a fixed order
// (strided per-thread sums, then a tree over the block); pass 3 adds the slices in order: deterministic.
constexpr int kLnSlices = 64;
__global__ __launch_bounds__(256) void layernorm_cf_bwd_param_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     const float* __restrict__ stats, float* __restrict__ partial,
                                                                     int n, int c, long long hw)
{
    __shared__ float sw[256], sb[256];
    const int k = blockIdx.x, slice = blockIdx.y;
    const long long total = (long long)n * hw;
    const long long per = (total + kLnSlices - 1) / kLnSlices;
    const long long lo = slice * per, hi = lo + per < total ? lo + per : total;
    float aw = 0.0f, ab = 0.0f;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const long long img = i / hw, pix = i - img * hw;
        const long long o = (img * c + k) * hw + pix;
        const float g = dy[o];
        aw += g * ((x[o] - stats[2 * i]) * stats[2 * i + 1]);
        ab += g;
    }
    sw[threadIdx.x] = aw; sb[threadIdx.x] = ab;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { sw[threadIdx.x] += sw[threadIdx.x + st]; sb[threadIdx.x] += sb[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[(k * kLnSlices + slice) * 2] = sw[0]; partial[(k * kLnSlices + slice) * 2 + 1] = sb[0]; }
}

__global__ void layernorm_cf_bwd_final_kernel(const float* __restrict__ partial, float* __restrict__ dweight,
                                              float* __restrict__ dbias, int c)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= c) return;
    float aw = 0.0f, ab = 0.0f;
    for (int sl = 0; sl < kLnSlices; ++sl) { aw += partial[(k * kLnSlices + sl) * 2]; ab += partial[(k * kLnSlices + sl) * 2 + 1]; }
    dweight[k] = aw; dbias[k] = ab;
}

}  // namespace

extern "C" {

SRHIP_API int srhip_disocclusion_f32(const float* cur_frame, const float* cur_buffer, const float* hist_frames,
                                     const float* hist_buffer, unsigned char* mask, float* marked_frames, float* aux,
                                     float* enhance_weight, int b, int f, int h, int w, void* stream)
{
    SRHIP_REQUIRE(cur_frame && cur_buffer && hist_frames && hist_buffer && marked_frames && aux && enhance_weight,
                  "disocclusion_f32: null pointer");
    SRHIP_REQUIRE(b >= 1 && f >= 1 && h >= 3 && w >= 3, "disocclusion_f32: bad shape (reflect padding of 2 needs h, w >= 3)");
    SRHIP_REQUIRE(h <= 65535 && b <= 65535 && (long long)b * f * 14 * h * w < (1ll << 40), "disocclusion_f32: shape too large");
    SRHIP_REQUIRE((long long)h * w * 14 < (1ll << 31), "disocclusion_f32: a frame must stay below 2^31 elements");
    hipLaunchKernelGGL(disocclusion_kernel, dim3(srhip_div_up(w, 64), h, b), dim3(64), 0, (hipStream_t)stream, cur_frame,
                       cur_buffer, hist_frames, hist_buffer, mask, marked_frames, aux, enhance_weight, f, h, w);
    return srhip_check_launch("disocclusion_f32");
}

SRHIP_API int srhip_depth_to_space2_nhwc(const void* src, void* dst, int n, int h, int w, int c, void* stream)
{
    SRHIP_REQUIRE(src && dst, "depth_to_space2_nhwc: null pointer");
    SRHIP_REQUIRE(n >= 1 && h >= 1 && w >= 1 && c >= 8 && c % 8 == 0, "depth_to_space2_nhwc: bad shape (c %% 8 == 0)");
    SRHIP_REQUIRE(2 * h <= 65535 && n <= 65535, "depth_to_space2_nhwc: at most 65535 output rows and images per call");
    const int q = c / 8;
    hipLaunchKernelGGL(depth_to_space2_nhwc_kernel, dim3(srhip_div_up((long long)2 * w * q, kThreads), 2 * h, n), dim3(kThreads), 0,
                       (hipStream_t)stream, (const uint4*)src, (uint4*)dst, h, w, q);
    return srhip_check_launch("depth_to_space2_nhwc");
}

SRHIP_API int srhip_crop_flip_f16_f32(const void* src, float* dst, int c, int h, int w, int x1, int y1, int crop_h, int crop_w,
                                      int flip_rows, int flip_cols, unsigned neg_mask, void* stream)
{
    SRHIP_REQUIRE(src && dst, "crop_flip_f16_f32: null pointer");
    SRHIP_REQUIRE(c >= 1 && c <= 32 && h >= 1 && w >= 1, "crop_flip_f16_f32: bad shape (at most 32 channels)");
    SRHIP_REQUIRE(x1 >= 0 && y1 >= 0 && crop_h >= 1 && crop_w >= 1 && x1 + crop_w <= w && y1 + crop_h <= h,
                  "crop_flip_f16_f32: crop window (%d,%d)+(%dx%d) outside the %dx%d image", x1, y1, crop_w, crop_h, w, h);
    SRHIP_REQUIRE(crop_h <= 65535, "crop_flip_f16_f32: at most 65535 rows");
    hipLaunchKernelGGL(crop_flip_f16_f32_kernel, dim3(srhip_div_up(crop_w, kThreads), crop_h, c), dim3(kThreads), 0,
                       (hipStream_t)stream, (const __half*)src, dst, h, w, x1, y1, crop_h, crop_w, flip_rows, flip_cols, neg_mask);
    return srhip_check_launch("crop_flip_f16_f32");
}

SRHIP_API int srhip_clamp_depth_f32(const float* src, float* dst, long long n, float thres1, float thres2, void* stream)
{
    SRHIP_REQUIRE(src && dst, "clamp_depth_f32: null pointer");
    SRHIP_REQUIRE(n >= 1 && thres2 > thres1, "clamp_depth_f32: bad arguments");
    hipLaunchKernelGGL(clamp_depth_kernel, dim3(grid_for(n, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream, src, dst,
                       n, thres1, thres2, log2f(thres2 - thres1));
    return srhip_check_launch("clamp_depth_f32");
}

SRHIP_API size_t srhip_layernorm_cf_bwd_workspace_bytes(int n, int c, int hw)
{
    if (n < 1 || c < 1 || hw < 1) return 0;
    return ((size_t)n * hw * 2 + (size_t)c * kLnSlices * 2) * sizeof(float);
}

// The launches of the backward on a scratch area of (2 n hw + 2 c kLnSlices) floats.  Up to 64 channels: the
// register-resident form, whose block partials (ceil(n hw / 256) x 64 x 2 floats <= n hw / 2 + 128) fit the same area;
// every word it reads is written first by the same call.
static bool ln_force_loops()   // SRHIP_LN_FORM=loops: the channel-loop kernels for every channel count (tests, A/B timing)
{
    const char* e = getenv("SRHIP_LN_FORM");
    return e != nullptr && e[0] == 'l';
}
static void layernorm_bwd_launches(const float* x, const float* weight, const float* dy, float* dx, float* dweight, float* dbias,
                                   float* scratch, int n, int c, int hw, float eps, hipStream_t s)
{
    if (c <= kLnRegC && !ln_force_loops()) {
        const long long nblocks = ((long long)n * hw + 255) / 256;
        hipLaunchKernelGGL(layernorm_cf_bwd_reg_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, x, weight, dy, dx, scratch, n, c,
                           (long long)hw, eps);
        hipLaunchKernelGGL(layernorm_cf_bwd_reg_final_kernel, dim3(c), dim3(256), 0, s, scratch, (int)nblocks, dweight, dbias);
        return;
    }
    float* stats = scratch;
    float* partial = scratch + (size_t)n * hw * 2;
    hipLaunchKernelGGL(layernorm_cf_bwd_dx_kernel, dim3(grid_for((long long)n * hw, srhip_num_cus())), dim3(kThreads), 0, s,
                       x, weight, dy, dx, stats, n, c, (long long)hw, eps);
    hipLaunchKernelGGL(layernorm_cf_bwd_param_kernel, dim3(c, kLnSlices), dim3(256), 0, s, x, dy, stats, partial, n, c, (long long)hw);
    hipLaunchKernelGGL(layernorm_cf_bwd_final_kernel, dim3(srhip_div_up(c, 64)), dim3(64), 0, s, partial, dweight, dbias, c);
}

SRHIP_API int srhip_layernorm_cf_bwd_f32(const float* x, const float* weight, const float* dy, float* dx, float* dweight,
                                         float* dbias, void* workspace, int n, int c, int hw, float eps, void* stream)
{
    SRHIP_REQUIRE(x && weight && dy && dx && dweight && dbias && workspace, "layernorm_cf_bwd_f32: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && hw >= 1, "layernorm_cf_bwd_f32: bad shape");
    SRHIP_REQUIRE(c <= 65535, "layernorm_cf_bwd_f32: at most 65535 channels");
    SRHIP_REQUIRE((long long)n * hw < (1ll << 31) * 256, "layernorm_cf_bwd_f32: too many pixels");
    layernorm_bwd_launches(x, weight, dy, dx, dweight, dbias, (float*)workspace, n, c, hw, eps, (hipStream_t)stream);
    return srhip_check_launch("layernorm_cf_bwd_f32");
}

// ---- the same backward for callers that must not allocate: its workspace is a fixed array in device memory.
// Each GPU the library runs on holds its own copy of the array (8.5 MB) at its own address until the process ends.
constexpr long long kLnStaticMaxPixels = 1ll << 20;   // n * hw
constexpr int kLnStaticMaxChannels = 256;
static __device__ __align__(256) float g_ln_scratch[kLnStaticMaxPixels * 2 + (long long)kLnStaticMaxChannels * kLnSlices * 2];
static std::mutex g_ln_scratch_mu;
static float* g_ln_scratch_ptr[kSrhipMaxDevices] = {};   // device address of the array, per device

SRHIP_API int srhip_layernorm_cf_bwd_f32_static(const float* x, const float* weight, const float* dy, float* dx,
                                                float* dweight, float* dbias, int n, int c, int hw, float eps,
                                                void* stream)
{
    SRHIP_REQUIRE(x && weight && dy && dx && dweight && dbias, "layernorm_cf_bwd_f32_static: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && hw >= 1, "layernorm_cf_bwd_f32_static: bad shape");
    SRHIP_REQUIRE((long long)n * hw <= kLnStaticMaxPixels && c <= kLnStaticMaxChannels,
                  "layernorm_cf_bwd_f32_static: n*hw = %lld pixels / %d channels exceed the fixed scratch (at most %lld pixels, "
                  "%d channels); use srhip_layernorm_cf_bwd_f32 with a caller-owned workspace",
                  (long long)n * hw, c, kLnStaticMaxPixels, kLnStaticMaxChannels);
    const int dev = srhip_current_device();
    float* scratch = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_ln_scratch_mu);
        if (g_ln_scratch_ptr[dev] == nullptr) {
            // the array's name is a host-side handle, not a device address: ask the runtime where THIS device keeps it
            float* ptr = nullptr;
            const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(g_ln_scratch));
            if (e != hipSuccess || ptr == nullptr) {
                srhip_set_error("layernorm_cf_bwd_f32_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
                return SRHIP_ERR_LAUNCH;
            }
            g_ln_scratch_ptr[dev] = ptr;
        }
        scratch = g_ln_scratch_ptr[dev];
    }
    // every word the later kernels read is written first by this call (stats for all n*hw pixels and all c * kLnSlices
    // partials, or the block partials of the register-resident form): the previous call's contents never need clearing
    layernorm_bwd_launches(x, weight, dy, dx, dweight, dbias, scratch, n, c, hw, eps, (hipStream_t)stream);
    return srhip_check_launch("layernorm_cf_bwd_f32_static");
}

SRHIP_API int srhip_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int n, int c, int h, int w, int cpad,
                                          void* stream)
{
    SRHIP_REQUIRE(src && dst, "nchw_f32_to_nhwc_bf16: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && h >= 1 && w >= 1 && c <= cpad, "nchw_f32_to_nhwc_bf16: bad shape");
    const long long hw = (long long)h * w;
    const int g = grid_for((long long)n * hw, srhip_num_cus());
    hipStream_t s = (hipStream_t)stream;
    switch (cpad) {
        case 16: hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel<16>, dim3(g), dim3(kThreads), 0, s, src, (unsigned short*)dst, n, c, hw); break;
        case 32: hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel<32>, dim3(g), dim3(kThreads), 0, s, src, (unsigned short*)dst, n, c, hw); break;
        case 48: hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel<48>, dim3(g), dim3(kThreads), 0, s, src, (unsigned short*)dst, n, c, hw); break;
        case 64: hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel<64>, dim3(g), dim3(kThreads), 0, s, src, (unsigned short*)dst, n, c, hw); break;
        default:
            SRHIP_REQUIRE(cpad % 8 == 0, "nchw_f32_to_nhwc_bf16: cpad must be a multiple of 8 (got %d)", cpad);
            if ((size_t)kLayoutTile * ((cpad / 8) | 1) * sizeof(uint4) <= 48 * 1024) {
                const long long ntiles = ((long long)n * hw + kLayoutTile - 1) / kLayoutTile;
                const long long cap = (long long)srhip_num_cus() * 16;
                hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_tiled_kernel, dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(256),
                                   (size_t)kLayoutTile * ((cpad / 8) | 1) * sizeof(uint4), s, src, (unsigned short*)dst, n, c, hw, cpad);
                break;
            }
            hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_generic_kernel, dim3(grid_for((long long)n * hw * (cpad / 8), srhip_num_cus())),
                               dim3(kThreads), 0, s, src, (unsigned short*)dst, n, c, hw, cpad);
            break;
    }
    return srhip_check_launch("nchw_f32_to_nhwc_bf16");
}

static int launch_pixel_unshuffle4(const float* src, void* dst, int n, int c, int h, int w, float* block_max, int max_cap, void* stream)
{
    SRHIP_REQUIRE(src && dst, "pixel_unshuffle4_nchw_f32_to_nhwc_bf16: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && h >= 1 && w >= 1, "pixel_unshuffle4_nchw_f32_to_nhwc_bf16: bad shape");
    SRHIP_REQUIRE((size_t)src % 16 == 0 && (size_t)dst % 16 == 0, "pixel_unshuffle4_nchw_f32_to_nhwc_bf16: 16-byte aligned tensors");
    const size_t lds = (size_t)kLayoutTile * (c * 4 + 2) * sizeof(uint2);
    SRHIP_REQUIRE(lds <= 48 * 1024, "pixel_unshuffle4_nchw_f32_to_nhwc_bf16: at most 23 input channels (got %d)", c);
    const long long ntiles = ((long long)n * h * w + kLayoutTile - 1) / kLayoutTile;
    long long cap = (long long)srhip_num_cus() * 16;
    if (block_max != nullptr) {
        SRHIP_REQUIRE(max_cap >= 1, "pixel_unshuffle4_nchw_f32_to_nhwc_bf16_max: max_cap must be at least 1");
        if (cap > max_cap) cap = max_cap;
        hipLaunchKernelGGL(pixel_unshuffle4_nchw_f32_to_nhwc_bf16_kernel<true>, dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(256),
                           lds, (hipStream_t)stream, src, (unsigned short*)dst, n, c, h, w, block_max, max_cap);
    } else {
        hipLaunchKernelGGL(pixel_unshuffle4_nchw_f32_to_nhwc_bf16_kernel<false>, dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(256),
                           lds, (hipStream_t)stream, src, (unsigned short*)dst, n, c, h, w, (float*)nullptr, 0);
    }
    return srhip_check_launch("pixel_unshuffle4_nchw_f32_to_nhwc_bf16");
}

SRHIP_API int srhip_pixel_unshuffle4_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int n, int c, int h, int w, void* stream)
{
    return launch_pixel_unshuffle4(src, dst, n, c, h, w, nullptr, 0, stream);
}

SRHIP_API int srhip_pixel_unshuffle4_nchw_f32_to_nhwc_bf16_max(const float* src, void* dst, int n, int c, int h, int w,
                                                               float* block_max, int max_cap, void* stream)
{
    SRHIP_REQUIRE(block_max != nullptr, "pixel_unshuffle4_nchw_f32_to_nhwc_bf16_max: null block_max");
    return launch_pixel_unshuffle4(src, dst, n, c, h, w, block_max, max_cap, stream);
}

SRHIP_API int srhip_nhwc_bf16_to_nchw_f32(const void* src, float* dst, int n, int c, int h, int w, int cstride,
                                          void* stream)
{
    SRHIP_REQUIRE(src && dst, "nhwc_bf16_to_nchw_f32: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && h >= 1 && w >= 1 && c <= cstride, "nhwc_bf16_to_nchw_f32: bad shape");
    const long long hw = (long long)h * w;
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, dim3(grid_for((long long)n * hw, srhip_num_cus())), dim3(kThreads), 0,
                       (hipStream_t)stream, (const unsigned short*)src, dst, n, c, hw, cstride);
    return srhip_check_launch("nhwc_bf16_to_nchw_f32");
}

SRHIP_API int srhip_upsample_grid_bilinear_nhwc(const void* src, void* dst, int n, int c, int sh, int sw, int th,
                                                int tw, void* stream)
{
    SRHIP_REQUIRE(src && dst, "upsample_nhwc: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 8 && c % 8 == 0 && sh >= 1 && sw >= 1 && th >= 1 && tw >= 1, "upsample_nhwc: bad shape");
    SRHIP_REQUIRE(th <= 65535 && n <= 65535, "upsample_grid_bilinear_nhwc: at most 65535 target rows and images per call");
    if (th == 4 * sh && tw == 4 * sw && sh <= 65535) {
        hipLaunchKernelGGL(upsample_int_nhwc_kernel<4>, dim3(srhip_div_up((long long)sw * (c / 8), kThreads), sh, n), dim3(kThreads), 0,
                           (hipStream_t)stream, (const unsigned short*)src, (unsigned short*)dst, c, sh, sw, coord_axis(tw), coord_axis(th));
        return srhip_check_launch("upsample_int_nhwc<4>");
    }
    if (th == 2 * sh && tw == 2 * sw && sh <= 65535) {
        hipLaunchKernelGGL(upsample_int_nhwc_kernel<2>, dim3(srhip_div_up((long long)sw * (c / 8), kThreads), sh, n), dim3(kThreads), 0,
                           (hipStream_t)stream, (const unsigned short*)src, (unsigned short*)dst, c, sh, sw, coord_axis(tw), coord_axis(th));
        return srhip_check_launch("upsample_int_nhwc<2>");
    }
    hipLaunchKernelGGL(upsample_nhwc_kernel, dim3(srhip_div_up((long long)tw * (c / 8), kThreads), th, n), dim3(kThreads), 0,
                       (hipStream_t)stream, (const unsigned short*)src, (unsigned short*)dst, c, sh, sw, th, tw, coord_axis(tw),
                       coord_axis(th));
    return srhip_check_launch("upsample_nhwc");
}

SRHIP_API int srhip_upsample_grid_bilinear_f32(const float* src, float* dst, int n, int c, int sh, int sw, int th,
                                               int tw, void* stream)
{
    SRHIP_REQUIRE(src && dst, "upsample_f32: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && sh >= 1 && sw >= 1 && th >= 1 && tw >= 1, "upsample_f32: bad shape");
    SRHIP_REQUIRE(th <= 65535 && n <= 65535, "upsample_f32: at most 65535 target rows and images per call");
    const long long chunks = srhip_div_up(tw, kThreads), units = chunks * th * n;
    SRHIP_REQUIRE(units < (1ll << 31), "upsample_f32: too many work-groups");
    hipLaunchKernelGGL(sample_f32_kernel<0>, dim3(xcd_band_grid(units)), dim3(kThreads), 0,
                       (hipStream_t)stream, src, (const float*)nullptr, dst, n, c, sh, sw, th, tw, (int)SRHIP_BILINEAR, 0,
                       coord_axis(tw), coord_axis(th), (unsigned)chunks, (unsigned)units);
    return srhip_check_launch("upsample_f32");
}

SRHIP_API int srhip_upsample_grid_bilinear_bwd_f32(const float* ddst, float* dsrc, int n, int c, int sh, int sw, int th,
                                                   int tw, void* stream)
{
    SRHIP_REQUIRE(ddst && dsrc, "upsample_bwd_f32: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && sh >= 1 && sw >= 1 && th >= 1 && tw >= 1, "upsample_bwd_f32: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(dsrc, 0, (size_t)n * c * sh * sw * sizeof(float), s) != hipSuccess) {
        srhip_set_error("upsample_bwd_f32: memset failed");
        return SRHIP_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(sample_bwd_f32_kernel<0>, dim3(grid_for((long long)n * th * tw, srhip_num_cus())), dim3(kThreads), 0, s,
                       (const float*)nullptr, (const float*)nullptr, ddst, dsrc, (float*)nullptr, n, c, sh, sw, th, tw, 0,
                       coord_axis(tw), coord_axis(th));
    return srhip_check_launch("upsample_bwd_f32");
}

SRHIP_API int srhip_backward_warp_f32(const float* frame, const float* motion, float* dst, int n, int c, int h, int w,
                                      int mode, int align_corners, void* stream)
{
    SRHIP_REQUIRE(frame && motion && dst, "backward_warp_f32: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && h >= 2 && w >= 2, "backward_warp_f32: bad shape (h, w must be >= 2)");
    SRHIP_REQUIRE(mode == SRHIP_BILINEAR || mode == SRHIP_NEAREST, "backward_warp_f32: bad mode %d", mode);
    SRHIP_REQUIRE(h <= 65535 && n <= 65535, "backward_warp_f32: at most 65535 rows and images per call");
    const long long chunks = srhip_div_up(w, kThreads), units = chunks * h * n;
    SRHIP_REQUIRE(units < (1ll << 31), "backward_warp_f32: too many work-groups");
    hipLaunchKernelGGL(sample_f32_kernel<1>, dim3(xcd_band_grid(units)), dim3(kThreads), 0,
                       (hipStream_t)stream, frame, motion, dst, n, c, h, w, h, w, mode, align_corners ? 1 : 0, CoordAxis{0, 0},
                       CoordAxis{0, 0}, (unsigned)chunks, (unsigned)units);
    return srhip_check_launch("backward_warp_f32");
}

SRHIP_API int srhip_warp_concat_nchw_f32_to_nhwc_bf16(const float* buffers, int cb, const float* prev, int cf, const float* motion,
                                                      void* dst, int n, int h, int w, int mode, int align_corners, void* stream)
{
    SRHIP_REQUIRE(buffers && dst, "warp_concat: null buffers / dst");
    SRHIP_REQUIRE(prev == nullptr || motion != nullptr, "warp_concat: a previous frame needs motion vectors");
    SRHIP_REQUIRE(cb >= 1 && cf >= 0 && cb + cf <= 16, "warp_concat: cb + cf must fit the 16-channel pixel (got %d + %d)", cb, cf);
    SRHIP_REQUIRE(n >= 1 && h >= 2 && w >= 2, "warp_concat: bad shape (h, w must be >= 2)");
    SRHIP_REQUIRE(mode == SRHIP_BILINEAR || mode == SRHIP_NEAREST, "warp_concat: bad mode %d", mode);
    SRHIP_REQUIRE(h <= 65535 && n <= 65535, "warp_concat: at most 65535 rows and images per call");
    const long long chunks = srhip_div_up(w, kThreads), units = chunks * h * n;
    SRHIP_REQUIRE(units < (1ll << 31), "warp_concat: too many work-groups");
    hipLaunchKernelGGL(warp_concat_nhwc16_kernel, dim3(xcd_band_grid(units)), dim3(kThreads), 0, (hipStream_t)stream, buffers, prev,
                       motion, (unsigned short*)dst, cb, cf, h, w, mode, align_corners ? 1 : 0, (unsigned)chunks, (unsigned)units);
    return srhip_check_launch("warp_concat_nchw_f32_to_nhwc_bf16");
}

SRHIP_API int srhip_backward_warp_nhwc(const void* frame, const float* motion, void* dst, int n, int c, int h, int w,
                                       int mode, int align_corners, void* stream)
{
    SRHIP_REQUIRE(frame && motion && dst, "backward_warp_nhwc: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 8 && c % 8 == 0 && h >= 2 && w >= 2, "backward_warp_nhwc: bad shape");
    SRHIP_REQUIRE(mode == SRHIP_BILINEAR || mode == SRHIP_NEAREST, "backward_warp_nhwc: bad mode %d", mode);
    SRHIP_REQUIRE(h <= 65535 && n <= 65535, "backward_warp_nhwc: at most 65535 rows and images per call");
    // one 16-byte channel group per thread: with two (half the motion / tap arithmetic per byte) consecutive lanes read
    // 32 bytes apart, every gather touches twice the cache lines, and the 4K warp took 0.47 ms instead of 0.28
    const int gpt = 1;
    const long long chunks = srhip_div_up((long long)w * (c / (8 * gpt)), kThreads), units = chunks * h * n;
    SRHIP_REQUIRE(units < (1ll << 31), "backward_warp_nhwc: too many work-groups");
    if (gpt == 2)
        hipLaunchKernelGGL(warp_nhwc_kernel<2>, dim3(xcd_band_grid(units)), dim3(kThreads), 0,
                           (hipStream_t)stream, (const unsigned short*)frame, motion, (unsigned short*)dst, c, h, w, mode,
                           align_corners ? 1 : 0, (unsigned)chunks, (unsigned)units);
    else
        hipLaunchKernelGGL(warp_nhwc_kernel<1>, dim3(xcd_band_grid(units)), dim3(kThreads), 0,
                           (hipStream_t)stream, (const unsigned short*)frame, motion, (unsigned short*)dst, c, h, w, mode,
                           align_corners ? 1 : 0, (unsigned)chunks, (unsigned)units);
    return srhip_check_launch("backward_warp_nhwc");
}

SRHIP_API int srhip_backward_warp_bwd_f32(const float* frame, const float* motion, const float* ddst, float* dframe,
                                          float* dmotion, int n, int c, int h, int w, int align_corners, void* stream)
{
    SRHIP_REQUIRE(frame && motion && ddst && dframe, "backward_warp_bwd_f32: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && h >= 2 && w >= 2, "backward_warp_bwd_f32: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(dframe, 0, (size_t)n * c * h * w * sizeof(float), s) != hipSuccess) {
        srhip_set_error("backward_warp_bwd_f32: memset failed");
        return SRHIP_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(sample_bwd_f32_kernel<1>, dim3(grid_for((long long)n * h * w, srhip_num_cus())), dim3(kThreads), 0, s,
                       frame, motion, ddst, dframe, dmotion, n, c, h, w, h, w, align_corners ? 1 : 0, CoordAxis{0, 0},
                       CoordAxis{0, 0});
    return srhip_check_launch("backward_warp_bwd_f32");
}

SRHIP_API int srhip_pixel_shuffle(const void* src, void* dst, int n, int c, int h, int w, int r, int elem_size,
                                  void* stream)
{
    SRHIP_REQUIRE(src && dst, "pixel_shuffle: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && h >= 1 && w >= 1 && r >= 1, "pixel_shuffle: bad shape");
    SRHIP_REQUIRE((long long)h * r <= 65535 && (long long)n * c <= 65535, "pixel_shuffle: at most 65535 output rows and 65535 planes per call");
    const dim3 grid(srhip_div_up(w, kThreads), h * r, n * c);
    const int vec = ((size_t)dst % 16 == 0) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (elem_size == 4) hipLaunchKernelGGL((pixel_shuffle_kernel<unsigned, false>), grid, dim3(kThreads), 0, s, (const unsigned*)src, (unsigned*)dst, c, h, w, r, vec);
    else if (elem_size == 2) hipLaunchKernelGGL((pixel_shuffle_kernel<unsigned short, false>), grid, dim3(kThreads), 0, s, (const unsigned short*)src, (unsigned short*)dst, c, h, w, r, vec);
    else { srhip_set_error("pixel_shuffle: elem_size must be 2 or 4"); return SRHIP_ERR_INVALID; }
    return srhip_check_launch("pixel_shuffle");
}

SRHIP_API int srhip_pixel_unshuffle(const void* src, void* dst, int n, int c, int h, int w, int r, int elem_size,
                                  void* stream)
{
    SRHIP_REQUIRE(src && dst, "pixel_unshuffle: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && h >= 1 && w >= 1 && r >= 1, "pixel_unshuffle: bad shape");
    SRHIP_REQUIRE((long long)h * r <= 65535 && (long long)n * c <= 65535, "pixel_unshuffle: at most 65535 output rows and 65535 planes per call");
    const dim3 grid(srhip_div_up(w, kThreads), h * r, n * c);
    const int vec = ((size_t)src % 16 == 0) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (elem_size == 4) hipLaunchKernelGGL((pixel_shuffle_kernel<unsigned, true>), grid, dim3(kThreads), 0, s, (const unsigned*)src, (unsigned*)dst, c, h, w, r, vec);
    else if (elem_size == 2) hipLaunchKernelGGL((pixel_shuffle_kernel<unsigned short, true>), grid, dim3(kThreads), 0, s, (const unsigned short*)src, (unsigned short*)dst, c, h, w, r, vec);
    else { srhip_set_error("pixel_unshuffle: elem_size must be 2 or 4"); return SRHIP_ERR_INVALID; }
    return srhip_check_launch("pixel_unshuffle");
}

SRHIP_API int srhip_pixel_shuffle_nhwc_to_nchw_f32(const void* src, float* dst, int n, int c, int h, int w, int r,
                                                   int cstride, void* stream)
{
    SRHIP_REQUIRE(src && dst, "pixel_shuffle_nhwc_to_nchw_f32: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && h >= 1 && w >= 1 && r >= 1 && cstride >= c * r * r, "pixel_shuffle_nhwc_to_nchw_f32: bad shape");
    SRHIP_REQUIRE((long long)h * r <= 65535 && (long long)n * c <= 65535, "pixel_shuffle_nhwc_to_nchw_f32: at most 65535 output rows and 65535 planes per call");
    const int vec = ((size_t)src % 8 == 0 && (size_t)dst % 16 == 0 && cstride % 4 == 0) ? 1 : 0;
    if (r == 4 && (size_t)src % 16 == 0 && (size_t)dst % 16 == 0 && cstride % 8 == 0 && h <= 65535 && n <= 65535) {
        hipLaunchKernelGGL(pixel_shuffle4_nhwc_to_nchw_f32_kernel, dim3(srhip_div_up(w, 64), h, n), dim3(64), 0,
                           (hipStream_t)stream, (const unsigned short*)src, dst, c, h, w, cstride);
        return srhip_check_launch("pixel_shuffle4_nhwc_to_nchw_f32");
    }
    hipLaunchKernelGGL(pixel_shuffle_nhwc_to_nchw_f32_kernel, dim3(srhip_div_up(w, kThreads), h * r, n * c), dim3(kThreads), 0,
                       (hipStream_t)stream, (const unsigned short*)src, dst, c, h, w, r, cstride, vec);
    return srhip_check_launch("pixel_shuffle_nhwc_to_nchw_f32");
}

SRHIP_API int srhip_nearest_down(const void* src, void* dst, int nc, int h, int w, int ho, int wo, double scale_factor,
                                 int elem_size, void* stream)
{
    SRHIP_REQUIRE(src && dst, "nearest_down: null pointer");
    SRHIP_REQUIRE(nc >= 1 && h >= 1 && w >= 1 && ho >= 1 && wo >= 1 && scale_factor > 0.0, "nearest_down: bad shape");
    const float inv = (float)(1.0 / scale_factor);
    SRHIP_REQUIRE(ho <= 65535 && nc <= 65535, "nearest_down: at most 65535 output rows and planes per call");
    const dim3 g(srhip_div_up(wo, kThreads), ho, nc);
    hipStream_t s = (hipStream_t)stream;
    if (elem_size == 4) hipLaunchKernelGGL(nearest_down_kernel<unsigned>, g, dim3(kThreads), 0, s, (const unsigned*)src, (unsigned*)dst, h, w, ho, wo, inv);
    else if (elem_size == 2) hipLaunchKernelGGL(nearest_down_kernel<unsigned short>, g, dim3(kThreads), 0, s, (const unsigned short*)src, (unsigned short*)dst, h, w, ho, wo, inv);
    else { srhip_set_error("nearest_down: elem_size must be 2 or 4"); return SRHIP_ERR_INVALID; }
    return srhip_check_launch("nearest_down");
}

SRHIP_API int srhip_rel_coord(float* dst, int th, int tw, int sh, int sw, void* stream)
{
    SRHIP_REQUIRE(dst, "rel_coord: null pointer");
    SRHIP_REQUIRE(th >= 1 && tw >= 1 && sh >= 1 && sw >= 1, "rel_coord: bad shape");
    hipLaunchKernelGGL(rel_coord_kernel, dim3(grid_for((long long)th * tw, srhip_num_cus())), dim3(kThreads), 0,
                       (hipStream_t)stream, dst, th, tw, sh, sw, coord_axis(tw), coord_axis(th), coord_axis(sw), coord_axis(sh));
    return srhip_check_launch("rel_coord");
}

SRHIP_API int srhip_make_coord(float* dst, int h, int w, void* stream)
{
    SRHIP_REQUIRE(dst, "make_coord: null pointer");
    SRHIP_REQUIRE(h >= 1 && w >= 1, "make_coord: bad shape");
    hipLaunchKernelGGL(make_coord_kernel, dim3(grid_for((long long)h * w, srhip_num_cus())), dim3(kThreads), 0,
                       (hipStream_t)stream, dst, h, w, coord_axis(w), coord_axis(h));
    return srhip_check_launch("make_coord");
}

SRHIP_API int srhip_upsample_grid_bilinear_bwd_nhwc(const void* ddst, const void* relu_out, void* dsrc, int n, int c,
                                                    int sh, int sw, int th, int tw, void* stream)
{
    SRHIP_REQUIRE(ddst && dsrc, "upsample_grid_bilinear_bwd_nhwc: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 8 && c % 8 == 0 && sh >= 1 && sw >= 1 && th >= 1 && tw >= 1, "upsample_grid_bilinear_bwd_nhwc: bad shape (c must be a multiple of 8)");
    const long long work = (long long)n * sh * sw * (c / 8);
    if (th == 4 * sh && tw == 4 * sw && getenv("SRHIP_NO_UPBWD_INT") == nullptr) {   // the network's ratio: 8 + 8 tap weights per thread
        SRHIP_REQUIRE(work < (1ll << 38), "upsample_grid_bilinear_bwd_nhwc: too many work items");
        hipLaunchKernelGGL(upsample_bwd_int_nhwc_kernel<4>, dim3((unsigned)(((work + kThreads - 1) / kThreads + 7) / 8 * 8)), dim3(kThreads), 0, (hipStream_t)stream,
                           (const unsigned short*)ddst, (const unsigned short*)relu_out, (unsigned short*)dsrc, n, c, sh, sw,
                           coord_axis(tw), coord_axis(th));
        return srhip_check_launch("upsample_bwd_int_nhwc");
    }
    hipLaunchKernelGGL(upsample_bwd_nhwc_kernel, dim3(grid_for(work, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const unsigned short*)ddst, (const unsigned short*)relu_out, (unsigned short*)dsrc, n, c, sh, sw, th, tw,
                       coord_axis(tw), coord_axis(th));
    return srhip_check_launch("upsample_bwd_nhwc");
}

SRHIP_API int srhip_relu_mask_bf16(const void* dy, const void* y, void* dst, size_t count, void* stream)
{
    SRHIP_REQUIRE(dy && y && dst, "relu_mask_bf16: null pointer");
    SRHIP_REQUIRE(count % 8 == 0, "relu_mask_bf16: count must be a multiple of 8");
    if (count == 0) return SRHIP_OK;
    const long long nvec = (long long)(count / 8);
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid_for(nvec, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4*)dy, (const uint4*)y, (uint4*)dst, nvec);
    return srhip_check_launch("relu_mask_bf16");
}

SRHIP_API int srhip_signbits_from_map_bf16(const void* map, void* bits, int n, int h, int w, void* stream)
{
    SRHIP_REQUIRE(map && bits, "signbits_from_map_bf16: null pointer");
    SRHIP_REQUIRE(n >= 1 && h >= 1 && w >= 1, "signbits_from_map_bf16: empty tensor");
    SRHIP_REQUIRE((size_t)map % 16 == 0 && (size_t)bits % 4 == 0, "signbits_from_map_bf16: misaligned pointer");
    const int groups = (w + 31) / 32;
    const long long rows = (long long)n * h;
    hipLaunchKernelGGL(signbits_from_map_kernel, dim3(grid_for(rows * groups * 64, srhip_num_cus())), dim3(kThreads), 0,
                       (hipStream_t)stream, (const unsigned short*)map, (unsigned*)bits, w, groups, rows);
    return srhip_check_launch("signbits_from_map_bf16");
}

SRHIP_API int srhip_add_bf16(const void* a, const void* b, void* dst, size_t count, void* stream)
{
    SRHIP_REQUIRE(a && b && dst, "add_bf16: null pointer");
    SRHIP_REQUIRE(count % 8 == 0, "add_bf16: count must be a multiple of 8");
    if (count == 0) return SRHIP_OK;
    const long long nvec = (long long)(count / 8);
    hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(nvec, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4*)a, (const uint4*)b, (uint4*)dst, nvec);
    return srhip_check_launch("add_bf16");
}

SRHIP_API int srhip_scale_add_bf16(const void* a, const void* y, const float* scale, void* dst, size_t count, void* stream)
{
    SRHIP_REQUIRE(a && y && scale && dst, "scale_add_bf16: null pointer");
    SRHIP_REQUIRE(count % 8 == 0, "scale_add_bf16: count must be a multiple of 8");
    SRHIP_REQUIRE((size_t)a % 16 == 0 && (size_t)y % 16 == 0 && (size_t)dst % 16 == 0, "scale_add_bf16: maps must be 16-byte aligned");
    if (count == 0) return SRHIP_OK;
    const long long nvec = (long long)(count / 8);
    hipLaunchKernelGGL(scale_add_bf16_kernel, dim3(grid_for(nvec, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4*)a, (const uint4*)y, scale, (uint4*)dst, nvec);
    return srhip_check_launch("scale_add_bf16");
}

SRHIP_API int srhip_scale_add_bwd_bf16(const void* dout, const void* y, const float* scale, void* dy, float* partials, int npartials,
                                       size_t count, void* stream)
{
    SRHIP_REQUIRE(dout && y && scale && dy && partials, "scale_add_bwd_bf16: null pointer");
    SRHIP_REQUIRE(count % 8 == 0, "scale_add_bwd_bf16: count must be a multiple of 8");
    SRHIP_REQUIRE(npartials >= 1 && npartials <= 65536, "scale_add_bwd_bf16: 1 to 65536 partial sums (got %d)", npartials);
    SRHIP_REQUIRE((size_t)dout % 16 == 0 && (size_t)y % 16 == 0 && (size_t)dy % 16 == 0, "scale_add_bwd_bf16: maps must be 16-byte aligned");
    // one partial per work-group, every work-group writes its own (zero when it has no element)
    hipLaunchKernelGGL(scale_add_bwd_bf16_kernel, dim3(npartials), dim3(kThreads), 0, (hipStream_t)stream, (const uint4*)dout,
                       (const uint4*)y, scale, (uint4*)dy, partials, (long long)(count / 8));
    return srhip_check_launch("scale_add_bwd_bf16");
}

SRHIP_API int srhip_max_pool2d_nhwc(const void* src, void* dst, int n, int c, int h, int w, int k, int s, void* stream)
{
    SRHIP_REQUIRE(src && dst, "max_pool2d_nhwc: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 8 && c % 8 == 0 && k >= 1 && s >= 1 && h >= k && w >= k, "max_pool2d_nhwc: bad shape (c %% 8 == 0, window inside the image)");
    const int ho = (h - k) / s + 1, wo = (w - k) / s + 1;
    const long long work = (long long)n * ho * wo * (c / 8);
    if (k == 7) {   // the ESA block's window: eight lanes per output
        hipLaunchKernelGGL(max_pool_rows_nhwc_kernel<7>, dim3(grid_for(work * 8, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                           (const unsigned short*)src, (unsigned short*)dst, n, c, h, w, ho, wo, s);
        return srhip_check_launch("max_pool2d_nhwc");
    }
    hipLaunchKernelGGL(max_pool_nhwc_kernel, dim3(grid_for(work, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const unsigned short*)src, (unsigned short*)dst, n, c, h, w, ho, wo, k, s);
    return srhip_check_launch("max_pool2d_nhwc");
}

SRHIP_API int srhip_resize_bilinear_nhwc(const void* src, const void* addend, void* dst, int n, int c, int h, int w, int ho,
                                         int wo, int align_corners, void* stream)
{
    SRHIP_REQUIRE(src && dst, "resize_bilinear_nhwc: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 8 && c % 8 == 0 && h >= 1 && w >= 1 && ho >= 1 && wo >= 1, "resize_bilinear_nhwc: bad shape (c %% 8 == 0)");
    const long long work = (long long)n * ho * wo * (c / 8);
    hipLaunchKernelGGL(resize_bilinear_nhwc_kernel, dim3(grid_for(work, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const unsigned short*)src, (const unsigned short*)addend, (unsigned short*)dst, n, c, h, w, ho, wo, align_corners);
    return srhip_check_launch("resize_bilinear_nhwc");
}

SRHIP_API int srhip_esa_tail_bf16(const void* x, const void* c1, const void* c3, const float* wf, const float* bf, const float* w4,
                                  const float* b4, void* dst, int n, int h, int w, int h3, int w3, int cfeat, int cstride, void* stream)
{
    SRHIP_REQUIRE(x && c1 && c3 && wf && bf && w4 && b4 && dst, "esa_tail_bf16: null pointer");
    SRHIP_REQUIRE(n >= 1 && h >= 1 && w >= 1 && h3 >= 1 && w3 >= 1, "esa_tail_bf16: empty tensor");
    SRHIP_REQUIRE(cfeat == 16 || cfeat == 32 || cfeat == 48 || cfeat == 64, "esa_tail_bf16: %d feature channels (16, 32, 48 or 64)", cfeat);
    SRHIP_REQUIRE(cstride == cfeat || cstride == 64, "esa_tail_bf16: channel stride %d (the %d feature channels, or 64)", cstride, cfeat);
    SRHIP_REQUIRE((size_t)x % 16 == 0 && (size_t)c1 % 16 == 0 && (size_t)c3 % 16 == 0 && (size_t)dst % 16 == 0, "esa_tail_bf16: maps must be 16-byte aligned");
    SRHIP_REQUIRE((long long)n * h * w < (1ll << 31), "esa_tail_bf16: too many pixels for 32-bit indices");
    const long long work = (long long)n * h * w * 2;   // a wave covers 32 pixels per step
    const dim3 grid(grid_for(work, srhip_num_cus())), block(kThreads);
    hipStream_t s = (hipStream_t)stream;
#define SRHIP_ESA_TAIL(CF)                                                                                                     \
    if (cstride == CF)                                                                                                         \
        hipLaunchKernelGGL((esa_tail_kernel<CF, CF>), grid, block, 0, s, (const unsigned short*)x, (const unsigned short*)c1,  \
                           (const unsigned short*)c3, wf, bf, w4, b4, (unsigned short*)dst, n, h, w, h3, w3);                  \
    else                                                                                                                       \
        hipLaunchKernelGGL((esa_tail_kernel<CF, 64>), grid, block, 0, s, (const unsigned short*)x, (const unsigned short*)c1,  \
                           (const unsigned short*)c3, wf, bf, w4, b4, (unsigned short*)dst, n, h, w, h3, w3)
    switch (cfeat) {
    case 16: SRHIP_ESA_TAIL(16); break;
    case 32: SRHIP_ESA_TAIL(32); break;
    case 48: SRHIP_ESA_TAIL(48); break;
    default: SRHIP_ESA_TAIL(64); break;
    }
#undef SRHIP_ESA_TAIL
    return srhip_check_launch("esa_tail_bf16");
}

SRHIP_API int srhip_sigmoid_gate_bf16(const void* x, const void* gate, void* dst, size_t count, void* stream)
{
    SRHIP_REQUIRE(x && gate && dst, "sigmoid_gate_bf16: null pointer");
    SRHIP_REQUIRE(count % 8 == 0, "sigmoid_gate_bf16: count must be a multiple of 8");
    if (count == 0) return SRHIP_OK;
    const long long nvec = (long long)(count / 8);
    hipLaunchKernelGGL(sigmoid_gate_kernel, dim3(grid_for(nvec, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4*)x, (const uint4*)gate, (uint4*)dst, nvec);
    return srhip_check_launch("sigmoid_gate_bf16");
}

SRHIP_API int srhip_attention_fuse_bf16(const void* cur, const void* fmap, const void* grad, const void* f_self, const void* g_cross,
                                        const void* g_self, const void* f_cross, int gate_stride, void* dst, size_t pixels, int c,
                                        void* stream)
{
    SRHIP_REQUIRE(cur && fmap && grad && f_self && g_cross && g_self && f_cross && dst, "attention_fuse_bf16: null pointer");
    SRHIP_REQUIRE(c >= 8 && c % 8 == 0 && gate_stride >= 1, "attention_fuse_bf16: c must be a multiple of 8, gate_stride >= 1");
    if (pixels == 0) return SRHIP_OK;
    const long long nvec = (long long)pixels * (c / 8);
    hipLaunchKernelGGL(attention_fuse_kernel, dim3(grid_for(nvec, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4*)cur, (const uint4*)fmap, (const uint4*)grad, (const unsigned short*)f_self,
                       (const unsigned short*)g_cross, (const unsigned short*)g_self, (const unsigned short*)f_cross, gate_stride,
                       (uint4*)dst, nvec, c / 8);
    return srhip_check_launch("attention_fuse_bf16");
}

SRHIP_API int srhip_attention_fuse_bwd_bf16(const void* d, const void* fmap, const void* grad, const void* f_self, const void* g_cross,
                                            const void* g_self, const void* f_cross, int gate_stride, void* dfmap, void* dgrad,
                                            void* d_f_self, void* d_g_cross, void* d_g_self, void* d_f_cross, int dgate_stride,
                                            size_t pixels, int c, void* stream)
{
    SRHIP_REQUIRE(d && fmap && grad && f_self && g_cross && g_self && f_cross && dfmap && dgrad && d_f_self && d_g_cross && d_g_self && d_f_cross,
                  "attention_fuse_bwd_bf16: null pointer");
    SRHIP_REQUIRE(c == 8 || c == 16 || c == 32 || c == 64, "attention_fuse_bwd_bf16: c must be 8, 16, 32 or 64 (got %d)", c);
    SRHIP_REQUIRE(gate_stride >= 1 && dgate_stride >= 8 && dgate_stride % 8 == 0, "attention_fuse_bwd_bf16: gate_stride >= 1, dgate_stride a multiple of 8");
    if (pixels == 0) return SRHIP_OK;
    const long long nvec = (long long)pixels * (c / 8);
    const dim3 grid(grid_for((nvec + 63) / 64 * 64, srhip_num_cus())), block(kThreads);
#define SRHIP_AFB_CASE(L) \
    if (c == 8 * L) hipLaunchKernelGGL(attention_fuse_bwd_kernel<L>, grid, block, 0, (hipStream_t)stream, (const uint4*)d, (const uint4*)fmap, \
        (const uint4*)grad, (const unsigned short*)f_self, (const unsigned short*)g_cross, (const unsigned short*)g_self, \
        (const unsigned short*)f_cross, gate_stride, (uint4*)dfmap, (uint4*)dgrad, (uint4*)d_f_self, (uint4*)d_g_cross, (uint4*)d_g_self, \
        (uint4*)d_f_cross, dgate_stride / 8, nvec);
    SRHIP_AFB_CASE(1) SRHIP_AFB_CASE(2) SRHIP_AFB_CASE(4) SRHIP_AFB_CASE(8)
#undef SRHIP_AFB_CASE
    return srhip_check_launch("attention_fuse_bwd_bf16");
}

SRHIP_API int srhip_lrelu_mask_bf16(const void* dy, const void* y, void* dst, size_t count, float slope, void* stream)
{
    SRHIP_REQUIRE(dy && y && dst, "lrelu_mask_bf16: null pointer");
    SRHIP_REQUIRE(count % 8 == 0, "lrelu_mask_bf16: count must be a multiple of 8");
    SRHIP_REQUIRE(slope >= 0.0f, "lrelu_mask_bf16: the mask is read from the activation's output, which needs slope >= 0");
    if (count == 0) return SRHIP_OK;
    const long long nvec = (long long)(count / 8);
    hipLaunchKernelGGL(lrelu_mask_kernel, dim3(grid_for(nvec, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4*)dy, (const uint4*)y, (uint4*)dst, nvec, slope);
    return srhip_check_launch("lrelu_mask_bf16");
}

SRHIP_API int srhip_sigmoid_gate_bwd_bf16(const void* x, const void* gate, const void* dy, void* dx, void* dgate, size_t count,
                                          void* stream)
{
    SRHIP_REQUIRE(x && gate && dy && dx && dgate, "sigmoid_gate_bwd_bf16: null pointer");
    SRHIP_REQUIRE(count % 8 == 0, "sigmoid_gate_bwd_bf16: count must be a multiple of 8");
    if (count == 0) return SRHIP_OK;
    const long long nvec = (long long)(count / 8);
    hipLaunchKernelGGL(sigmoid_gate_bwd_kernel, dim3(grid_for(nvec, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4*)x, (const uint4*)gate, (const uint4*)dy, (uint4*)dx, (uint4*)dgate, nvec);
    return srhip_check_launch("sigmoid_gate_bwd_bf16");
}

SRHIP_API int srhip_max_pool2d_bwd_nhwc(const void* src, const void* dy, void* dx, int n, int c, int h, int w, int k, int s,
                                        void* stream)
{
    SRHIP_REQUIRE(src && dy && dx, "max_pool2d_bwd_nhwc: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 8 && c % 8 == 0 && k >= 1 && s >= 1 && h >= k && w >= k, "max_pool2d_bwd_nhwc: bad shape");
    const int ho = (h - k) / s + 1, wo = (w - k) / s + 1;
    const long long work = (long long)n * h * w * (c / 8);
    hipLaunchKernelGGL(max_pool_bwd_nhwc_kernel, dim3(grid_for(work, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const unsigned short*)src, (const unsigned short*)dy, (unsigned short*)dx, n, c, h, w, ho, wo, k, s);
    return srhip_check_launch("max_pool2d_bwd_nhwc");
}

SRHIP_API int srhip_max_pool2d_arg_nhwc(const void* src, void* dst, void* arg, int n, int c, int h, int w, int k, int s, void* stream)
{
    SRHIP_REQUIRE(src && dst && arg, "max_pool2d_arg_nhwc: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 8 && c % 8 == 0 && k >= 1 && k <= 15 && s >= 1 && h >= k && w >= k,
                  "max_pool2d_arg_nhwc: bad shape (c %% 8 == 0, k <= 15, window inside the image)");
    const int ho = (h - k) / s + 1, wo = (w - k) / s + 1;
    const long long work = (long long)n * ho * wo * (c / 8);
    hipLaunchKernelGGL(max_pool_arg_nhwc_kernel, dim3(grid_for(work, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const unsigned short*)src, (unsigned short*)dst, (unsigned char*)arg, n, c, h, w, ho, wo, k, s);
    return srhip_check_launch("max_pool2d_arg_nhwc");
}

SRHIP_API int srhip_max_pool2d_bwd_arg_nhwc(const void* arg, const void* dy, void* dx, int n, int c, int h, int w, int k, int s,
                                            void* stream)
{
    SRHIP_REQUIRE(arg && dy && dx, "max_pool2d_bwd_arg_nhwc: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 8 && c % 8 == 0 && k >= 1 && k <= 15 && s >= 1 && h >= k && w >= k, "max_pool2d_bwd_arg_nhwc: bad shape");
    const int ho = (h - k) / s + 1, wo = (w - k) / s + 1;
    const long long work = (long long)n * h * w * (c / 8);
    hipLaunchKernelGGL(max_pool_bwd_arg_nhwc_kernel, dim3(grid_for(work, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const unsigned char*)arg, (const unsigned short*)dy, (unsigned short*)dx, n, c, h, w, ho, wo, k, s);
    return srhip_check_launch("max_pool2d_bwd_arg_nhwc");
}

SRHIP_API int srhip_resize_bilinear_bwd_nhwc(const void* dy, void* dx, int n, int c, int h, int w, int ho, int wo,
                                             int align_corners, void* stream)
{
    SRHIP_REQUIRE(dy && dx, "resize_bilinear_bwd_nhwc: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 8 && c % 8 == 0 && h >= 1 && w >= 1 && ho >= 1 && wo >= 1, "resize_bilinear_bwd_nhwc: bad shape");
    const long long work = (long long)n * h * w * (c / 8);
    hipLaunchKernelGGL(resize_bilinear_bwd_nhwc_kernel, dim3(grid_for(work, srhip_num_cus())), dim3(kThreads), 0,
                       (hipStream_t)stream, (const unsigned short*)dy, (unsigned short*)dx, n, c, h, w, ho, wo, align_corners ? 1 : 0);
    return srhip_check_launch("resize_bilinear_bwd_nhwc");
}

SRHIP_API int srhip_gradient_map_f32(const float* src, float* dst, int planes, int h, int w, float eps, void* stream)
{
    SRHIP_REQUIRE(src && dst, "gradient_map_f32: null pointer");
    SRHIP_REQUIRE(planes >= 1 && h >= 1 && w >= 1, "gradient_map_f32: bad shape");
    const bool grid3 = h <= 65535 && planes <= 65535 && (long long)h * w < (1ll << 31);
    if (grid3 && w % 4 == 0 && ((size_t)src | (size_t)dst) % 16 == 0)
        hipLaunchKernelGGL(gradient_map_rows4_kernel, dim3(srhip_div_up(w / 4, kThreads), h, planes), dim3(kThreads), 0,
                           (hipStream_t)stream, src, dst, h, w, eps);
    else if (grid3)
        hipLaunchKernelGGL(gradient_map_rows_kernel, dim3(srhip_div_up(w, kThreads), h, planes), dim3(kThreads), 0,
                           (hipStream_t)stream, src, dst, h, w, eps);
    else
        hipLaunchKernelGGL(gradient_map_kernel, dim3(grid_for((long long)planes * h * w, srhip_num_cus())), dim3(kThreads), 0,
                           (hipStream_t)stream, src, dst, (long long)planes, h, w, eps);
    return srhip_check_launch("gradient_map_f32");
}

SRHIP_API int srhip_gradient_map_unshuffle4_nhwc_bf16(const float* src, int src_planes, const int* planes, int nplanes, void* dst, int n,
                                                      int h, int w, float eps, void* stream)
{
    SRHIP_REQUIRE(src && dst && planes, "gradient_map_unshuffle4_nhwc_bf16: null pointer");
    SRHIP_REQUIRE(n >= 1 && h >= 1 && w >= 1 && src_planes >= 1, "gradient_map_unshuffle4_nhwc_bf16: bad shape");
    SRHIP_REQUIRE(nplanes >= 1 && nplanes <= 23, "gradient_map_unshuffle4_nhwc_bf16: 1 to 23 planes (got %d)", nplanes);
    SRHIP_REQUIRE((size_t)src % 16 == 0 && (size_t)dst % 16 == 0, "gradient_map_unshuffle4_nhwc_bf16: 16-byte aligned tensors");
    SRHIP_REQUIRE((long long)h * w * 16 < (1ll << 31), "gradient_map_unshuffle4_nhwc_bf16: at most 2^31 pixels per plane");
    GradPlanes gp;
    for (int k = 0; k < 24; ++k) gp.idx[k] = 0;
    for (int k = 0; k < nplanes; ++k) {
        SRHIP_REQUIRE(planes[k] >= 0 && planes[k] < src_planes, "gradient_map_unshuffle4_nhwc_bf16: plane index %d outside the tensor", planes[k]);
        gp.idx[k] = planes[k];
    }
    const size_t lds = (size_t)kLayoutTile * (nplanes * 4 + 2) * sizeof(uint2);
    const long long ntiles = ((long long)n * h * w + kLayoutTile - 1) / kLayoutTile;
    const long long cap = (long long)srhip_num_cus() * 16;
    hipLaunchKernelGGL(gradient_map_unshuffle4_nhwc_bf16_kernel, dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(256), lds,
                       (hipStream_t)stream, src, (unsigned short*)dst, n, src_planes, gp, nplanes, h, w, eps);
    return srhip_check_launch("gradient_map_unshuffle4_nhwc_bf16");
}

SRHIP_API int srhip_pixel_shuffle4_conv3x3_c3_nchw_f32(const void* src, const float* weight, const float* bias, float* dst, int n,
                                                       int h, int w, int cstride, int act, float slope, void* stream)
{
    SRHIP_REQUIRE(src && weight && dst, "pixel_shuffle4_conv3x3_c3_nchw_f32: null pointer");
    SRHIP_REQUIRE(n >= 1 && h >= 1 && w >= 1 && cstride >= 48 && cstride % 8 == 0,
                  "pixel_shuffle4_conv3x3_c3_nchw_f32: input pixels of at least 48 channels (a multiple of 8)");
    SRHIP_REQUIRE((size_t)src % 16 == 0 && (size_t)dst % 16 == 0, "pixel_shuffle4_conv3x3_c3_nchw_f32: 16-byte aligned tensors");
    SRHIP_REQUIRE(act >= 0 && act <= 2, "pixel_shuffle4_conv3x3_c3_nchw_f32: act must be 0 (none), 1 (ReLU) or 2 (LeakyReLU)");
    const float s = act == 0 ? 1.0f : (act == 1 ? 0.0f : slope);
    const long long ntiles = (long long)n * srhip_div_up(h, kTailLY) * srhip_div_up(w, kTailLX);
    const long long cap = (long long)srhip_num_cus() * 8;
    hipLaunchKernelGGL(pixel_shuffle4_conv3x3_c3_kernel, dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)src, weight, bias, dst, n, h, w, cstride, s);
    return srhip_check_launch("pixel_shuffle4_conv3x3_c3_nchw_f32");
}

SRHIP_API int srhip_zero_insert2_nhwc_bf16(const void* src, void* dst, int n, int ho, int wo, int h, int w, int c, void* stream)
{
    SRHIP_REQUIRE(src && dst, "zero_insert2_nhwc_bf16: null pointer");
    SRHIP_REQUIRE(n >= 1 && ho >= 1 && wo >= 1 && c >= 8 && c % 8 == 0, "zero_insert2_nhwc_bf16: bad shape (channels in multiples of 8)");
    SRHIP_REQUIRE((h + 1) / 2 == ho && (w + 1) / 2 == wo, "zero_insert2_nhwc_bf16: (%d x %d) is not the even positions of (%d x %d)", ho, wo, h, w);
    SRHIP_REQUIRE(((size_t)src | (size_t)dst) % 16 == 0, "zero_insert2_nhwc_bf16: aligned tensors");
    const long long total = (long long)n * h * w * (c / 8);
    hipLaunchKernelGGL(zero_insert2_nhwc_kernel, dim3(grid_for(total, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4*)src, (uint4*)dst, ho, wo, h, w, c / 8, total);
    return srhip_check_launch("zero_insert2_nhwc_bf16");
}

SRHIP_API int srhip_pixel_shuffle4_nhwc16_bwd_bf16(const void* d, void* dt, int n, int c, int h, int w, int cstride, void* stream)
{
    SRHIP_REQUIRE(d && dt, "pixel_shuffle4_nhwc16_bwd_bf16: null pointer");
    SRHIP_REQUIRE(n >= 1 && h >= 1 && w >= 1, "pixel_shuffle4_nhwc16_bwd_bf16: bad shape");
    SRHIP_REQUIRE(c >= 1 && c <= 4 && cstride >= 16 * c && cstride % 8 == 0,
                  "pixel_shuffle4_nhwc16_bwd_bf16: 1 to 4 colours, feature pixels of at least 16 c channels (a multiple of 8)");
    SRHIP_REQUIRE((size_t)dt % 16 == 0, "pixel_shuffle4_nhwc16_bwd_bf16: aligned tensors");
    const long long work = (long long)n * h * w * (cstride / 8);
    hipLaunchKernelGGL(pixel_shuffle4_nhwc16_bwd_kernel, dim3(grid_for(work, srhip_num_cus())), dim3(kThreads), 0,
                       (hipStream_t)stream, (const unsigned short*)d, (uint4*)dt, n, c, h, w, cstride);
    return srhip_check_launch("pixel_shuffle4_nhwc16_bwd_bf16");
}

SRHIP_API int srhip_pixel_shuffle4_nhwc_bf16_to_nhwc16(const void* src, void* dst, int n, int c, int h, int w, int cstride, void* stream)
{
    SRHIP_REQUIRE(src && dst, "pixel_shuffle4_nhwc_bf16_to_nhwc16: null pointer");
    SRHIP_REQUIRE(n >= 1 && h >= 1 && w >= 1, "pixel_shuffle4_nhwc_bf16_to_nhwc16: bad shape");
    SRHIP_REQUIRE(c >= 1 && c <= 4 && cstride >= 16 * c && cstride % 4 == 0,
                  "pixel_shuffle4_nhwc_bf16_to_nhwc16: 1 to 4 output channels, input pixels of at least 16 c channels");
    SRHIP_REQUIRE((size_t)src % 8 == 0 && (size_t)dst % 16 == 0, "pixel_shuffle4_nhwc_bf16_to_nhwc16: aligned tensors");
    const long long work = (long long)n * h * 4 * w * 4 * 2;
    hipLaunchKernelGGL(pixel_shuffle4_nhwc_bf16_to_nhwc16_kernel, dim3(grid_for(work, srhip_num_cus())), dim3(kThreads), 0,
                       (hipStream_t)stream, (const unsigned short*)src, (uint4*)dst, n, c, h, w, cstride);
    return srhip_check_launch("pixel_shuffle4_nhwc_bf16_to_nhwc16");
}

SRHIP_API int srhip_dilate_velocity_f32(const float* motion, const float* depth, float* dst, int b, int h, int w, int ksize,
                                        void* stream)
{
    SRHIP_REQUIRE(motion && depth && dst, "dilate_velocity_f32: null pointer");
    SRHIP_REQUIRE(b >= 1 && b <= 65535 && h >= 1 && h <= 65535 && w >= 1 && (long long)h * w < (1ll << 30),
                  "dilate_velocity_f32: bad shape (at most 65535 rows and samples)");
    SRHIP_REQUIRE(ksize >= 1 && ksize % 2 == 1 && ksize <= 15, "dilate_velocity_f32: ksize must be odd, 1..15");
    SRHIP_REQUIRE(((size_t)motion | (size_t)dst) % 8 == 0, "dilate_velocity_f32: motion and dst must be 8-byte aligned");
    hipLaunchKernelGGL(dilate_velocity_kernel, dim3(srhip_div_up(w, kThreads), h, b), dim3(kThreads), 0, (hipStream_t)stream,
                       (const float2*)motion, depth, (float2*)dst, h, w, ksize);
    return srhip_check_launch("dilate_velocity_f32");
}

SRHIP_API int srhip_concat_nhwc(const void* a, const void* b, void* dst, size_t pixels, int ca, int cb, void* stream)
{
    SRHIP_REQUIRE(a && b && dst, "concat_nhwc: null pointer");
    SRHIP_REQUIRE(ca >= 8 && cb >= 8 && ca % 8 == 0 && cb % 8 == 0, "concat_nhwc: channels must be multiples of 8");
    const long long work = (long long)pixels * ((ca + cb) / 8);
    hipLaunchKernelGGL(concat_nhwc_kernel, dim3(grid_for(work, srhip_num_cus())), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4*)a, (const uint4*)b, (uint4*)dst, (long long)pixels, ca / 8, cb / 8);
    return srhip_check_launch("concat_nhwc");
}

SRHIP_API int srhip_layernorm_cf_f32(const float* x, const float* weight, const float* bias, float* y, int n, int c, int hw,
                                     float eps, void* stream)
{
    SRHIP_REQUIRE(x && weight && bias && y, "layernorm_cf_f32: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && hw >= 1, "layernorm_cf_f32: bad shape");
    if (c <= kLnRegC && !ln_force_loops())
        hipLaunchKernelGGL(layernorm_cf_reg_kernel, dim3((unsigned)(((long long)n * hw + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, x, weight, bias, y, n, c, (long long)hw, eps);
    else
        hipLaunchKernelGGL(layernorm_cf_kernel, dim3(grid_for((long long)n * hw, srhip_num_cus())), dim3(kThreads), 0,
                           (hipStream_t)stream, x, weight, bias, y, n, c, (long long)hw, eps);
    return srhip_check_launch("layernorm_cf_f32");
}

}  // extern "C"
