// Fused L1 + SSIM loss, forward and backward, for gfx950.
//
// Replaces LossGroup.forward + loss.backward() of the default trainer configuration
// (core/models/model.py:183-207; SSIMLoss core/modules/loss.py:12-29; _ssim
// core/modules/functional.py:169-189) together with the pixel clip in front of it
// (core/trainer.py:154-156):
//
//   xc   = clamp(sr, clip_min, clip_max)
//   loss = lambda_l1 * mean|xc - y| + lambda_ssim * (1 - mean(ssim_map(xc, y)))
//
// ssim_map uses five 11x11 Gaussian (sigma 1.5) depthwise filters with zero padding:
// mu_x, mu_y, E[x^2], E[y^2], E[xy].  The window is an outer product, so the filters run as a
// horizontal and a vertical 11-tap pass through LDS.  Because the loss is a mean, dL/d(ssim_map)
// and dL/d|xc-y| are constants known up front, so the backward needs no second look at the loss:
//
//   pass 1 (ssim_fwd_kernel):  per pixel S and a = dS/dmu_x, b = dS/dE[x^2], c = dS/dE[xy];
//                              block sums of S and |xc-y| -> partials;  a, b, c -> workspace
//   pass 2 (ssim_bwd_kernel):  dsr = clipmask * ( -ls/N * (G*a + 2 xc (G*b) + y (G*c))
//                                                 + l1/N * sign(xc - y) )
//   finalize: sums the partials in a fixed order -> out[3] = {l1, ssim_loss, total}
//
// HBM-bound: 2 + 3 reads/writes in pass 1, 5 + 1 in pass 2 = 11 floats per pixel.
#include "srhip_common.h"

namespace {

constexpr int WIN = 11, HALO = 5, TILE = 32, EXT = TILE + 2 * HALO;  // 42

struct Gauss { float w[WIN]; };

struct LossParams {
    const float* sr;
    const float* y;
    float* abc;      // workspace: 3 maps of planes*h*w floats
    float* partial;  // workspace: 2 floats per block
    float* dsr;
    int planes, h, w, tiles_x, tiles_y, ntiles;
    float clip_min, clip_max;
    float g_ssim, g_l1;  // -lambda_ssim / N,  lambda_l1 / N
    Gauss g;
};

// torch.clamp semantics: a NaN stays a NaN (fminf/fmaxf alone would return the bound and hide it)
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v != v ? v : fminf(fmaxf(v, lo), hi); }

typedef float lf2 __attribute__((ext_vector_type(2)));   // two filtered quantities per packed fp32 instruction

// Both passes filter with a register window: a thread produces FOUR neighbouring outputs of a separable 11-tap pass from
// the 14 inputs they share (14 LDS reads instead of 44), quantities travel in pairs (x, y), (x^2, y^2) so that one
// packed instruction serves two of the five filters, and the products x^2, y^2, xy are formed once per loaded value
// instead of once per tap.  (One output per thread with one LDS read per tap per quantity was bound by those reads and
// by vector-ALU issue: 0.62 ms for a 4K frame pair at 1.7 TB/s while moving exactly the algorithmic bytes.)
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const LossParams p)
{
    __shared__ lf2 sxy[EXT][EXT + 1];                       // (clamped sr, target)
    __shared__ lf2 h01[EXT][TILE + 1], h23[EXT][TILE + 1];  // horizontal pass: (mu_x, mu_y), (E[x^2], E[y^2])
    __shared__ float h4[EXT][TILE + 1];                     //                  E[xy]
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    // Tiles in XCD bands: work-group g takes tile (g % 8) * ceil(tiles / 8) + g / 8, so each of the 8 XCDs (one L2
    // each, work-groups dealt round-robin) owns a contiguous run of tiles and the 5-pixel halos of neighbouring tiles
    // are served by its own L2 (with the natural order the fetch traffic was 3x the algorithmic reads).
    const int per = (p.ntiles + 7) >> 3;
    int b = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (b >= p.ntiles) return;
    const int tile_id = b;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int plane = b / p.tiles_y;
    const int x0 = tx * TILE, y0 = ty * TILE;
    const long long base = (long long)plane * p.h * p.w;

    for (int i = tid; i < EXT * EXT; i += 256) {
        const int r = i / EXT, c = i % EXT;
        const int yy = y0 - HALO + r, xx = x0 - HALO + c;
        float vx = 0.0f, vy = 0.0f;
        if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) {
            vx = clampf(p.sr[base + (long long)yy * p.w + xx], p.clip_min, p.clip_max);
            vy = p.y[base + (long long)yy * p.w + xx];
        }
        sxy[r][c] = lf2{vx, vy};
    }
    __syncthreads();
    // horizontal pass: EXT rows x (TILE / 4) groups of four columns
    for (int i = tid; i < EXT * (TILE / 4); i += 256) {
        const int r = i / (TILE / 4), c0 = (i % (TILE / 4)) * 4;
        lf2 v[WIN + 3], sq[WIN + 3];
        float xy[WIN + 3];
#pragma unroll
        for (int k = 0; k < WIN + 3; ++k) {
            v[k] = sxy[r][c0 + k];
            sq[k] = v[k] * v[k];
            xy[k] = v[k].x * v[k].y;
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            lf2 q01 = lf2{0.0f, 0.0f}, q23 = lf2{0.0f, 0.0f};
            float q4 = 0.0f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float g = p.g.w[k];
                q01 += g * v[o + k]; q23 += g * sq[o + k]; q4 += g * xy[o + k];
            }
            h01[r][c0 + o] = q01; h23[r][c0 + o] = q23; h4[r][c0 + o] = q4;
        }
    }
    __syncthreads();
    // vertical pass: one thread = one column x four rows (TILE / 4 row groups x TILE columns = 256 threads)
    float s_sum = 0.0f, l_sum = 0.0f;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    {
        const int c = tid % TILE, r0 = (tid / TILE) * 4;
        lf2 a01[WIN + 3], a23[WIN + 3];
        float a4[WIN + 3];
#pragma unroll
        for (int k = 0; k < WIN + 3; ++k) { a01[k] = h01[r0 + k][c]; a23[k] = h23[r0 + k][c]; a4[k] = h4[r0 + k][c]; }
        const long long map = (long long)p.planes * p.h * p.w;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            lf2 m = lf2{0.0f, 0.0f}, e2 = lf2{0.0f, 0.0f};
            float exy = 0.0f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float g = p.g.w[k];
                m += g * a01[o + k]; e2 += g * a23[o + k]; exy += g * a4[o + k];
            }
            const int yy = y0 + r0 + o, xx = x0 + c;
            if (yy < p.h && xx < p.w) {
                const float mx = m.x, my = m.y, exx = e2.x, eyy = e2.y;
                const float mx2 = mx * mx, my2 = my * my, mxy = mx * my;
                const float A1 = 2.0f * mxy + C1, A2 = 2.0f * (exy - mxy) + C2;
                const float B1 = mx2 + my2 + C1, B2 = (exx - mx2) + (eyy - my2) + C2;
                const float inv = 1.0f / (B1 * B2);
                const float S = A1 * A2 * inv;
                const float da = 2.0f * my * (A2 - A1) * inv - S * 2.0f * mx * (B2 - B1) * inv;
                const float db = -S / B2;
                const float dc = 2.0f * A1 * inv;
                const long long oidx = base + (long long)yy * p.w + xx;
                p.abc[oidx] = da; p.abc[map + oidx] = db; p.abc[2 * map + oidx] = dc;
                s_sum += S;
                const lf2 ctr = sxy[r0 + o + HALO][c + HALO];
                l_sum += fabsf(ctr.x - ctr.y);
            }
        }
    }
    // block reduction in a fixed order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s_sum += __shfl_down(s_sum, off);
        l_sum += __shfl_down(l_sum, off);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s_sum; red[1][tid >> 6] = l_sum; }
    __syncthreads();
    if (tid == 0) {
        p.partial[2 * tile_id] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        p.partial[2 * tile_id + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(const LossParams p)
{
    __shared__ lf2 sab[EXT][EXT + 1];      // (a, b)
    __shared__ float sc[EXT][EXT + 1];     // c
    __shared__ lf2 hab[EXT][TILE + 1];
    __shared__ float hc[EXT][TILE + 1];
    const int tid = threadIdx.x;
    const int per = (p.ntiles + 7) >> 3;   // XCD bands, as in the forward pass
    int b = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (b >= p.ntiles) return;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int plane = b / p.tiles_y;
    const int x0 = tx * TILE, y0 = ty * TILE;
    const long long base = (long long)plane * p.h * p.w;
    const long long map = (long long)p.planes * p.h * p.w;
    for (int i = tid; i < EXT * EXT; i += 256) {
        const int r = i / EXT, c = i % EXT;
        const int yy = y0 - HALO + r, xx = x0 - HALO + c;
        float a = 0.0f, bb = 0.0f, cc = 0.0f;
        if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) {
            const long long o = base + (long long)yy * p.w + xx;
            a = p.abc[o]; bb = p.abc[map + o]; cc = p.abc[2 * map + o];
        }
        sab[r][c] = lf2{a, bb}; sc[r][c] = cc;
    }
    __syncthreads();
    for (int i = tid; i < EXT * (TILE / 4); i += 256) {
        const int r = i / (TILE / 4), c0 = (i % (TILE / 4)) * 4;
        lf2 v[WIN + 3];
        float w[WIN + 3];
#pragma unroll
        for (int k = 0; k < WIN + 3; ++k) { v[k] = sab[r][c0 + k]; w[k] = sc[r][c0 + k]; }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            lf2 q = lf2{0.0f, 0.0f};
            float q2 = 0.0f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float g = p.g.w[k];
                q += g * v[o + k]; q2 += g * w[o + k];
            }
            hab[r][c0 + o] = q; hc[r][c0 + o] = q2;
        }
    }
    __syncthreads();
    {
        const int c = tid % TILE, r0 = (tid / TILE) * 4;
        lf2 v[WIN + 3];
        float w[WIN + 3];
#pragma unroll
        for (int k = 0; k < WIN + 3; ++k) { v[k] = hab[r0 + k][c]; w[k] = hc[r0 + k][c]; }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            lf2 gab = lf2{0.0f, 0.0f};
            float gc = 0.0f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float g = p.g.w[k];
                gab += g * v[o + k]; gc += g * w[o + k];
            }
            const int yy = y0 + r0 + o, xx = x0 + c;
            if (yy < p.h && xx < p.w) {
                const long long oidx = base + (long long)yy * p.w + xx;
                const float raw = p.sr[oidx], yv = p.y[oidx];
                const float xc = clampf(raw, p.clip_min, p.clip_max);
                const float diff = xc - yv;
                const float sgn = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
                float d = p.g_ssim * (gab.x + 2.0f * xc * gab.y + yv * gc) + p.g_l1 * sgn;
                if (!(raw >= p.clip_min && raw <= p.clip_max)) d = 0.0f;  // clamp backward
                p.dsr[oidx] = d;
            }
        }
    }
}

// One work-group of 1024 threads; every thread sums its partials four at a time (independent loads in flight: a
// dependent chain of ~100 global loads per thread took 41 us for the 24,300 tiles of a 4K frame), then a fixed-order
// tree in LDS: deterministic from run to run.
__global__ __launch_bounds__(1024) void loss_finalize_kernel(const float* __restrict__ partial, int nblocks, float inv_n, float lambda_l1,
                                                             float lambda_ssim, float* __restrict__ out)
{
    __shared__ double rs[1024], rl[1024];
    struct __attribute__((packed, aligned(4))) Pair { float x, y; };   // the partials start at a 4-byte-aligned address
    const Pair* p2 = (const Pair*)partial;       // (ssim sum, l1 sum) per tile
    double s = 0.0, l = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * 1024 < nblocks; i += 4 * 1024) {
        const Pair a = p2[i], b = p2[i + 1024], c = p2[i + 2048], d = p2[i + 3072];
        s += ((double)a.x + (double)b.x) + ((double)c.x + (double)d.x);
        l += ((double)a.y + (double)b.y) + ((double)c.y + (double)d.y);
    }
    for (; i < nblocks; i += 1024) { const Pair a = p2[i]; s += a.x; l += a.y; }
    rs[threadIdx.x] = s; rl[threadIdx.x] = l;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { rs[threadIdx.x] += rs[threadIdx.x + off]; rl[threadIdx.x] += rl[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l1 = (float)(rl[0] * inv_n);
        const float ssim = 1.0f - (float)(rs[0] * inv_n);
        out[0] = l1; out[1] = ssim; out[2] = lambda_l1 * l1 + lambda_ssim * ssim;
    }
}

}  // namespace

extern "C" {

SRHIP_API size_t srhip_l1_ssim_workspace_bytes(int planes, int h, int w)
{
    const size_t px = (size_t)planes * h * w;
    const size_t blocks = (size_t)planes * srhip_div_up(h, TILE) * srhip_div_up(w, TILE);
    return (3 * px + 2 * blocks) * sizeof(float);
}

SRHIP_API int srhip_l1_ssim_loss(const float* sr, const float* target, float* dsr, float* out3, void* workspace,
                                 int planes, int h, int w, float lambda_l1, float lambda_ssim, float clip_min,
                                 float clip_max, void* stream)
{
    SRHIP_REQUIRE(sr && target && dsr && out3 && workspace, "l1_ssim_loss: null pointer");
    SRHIP_REQUIRE(planes >= 1 && h >= 1 && w >= 1, "l1_ssim_loss: empty tensor");
    SRHIP_REQUIRE(clip_min <= clip_max, "l1_ssim_loss: clip_min > clip_max");
    LossParams p;
    p.sr = sr; p.y = target; p.dsr = dsr;
    p.planes = planes; p.h = h; p.w = w;
    p.tiles_x = srhip_div_up(w, TILE); p.tiles_y = srhip_div_up(h, TILE);
    const long long nblocks = (long long)planes * p.tiles_x * p.tiles_y;
    SRHIP_REQUIRE(nblocks < (1ll << 31), "l1_ssim_loss: too many tiles");
    const size_t px = (size_t)planes * h * w;
    p.abc = (float*)workspace;
    p.partial = p.abc + 3 * px;
    p.clip_min = clip_min; p.clip_max = clip_max;
    const double n = (double)px;
    p.g_ssim = (float)(-(double)lambda_ssim / n);
    p.g_l1 = (float)((double)lambda_l1 / n);
    // core/modules/functional.py:6-9 in fp32: exp(-(i - 5)^2 / (2 sigma^2)), normalised by its sum
    float gs[WIN], sum = 0.0f;
    for (int i = 0; i < WIN; ++i) {
        const float d = (float)(i - WIN / 2);
        gs[i] = expf(-(d * d) / (2.0f * 1.5f * 1.5f));
        sum += gs[i];
    }
    for (int i = 0; i < WIN; ++i) p.g.w[i] = gs[i] / sum;
    hipStream_t s = (hipStream_t)stream;
    p.ntiles = (int)nblocks;
    const unsigned grid = (unsigned)((nblocks + 7) / 8 * 8);
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(grid), dim3(256), 0, s, p);
    int rc = srhip_check_launch("ssim_fwd_kernel");
    if (rc != SRHIP_OK) return rc;
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3(grid), dim3(256), 0, s, p);
    rc = srhip_check_launch("ssim_bwd_kernel");
    if (rc != SRHIP_OK) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, s, p.partial, (int)nblocks, (float)(1.0 / n), lambda_l1,
                       lambda_ssim, out3);
    return srhip_check_launch("loss_finalize_kernel");
}

}  // extern "C"
