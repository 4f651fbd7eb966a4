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
#include <cstdlib>

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

// ------------------------------------------------------------------------------------------------------------------
// Streaming forms of the two passes.  The tile kernels above run load -> barrier -> filter -> barrier -> filter -> store
// once per 32 x 32 tile with three 42-KB tiles per CU in flight, and that latency chain, not the traffic (exactly the
// algorithmic bytes) or the arithmetic, sets their pace.  Here a work-group owns a strip of 256 columns (one thread per
// column) and walks a segment of rows top to bottom:
//   * each input row goes through a small LDS row buffer (266 values, double buffered, ONE barrier per row) for the
//     horizontal 11-tap pass -- consecutive threads read consecutive addresses;
//   * the vertical pass never touches LDS: a thread keeps the eleven output rows its column is still collecting in
//     registers and adds g[k] * (this row's horizontal result) to output row r - k.  A row's first contribution is k = 0
//     and its last k = 10, so each sum runs over k in ascending order, as in the tile kernels;
//   * the row loop is unrolled eleven times so that "row r mod 11" names a register, not an index;
//   * the next row's values are requested before the current row is filtered.
// A segment re-filters the 10 halo rows above and below it (the host picks the segment height so that the launch has
// about three work-groups per resident slot).
constexpr int SW = 256;                 // columns per strip = threads per work-group
constexpr int SROW = SW + 2 * HALO;     // values per staged row

struct StreamParams {
    LossParams p;
    int strips, segs, seg_h, nunits;
};

template <int V>
struct LInt { static constexpr int value = V; };
template <int B, int E, class F>
__device__ __forceinline__ void l_for(F&& f)
{
    if constexpr (B < E) {
        f(LInt<B>{});
        l_for<B + 1, E>(f);
    }
}

__global__ __launch_bounds__(SW) void ssim_fwd_stream_kernel(const StreamParams sp)
{
    const LossParams& p = sp.p;
    __shared__ lf2 rowbuf[2][SROW + 2];
    __shared__ float red[2][SW / 64];
    const int tid = threadIdx.x;
    const int per = (sp.nunits + 7) >> 3;   // XCD bands: vertical neighbours (which share halo rows) stay on one L2
    int b = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (b >= sp.nunits) return;
    const int unit = b;
    const int seg = b % sp.segs; b /= sp.segs;
    const int strip = b % sp.strips;
    const int plane = b / sp.strips;
    const int x0 = strip * SW, ys = seg * sp.seg_h;
    const int nrows = min(sp.seg_h, p.h - ys);          // output rows of this segment
    const int nsteps = nrows + 2 * HALO;                // input rows ys - 5 .. ys + nrows + 4
    const long long base = (long long)plane * p.h * p.w;
    const long long map = (long long)p.planes * p.h * p.w;
    const int xc = x0 + tid;                            // this thread's output column
    const int xa = x0 - HALO + tid, xb = x0 - HALO + SW + tid;   // the columns it stages (xb: threads 0..9 only)
    const bool has_b = tid < 2 * HALO;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

    auto fetch = [&](const int r, lf2& va, lf2& vb) {   // input row r of the segment: image row ys - 5 + r
        const int yy = ys - HALO + r;
        va = lf2{0.0f, 0.0f}; vb = lf2{0.0f, 0.0f};
        if (r < nsteps && yy >= 0 && yy < p.h) {
            const long long ro = base + (long long)yy * p.w;
            if (xa >= 0 && xa < p.w) va = lf2{clampf(p.sr[ro + xa], p.clip_min, p.clip_max), p.y[ro + xa]};
            if (has_b && xb < p.w) vb = lf2{clampf(p.sr[ro + xb], p.clip_min, p.clip_max), p.y[ro + xb]};
        }
    };
    lf2 A01[WIN], A23[WIN], ctr[WIN];
    float A4[WIN];
#pragma unroll
    for (int k = 0; k < WIN; ++k) { A01[k] = lf2{0.0f, 0.0f}; A23[k] = lf2{0.0f, 0.0f}; A4[k] = 0.0f; ctr[k] = lf2{0.0f, 0.0f}; }
    float s_sum = 0.0f, l_sum = 0.0f;
    // three rows are in flight ahead of the one being filtered (a row step is a few hundred cycles, a load a few thousand)
    lf2 na, nb, na1, nb1, na2, nb2;
    fetch(0, na, nb);
    fetch(1, na1, nb1);
    fetch(2, na2, nb2);
    for (int rb = 0; rb < nsteps; rb += WIN) {
        l_for<0, WIN>([&](auto ph_c) {
            constexpr int ph = decltype(ph_c)::value;
            const int r = rb + ph;
            if (r < nsteps) {   // uniform over the work-group
                rowbuf[r & 1][tid] = na;
                if (has_b) rowbuf[r & 1][SW + tid] = nb;
                na = na1; nb = nb1; na1 = na2; nb1 = nb2;
                fetch(r + 3, na2, nb2);
                __syncthreads();
                lf2 q01 = lf2{0.0f, 0.0f}, q23 = lf2{0.0f, 0.0f};
                float q4 = 0.0f;
#pragma unroll
                for (int k = 0; k < WIN; ++k) {
                    const lf2 v = rowbuf[r & 1][tid + k];
                    const float g = p.g.w[k];
                    q01 += g * v; q23 += g * (v * v); q4 += g * (v.x * v.y);
                    if (k == HALO) ctr[ph] = v;
                }
                // this row is tap k of output row r - k, which lives in slot (ph - k) mod 11; tap 0 starts a new sum
#pragma unroll
                for (int k = 0; k < WIN; ++k) {
                    constexpr int dummy = 0; (void)dummy;
                    const int slot = (ph - k + WIN) % WIN;
                    const float g = p.g.w[k];
                    if (k == 0) { A01[slot] = g * q01; A23[slot] = g * q23; A4[slot] = g * q4; }
                    else { A01[slot] += g * q01; A23[slot] += g * q23; A4[slot] += g * q4; }
                }
                // output row o = r - 10 is complete (slot (ph + 1) mod 11); its centre pixel came with input row r - 5
                const int o = r - 2 * HALO;
                if (o >= 0 && xc < p.w) {
                    constexpr int so = (ph + 1) % WIN, sc = (ph - HALO + WIN) % WIN;
                    const float mx = A01[so].x, my = A01[so].y, exx = A23[so].x, eyy = A23[so].y, exy = A4[so];
                    const float mx2 = mx * mx, my2 = my * my, mxy = mx * my;
                    const float A1 = 2.0f * mxy + C1, A2 = 2.0f * (exy - mxy) + C2;
                    const float B1 = mx2 + my2 + C1, B2 = (exx - mx2) + (eyy - my2) + C2;
                    const float inv = 1.0f / (B1 * B2);
                    const float S = A1 * A2 * inv;
                    const float da = 2.0f * my * (A2 - A1) * inv - S * 2.0f * mx * (B2 - B1) * inv;
                    const float db = -S / B2;
                    const float dc = 2.0f * A1 * inv;
                    const long long oidx = base + (long long)(ys + o) * p.w + xc;
                    p.abc[oidx] = da; p.abc[map + oidx] = db; p.abc[2 * map + oidx] = dc;
                    s_sum += S;
                    l_sum += fabsf(ctr[sc].x - ctr[sc].y);
                }
            }
        });
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s_sum += __shfl_down(s_sum, off);
        l_sum += __shfl_down(l_sum, off);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s_sum; red[1][tid >> 6] = l_sum; }
    __syncthreads();
    if (tid == 0) {
        p.partial[2 * unit] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        p.partial[2 * unit + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(SW) void ssim_bwd_stream_kernel(const StreamParams sp)
{
    const LossParams& p = sp.p;
    __shared__ lf2 rowab[2][SROW + 2];
    __shared__ float rowc[2][SROW + 2];
    const int tid = threadIdx.x;
    const int per = (sp.nunits + 7) >> 3;
    int b = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (b >= sp.nunits) return;
    const int seg = b % sp.segs; b /= sp.segs;
    const int strip = b % sp.strips;
    const int plane = b / sp.strips;
    const int x0 = strip * SW, ys = seg * sp.seg_h;
    const int nrows = min(sp.seg_h, p.h - ys);
    const int nsteps = nrows + 2 * HALO;
    const long long base = (long long)plane * p.h * p.w;
    const long long map = (long long)p.planes * p.h * p.w;
    const int xc = x0 + tid;
    const int xa = x0 - HALO + tid, xb = x0 - HALO + SW + tid;
    const bool has_b = tid < 2 * HALO;

    auto fetch = [&](const int r, lf2& va, float& ca, lf2& vb, float& cb) {
        const int yy = ys - HALO + r;
        va = lf2{0.0f, 0.0f}; vb = lf2{0.0f, 0.0f}; ca = 0.0f; cb = 0.0f;
        if (r < nsteps && yy >= 0 && yy < p.h) {
            const long long ro = base + (long long)yy * p.w;
            if (xa >= 0 && xa < p.w) { va = lf2{p.abc[ro + xa], p.abc[map + ro + xa]}; ca = p.abc[2 * map + ro + xa]; }
            if (has_b && xb < p.w) { vb = lf2{p.abc[ro + xb], p.abc[map + ro + xb]}; cb = p.abc[2 * map + ro + xb]; }
        }
    };
    // sr and target of output row o (needed when that row's sums are complete): requested one step ahead
    auto fetch_out = [&](const int o, float& raw, float& yv) {
        raw = 0.0f; yv = 0.0f;
        if (o >= 0 && o < nrows && xc < p.w) {
            const long long oidx = base + (long long)(ys + o) * p.w + xc;
            raw = p.sr[oidx]; yv = p.y[oidx];
        }
    };
    lf2 Aab[WIN];
    float Ac[WIN];
#pragma unroll
    for (int k = 0; k < WIN; ++k) { Aab[k] = lf2{0.0f, 0.0f}; Ac[k] = 0.0f; }
    lf2 na, nb, na1, nb1, na2, nb2;
    float nca, ncb, nca1, ncb1, nca2, ncb2, oraw, oy, oraw1, oy1, oraw2, oy2;
    fetch(0, na, nca, nb, ncb);
    fetch(1, na1, nca1, nb1, ncb1);
    fetch(2, na2, nca2, nb2, ncb2);
    fetch_out(-2 * HALO, oraw, oy);
    fetch_out(1 - 2 * HALO, oraw1, oy1);
    fetch_out(2 - 2 * HALO, oraw2, oy2);
    for (int rb = 0; rb < nsteps; rb += WIN) {
        l_for<0, WIN>([&](auto ph_c) {
            constexpr int ph = decltype(ph_c)::value;
            const int r = rb + ph;
            if (r < nsteps) {
                rowab[r & 1][tid] = na; rowc[r & 1][tid] = nca;
                if (has_b) { rowab[r & 1][SW + tid] = nb; rowc[r & 1][SW + tid] = ncb; }
                const float raw = oraw, yv = oy;        // of output row r - 10
                na = na1; nca = nca1; nb = nb1; ncb = ncb1; na1 = na2; nca1 = nca2; nb1 = nb2; ncb1 = ncb2;
                oraw = oraw1; oy = oy1; oraw1 = oraw2; oy1 = oy2;
                fetch(r + 3, na2, nca2, nb2, ncb2);
                fetch_out(r + 3 - 2 * HALO, oraw2, oy2);
                __syncthreads();
                lf2 q = lf2{0.0f, 0.0f};
                float q2 = 0.0f;
#pragma unroll
                for (int k = 0; k < WIN; ++k) {
                    const float g = p.g.w[k];
                    q += g * rowab[r & 1][tid + k]; q2 += g * rowc[r & 1][tid + k];
                }
#pragma unroll
                for (int k = 0; k < WIN; ++k) {
                    const int slot = (ph - k + WIN) % WIN;
                    const float g = p.g.w[k];
                    if (k == 0) { Aab[slot] = g * q; Ac[slot] = g * q2; }
                    else { Aab[slot] += g * q; Ac[slot] += g * q2; }
                }
                const int o = r - 2 * HALO;
                if (o >= 0 && xc < p.w) {
                    constexpr int so = (ph + 1) % WIN;
                    const float xcv = clampf(raw, p.clip_min, p.clip_max);
                    const float diff = xcv - yv;
                    const float sgn = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
                    float d = p.g_ssim * (Aab[so].x + 2.0f * xcv * Aab[so].y + yv * Ac[so]) + p.g_l1 * sgn;
                    if (!(raw >= p.clip_min && raw <= p.clip_max)) d = 0.0f;  // clamp backward
                    p.dsr[base + (long long)(ys + o) * p.w + xc] = d;
                }
            }
        });
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

// ---- pixel-weighted L1: loss = mean((1 + weight at the pixel's low-resolution position) * |sr - target|) ----
// The temporal network's second output is a per-pixel loss weight at the input resolution
// (core/models/model_final.py:499-505: enhance_weight); the loss that uses it broadcasts it over the channels and the
// scale x scale pixels it covers (nearest).  One pass reads sr and target once and writes dsr = d loss / d sr and one
// partial sum per work-group; a second launch adds the partial sums in a fixed order (in double).
constexpr int WL1_THREADS = 256;
__global__ __launch_bounds__(WL1_THREADS) void weighted_l1_kernel(const float4* __restrict__ sr, const float4* __restrict__ y,
                                                                  const float* __restrict__ wt, float4* __restrict__ dsr,
                                                                  float* __restrict__ partial, int c, int h, int w4, int wh, int ww,
                                                                  int ry, int rx, long long nvec, float inv_n)
{
    __shared__ float red[WL1_THREADS / 64];
    float acc = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        const int x4 = (int)(i % w4);
        long long t = i / w4;
        const int yy = (int)(t % h);
        t /= h;                                   // plane = image * c + channel
        const long long img = t / c;
        const float* wrow = wt + (img * wh + yy / ry) * ww;
        const float4 a = sr[i], b = y[i];
        const float dv[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float k = 1.0f + wrow[(4 * x4 + e) / rx];
            acc += k * fabsf(dv[e]);
            o[e] = dv[e] > 0.0f ? k * inv_n : (dv[e] < 0.0f ? -k * inv_n : 0.0f);
        }
        dsr[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void weighted_l1_sum_kernel(const float* __restrict__ partial, int nparts, double inv_n, float* __restrict__ out)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * inv_n);
}

}  // namespace

extern "C" {

SRHIP_API size_t srhip_weighted_l1_workspace_bytes(void) { return (size_t)srhip_num_cus() * 8 * sizeof(float); }

SRHIP_API int srhip_weighted_l1_loss(const float* sr, const float* target, const float* weight, float* dsr, float* out1,
                                     void* workspace, int n, int c, int h, int w, int wh, int ww, void* stream)
{
    SRHIP_REQUIRE(sr && target && weight && dsr && out1 && workspace, "weighted_l1_loss: null pointer");
    SRHIP_REQUIRE(n >= 1 && c >= 1 && h >= 1 && w >= 1 && wh >= 1 && ww >= 1, "weighted_l1_loss: empty tensor");
    SRHIP_REQUIRE(h % wh == 0 && w % ww == 0, "weighted_l1_loss: the weight map (%d x %d) must divide the image (%d x %d)", wh, ww, h, w);
    SRHIP_REQUIRE(w % 4 == 0, "weighted_l1_loss: the image width must be a multiple of 4");
    SRHIP_REQUIRE(((size_t)sr | (size_t)target | (size_t)dsr) % 16 == 0, "weighted_l1_loss: sr, target and dsr must be 16-byte aligned");
    const long long count = (long long)n * c * h * w;
    const long long nvec = count / 4;
    const int blocks = (int)(nvec < (long long)srhip_num_cus() * 8 * WL1_THREADS ? (nvec + WL1_THREADS - 1) / WL1_THREADS : (long long)srhip_num_cus() * 8);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(weighted_l1_kernel, dim3(blocks), dim3(WL1_THREADS), 0, s, (const float4*)sr, (const float4*)target, weight,
                       (float4*)dsr, (float*)workspace, c, h, w / 4, wh, ww, h / wh, w / ww, nvec, (float)(1.0 / (double)count));
    int rc = srhip_check_launch("weighted_l1_kernel");
    if (rc != SRHIP_OK) return rc;
    hipLaunchKernelGGL(weighted_l1_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, blocks, 1.0 / (double)count, out1);
    return srhip_check_launch("weighted_l1_sum_kernel");
}

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
    // Streaming form for images that fill the chip with strips of 256 columns (the training crops, 4K frames): segments
    // so that about three work-groups wait for each of the chip's resident slots (4 per CU), at least 32 rows each.
    const int strips = srhip_div_up(w, SW);
    const char* form = getenv("SRHIP_LOSS_FORM");   // "tiles" / "stream" force a form (tests, A/B timing)
    const bool stream_auto = (long long)planes * strips * srhip_div_up(h, 32) >= 2ll * srhip_num_cus();
    if (form != nullptr ? form[0] == 's' : stream_auto) {
        StreamParams sp;
        long long want = (12ll * srhip_num_cus() + (long long)planes * strips - 1) / ((long long)planes * strips);
        const long long most = srhip_div_up(h, 32);
        if (want > most) want = most;
        if (want < 1) want = 1;
        sp.seg_h = srhip_div_up(h, want);
        sp.segs = srhip_div_up(h, sp.seg_h);
        sp.strips = strips;
        const long long units = (long long)planes * strips * sp.segs;
        // (the partials of the streaming form: one pair per work-group; the workspace holds one per 32 x 32 tile, never fewer)
        SRHIP_REQUIRE(units <= nblocks, "l1_ssim_loss: internal: more segments than tile partials");
        sp.nunits = (int)units;
        sp.p = p;
        const unsigned sgrid = (unsigned)((units + 7) / 8 * 8);
        hipLaunchKernelGGL(ssim_fwd_stream_kernel, dim3(sgrid), dim3(SW), 0, s, sp);
        int rcs = srhip_check_launch("ssim_fwd_stream_kernel");
        if (rcs != SRHIP_OK) return rcs;
        hipLaunchKernelGGL(ssim_bwd_stream_kernel, dim3(sgrid), dim3(SW), 0, s, sp);
        rcs = srhip_check_launch("ssim_bwd_stream_kernel");
        if (rcs != SRHIP_OK) return rcs;
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, s, p.partial, (int)units, (float)(1.0 / n), lambda_l1,
                           lambda_ssim, out3);
        return srhip_check_launch("loss_finalize_kernel");
    }
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
