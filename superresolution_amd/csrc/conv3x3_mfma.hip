// 3x3 / stride 1 / pad 1 convolution for gfx950 as a streaming implicit GEMM on MFMA.
//
// Mapping (M = output channels, N = pixels, K = 9 taps x input channels):
//   * v_mfma_f32_32x32x16_bf16 with A = weights (row = cout), B = pixels (col = pixel), so every
//     lane ends up holding, for ONE pixel, runs of 4 consecutive output channels (NHWC friendly).
//   * The whole weight tensor of the layer lives in REGISTERS for the life of the workgroup
//     (9 * KC * NCT fragments of 4 VGPRs; the unified 512-entry VGPR/AGPR file at one wave per
//     SIMD holds the 288 registers of a 64->64 layer), so LDS carries only activations.
//   * A workgroup walks a vertical strip of TW output columns top to bottom.  Input rows stream
//     HBM -> LDS through a ring of rows filled by global_load_lds (LDS-DMA, 16 B per lane); every
//     input row is fetched once per strip.  One wave = 32 consecutive pixels of the current row.
//   * LDS row image: [16-channel plane][pixel][32 B]; the two 16-B halves of a pixel are swapped
//     when (pixel >> 3) & 1, which makes the ds_read_b128 of 32 consecutive pixels conflict free.
//     LDS-DMA writes linearly, so the swap is applied to the per-lane SOURCE address.
//   * Channel concat of two source tensors is folded into the K loop (planes of A then planes of B).
//   * Epilogue: bias (as initial accumulator), activation, alpha, residual add, bf16 pack, store.
#include "srhip_common.h"

namespace {

struct ConvParams {
    const char* src_a;
    const char* src_b;
    const bf16x8* wpack;
    const float* bias;
    const char* res;
    char* dst;
    const char* zeros;
    int n, h, w;
    int ca, cb;  // padded channels of the two sources
    int kc_a;    // 16-channel planes that come from src_a
    int cout, cout_stride;
    int act;
    float slope, alpha;
    int dst_nchw_f32;
    int nstrips, nsegs, seg_h, total_items;
};

template <int KC_, int CT_TOTAL_, int CSPLIT_>
struct Cfg {
    static constexpr int KC = KC_;              // 16-channel K planes
    static constexpr int CT_TOTAL = CT_TOTAL_;  // 32-wide output-channel tiles of the layer
    static constexpr int CSPLIT = CSPLIT_;      // waves that share a pixel group (split over cout tiles)
    static constexpr int NCT = CT_TOTAL / CSPLIT;
    static constexpr int NWAVES = 4;
    static constexpr int NPG = NWAVES / CSPLIT;  // pixel groups (32 px) per row
    static constexpr int TW = 32 * NPG;          // strip width
    static constexpr int PW = (TW + 2 + 7) / 8 * 8;  // LDS row pitch in pixels (1 halo px each side)
    static constexpr int PLANE_BYTES = PW * 32;
    static constexpr int SLOT_BYTES = KC * PLANE_BYTES;
    static constexpr int SLOT_CHUNKS = SLOT_BYTES / 16;
    static constexpr int NPIECE = (SLOT_CHUNKS + 63) / 64;          // 1 KiB LDS-DMA pieces per row
    static constexpr int MAXP = (NPIECE + NWAVES - 1) / NWAVES;     // pieces per wave per row
    static constexpr int LA = 3;                                    // rows issued ahead
    static constexpr int RING = 4;                                  // LDS row slots (power of two)
    static constexpr int LDS_BYTES = RING * SLOT_BYTES;
};

template <class C>
__global__ __launch_bounds__(C::NWAVES * 64, 1) void conv3x3_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs = wave % C::CSPLIT;  // which cout tiles this wave owns
    const int pg = wave / C::CSPLIT;  // which 32-pixel group of the row
    const int hl = lane >> 5;         // k half of the MFMA fragment / cout sub-block
    const int pl = lane & 31;         // pixel within the group

    // ---- weights -> registers (fragment order prepared by the pack kernel) ----
    bf16x8 wreg[9][C::KC][C::NCT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < C::KC; ++kc)
#pragma unroll
            for (int ct = 0; ct < C::NCT; ++ct)
                wreg[t][kc][ct] = p.wpack[((t * C::KC + kc) * C::CT_TOTAL + cs * C::NCT + ct) * 64 + lane];

    // ---- bias -> registers, in accumulator order: reg r <-> cout = tile*32 + 8*(r>>2) + 4*hl + (r&3)
    float breg[C::NCT][16];
#pragma unroll
    for (int ct = 0; ct < C::NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (cs * C::NCT + ct) * 32 + 8 * (r >> 2) + 4 * hl + (r & 3);
            breg[ct][r] = (p.bias != nullptr && co < p.cout) ? p.bias[co] : 0.0f;
        }

    // ---- per-lane LDS read offsets inside a row slot, one per horizontal tap ----
    int rd[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = 32 * pg + pl + kx;
        rd[kx] = px * 32 + (((hl ^ (px >> 3)) & 1) << 4);
    }

    const int H = p.h, W = p.w;
    const long long row_a_bytes = (long long)W * p.ca * 2;
    const long long row_b_bytes = (long long)W * p.cb * 2;

    for (int item = blockIdx.x; item < p.total_items; item += gridDim.x) {
        const int strip = item % p.nstrips;
        const int tseg = item / p.nstrips;
        const int seg = tseg % p.nsegs;
        const int n = tseg / p.nsegs;
        const int x0 = strip * C::TW;
        const int y0 = seg * p.seg_h;
        const int y1 = min(y0 + p.seg_h, H);
        const int nout = y1 - y0;
        const int nin = nout + 2;  // input rows r = 0..nin-1  <->  y = y0 - 1 + r

        // ---- per-lane source offsets of this wave's LDS-DMA pieces (constant over the strip) ----
        int poff[C::MAXP];
        int pflag[C::MAXP];  // bit0: lane participates, bit1: x inside the image, bit2: source B
#pragma unroll
        for (int m = 0; m < C::MAXP; ++m) {
            const int q = wave + m * C::NWAVES;
            const int j = q * 64 + lane;
            const bool part = (q < C::NPIECE) && (j < C::SLOT_CHUNKS);
            const int plane = j / (2 * C::PW);
            const int rem = j - plane * (2 * C::PW);
            const int px = rem >> 1;
            const int hf = (rem & 1) ^ ((px >> 3) & 1);
            const int x = x0 - 1 + px;
            const bool xin = (x >= 0) && (x < W);
            const bool is_b = plane >= p.kc_a;
            const int off = is_b ? (x * p.cb + (plane - p.kc_a) * 16 + hf * 8) * 2
                                 : (x * p.ca + plane * 16 + hf * 8) * 2;
            poff[m] = off;
            pflag[m] = (part ? 1 : 0) | (xin ? 2 : 0) | (is_b ? 4 : 0);
        }

        auto issue_row = [&](int r) {
            const int y = y0 - 1 + r;
            const bool yin = (y >= 0) && (y < H);
            const char* row_a = p.src_a + ((long long)n * H + y) * row_a_bytes;
            const char* row_b = p.src_b + ((long long)n * H + y) * row_b_bytes;
            char* slot = smem + (r & (C::RING - 1)) * C::SLOT_BYTES;
#pragma unroll
            for (int m = 0; m < C::MAXP; ++m) {
                const int q = wave + m * C::NWAVES;
                if (pflag[m] & 1) {
                    const char* g = (pflag[m] & 4) ? row_b : row_a;
                    g = (yin && (pflag[m] & 2)) ? g + poff[m] : p.zeros;
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)g,
                        (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
                }
            }
        };

        __syncthreads();  // every wave is done reading the previous item's rows
#pragma unroll 1
        for (int r = 0; r < C::LA && r < nin; ++r) issue_row(r);

#pragma unroll 1
        for (int i = 0; i < nout; ++i) {
            // rows i..i+2 were issued at least one iteration ago; wait for this wave's pieces,
            // then the barrier makes every wave's pieces visible (and frees row i-1's slot).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (i + C::LA < nin) issue_row(i + C::LA);

            f32x16 acc[C::NCT];
#pragma unroll
            for (int ct = 0; ct < C::NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = breg[ct][r];

#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const char* rowp = smem + ((i + ky) & (C::RING - 1)) * C::SLOT_BYTES;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const char* ap = rowp + rd[kx];
#pragma unroll
                    for (int kc = 0; kc < C::KC; ++kc) {
                        const bf16x8 b = *(const bf16x8*)(ap + kc * C::PLANE_BYTES);
#pragma unroll
                        for (int ct = 0; ct < C::NCT; ++ct)
                            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[ky * 3 + kx][kc][ct], b,
                                                                              acc[ct], 0, 0, 0);
                    }
                }
            }

            // ---- epilogue ----
            const int x = x0 + 32 * pg + pl;
            const int y = y0 + i;
            if (x < W) {
                if (!p.dst_nchw_f32) {
                    const long long pix = ((long long)n * H + y) * W + x;
                    char* drow = p.dst + pix * p.cout_stride * 2;
                    const char* rrow = p.res ? p.res + pix * p.cout_stride * 2 : nullptr;
#pragma unroll
                    for (int ct = 0; ct < C::NCT; ++ct)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c0 = (cs * C::NCT + ct) * 32 + 8 * g + 4 * hl;
                            if (c0 < p.cout_stride) {
                                float v[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float t = acc[ct][4 * g + e];
                                    if (p.act == SRHIP_ACT_RELU) t = fmaxf(t, 0.0f);
                                    else if (p.act == SRHIP_ACT_LRELU) t = t > 0.0f ? t : t * p.slope;
                                    v[e] = t * p.alpha;
                                }
                                if (rrow) {
                                    const uint2 rr = *(const uint2*)(rrow + c0 * 2);
                                    v[0] += bf16_lo(rr.x); v[1] += bf16_hi(rr.x);
                                    v[2] += bf16_lo(rr.y); v[3] += bf16_hi(rr.y);
                                }
                                uint2 o;
                                o.x = pack_bf16x2(v[0], v[1]);
                                o.y = pack_bf16x2(v[2], v[3]);
                                *(uint2*)(drow + c0 * 2) = o;
                            }
                        }
                } else {
                    // fp32 NCHW output (network output / fp32-accumulate checks)
                    float* d = (float*)p.dst;
#pragma unroll
                    for (int ct = 0; ct < C::NCT; ++ct)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int co = (cs * C::NCT + ct) * 32 + 8 * (r >> 2) + 4 * hl + (r & 3);
                            if (co < p.cout) {
                                float t = acc[ct][r];
                                if (p.act == SRHIP_ACT_RELU) t = fmaxf(t, 0.0f);
                                else if (p.act == SRHIP_ACT_LRELU) t = t > 0.0f ? t : t * p.slope;
                                d[(((long long)n * p.cout + co) * H + y) * W + x] = t * p.alpha;
                            }
                        }
                }
            }
        }
    }
}

// fp32 OIHW master weights -> bf16 MFMA fragment order [tap][kc][cout tile][lane][8].
__global__ void pack_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int w_o,
                                    int w_i, int cout, int cout_off, int cin_a_real, int ca_pad,
                                    int cin_b_real, int cb_pad, int transpose_flip, int kc_total,
                                    int ct_total, int total)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 7;
    const int lane = (idx >> 3) & 63;
    int rest = idx >> 9;
    const int ct = rest % ct_total; rest /= ct_total;
    const int kc = rest % kc_total; rest /= kc_total;
    const int tap = rest;
    const int ky = tap / 3, kx = tap % 3;
    const int co = ct * 32 + (lane & 31);
    const int cip = kc * 16 + 8 * (lane >> 5) + j;
    int ci = -1;
    if (cip < ca_pad) { if (cip < cin_a_real) ci = cip; }
    else { const int cbi = cip - ca_pad; if (cbi < cin_b_real) ci = cin_a_real + cbi; }
    float v = 0.0f;
    if (co < cout && ci >= 0) {
        if (!transpose_flip) {
            // forward: W[cout_off + co][ci][ky][kx]
            v = w[(((long long)(cout_off + co) * w_i + ci) * 3 + ky) * 3 + kx];
        } else {
            // data gradient: W'[co][ci][ky][kx] = W[ci][cout_off + co][2-ky][2-kx]
            v = w[(((long long)ci * w_i + cout_off + co) * 3 + (2 - ky)) * 3 + (2 - kx)];
        }
    }
    (void)w_o;
    out[idx] = f32_to_bf16_bits(v);
}

template <class C>
int launch_conv(ConvParams p, hipStream_t stream)
{
    static int blocks_per_cu = 0;  // immutable after first call for this instantiation
    if (blocks_per_cu == 0) {
        if (C::LDS_BYTES > 48 * 1024) {
            if (hipFuncSetAttribute((const void*)conv3x3_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    C::LDS_BYTES) != hipSuccess) {
                srhip_set_error("conv3x3: cannot raise the dynamic LDS limit to %d", C::LDS_BYTES);
                return SRHIP_ERR_LAUNCH;
            }
        }
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv3x3_kernel<C>, C::NWAVES * 64, C::LDS_BYTES) !=
                hipSuccess || nb < 1)
            nb = 1;
        blocks_per_cu = nb > 2 ? 2 : nb;
    }
    // ---- work decomposition: items = images x vertical segments x strips ----
    const int ncu = srhip_num_cus();
    const int slots = ncu * blocks_per_cu;
    const int nstrips = srhip_div_up(p.w, C::TW);
    // pick the number of vertical segments per image so that items fill an integer number of
    // rounds over the resident workgroups with little idle tail and little halo re-read
    int best_segs = 1;
    double best_cost = 1e30;
    for (int rounds = 1; rounds <= 16; ++rounds) {
        int segs = (int)(((long long)rounds * slots) / ((long long)nstrips * p.n));
        if (segs < 1) segs = 1;
        if (segs > p.h) segs = p.h;
        const int seg_h = srhip_div_up(p.h, segs);
        segs = srhip_div_up(p.h, seg_h);
        const long long items = (long long)segs * nstrips * p.n;
        const int r = srhip_div_up(items, slots);
        // time ~ rounds * (rows per item + fill latency of ~6 row-times)
        const double cost = (double)r * (seg_h + 2 + 6.0);
        if (cost < best_cost) { best_cost = cost; best_segs = segs; }
    }
    p.nsegs = best_segs;
    p.seg_h = srhip_div_up(p.h, best_segs);
    p.nsegs = srhip_div_up(p.h, p.seg_h);
    p.nstrips = nstrips;
    const long long total = (long long)p.nsegs * nstrips * p.n;
    SRHIP_REQUIRE(total < (1ll << 30), "conv3x3: too many work items");
    p.total_items = (int)total;
    const int grid = (int)(total < slots ? total : slots);
    hipLaunchKernelGGL(conv3x3_kernel<C>, dim3(grid), dim3(C::NWAVES * 64), C::LDS_BYTES, stream, p);
    return srhip_check_launch("conv3x3_kernel");
}

}  // namespace

extern "C" {

SRHIP_API size_t srhip_conv3x3_packed_bytes(int cin_padded, int cout)
{
    const int kc = cin_padded / 16;
    const int ct = (cout + 31) / 32;
    return (size_t)9 * kc * ct * 64 * 8 * 2;
}

SRHIP_API int srhip_conv3x3_pack_weights(const float* w, void* wpack, int w_o, int w_i, int cout, int cout_off,
                                         int cin_a_real, int ca_pad, int cin_b_real, int cb_pad,
                                         int transpose_flip, void* stream)
{
    SRHIP_REQUIRE(w && wpack, "pack_weights: null pointer");
    SRHIP_REQUIRE(ca_pad % 16 == 0 && cb_pad % 16 == 0 && ca_pad > 0 && cb_pad >= 0, "pack_weights: padded channels must be multiples of 16");
    SRHIP_REQUIRE(cin_a_real <= ca_pad && cin_b_real <= cb_pad, "pack_weights: real channels exceed padded");
    SRHIP_REQUIRE(cout >= 1 && cout <= 64, "pack_weights: cout must be in [1,64]");
    const int cin_real = cin_a_real + cin_b_real;
    if (!transpose_flip)
        SRHIP_REQUIRE(cout_off + cout <= w_o && cin_real == w_i, "pack_weights: weight shape (%d,%d) does not match cout %d+%d / cin %d", w_o, w_i, cout_off, cout, cin_real);
    else
        SRHIP_REQUIRE(cout_off + cout <= w_i && cin_real == w_o, "pack_weights: weight shape (%d,%d) does not match the transposed use", w_o, w_i);
    const int kc = (ca_pad + cb_pad) / 16, ct = (cout + 31) / 32;
    const int total = 9 * kc * ct * 64 * 8;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(srhip_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned short*)wpack, w_o, w_i, cout, cout_off, cin_a_real, ca_pad, cin_b_real, cb_pad,
                       transpose_flip, kc, ct, total);
    return srhip_check_launch("pack_weights_kernel");
}

SRHIP_API int srhip_conv3x3_fwd(const srhip_conv3x3_args* a, void* stream)
{
    SRHIP_REQUIRE(a != nullptr, "conv3x3: null args");
    SRHIP_REQUIRE(a->src_a && a->wpack && a->dst, "conv3x3: null src_a / wpack / dst");
    SRHIP_REQUIRE(a->n >= 1 && a->h >= 1 && a->w >= 1, "conv3x3: empty tensor");
    SRHIP_REQUIRE(a->ca > 0 && a->ca % 16 == 0 && a->cb >= 0 && a->cb % 16 == 0, "conv3x3: channels must be multiples of 16 (ca=%d cb=%d)", a->ca, a->cb);
    SRHIP_REQUIRE((a->cb == 0) == (a->src_b == nullptr), "conv3x3: src_b and cb disagree");
    SRHIP_REQUIRE(a->cout >= 1 && a->cout <= 64, "conv3x3: cout %d not in [1,64]", a->cout);
    SRHIP_REQUIRE((long long)a->w * (a->ca > a->cb ? a->ca : a->cb) * 2 < (1ll << 31), "conv3x3: row too long");
    if (a->dst_nchw_f32) {
        SRHIP_REQUIRE(a->res == nullptr, "conv3x3: fp32 NCHW output takes no residual");
    } else {
        SRHIP_REQUIRE(a->cout_stride % 8 == 0 && a->cout_stride >= a->cout && a->cout_stride <= 64, "conv3x3: bad cout_stride %d", a->cout_stride);
    }
    int st = SRHIP_OK;
    ConvParams p;
    p.zeros = (const char*)srhip_zero_page(&st);
    if (st != SRHIP_OK) return st;
    p.src_a = (const char*)a->src_a;
    p.src_b = a->src_b ? (const char*)a->src_b : (const char*)a->src_a;
    p.wpack = (const bf16x8*)a->wpack;
    p.bias = a->bias;
    p.res = (const char*)a->res;
    p.dst = (char*)a->dst;
    p.n = a->n; p.h = a->h; p.w = a->w;
    p.ca = a->ca; p.cb = a->cb; p.kc_a = a->ca / 16;
    p.cout = a->cout; p.cout_stride = a->cout_stride;
    p.act = a->act; p.slope = a->slope; p.alpha = a->alpha;
    p.dst_nchw_f32 = a->dst_nchw_f32;
    const int kc = (a->ca + a->cb) / 16;
    const int ct = (a->cout + 31) / 32;
    hipStream_t s = (hipStream_t)stream;
#define SRHIP_CONV_CASE(KC, CT, CSPLIT) \
    if (kc == KC && ct == CT) return launch_conv<Cfg<KC, CT, CSPLIT>>(p, s);
    SRHIP_CONV_CASE(1, 1, 1)
    SRHIP_CONV_CASE(2, 1, 1)
    SRHIP_CONV_CASE(3, 1, 1)
    SRHIP_CONV_CASE(4, 1, 1)
    SRHIP_CONV_CASE(6, 1, 1)
    SRHIP_CONV_CASE(1, 2, 1)
    SRHIP_CONV_CASE(2, 2, 1)
    SRHIP_CONV_CASE(3, 2, 1)
    SRHIP_CONV_CASE(4, 2, 1)
    SRHIP_CONV_CASE(6, 2, 2)
#undef SRHIP_CONV_CASE
    srhip_set_error("conv3x3: unsupported channel configuration cin=%d cout=%d", a->ca + a->cb, a->cout);
    return SRHIP_ERR_INVALID;
}

}  // extern "C"
