// Weight gradient of the 3x3 / stride 1 / pad 1 convolution for gfx950 (autograd backward of
// nn.Conv2d at core/models/model.py:12,26-28,112-114, reached through loss.backward() at
// core/trainer.py:160):
//
//   dW[co][ci][ky][kx] = sum over (n, y, x) of dz[n][y][x][co] * xin[n][y+ky-1][x+kx-1][ci]
//   db[co]             = sum over (n, y, x) of dz[n][y][x][co]
//
// A GEMM whose contraction runs over PIXELS: M = co, N = ci, K = pixels, one 32x32 output tile
// per (co tile, ci tile, tap).  Both MFMA operands therefore need, per lane, runs of 8 consecutive
// pixels of ONE channel, while HBM and LDS hold pixel-major NHWC rows; the transposing LDS read
// ds_read_b64_tr_b16 delivers exactly that from the row-major image.
//
//   * A workgroup (4 waves, one per SIMD) walks vertical strips of 128 columns like the forward
//     kernel.  Rows of xin (with a one-pixel halo) and of dz stream HBM -> LDS by buffer LDS-DMA
//     into two rings; rows and columns outside the image fail the bounds check and land as zeros,
//     which is the zero padding and also makes partial strips contribute nothing.
//   * Wave w owns output tile (w % NT) for all 9 taps (9 accumulators of 16 registers) plus a share of
//     the bias-gradient accumulator (dz times an all-ones operand; the waves of one co tile split the k-steps).  Layers with fewer
//     than 4 tiles split the strip's pixels between the waves that share a tile.
//   * Accumulators persist over every work item of the workgroup; at the end each wave writes its
//     partial tiles to the workspace and a second kernel sums them in a fixed order
//     (deterministic, no atomics) into the fp32 OIHW gradient.
//   * LDS image [pixel][channel]; for 64-channel rows the two 64-byte halves of a pixel are swapped
//     on every second pixel pair (applied on the SOURCE side of the DMA) so that the transposing
//     reads of 4 consecutive pixels x 32 channels touch every bank once.
#include <atomic>

#include "srhip_common.h"
#define SRHIP_CAPTURE(x) (void)&(x)   // asm operands alone do not make a generic lambda capture

namespace {

// One launch serves up to kMaxJobs jobs = (input window, gradient window) pairs of ONE layer (same image size, same tile
// shape): work-groups [j * wgs_per_job, (j + 1) * wgs_per_job) walk job j's work items.  The small maps of a wide layer
// (384 -> 256 at 1/8 resolution = 24 channel-slice pairs of a few microseconds each) would otherwise pay a launch, a
// pipeline fill, a full grid of partial tiles and a reduction per pair.
constexpr int kMaxJobs = 32;
struct WgradJob {
    const char* x;
    const char* dz;
    // x and dz may be channel windows of wider maps: bytes per pixel of the map, byte offset of the window's first channel
    int xs, xco, zs, zco;
};
struct WgradParams {
    float* partial;
    int n, h, w;
    int nstrips, nsegs, seg_h, total_items;
    int njobs, wgs_per_job;
    WgradJob jobs[kMaxJobs];
};
struct WgradOut {
    float* dw;   // first output row of the job's co slice
    float* db;   // or null
    int ci_off, ci_real, co_real, pad;
};
struct ReduceParams {
    const float* partial;
    int nwg;              // work-groups (partial tile sets) per job
    int cit_n, nt, w_i, accumulate;
    float scale;
    int nwblocks, blocks_per_job;
    int tap0, ntaps, dw_taps;   // partial tiles [tap0, tap0 + ntaps) -> dw[co][ci][dw_taps] (9: tap index = position; 1: position 0)
    WgradOut out[kMaxJobs];
};

constexpr int round_up_i(int a, int b) { return (a + b - 1) / b * b; }
constexpr unsigned kOob = 0x80000000u;

// TAPS = 9: the 3x3 layer.  TAPS = 1: a 1x1 layer (core/models/model_final.py:237-246) = the centre tap alone -- one
// accumulator tile instead of nine, 8 MFMAs per row and wave instead of 72 (as the centre tap of the full kernel a 1x1
// layer paid for eight taps it threw away).
template <int CI_, int CO_, int TAPS_ = 9>
struct WCfg {
    static constexpr int CI = CI_, CO = CO_, TAPS = TAPS_;
    static constexpr int KX0 = TAPS == 9 ? 0 : 1, KX1 = TAPS == 9 ? 3 : 2;   // tap columns walked
    static_assert(TAPS == 9 || TAPS == 1, "3x3 or its centre tap");
    static constexpr int NWAVES = 4, TW = 128;
    static constexpr int CIT = CI >= 32 ? CI / 32 : 1;
    static constexpr int COT = CO >= 32 ? CO / 32 : 1;
    static constexpr int NT = CIT * COT;          // output tiles of the layer (1, 2 or 4)
    static constexpr int KS = NWAVES / NT;        // waves sharing a tile split the strip's pixels
    static constexpr int PXW = TW / KS;           // pixels per wave per row
    static constexpr int NKS = PXW / 16;          // MFMA k-steps per wave per row and tap
    static constexpr int XPB = CI * 2, ZPB = CO * 2;              // bytes per pixel
    static constexpr int XPP = 1024 / XPB, ZPP = 1024 / ZPB;      // pixels per 1-KiB piece
    static constexpr int XNP = round_up_i(TW + 2, XPP) / XPP;     // pieces per row
    static constexpr int ZNP = TW / ZPP;
    static constexpr int X_SLOT = XNP * 1024, Z_SLOT = ZNP * 1024;
    static constexpr int XA = 3, ZA = 3;          // rows issued ahead
    static constexpr int XRING = XA + 1, ZRING = ZA + 1;
    static constexpr int Z_BASE = XRING * X_SLOT;
    static constexpr int SINK_OFF = Z_BASE + ZRING * Z_SLOT;     // 1 KiB sink of the DMA pieces a wave does not own
    static constexpr int LDS_BYTES = SINK_OFF + 1024;
    static constexpr int MAXPX = (XNP + NWAVES - 1) / NWAVES, MINPX = XNP / NWAVES;
    static constexpr int MAXPZ = (ZNP + NWAVES - 1) / NWAVES, MINPZ = ZNP / NWAVES;
    // Every wave issues exactly MAXPX + MAXPZ pieces per iteration (a piece it does not own goes,
    // all-out-of-range, to the sink), x pieces first.  Top of iteration r needs x row r and dz row r,
    // both issued in iteration r-XA: younger are exactly XA-1 whole iterations.
    static constexpr int TOP_WAIT = (XA - 1) * (MAXPX + MAXPZ);
    static_assert(XA == ZA && XA >= 2, "wait counts assume both rings run the same distance ahead");
    static_assert(LDS_BYTES <= 160 * 1024, "rings exceed the CU's LDS");
};

__device__ __forceinline__ void wg_lds_dma(__amdgpu_buffer_rsrc_t rsrc, __attribute__((address_space(3))) void* dst,
                                           unsigned voffset, unsigned soffset)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, (int)voffset, (int)soffset, 0, 0);
}
template <int N>
__device__ __forceinline__ void wg_vm_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wg_lds_wait()
{
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N));
}
// Transposing LDS read (asm: the waits are counted by hand and the compiler must not order it
// against the LDS-DMA writes, which it would do by draining the whole vector-memory queue).
template <int OFF>
__device__ __forceinline__ bf16x4 lds_read_tr(unsigned addr)
{
    bf16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
    return r;
}
template <int V>
struct WIntC { static constexpr int value = V; };
template <int B, int E, class F>
__device__ __forceinline__ void wg_static_for(F&& f)
{
    if constexpr (B < E) {
        f(WIntC<B>{});
        wg_static_for<B + 1, E>(f);
    }
}
// MFMA as asm with the accumulator of tile T pinned to the fixed registers a[16T : 16T+15]: accumulators
// never move between register files (a copy right before an asm MFMA would be a hazard the compiler cannot
// see), and no hazard padding is added between the empty asm that ties the fragment registers to the
// counted LDS wait and the instruction that consumes them.  The leading s_nop covers a register copy the
// compiler may still place right in front of the operands.
template <int T>
__device__ __forceinline__ void wg_mfma(f32x16& acc, const bf16x8& a, const bf16x8& b)
{
#define WG_ACC_CASE(TI, RANGE) \
    if constexpr (T == TI) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+{a[" RANGE "]}"(acc) : "v"(a), "v"(b));
    WG_ACC_CASE(0, "0:15") WG_ACC_CASE(1, "16:31") WG_ACC_CASE(2, "32:47") WG_ACC_CASE(3, "48:63") WG_ACC_CASE(4, "64:79")
    WG_ACC_CASE(5, "80:95") WG_ACC_CASE(6, "96:111") WG_ACC_CASE(7, "112:127") WG_ACC_CASE(8, "128:143") WG_ACC_CASE(9, "144:159")
#undef WG_ACC_CASE
}
__device__ __forceinline__ bf16x8 join(bf16x4 lo, bf16x4 hi)
{
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// byte offset of (pixel px, channel byte cb) inside a row slot with PB bytes per pixel
template <int PB>
__device__ __forceinline__ unsigned img_off(int px, int cb)
{
    if (PB == 128) return (unsigned)(px * 128 + (cb ^ (((px >> 1) & 1) << 6)));
    return (unsigned)(px * PB + cb);
}

template <class C>
__global__ __launch_bounds__(C::NWAVES * 64, 1) void conv3x3_wgrad_kernel(const WgradParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = wave % C::NT;
    const int cot = tile / C::CIT, cit = tile % C::CIT;
    const int part = wave / C::NT;       // which PXW-pixel part of the strip
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- transposed-read lane roles: group g (16 lanes) reads the block of 4 pixels x 16 channels
    // whose row q / channel quad p this lane addresses; lanes 0..31 hold k = 0..7, lanes 32..63 k = 8..15.
    const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
    const int chsel = 16 * (g & 1) + 4 * pq;
    unsigned xaddr[3][2], zaddr[2];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
        const int pxl = part * C::PXW + 8 * (g >> 1) + 4 * j2 + q;
        const int zcb = ((32 * cot + chsel) & (C::CO - 1)) * 2;
        zaddr[j2] = lds_base + C::Z_BASE + img_off<C::ZPB>(pxl, zcb);
        const int xcb = ((32 * cit + chsel) & (C::CI - 1)) * 2;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) xaddr[kx][j2] = lds_base + img_off<C::XPB>(pxl + kx, xcb);
    }

    const int H = p.h, W = p.w;
    f32x16 acc[9];
    f32x16 accb;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;   // (TAPS = 1 touches acc[4] only: the others are dead code)
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.0f;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3f80;
    asm volatile("" : "+v"(ones));  // opaque: stays in four registers instead of being rebuilt next to its (asm) consumer
    bf16x4 alo[3][C::NKS], ahi[3][C::NKS];   // dz fragments of the three newest rows

    const int job = __builtin_amdgcn_readfirstlane((int)blockIdx.x / p.wgs_per_job);
    const int first_item = (int)blockIdx.x - job * p.wgs_per_job;
    const WgradJob jb = p.jobs[job];
    for (int item = first_item; item < p.total_items; item += p.wgs_per_job) {
        const int strip = item % p.nstrips;
        const int tseg = item / p.nstrips;
        const int seg = tseg % p.nsegs;
        const int n = tseg / p.nsegs;
        const int x0 = strip * C::TW;
        const int y0 = seg * p.seg_h;
        const int y1 = min(y0 + p.seg_h, H);
        const int nout = y1 - y0;

        const unsigned img_x = (unsigned)(H * W) * (unsigned)jb.xs;
        const unsigned img_z = (unsigned)(H * W) * (unsigned)jb.zs;
        const __amdgpu_buffer_rsrc_t rsrc_x =
            __builtin_amdgcn_make_buffer_rsrc((void*)(jb.x + (long long)n * img_x), 0, (int)img_x, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_z =
            __builtin_amdgcn_make_buffer_rsrc((void*)(jb.dz + (long long)n * img_z), 0, (int)img_z, 0x00020000);

        // per-lane source offsets of this wave's DMA pieces (constant over the strip)
        unsigned xoff[C::MAXPX], zoff[C::MAXPZ];
#pragma unroll
        for (int m = 0; m < C::MAXPX; ++m) {
            const int L = (wave + m * C::NWAVES) * 64 + lane;  // 16-B chunk index inside the row slot
            constexpr int CPP = C::XPB / 16;
            const int px = L / CPP, s = L % CPP;
            const int c = (C::XPB == 128) ? (s ^ (((px >> 1) & 1) << 2)) : s;
            const int col = x0 - 1 + px;
            const bool ok = col >= 0 && col < W && px < C::TW + 2;
            xoff[m] = ok ? (unsigned)(col * jb.xs + jb.xco + c * 16) : kOob;
        }
#pragma unroll
        for (int m = 0; m < C::MAXPZ; ++m) {
            const int L = (wave + m * C::NWAVES) * 64 + lane;
            constexpr int CPP = C::ZPB / 16;
            const int px = L / CPP, s = L % CPP;
            const int c = (C::ZPB == 128) ? (s ^ (((px >> 1) & 1) << 2)) : s;
            const int col = x0 + px;
            zoff[m] = col < W ? (unsigned)(col * jb.zs + jb.zco + c * 16) : kOob;
        }
        // x row r <-> image row y0 - 1 + r (r = 0 .. nout+1); dz row r <-> image row y0 + r.
        // Piece m (0 .. MAXPX+MAXPZ-1) of a row pair: x pieces first, then dz pieces; branch-free.
        auto x_soff = [&](const int r) {
            const int y = y0 - 1 + r;
            const bool in = (y >= 0) & (y < H) & (r < nout + 2);
            return in ? (unsigned)(y * W) * (unsigned)jb.xs : kOob;
        };
        auto z_soff = [&](const int r) { return (r < nout) ? (unsigned)((y0 + r) * W) * (unsigned)jb.zs : kOob; };
        auto issue_piece = [&](auto m_c, const unsigned xsoff, const unsigned xslot, const unsigned zsoff, const unsigned zslot) {
            constexpr int m = decltype(m_c)::value;
            if constexpr (m < C::MAXPX) {
                const int qq = wave + m * C::NWAVES;
                const unsigned off = (qq < C::XNP) ? xslot + (unsigned)qq * 1024u : (unsigned)C::SINK_OFF;
                wg_lds_dma(rsrc_x, (__attribute__((address_space(3))) void*)(smem + off), (qq < C::XNP) ? xoff[m] : kOob, xsoff);
            } else {
                constexpr int mz = m - C::MAXPX;
                const int qq = wave + mz * C::NWAVES;
                const unsigned off = (qq < C::ZNP) ? zslot + (unsigned)qq * 1024u : (unsigned)C::SINK_OFF;
                wg_lds_dma(rsrc_z, (__attribute__((address_space(3))) void*)(smem + off), (qq < C::ZNP) ? zoff[mz] : kOob, zsoff);
            }
        };

        __builtin_amdgcn_s_barrier();  // every wave is done reading the previous item's rows
#pragma unroll 1
        for (int r = 0; r < C::XA; ++r) {
            const unsigned so = x_soff(r);
            wg_static_for<0, C::MAXPX>([&](auto m_c) { issue_piece(m_c, so, (unsigned)(r * C::X_SLOT), 0u, 0u); });
        }
#pragma unroll 1
        for (int r = 0; r < C::ZA; ++r) {
            const unsigned so = z_soff(r);
            wg_static_for<C::MAXPX, C::MAXPX + C::MAXPZ>([&](auto m_c) { issue_piece(m_c, 0u, 0u, so, (unsigned)(C::Z_BASE + r * C::Z_SLOT)); });
        }
        // Iteration r works on x row r (image row y0 - 1 + r, r = 0 .. nout + 1): its fragments for the three column
        // shifts are read from LDS ONCE and multiplied with the dz rows r, r-1, r-2 (taps ky = 0, 1, 2), whose fragments
        // stay in registers for three iterations (register slot = row % 3; the loop is unrolled by three so that every
        // slot is a fixed set of registers).  That is one transposing LDS read pair per three MFMAs instead of one per MFMA: the
        // LDS pipe, which four waves reading a fragment per MFMA keep busy all the time, is left with a third of the load.
        // dz rows before the item's first are zeros; rows behind its last arrive as zeros from the DMA.
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < C::NKS; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { alo[j][k][e] = 0; ahi[j][k][e] = 0; }

        int ring = 0;   // ring slot of x row r and of dz row r (both rings have XRING slots)
        auto row_body = [&](auto ph_c, const int r) {
            SRHIP_CAPTURE(alo); SRHIP_CAPTURE(ahi); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(accb);
            constexpr int PH = decltype(ph_c)::value;   // r % 3: register slot of dz row r
            if (r < C::XA) wg_vm_wait<0>();
            else wg_vm_wait<C::TOP_WAIT>();
            __builtin_amdgcn_s_barrier();
            // rows fetched during this iteration go to the slots that row r-1 just left
            const unsigned dxs = x_soff(r + C::XA), dzs = z_soff(r + C::ZA);
            const unsigned dxslot = (unsigned)((ring == 0 ? C::XRING - 1 : ring - 1) * C::X_SLOT);
            const unsigned dzslot = (unsigned)(C::Z_BASE + (ring == 0 ? C::ZRING - 1 : ring - 1) * C::Z_SLOT);
            // per-row read bases: everything else is an immediate offset of the LDS instruction
            unsigned xa[3][2], za[2];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                za[j2] = zaddr[j2] + (unsigned)(ring * C::Z_SLOT);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) xa[kx][j2] = xaddr[kx][j2] + (unsigned)(ring * C::X_SLOT);
            }
            ring = (ring + 1 == C::XRING) ? 0 : ring + 1;

            // dz fragments of row r -> register slot PH
            wg_static_for<0, C::NKS>([&](auto s_c) {
                SRHIP_CAPTURE(alo); SRHIP_CAPTURE(ahi);
                constexpr int sk = decltype(s_c)::value;
                alo[PH][sk] = lds_read_tr<sk * 16 * C::ZPB>(za[0]);
                ahi[PH][sk] = lds_read_tr<sk * 16 * C::ZPB>(za[1]);
            });
            // x fragments run two steps ahead of the MFMAs
            // steps [ST0, NSTEP) = (tap column kx, k-step): all three columns, or the centre column alone
            constexpr int ST0 = C::KX0 * C::NKS, NSTEP = C::KX1 * C::NKS;
            constexpr int NP = C::MAXPX + C::MAXPZ, PPS = (NP + (NSTEP - ST0) - 1) / (NSTEP - ST0);
            bf16x4 blo[3], bhi[3];
            wg_static_for<ST0, ST0 + 2>([&](auto st_c) {
                SRHIP_CAPTURE(blo); SRHIP_CAPTURE(bhi);
                constexpr int st = decltype(st_c)::value;
                blo[st % 3] = lds_read_tr<(st % C::NKS) * 16 * C::XPB>(xa[st / C::NKS][0]);
                bhi[st % 3] = lds_read_tr<(st % C::NKS) * 16 * C::XPB>(xa[st / C::NKS][1]);
            });
            wg_static_for<ST0, NSTEP>([&](auto st_c) {
                SRHIP_CAPTURE(alo); SRHIP_CAPTURE(ahi); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(accb); SRHIP_CAPTURE(blo); SRHIP_CAPTURE(bhi);
                constexpr int st = decltype(st_c)::value;
                constexpr int kx = st / C::NKS, sk = st % C::NKS;
                constexpr int cur = st % 3;
                if constexpr (st + 2 < NSTEP) {
                    constexpr int sn = st + 2, nb = sn % 3;
                    blo[nb] = lds_read_tr<(sn % C::NKS) * 16 * C::XPB>(xa[sn / C::NKS][0]);
                    bhi[nb] = lds_read_tr<(sn % C::NKS) * 16 * C::XPB>(xa[sn / C::NKS][1]);
                    wg_lds_wait<4>();
                } else if constexpr (st + 1 < NSTEP) {
                    wg_lds_wait<2>();
                } else {
                    wg_lds_wait<0>();   // nothing stays in flight across the end of a row
                }
                if constexpr (st == ST0) {  // the dz reads are older than every x read
#pragma unroll
                    for (int k = 0; k < C::NKS; ++k) { SRHIP_LOAD_FENCE(alo[PH][k]); SRHIP_LOAD_FENCE(ahi[PH][k]); }
                }
                SRHIP_LOAD_FENCE(blo[cur]);  // consumers stay behind the wait
                SRHIP_LOAD_FENCE(bhi[cur]);
                const bf16x8 b = join(blo[cur], bhi[cur]);
                wg_static_for<(C::TAPS == 9 ? 0 : 1), (C::TAPS == 9 ? 3 : 2)>([&](auto ky_c) {
                    SRHIP_CAPTURE(alo); SRHIP_CAPTURE(ahi); SRHIP_CAPTURE(acc);
                    constexpr int ky = decltype(ky_c)::value;
                    constexpr int sl = (PH + 3 - ky) % 3;    // dz row r - ky
                    wg_mfma<ky * 3 + kx>(acc[ky * 3 + kx], join(alo[sl][sk], ahi[sl][sk]), b);
                });
                if constexpr (kx == 1) {
                    // bias gradient: the waves that share an output-channel tile take turns over the k-steps, so that
                    // every wave issues the same number of MFMAs per row (the reduction adds their partial sums)
                    if (cit == sk % C::CIT) wg_mfma<9>(accb, join(alo[PH][sk], ahi[PH][sk]), ones);
                }
                // the LDS-DMA pieces of the rows two iterations ahead, spread over the first steps
                wg_static_for<(st - ST0) * PPS, ((st - ST0 + 1) * PPS < NP ? (st - ST0 + 1) * PPS : NP)>([&](auto m_c) {
                    issue_piece(WIntC<decltype(m_c)::value + 0 * st>{}, dxs, dxslot, dzs, dzslot);
                });
            });
        };
#pragma unroll 1
        for (int r = 0; r < nout + 2; r += 3) {
            row_body(WIntC<0>{}, r);
            if (r + 1 < nout + 2) row_body(WIntC<1>{}, r + 1);
            if (r + 2 < nout + 2) row_body(WIntC<2>{}, r + 2);
        }
        wg_vm_wait<0>();  // look-ahead rows past the item's end were issued as all-zero rows
    }

    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // last MFMA results -> the reads below (hazard invisible inside asm)
    // ---- partial tiles -> workspace [workgroup][wave][10][16 registers][64 lanes]
    float* out = p.partial + ((size_t)(blockIdx.x * C::NWAVES + wave) * 10) * 1024 + lane;
#pragma unroll
    for (int t = (C::TAPS == 9 ? 0 : 4); t < (C::TAPS == 9 ? 9 : 5); ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(t * 16 + r) * 64] = acc[t][r];
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(9 * 16 + r) * 64] = accb[r];
}

// Sums the partial tiles in a fixed order and scatters them into the fp32 OIHW gradient.
// Accumulator register r of lane l holds row (r&3) + 8(r>>2) + 4(l>>5) (= co) and column l&31 (= ci).
// A partial tile of one tap is 4 KiB contiguous ([16 registers][64 lanes]).  A weight block = (output tile, tap, quarter q of
// the tile = registers 4q..4q+3 = rows 8q..8q+7): 16 thread groups of 64 lanes, group g sums partials g, g+16, ..., every
// lane one float4 of the quarter's contiguous 1 KiB (a 41 MB workspace read in 1-KiB pieces; as 128-byte runs, one per
// (co, tap, 32 ci), the same sum ran at 1.4 TB/s), then the 16 sums are added in a fixed order.  The blocks behind the
// weight blocks reduce the bias gradient, one element per lane (32 groups of 32 lanes).
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const ReduceParams rp)
{
    __shared__ float4 red[16][64];
    const int job = (int)blockIdx.x / rp.blocks_per_job, jblock = (int)blockIdx.x - job * rp.blocks_per_job;
    const WgradOut jo = rp.out[job];
    const float* __restrict__ partial = rp.partial + (size_t)job * rp.nwg * 4 * 10240;
    float* __restrict__ dw = jo.dw;
    float* __restrict__ db = jo.db;
    const int nwg = rp.nwg, cit_n = rp.cit_n, nt = rp.nt, w_i = rp.w_i, accumulate = rp.accumulate, nwblocks = rp.nwblocks;
    const int ci_off = jo.ci_off, ci_real = jo.ci_real, co_real = jo.co_real;
    const float scale = rp.scale;
    const int ks = 4 / nt;
    const int total = nwg * ks;
    if (jblock < nwblocks) {
        int b = jblock;
        const int q = b & 3; b >>= 2;
        const int tap = rp.tap0 + b % rp.ntaps, tile = b / rp.ntaps;
        const int cot = tile / cit_n, cit = tile - cot * cit_n;
        if (32 * cot + 8 * q >= co_real) return;   // the quarter's rows are all padding (uniform over the block)
        const int l64 = threadIdx.x & 63, grp = threadIdx.x >> 6;
        const int r = 4 * q + (l64 >> 4), l0 = (l64 & 15) * 4;
        const int ci0 = 32 * cit + (l0 & 31);
        const int co = 32 * cot + (r & 3) + 8 * (r >> 2) + 4 * (l0 >> 5);
        const bool live = ci0 < ci_real && co < co_real;
        float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (live) {
            const float* src = partial + (size_t)tap * 1024 + r * 64 + l0;
#pragma unroll 8
            for (int k = grp; k < total; k += 16) {   // unrolled: eight loads in flight, added in the same order
                const int wg = k / ks, wv = tile + (k % ks) * nt;
                const float4 v = *(const float4*)(src + (size_t)(wg * 4 + wv) * 10240);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        red[grp][l64] = s;
        __syncthreads();
        if (grp == 0 && live) {
            float4 t = red[0][l64];
#pragma unroll
            for (int k = 1; k < 16; ++k) { const float4 v = red[k][l64]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            const float tv[4] = {t.x * scale, t.y * scale, t.z * scale, t.w * scale};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ci0 + j < ci_real) {
                    float* dst = dw + ((size_t)co * w_i + ci_off + ci0 + j) * rp.dw_taps + (rp.dw_taps == 9 ? tap : 0);
                    *dst = accumulate ? *dst + tv[j] : tv[j];
                }
        }
        return;
    }
    // bias gradient: the dz x ones accumulators of the waves of each co tile (every ci tile holds a share)
    float* redb = (float*)red;   // [32 groups][32 lanes]
    const int o = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int co = (jblock - nwblocks) * 32 + o;
    int tile = 0, elem = -1;
    if (db != nullptr && co < co_real) {
        const int m = co & 31;
        tile = (co >> 5) * cit_n;
        elem = 9 * 1024 + ((m & 3) + 4 * (m >> 3)) * 64 + 32 * ((m >> 2) & 1);
    }
    float sb = 0.0f;
    if (elem >= 0) {
        for (int k = part; k < total; k += 32) {
            const int wg = k / ks, wv = tile + (k % ks) * nt;
            for (int c = 0; c < cit_n; ++c) sb += partial[(size_t)(wg * 4 + wv + c) * 10240 + elem];
        }
    }
    redb[part * 32 + o] = sb;
    __syncthreads();
    if (part == 0 && elem >= 0) {
        float t = redb[o];
#pragma unroll
        for (int k = 1; k < 32; ++k) t += redb[k * 32 + o];
        t *= scale;
        db[co] = accumulate ? db[co] + t : t;
    }
}

template <class C>
int launch_wgrad(WgradParams p, const WgradOut* outs, int w_i, int accumulate, float scale, hipStream_t stream)
{
    static std::atomic<bool> configured_dev[kSrhipMaxDevices];   // per device
    const int dev = srhip_current_device();
    if (!configured_dev[dev].load(std::memory_order_acquire)) {
        if (C::LDS_BYTES > 48 * 1024 &&
            hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS_BYTES) != hipSuccess) {
            srhip_set_error("conv3x3_wgrad: cannot raise the dynamic LDS limit to %d", C::LDS_BYTES);
            return SRHIP_ERR_LAUNCH;
        }
        configured_dev[dev].store(true, std::memory_order_release);
    }
    // every job gets the same share of the CUs (and of the workspace, which holds one partial tile set per CU)
    const int slots = srhip_num_cus() / p.njobs > 0 ? srhip_num_cus() / p.njobs : 1;
    const int nstrips = srhip_div_up(p.w, C::TW);
    // Segments per strip: the cost of a launch in row steps = rounds x (rows of a segment + halo + pipeline fill) + what
    // its partial tiles cost.  Every work-group that runs writes 160 KB of partial sums and the reduction reads them back
    // (about 10 us of the gradient launch and 10 us of the reduction for a full grid): a few hundredths of a 1.1 us row
    // step per work-group; 0.06 measured best on the temporal network's training step (0.03-0.16 within 1 % of each other,
    // no partial term at all 3 % slower).  On the small maps of that network and of RepRFN's crops the term outweighs the
    // arithmetic, and a grid of 64-128 work-groups is cheaper than one per CU.
    int best_segs = 1;
    double best_cost = 1e30;
    static const double pcost = [] { const char* e = getenv("SRHIP_WGRAD_PCOST"); return e ? atof(e) : 0.06; }();
    const long long base_items = (long long)nstrips * p.n;
    long long max_segs = (16ll * slots + base_items - 1) / base_items;
    if (max_segs > p.h) max_segs = p.h;
    if (max_segs < 1) max_segs = 1;
    int last_seg_h = -1;
    for (int want = 1; want <= (int)max_segs; ++want) {
        const int seg_h = srhip_div_up(p.h, want);
        if (seg_h == last_seg_h) continue;
        last_seg_h = seg_h;
        const int segs = srhip_div_up(p.h, seg_h);
        const long long items = (long long)segs * base_items;
        const int r = srhip_div_up(items, slots);
        const long long wgs = items < slots ? items : slots;
        const double cost = (double)r * (seg_h + 2 + 6.0) + pcost * (double)wgs * p.njobs;
        if (cost < best_cost) { best_cost = cost; best_segs = segs; }
    }
    p.seg_h = srhip_div_up(p.h, best_segs);
    p.nsegs = srhip_div_up(p.h, p.seg_h);
    p.nstrips = nstrips;
    const long long total = (long long)p.nsegs * nstrips * p.n;
    SRHIP_REQUIRE(total < (1ll << 30), "conv3x3_wgrad: too many work items");
    p.total_items = (int)total;
    p.wgs_per_job = (int)(total < slots ? total : slots);
    const int grid = p.wgs_per_job * p.njobs;
    hipLaunchKernelGGL(conv3x3_wgrad_kernel<C>, dim3(grid), dim3(C::NWAVES * 64), C::LDS_BYTES, stream, p);
    int rc = srhip_check_launch("conv3x3_wgrad_kernel");
    if (rc != SRHIP_OK) return rc;
    ReduceParams rp;
    rp.partial = p.partial;
    rp.nwg = p.wgs_per_job;
    rp.cit_n = C::CIT; rp.nt = C::NT; rp.w_i = w_i; rp.accumulate = accumulate; rp.scale = scale;
    rp.tap0 = C::TAPS == 9 ? 0 : 4; rp.ntaps = C::TAPS; rp.dw_taps = C::TAPS;
    rp.nwblocks = C::NT * C::TAPS * 4;
    rp.blocks_per_job = rp.nwblocks + C::COT;   // + one bias block per 32 output channels
    for (int j = 0; j < p.njobs; ++j) rp.out[j] = outs[j];
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rp.blocks_per_job * p.njobs), dim3(1024), 0, stream, rp);
    return srhip_check_launch("wgrad_reduce_kernel");
}

}  // namespace

extern "C" {

SRHIP_API size_t srhip_conv3x3_wgrad_workspace_bytes(void)
{
    return (size_t)srhip_num_cus() * 4 * 10 * 1024 * sizeof(float);
}

static int wgrad_multi(const srhip_wgrad_job* jobs, int njobs, void* workspace, int n, int h, int w, int ci_pad, int co_pad,
                       int w_i, int accumulate, float scale, int taps, void* stream)
{
    SRHIP_REQUIRE(jobs && workspace, "conv3x3_wgrad: null pointer");
    SRHIP_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "conv3x3_wgrad: %d jobs (1 .. %d per call)", njobs, kMaxJobs);
    SRHIP_REQUIRE(n >= 1 && h >= 1 && w >= 1, "conv3x3_wgrad: empty tensor");
    WgradParams p;
    WgradOut outs[kMaxJobs];
    p.partial = (float*)workspace;
    p.n = n; p.h = h; p.w = w;
    p.njobs = njobs;
    for (int j = 0; j < njobs; ++j) {
        const srhip_wgrad_job& a = jobs[j];
        SRHIP_REQUIRE(a.x && a.dz && a.dw, "conv3x3_wgrad: null pointer (job %d)", j);
        SRHIP_REQUIRE(a.ci_real >= 1 && a.ci_real <= ci_pad && a.co_real >= 1 && a.co_real <= co_pad, "conv3x3_wgrad: real channels exceed padded (job %d)", j);
        SRHIP_REQUIRE(a.ci_off >= 0 && a.ci_off + a.ci_real <= w_i, "conv3x3_wgrad: input-channel window [%d,%d) outside the weight's %d", a.ci_off, a.ci_off + a.ci_real, w_i);
        SRHIP_REQUIRE(a.x_cstride >= 8 && a.x_cstride % 8 == 0 && a.x_coff >= 0 && a.x_coff % 8 == 0 && a.x_coff + a.ci_real <= a.x_cstride,
                      "conv3x3_wgrad: x window [%d,%d) outside the map's %d channels (offsets and widths in multiples of 8)", a.x_coff, a.x_coff + a.ci_real, a.x_cstride);
        SRHIP_REQUIRE(a.dz_cstride >= 8 && a.dz_cstride % 8 == 0 && a.dz_coff >= 0 && a.dz_coff % 8 == 0 && a.dz_coff + a.co_real <= a.dz_cstride,
                      "conv3x3_wgrad: dz window [%d,%d) outside the map's %d channels (offsets and widths in multiples of 8)", a.dz_coff, a.dz_coff + a.co_real, a.dz_cstride);
        SRHIP_REQUIRE((long long)h * w * (a.x_cstride > a.dz_cstride ? a.x_cstride : a.dz_cstride) * 2 < (1ll << 31), "conv3x3_wgrad: image too large for 32-bit offsets");
        p.jobs[j].x = (const char*)a.x; p.jobs[j].dz = (const char*)a.dz;
        p.jobs[j].xs = a.x_cstride * 2; p.jobs[j].xco = a.x_coff * 2; p.jobs[j].zs = a.dz_cstride * 2; p.jobs[j].zco = a.dz_coff * 2;
        outs[j].dw = a.dw; outs[j].db = a.db; outs[j].ci_off = a.ci_off; outs[j].ci_real = a.ci_real; outs[j].co_real = a.co_real; outs[j].pad = 0;
    }
    for (int j = njobs; j < kMaxJobs; ++j) { p.jobs[j] = p.jobs[0]; }
    hipStream_t s = (hipStream_t)stream;
#define SRHIP_WGRAD_CASE(CI, CO) \
    if (ci_pad == CI && co_pad == CO) \
        return taps == 9 ? launch_wgrad<WCfg<CI, CO, 9>>(p, outs, w_i, accumulate, scale, s) \
                         : launch_wgrad<WCfg<CI, CO, 1>>(p, outs, w_i, accumulate, scale, s);
    SRHIP_WGRAD_CASE(64, 64)
    SRHIP_WGRAD_CASE(32, 64)
    SRHIP_WGRAD_CASE(16, 64)
    SRHIP_WGRAD_CASE(64, 32)
    SRHIP_WGRAD_CASE(32, 32)
    SRHIP_WGRAD_CASE(16, 32)
    SRHIP_WGRAD_CASE(64, 16)
    SRHIP_WGRAD_CASE(32, 16)
    SRHIP_WGRAD_CASE(16, 16)
#undef SRHIP_WGRAD_CASE
    srhip_set_error("conv3x3_wgrad: unsupported padded channels ci=%d co=%d (each must be 16, 32 or 64)", ci_pad, co_pad);
    return SRHIP_ERR_INVALID;
}

SRHIP_API int srhip_conv3x3_wgrad_multi(const srhip_wgrad_job* jobs, int njobs, void* workspace, int n, int h, int w,
                                        int ci_pad, int co_pad, int w_i, int accumulate, float scale, void* stream)
{
    return wgrad_multi(jobs, njobs, workspace, n, h, w, ci_pad, co_pad, w_i, accumulate, scale, 9, stream);
}

SRHIP_API int srhip_conv1x1_wgrad_multi(const srhip_wgrad_job* jobs, int njobs, void* workspace, int n, int h, int w,
                                        int ci_pad, int co_pad, int w_i, int accumulate, float scale, void* stream)
{
    return wgrad_multi(jobs, njobs, workspace, n, h, w, ci_pad, co_pad, w_i, accumulate, scale, 1, stream);
}

SRHIP_API int srhip_conv3x3_wgrad_windows(const void* x, const void* dz, float* dw, float* db, void* workspace, int n, int h,
                                          int w, int ci_pad, int ci_real, int ci_off, int w_i, int co_pad, int co_real,
                                          int accumulate, float scale, int x_cstride, int x_coff, int dz_cstride, int dz_coff,
                                          void* stream)
{
    srhip_wgrad_job job;
    job.x = x; job.dz = dz; job.dw = dw; job.db = db;
    job.ci_real = ci_real; job.ci_off = ci_off; job.co_real = co_real;
    job.x_cstride = x_cstride; job.x_coff = x_coff; job.dz_cstride = dz_cstride; job.dz_coff = dz_coff;
    job.reserved = 0;
    return srhip_conv3x3_wgrad_multi(&job, 1, workspace, n, h, w, ci_pad, co_pad, w_i, accumulate, scale, stream);
}

SRHIP_API int srhip_conv3x3_wgrad(const void* x, const void* dz, float* dw, float* db, void* workspace, int n, int h,
                                  int w, int ci_pad, int ci_real, int ci_off, int w_i, int co_pad, int co_real,
                                  int accumulate, float scale, void* stream)
{
    return srhip_conv3x3_wgrad_windows(x, dz, dw, db, workspace, n, h, w, ci_pad, ci_real, ci_off, w_i, co_pad, co_real, accumulate, scale,
                                       ci_pad, 0, co_pad, 0, stream);
}

}  // extern "C"
