// General convolution as an implicit GEMM on the matrix cores, for the layers that are not the 3x3 / stride 1 /
// <= 64-channel shape of conv3x3_mfma.hip: the temporal network's wide, strided and 1x1 layers
// (core/models/model_final.py:237-246 G-buffer encoder 160 -> 160 1x1, :272-276 / :329-363 encoder-decoder layers
// with up to 384 input and 256 output channels, 3x3 stride 2 down-sampling) and RepRFN's ESA 1x1 / strided layers
// (core/arch/RepRFN.py:285-291).
//
//   dst = alpha * act(conv_{k x k, stride s, pad p}(concat(src_0, src_1, src_2)) + bias) + res
//
// NHWC bf16 activations, fp32 accumulation over the WHOLE reduction (all taps, all input channels of all sources) in
// one launch: no partial sums leave the registers.  k in {1, 3}, s in {1, 2}, p in {0, 1}.
//
//   * A work-group (4 waves, two work-groups per CU) computes 2 output rows x 128 pixels x up to 96 output channels
//     (NT <= 3 MFMA tiles of 32; wider layers run as channel groups).  Wave w owns row w / 2, pixels 64 (w % 2) .. + 63 as
//     two MFMA pixel tiles: 2 x NT accumulators of 16 registers, and every weight fragment read from LDS feeds two MFMAs.
//   * The reduction runs over stages of 16 input channels of the concatenated input (64 for 1x1 layers).  Per stage the
//     work-group stages in LDS the input window (((2-1)s + k) rows x ((128-1)s + k) pixels, rows and columns outside the
//     image as zeros: buffer loads whose offset fails the bounds check) and the stage's weights for all k*k taps (1 KiB
//     fragments already in MFMA operand order), then every wave runs k*k steps of (2 + NT fragment reads -> 2 NT MFMAs),
//     the reads of step i + 1 requested before the MFMAs of step i with counted waits.  The next stage's window loads are
//     issued in front of the MFMA section and held in registers, so L2 latency overlaps the matrix work; the second
//     work-group of the CU covers the commit phase and the barriers.
//   * The weights do not pass through registers: a fragment is one buffer LDS-DMA instruction (1 KiB, contiguous in the
//     packed weights) into the weight buffer of the next stage's parity, issued beside the window loads.  (A 16-byte LDS
//     store costs 13 cycles of the register-to-LDS path per wave instruction, a third of the rate of the reads: with the
//     weights committed from registers the commit phase was a fifth of a 160 -> 128 launch.)
//   * LDS image of the window: [row][chunk][plane][slot][16 B], plane = which 8 of the 16 channels (x pixel parity for
//     stride 2): each half-wave's fragment read is 512 contiguous bytes, free of bank conflicts.
//   * Work-groups walk XCD bands (unit = (g % 8) * ceil(units / 8) + g / 8, channel group fastest): input rows shared by
//     neighbouring output rows and by a tile's channel groups hit in the XCD's own L2.
//   * The concatenation of up to three maps (skip connections, model_final.py:375-379) is read in place: a chunk
//     comes from the map that holds it.
//   * Epilogue straight from the accumulators: lane (pixel, 4 consecutive channels) -> bias, activation, alpha,
//     residual, bf16 pack, one 8-byte store.
// -DGEMM_EXP=<bits> builds timing-only variants: 1 no weight loads, 2 no input loads after the first stage, 4 no stores.
#include <atomic>
#include <cstdlib>

#include "srhip_common.h"

namespace {

constexpr int TPX = 128;    // output pixels per row of a work-group's tile
constexpr int MAXNT = 3;    // most 32-channel output tiles per work-group (a template parameter of the kernel)
constexpr int NTHREADS = 256;
constexpr unsigned kOobOff = 0x80000000u;   // buffer offset that fails the bounds check: the load returns zeros

struct GemmSrc {
    const char* p;
    int c;         // channels read from this map (multiple of 16)
    int cstride;   // channels per pixel of the tensor it lives in
    unsigned bytes;   // size of the tensor (< 2^31)
};

struct GemmParams {
    GemmSrc src[3];
    int nsrc;
    const char* wpack;
    const float* bias;
    const char* res;
    char* dst;
    int n, h, w, ho, wo;
    int kc_total, ct_total;   // 16-channel input chunks, 32-channel output tiles of the packed weights
    int cout, cout_w, cout_stride;
    int pad;
    float slope, alpha;
    unsigned wbytes;   // size of the packed weights
    unsigned units;    // work-groups that have work
    int groups;        // output-channel groups of NT tiles
    int vec16;         // dst pixel rows are 16-byte aligned: the epilogue goes through LDS to 16-byte pixel-major stores
    int res_mask;      // res is a ReLU mask (result zeroed where res <= 0) instead of an addend
};

// ROWS = output rows per work-group: 2 (wave w: row w / 2, 64 pixels = two MFMA pixel tiles) or 4 (wave w: row w, all 128
// pixels = four pixel tiles; 3x3 stride 1 layers on large images: a weight fragment feeds four MFMAs, the window is 6
// rows for 4 instead of 4 for 2, and a stage holds twice the matrix work per barrier pair)
template <int KS, int S, int NT, int ROWS_ = 2>
struct GCfg {
    static constexpr int ROWS = ROWS_;
    static constexpr int NPT = ROWS == 2 ? 2 : 4;                   // MFMA pixel tiles per wave
    static_assert(ROWS == 2 || (ROWS == 4 && KS == 3 && S == 1 && NT <= 2), "four-row tiles: 3x3 stride 1, at most 64 output channels per wave");
    static constexpr int T = KS * KS;
    // 16-channel chunks per LDS stage: 4 for 1x1 layers (64 channels), 1 for 3x3 (two for the narrow-output stride-1 layers,
    // whole 64-byte sectors per pixel, was slower: 0.295 vs 0.245 ms on the 160 -> 128 layer)
    static constexpr int KSUB = KS == 1 ? 4 : 1;
    static constexpr int NSTEP = T * KSUB;                          // (chunk, tap) steps per stage, NT MFMAs each
    static constexpr int PX = (TPX - 1) * S + KS;                   // input pixels per staged row
    static constexpr int KR = (ROWS - 1) * S + KS;                  // staged input rows
    static constexpr int XP1 = KR * PX * 2;                         // 16-byte pieces of the input window of ONE chunk
    static constexpr int NXP1 = (XP1 + NTHREADS - 1) / NTHREADS;    // per thread and chunk
    static constexpr int NXP = NXP1 * KSUB;
    static constexpr int WFRAGS = NSTEP * NT;                       // 1-KiB weight fragments per stage
    static constexpr int NWP = (WFRAGS + 3) / 4;                    // LDS-DMA pieces per wave and stage
    // LDS image of the input window: [row][chunk][plane][slot][16 B].  A lane's fragment is the 16 bytes (8 channels) of
    // one pixel; lanes 0..31 of a wave read 32 consecutive output pixels, lanes 32..63 the other 8 channels of the same
    // pixels.  plane = which 8 channels (x pixel parity for stride 2), slot = pixel (pixel / 2 for stride 2), so each
    // half-wave reads 512 contiguous bytes: no bank conflicts (with pixel-major 32-byte cells every read was 2-way, for
    // stride 2 4-way, conflicted: 86 % of the LDS cycles of the first version).
    static constexpr int NPL = 2 * S;
    static constexpr int PSL = (PX + S - 1) / S;
    static constexpr int X_BYTES = KSUB * KR * NPL * PSL * 16;
    static constexpr int W_BYTES = WFRAGS * 1024;                   // [chunk][tap][tile][1 KiB], two buffers (stage parity)
    static constexpr int STAGE_LDS = X_BYTES + 2 * W_BYTES;
    // epilogue staging: per wave 64 pixels x (NT * 32 channels bf16 + 16 B padding), pixel-major
    static constexpr int OUT_ROWB = NT * 64 + 16;
    static constexpr int OUT_WAVE = 64 * OUT_ROWB;
    static constexpr int BIAS_OFF = STAGE_LDS > 4 * OUT_WAVE ? STAGE_LDS : 4 * OUT_WAVE;   // the group's NT * 32 biases (fp32)
    static constexpr int LDS_BYTES = BIAS_OFF + NT * 32 * 4;
    static_assert(W_BYTES < 65536, "weight fragments are addressed by the LDS instruction's 16-bit offset");
    static_assert(2 * LDS_BYTES <= 160 * 1024 || (KS == 1 && S == 2), "two work-groups share a CU's LDS (the strided 1x1 form stages three rows of 64 channels: one work-group per CU)");
};

// loops whose index must be a compile-time constant from the start (register arrays stay registers)
template <int V>
struct GInt { static constexpr int value = V; };
template <int B, int E, class F>
__device__ __forceinline__ void g_for(F&& f)
{
    if constexpr (B < E) {
        f(GInt<B>{});
        g_for<B + 1, E>(f);
    }
}

// LDS fragment read with an immediate offset, as asm: the waits of the MFMA section are counted by hand (the compiler
// keeps only one or two reads in flight ahead of each MFMA, far less than the LDS latency)
template <int OFF>
__device__ __forceinline__ bf16x8 g_lds_read(unsigned addr)
{
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
    return r;
}
template <int N>
__device__ __forceinline__ void g_lds_wait()
{
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gemm_rsrc(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 gemm_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int KS, int S, int NT, int ROWS>
__global__ __launch_bounds__(NTHREADS, 2) void conv_gemm_kernel(const GemmParams p)
{
    using C = GCfg<KS, S, NT, ROWS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xt = smem;
    char* wt = smem + C::X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // work-group -> unit = (output-channel group, strip of the output rows, pair of output rows, image), group fastest.
    // Units are walked in XCD bands: the dispatcher deals consecutive work-groups round-robin to the 8 XCDs (one L2
    // each), so work-group g takes unit (g % 8) * ceil(units / 8) + g / 8 and every XCD owns a contiguous band of rows:
    // the input rows that neighbouring output rows share, and the re-read of a tile's input by its channel groups, hit in
    // that XCD's L2 (with the natural order a 167 MB input was fetched as 506 MB and the loads' latency showed).
    const int strips = (p.wo + TPX - 1) / TPX, rowpairs = (p.ho + ROWS - 1) / ROWS;   // (groups of ROWS output rows)
    const unsigned per = (p.units + 7u) >> 3;
    const unsigned unit = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (unit >= p.units) return;
    int b = (int)unit;
    const int grp = b % p.groups; b /= p.groups;
    const int strip = b % strips; b /= strips;
    const int yo = (b % rowpairs) * ROWS;
    const int img = b / rowpairs;
    const int xo0 = strip * TPX;
    const int ct0 = grp * NT;
    const int nt = min(NT, p.ct_total - ct0);   // < NT only in the last group: its missing tiles read zero weights

    // the group's biases -> LDS: the epilogue reads them there (a global load per accumulator quad would be a memory round
    // trip of its own at a point where nothing is left to hide it)
    if (tid < NT * 32) {
        const int c = ct0 * 32 + tid;
        ((float*)(smem + C::BIAS_OFF))[tid] = (p.bias != nullptr && c < p.cout) ? p.bias[c] : 0.0f;
    }
    // All global loads are buffer loads whose out-of-image / out-of-range pieces carry an offset that fails the bounds
    // check and come back as zeros: no branches in the load phase.
    // per-thread pieces of the input window: pixel index inside the image, constant over the chunks
    // (the same table serves every chunk of a stage: piece m of chunk `sub` is register m + sub * NXP1)
    int xpix[C::NXP1];
    unsigned xhalf[C::NXP1], xlds[C::NXP1];
    g_for<0, C::NXP1>([&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        const int q = tid + m * NTHREADS;
        const int ky = q / (C::PX * 2), rem = q - ky * (C::PX * 2);
        const int px = rem >> 1;
        const int yy = yo * S + ky - p.pad, xx = xo0 * S + px - p.pad;
        const bool ok = (q < C::XP1) & (yy >= 0) & (yy < p.h) & (xx >= 0) & (xx < p.w);
        xpix[m] = ok ? (img * p.h + yy) * p.w + xx : -1;
        xhalf[m] = (unsigned)(rem & 1) * 16u;
        const int plane = (rem & 1) * S + (S == 2 ? (px & 1) : 0), slot = S == 2 ? (px >> 1) : px;
        xlds[m] = (unsigned)(((ky * C::KSUB * C::NPL + plane) * C::PSL + slot) * 16);   // + sub * NPL * PSL * 16
    });
    // The stage's weights go global -> LDS by buffer LDS-DMA, 1 KiB (one MFMA fragment, already in operand order) per
    // instruction: fragment f = (chunk, tap, tile) of the stage belongs to wave f % 4.  Per-lane source offset inside the
    // packed weights (a tile the last group does not have fails the bounds check: zeros), constant over the stages.
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    unsigned woff[C::NWP];
    int wsub[C::NWP];
    g_for<0, C::NWP>([&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        const int f = wv + 4 * m;
        const int step = f / NT, ct = f - step * NT;
        const int sub = step / C::T, tap = step - sub * C::T;
        const bool ok = (f < C::WFRAGS) & (ct < nt);
        woff[m] = ok ? (unsigned)(((tap * p.kc_total + sub) * p.ct_total + ct0 + ct) * 1024 + lane * 16) : kOobOff;
        wsub[m] = sub;
    });
    const __amdgpu_buffer_rsrc_t wsrd = gemm_rsrc(p.wpack, p.wbytes);
    const __amdgpu_buffer_rsrc_t xsrd0 = gemm_rsrc(p.src[0].p, p.src[0].bytes);
    const __amdgpu_buffer_rsrc_t xsrd1 = gemm_rsrc(p.nsrc > 1 ? p.src[1].p : p.src[0].p, p.nsrc > 1 ? p.src[1].bytes : 0u);
    const __amdgpu_buffer_rsrc_t xsrd2 = gemm_rsrc(p.nsrc > 2 ? p.src[2].p : p.src[0].p, p.nsrc > 2 ? p.src[2].bytes : 0u);

    // scalars of the argument block that the chunk loop needs (plain locals: the block itself is never indexed there)
    const int nsrc = p.nsrc, sc0 = p.src[0].c, sc1 = p.src[1].c;
    const int st0 = p.src[0].cstride, st1 = p.src[1].cstride, st2 = p.src[2].cstride;
    const int kc_total = p.kc_total, ct_total = p.ct_total;
    uint4 xreg[C::NXP];

    // wave w: output row yo + (w >> 1), pixels 64 (w & 1) .. + 63 of the strip as two MFMA pixel tiles; every weight
    // fragment read from LDS feeds two MFMAs
    f32x16 acc[C::NPT][NT];
#pragma unroll
    for (int pt = 0; pt < C::NPT; ++pt)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pt][ct][r] = 0.0f;

    const int wrow = ROWS == 2 ? wave >> 1 : wave;
    const int wpx0 = ROWS == 2 ? (wave & 1) * 64 : 0;   // the wave's first pixel inside the strip
    const int pl = wpx0 + (lane & 31);              // this lane's first output pixel inside the strip (the others: + 32 each)
    const int hl = lane >> 5;                       // which 8 of the chunk's 16 channels this lane feeds

    // Iteration st (a stage = KSUB chunks): [commit stage st to LDS] -> issue the global loads of stage st + 1 -> MFMAs
    // of stage st.  Iteration -1 only issues the loads of stage 0.  (Load and commit code exist once, inline: as lambdas
    // called from two places the compiler indexed their closure dynamically and every register array behind it moved to
    // scratch.)
    const int nstage = (kc_total + C::KSUB - 1) / C::KSUB;
    const unsigned baddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem +
                           (unsigned)((((wrow * S * C::KSUB * 2 + hl) * S) * C::PSL + pl) * 16);
    const unsigned aaddr0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(C::X_BYTES + lane * 16);
    for (int st = -1; st < nstage; ++st) {
        if (st >= 0) {
            // this wave's weight fragments of stage st have landed in LDS (and its window pieces in registers); the
            // barriers below publish them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();   // everyone is done reading the previous stage
            g_for<0, C::NXP>([&](auto m_c) {
                constexpr int m = decltype(m_c)::value;
                constexpr int sub = m / C::NXP1, m1 = m % C::NXP1;
                const int q = tid + m1 * NTHREADS;
                if (C::XP1 % NTHREADS == 0 || q < C::XP1) *(uint4*)(xt + xlds[m1] + sub * C::NPL * C::PSL * 16) = xreg[m];
            });
            __syncthreads();
        }
        {
            // the last iteration repeats its own stage: harmless, and it keeps the code free of a branch around the loads
            const int ns = st + 1 < nstage ? st + 1 : st;
            g_for<0, C::KSUB>([&](auto s_c) {
                constexpr int sub = decltype(s_c)::value;
                // the map that holds input chunk kc, and the chunk's first channel inside it
                const int kc = ns * C::KSUB + sub;
                int coff = kc * 16;
                int cstride = st0;
                __amdgpu_buffer_rsrc_t xs = xsrd0;
                if (nsrc > 1 && coff >= sc0) {
                    coff -= sc0; cstride = st1; xs = xsrd1;
                    if (nsrc > 2 && coff >= sc1) { coff -= sc1; cstride = st2; xs = xsrd2; }
                }
                const bool live = kc < kc_total;   // a stage may reach past the last chunk: zeros
                const unsigned cbytes = (unsigned)cstride * 2u, soff = (unsigned)coff * 2u;
#if defined(GEMM_EXP) && (GEMM_EXP & 2)
                if (st < 0)
#endif
                g_for<0, C::NXP1>([&](auto m_c) {
                    constexpr int m = decltype(m_c)::value;
                    xreg[sub * C::NXP1 + m] = gemm_load16(xs, (live && xpix[m] >= 0) ? (unsigned)xpix[m] * cbytes + xhalf[m] : kOobOff, soff);
                });
            });
            // weights of stage st + 1 -> the buffer that stage st - 1 used (every wave is past this iteration's barriers)
            const unsigned wsoff = (unsigned)(ns * C::KSUB * ct_total) * 1024u;
#if defined(GEMM_EXP) && (GEMM_EXP & 1)
            if (st < 0)
#endif
            if (st + 1 < nstage) {
                char* wdst = wt + ((st + 1) & 1) * C::W_BYTES + wv * 1024;
                g_for<0, C::NWP>([&](auto m_c) {
                    constexpr int m = decltype(m_c)::value;
                    const bool live = C::KSUB == 1 || ns * C::KSUB + wsub[m] < kc_total;
                    if (C::WFRAGS % 4 == 0 || wv + 4 * m < C::WFRAGS)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (__attribute__((address_space(3))) void*)(wdst + m * 4096), 16,
                                                                 (int)(live ? woff[m] : kOobOff), (int)wsoff, 0, 0);
                });
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // the loads above stay in front of the MFMA section
        if (st < 0) continue;
        // NSTEP steps of (2 pixel fragments + NT weight fragments -> 2 NT MFMAs); the fragments of step i + 1 are
        // requested before the MFMAs of step i, with counted waits
        bf16x8 af[2][NT], bq[2][C::NPT];
        const unsigned aaddr = aaddr0 + (unsigned)((st & 1) * C::W_BYTES);
        g_for<0, C::NSTEP>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            auto issue = [&](auto j_c) __attribute__((always_inline)) {
                constexpr int j = decltype(j_c)::value;
                constexpr int sub = j / C::T, tap = j % C::T, ky = tap / KS, kx = tap % KS;
                constexpr int boff = (((ky * C::KSUB + sub) * C::NPL + (S == 2 ? (kx & 1) : 0)) * C::PSL + (S == 2 ? (kx >> 1) : kx)) * 16;
                g_for<0, C::NPT>([&](auto pt_c) {
                    constexpr int pt = decltype(pt_c)::value;
                    bq[j & 1][pt] = g_lds_read<boff + pt * 32 * 16>(baddr);
                });
                g_for<0, NT>([&](auto ct_c) {
                    constexpr int ct = decltype(ct_c)::value;
                    af[j & 1][ct] = g_lds_read<(j * NT + ct) * 1024>(aaddr);
                });
            };
            if constexpr (i == 0) issue(GInt<0>{});
            if constexpr (i + 1 < C::NSTEP) {
                issue(GInt<i + 1>{});
                g_lds_wait<NT + C::NPT>();
            } else {
                g_lds_wait<0>();   // nothing stays in flight across the barrier
            }
            g_for<0, C::NPT>([&](auto pt_c) { (void)&bq; SRHIP_LOAD_FENCE(bq[i & 1][decltype(pt_c)::value]); });   // (void)&: asm operands alone do not make a generic lambda capture
            g_for<0, NT>([&](auto ct_c) {
                constexpr int ct = decltype(ct_c)::value;
                SRHIP_LOAD_FENCE(af[i & 1][ct]);
                g_for<0, C::NPT>([&](auto pt_c) {
                    constexpr int pt = decltype(pt_c)::value;
                    acc[pt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i & 1][ct], bq[i & 1][pt], acc[pt][ct], 0, 0, 0);
                });
            });
        });
    }

    // epilogue: accumulator register r of lane l is output channel 32 ct + 8 (r >> 2) + 4 (l >> 5) + (r & 3) of pixel l & 31
    const int yrow = yo + wrow;
    if (p.vec16) {
        // Straight from the accumulators a lane holds 4 channels of one pixel, and a store instruction would write 64
        // separate 8-byte pieces (measured: the stores of a 160 -> 160 1x1 layer cost a third of its launch).  Instead the
        // wave transposes its 64 x (NT * 32) results through LDS and writes 16 bytes per lane, consecutive lanes =
        // consecutive channels of a pixel.
        __syncthreads();   // every wave is done with the last stage's window and weights
        char* ow = smem + wave * C::OUT_WAVE;
        // 64 pixels (two pixel tiles) at a time through the wave's own staging area: LDS operations of one wave run in
        // order, so the second half's writes follow the first half's read-back
#pragma unroll
        for (int half = 0; half < C::NPT / 2; ++half) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int pt = 2 * half + pp;
            const int px = 32 * pp + (lane & 31);
            const int xo = xo0 + pl + 32 * pt;
            const long long opix = ((long long)img * p.ho + yrow) * p.wo + xo;
            const bool live = xo < p.wo && yrow < p.ho;
            const char* rp = (p.res && live) ? p.res + opix * p.cout_stride * 2 : nullptr;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = ct * 32 + 8 * g + 4 * hl, c0 = ct0 * 32 + cl;
                    float v[4];
                    const float4 bia = *(const float4*)(smem + C::BIAS_OFF + cl * 4);
                    const float biav[4] = {bia.x, bia.y, bia.z, bia.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = acc[pt][ct][4 * g + e] + biav[e];
                        v[e] = fmaxf(t, t * p.slope);   // act(t) = max(t, slope * t), slope in {1, 0, s}
                    }
                    if (rp != nullptr && c0 < p.cout_w) {
                        const uint2 r = *(const uint2*)(rp + c0 * 2);
                        const float rf[4] = {bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y)};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = p.res_mask ? (rf[e] > 0.0f ? v[e] * p.alpha : 0.0f)
                                              : __builtin_fmaf(v[e], p.alpha, rf[e]);   // one rounding, whatever the contraction mode
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
                    }
                    *(uint2*)(ow + px * C::OUT_ROWB + cl * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                }
            }
        }
        // (the wave reads back only what it wrote itself: no barrier)
        constexpr int PPP = NT * 4;   // 16-byte pieces per pixel
        if (yrow < p.ho) {
#pragma unroll
        for (int j = 0; j < PPP; ++j) {
            const int i = lane + 64 * j;
            const int px = i / PPP, cg = i - px * PPP;
            const int xo = xo0 + wpx0 + 64 * half + px;
            const int c0 = ct0 * 32 + cg * 8;
            if (xo < p.wo && c0 < p.cout_w) {
                const long long opix = ((long long)img * p.ho + yrow) * p.wo + xo;
#if defined(GEMM_EXP) && (GEMM_EXP & 4)
                if (xo == -12345)
#endif
                *(uint4*)(p.dst + (opix * p.cout_stride + c0) * 2) = *(const uint4*)(ow + px * C::OUT_ROWB + cg * 16);
            }
        }
        }
        }
        return;
    }
    if (yrow >= p.ho) return;
#pragma unroll
    for (int pt = 0; pt < C::NPT; ++pt) {
        const int xo = xo0 + pl + 32 * pt;
        if (xo >= p.wo) continue;
        const long long opix = ((long long)img * p.ho + yrow) * p.wo + xo;
        char* dp = p.dst + opix * p.cout_stride * 2;
        const char* rp = p.res ? p.res + opix * p.cout_stride * 2 : nullptr;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = (ct0 + ct) * 32 + 8 * g + 4 * hl;
                if (c0 >= p.cout_w) continue;
                float v[4];
                const float4 bia = *(const float4*)(smem + C::BIAS_OFF + (c0 - ct0 * 32) * 4);
                const float biav[4] = {bia.x, bia.y, bia.z, bia.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[pt][ct][4 * g + e] + biav[e];
                    v[e] = fmaxf(t, t * p.slope);   // act(t) = max(t, slope * t), slope in {1, 0, s}
                }
                if (rp != nullptr) {
                    const uint2 r = *(const uint2*)(rp + c0 * 2);
                    const float rf[4] = {bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = p.res_mask ? (rf[e] > 0.0f ? v[e] * p.alpha : 0.0f) : __builtin_fmaf(v[e], p.alpha, rf[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
                }
                *(uint2*)(dp + c0 * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
    }
}

// fp32 OIHW (or OI11) master weights -> bf16 MFMA fragment order [tap][16-channel chunk][32-channel tile][lane][8].
// transposed: the packed layer is the data-gradient layer of the stored one -- its output channel co is the stored
// weight's input channel row0 + co, its input channel ci the stored output channel, its kernel the stored one flipped
// in both directions (w_i = input channels of the stored weight).
__global__ void gemm_pack_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout, int cin,
                                         int ks, int kc_total, int ct_total, int total, int transposed, int row0, int w_i)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 7;
    const int lane = (idx >> 3) & 63;
    int rest = idx >> 9;
    const int ct = rest % ct_total; rest /= ct_total;
    const int kc = rest % kc_total; rest /= kc_total;
    const int tap = rest;
    const int ky = tap / ks, kx = tap - ky * ks;
    const int co = ct * 32 + (lane & 31);
    const int ci = kc * 16 + 8 * (lane >> 5) + j;
    float v = 0.0f;
    if (co < cout && ci < cin)
        v = transposed ? w[(((long long)ci * w_i + row0 + co) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)]
                       : w[(((long long)co * cin + ci) * ks + ky) * ks + kx];
    out[idx] = f32_to_bf16_bits(v);
}

template <int KS, int S, int NT, int ROWS = 2>
int launch_gemm_nt(GemmParams p, int groups, hipStream_t stream)
{
    using C = GCfg<KS, S, NT, ROWS>;
    static std::atomic<bool> configured_dev[kSrhipMaxDevices];
    const int dev = srhip_current_device();
    if (!configured_dev[dev].load(std::memory_order_acquire)) {
        if (C::LDS_BYTES > 48 * 1024 &&
            hipFuncSetAttribute((const void*)conv_gemm_kernel<KS, S, NT, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
            srhip_set_error("conv_gemm: cannot raise the dynamic LDS limit to %d", C::LDS_BYTES);
            return SRHIP_ERR_LAUNCH;
        }
        configured_dev[dev].store(true, std::memory_order_release);
    }
    const long long units = (long long)srhip_div_up(p.wo, TPX) * srhip_div_up(p.ho, ROWS) * p.n * groups;
    SRHIP_REQUIRE(units < (1ll << 30), "conv_gemm: too many work-groups");
    p.units = (unsigned)units;
    p.groups = groups;
    hipLaunchKernelGGL((conv_gemm_kernel<KS, S, NT, ROWS>), dim3((unsigned)((units + 7) / 8 * 8)), dim3(NTHREADS), C::LDS_BYTES, stream, p);
    return srhip_check_launch("conv_gemm_kernel");
}

// Output tiles are spread evenly over the fewest groups of at most MAXNT tiles (5 tiles -> one group of 5, 8 -> two of 4,
// 7 -> 4 + 3: the last group's missing tile multiplies zero weights and stores nothing).
template <int KS, int S>
int launch_gemm(const GemmParams& p, hipStream_t stream)
{
    // Small images: fewer output tiles per work-group give more work-groups (each re-stages the input window, which the
    // XCD's L2 holds).  Measured at 68 x 120 and 136 x 240 (256 -> 256: 36 -> 28 us with one tile per group, 192 -> 192:
    // 45 -> 38 us with two); from about 400 work-groups on, the widest grouping wins again.
    // (stride 2: the staged window is 41 KB, and with two weight buffers of three tiles a work-group would take more than
    // half of the CU's LDS)
    if constexpr (KS == 3 && S == 1) {
        // Four-row tiles (at most two output tiles per wave: 128 accumulator registers beside the fragments) where they
        // were measured faster: 5 output tiles on a large image, which the two-row form runs as groups of 3 + 2
        // (160 -> 160 at 544 x 960: 340 -> 314 us; with 4 tiles, 2 + 2 either way, the two-row form is as fast or faster:
        // 160 -> 128 223 / 220 us, 112 -> 112 151 / 157, 128 -> 64 91 / 98).  SRHIP_GEMM_ROWS4 = 1 / 0 forces / forbids it.
        const char* e = getenv("SRHIP_GEMM_ROWS4");
        const int groups4 = srhip_div_up(p.ct_total, 2);
        const long long units4 = (long long)srhip_div_up(p.wo, TPX) * srhip_div_up(p.ho, 4) * p.n * groups4;
        const bool use4 = e ? atoi(e) == 1 : (p.ct_total == 5 && units4 >= 1024);
        if (use4) {
            if (srhip_div_up(p.ct_total, groups4) == 1) return launch_gemm_nt<KS, S, 1, 4>(p, groups4, stream);
            return launch_gemm_nt<KS, S, 2, 4>(p, groups4, stream);
        }
    }
    int maxnt = S == 2 ? 2 : MAXNT;
    const long long tiles = (long long)srhip_div_up(p.wo, TPX) * srhip_div_up(p.ho, 2) * p.n;
    while (maxnt > (S == 2 ? 2 : 1) && tiles * srhip_div_up(p.ct_total, maxnt) < 400) --maxnt;
    if (const char* e = getenv("SRHIP_GEMM_NT")) maxnt = atoi(e) >= 1 && atoi(e) <= (S == 2 ? 2 : MAXNT) ? atoi(e) : maxnt;   // timing experiments
    const int groups = srhip_div_up(p.ct_total, maxnt);
    const int nt = srhip_div_up(p.ct_total, groups);
    switch (nt) {
    case 1: return launch_gemm_nt<KS, S, 1>(p, groups, stream);
    case 2: return launch_gemm_nt<KS, S, 2>(p, groups, stream);
    default:
        if constexpr (S == 1) return launch_gemm_nt<KS, S, 3>(p, groups, stream);
        else return launch_gemm_nt<KS, S, 2>(p, groups, stream);
    }
}

}  // namespace

extern "C" {

SRHIP_API size_t srhip_conv_gemm_packed_bytes(int cin_padded, int cout, int ksize)
{
    if (cin_padded < 16 || cin_padded % 16 || cout < 1 || (ksize != 1 && ksize != 3)) return 0;
    return (size_t)ksize * ksize * (cin_padded / 16) * ((cout + 31) / 32) * 1024;
}

SRHIP_API int srhip_conv_gemm_pack_weights(const float* weight, void* packed, int cout, int cin, int ksize, int cin_padded,
                                           void* stream)
{
    SRHIP_REQUIRE(weight && packed, "conv_gemm_pack_weights: null pointer");
    SRHIP_REQUIRE(cout >= 1 && cin >= 1 && (ksize == 1 || ksize == 3), "conv_gemm_pack_weights: bad shape");
    SRHIP_REQUIRE(cin_padded >= cin && cin_padded % 16 == 0, "conv_gemm_pack_weights: cin_padded must be a multiple of 16 and >= cin");
    const int kc_total = cin_padded / 16, ct_total = (cout + 31) / 32;
    const long long total = (long long)ksize * ksize * kc_total * ct_total * 512;
    SRHIP_REQUIRE(total < (1ll << 31), "conv_gemm_pack_weights: weight too large");
    hipLaunchKernelGGL(gemm_pack_weights_kernel, dim3(srhip_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, weight,
                       (unsigned short*)packed, cout, cin, ksize, kc_total, ct_total, (int)total, 0, 0, 0);
    return srhip_check_launch("gemm_pack_weights_kernel");
}

SRHIP_API int srhip_conv_gemm_pack_weights_transposed(const float* weight, void* packed, int w_o, int w_i, int ksize, int row0,
                                                      int rows, int cin_padded, void* stream)
{
    SRHIP_REQUIRE(weight && packed, "conv_gemm_pack_weights_transposed: null pointer");
    SRHIP_REQUIRE(w_o >= 1 && w_i >= 1 && (ksize == 1 || ksize == 3), "conv_gemm_pack_weights_transposed: bad shape");
    SRHIP_REQUIRE(row0 >= 0 && rows >= 1 && row0 + rows <= w_i, "conv_gemm_pack_weights_transposed: rows [%d,%d) outside the weight's %d input channels", row0, row0 + rows, w_i);
    SRHIP_REQUIRE(cin_padded >= w_o && cin_padded % 16 == 0, "conv_gemm_pack_weights_transposed: cin_padded must be a multiple of 16 and >= the weight's output channels");
    const int kc_total = cin_padded / 16, ct_total = (rows + 31) / 32;
    const long long total = (long long)ksize * ksize * kc_total * ct_total * 512;
    SRHIP_REQUIRE(total < (1ll << 31), "conv_gemm_pack_weights_transposed: weight too large");
    hipLaunchKernelGGL(gemm_pack_weights_kernel, dim3(srhip_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, weight,
                       (unsigned short*)packed, rows, w_o, ksize, kc_total, ct_total, (int)total, 1, row0, w_i);
    return srhip_check_launch("gemm_pack_weights_kernel");
}

SRHIP_API int srhip_conv_gemm_fwd(const srhip_conv_gemm_args* a, void* stream)
{
    SRHIP_REQUIRE(a != nullptr, "conv_gemm: null argument block");
    SRHIP_REQUIRE(a->nsrc >= 1 && a->nsrc <= 3 && a->wpack && a->dst, "conv_gemm: null pointer or bad source count");
    SRHIP_REQUIRE(a->n >= 1 && a->h >= 1 && a->w >= 1 && a->cout >= 1, "conv_gemm: empty tensor");
    SRHIP_REQUIRE((a->ksize == 1 || a->ksize == 3) && (a->stride == 1 || a->stride == 2) && (a->pad == 0 || a->pad == 1) && a->pad <= a->ksize / 2,
                  "conv_gemm: kernel %d stride %d pad %d not supported (k in {1,3}, s in {1,2}, p in {0,1})", a->ksize, a->stride, a->pad);
    SRHIP_REQUIRE(a->act == SRHIP_ACT_NONE || a->act == SRHIP_ACT_RELU || a->act == SRHIP_ACT_LRELU, "conv_gemm: bad activation %d", a->act);
    SRHIP_REQUIRE(a->alpha > 0.0f, "conv_gemm: alpha must be positive");
    GemmParams p;
    int cin_pad = 0;
    for (int k = 0; k < 3; ++k) {
        p.src[k].p = nullptr; p.src[k].c = 0; p.src[k].cstride = 0; p.src[k].bytes = 0;
        if (k < a->nsrc) {
            SRHIP_REQUIRE(a->src[k] != nullptr && a->c[k] >= 16 && a->c[k] % 16 == 0 && a->cstride[k] >= a->c[k] && a->cstride[k] % 8 == 0,
                          "conv_gemm: source %d: channels %d (multiple of 16) of a tensor with %d per pixel (multiple of 8)", k, a->c[k], a->cstride[k]);
            SRHIP_REQUIRE((size_t)a->src[k] % 16 == 0, "conv_gemm: source %d is not 16-byte aligned", k);
            SRHIP_REQUIRE((long long)a->n * a->h * a->w < (1ll << 31), "conv_gemm: too many pixels for 32-bit indices");
            p.src[k].p = (const char*)a->src[k]; p.src[k].c = a->c[k]; p.src[k].cstride = a->cstride[k];
            const long long bytes = (long long)a->n * a->h * a->w * a->cstride[k] * 2;
            SRHIP_REQUIRE(bytes < (1ll << 31), "conv_gemm: source %d larger than 2 GiB", k);
            p.src[k].bytes = (unsigned)bytes;
            cin_pad += a->c[k];
        }
    }
    p.nsrc = a->nsrc;
    p.wpack = (const char*)a->wpack; p.bias = a->bias; p.res = (const char*)a->res; p.dst = (char*)a->dst;
    p.n = a->n; p.h = a->h; p.w = a->w;
    p.ho = (a->h + 2 * a->pad - a->ksize) / a->stride + 1;
    p.wo = (a->w + 2 * a->pad - a->ksize) / a->stride + 1;
    SRHIP_REQUIRE(p.ho >= 1 && p.wo >= 1, "conv_gemm: the kernel does not fit the image");
    p.kc_total = cin_pad / 16;
    p.ct_total = (a->cout + 31) / 32;
    p.cout = a->cout;
    p.cout_w = (a->cout + 15) / 16 * 16;
    p.cout_stride = a->cout_stride;
    SRHIP_REQUIRE(a->cout_stride >= p.cout_w && a->cout_stride % 4 == 0, "conv_gemm: cout_stride %d must hold pad16(cout) = %d channels", a->cout_stride, p.cout_w);
    SRHIP_REQUIRE((size_t)a->dst % 8 == 0 && (a->res == nullptr || (size_t)a->res % 8 == 0), "conv_gemm: dst / res must be 8-byte aligned");
    p.pad = a->pad;
    p.vec16 = ((size_t)a->dst % 16 == 0 && a->cout_stride % 8 == 0) ? 1 : 0;
    p.slope = a->act == SRHIP_ACT_NONE ? 1.0f : (a->act == SRHIP_ACT_RELU ? 0.0f : a->slope);
    p.alpha = a->alpha;
    SRHIP_REQUIRE(a->res_mode == SRHIP_RES_ADD || a->res_mode == SRHIP_RES_MASK, "conv_gemm: res_mode %d (add or mask)", a->res_mode);
    SRHIP_REQUIRE(a->res_mode == SRHIP_RES_ADD || a->res != nullptr, "conv_gemm: the mask form needs `res`");
    p.res_mask = a->res_mode == SRHIP_RES_MASK ? 1 : 0;
    {
        const size_t wb = srhip_conv_gemm_packed_bytes(cin_pad, a->cout, a->ksize);
        SRHIP_REQUIRE(wb > 0 && wb < (1ull << 31), "conv_gemm: packed weights too large");
        p.wbytes = (unsigned)wb;
    }
    hipStream_t s = (hipStream_t)stream;
    if (a->ksize == 3 && a->stride == 1) return launch_gemm<3, 1>(p, s);
    if (a->ksize == 3 && a->stride == 2) return launch_gemm<3, 2>(p, s);
    if (a->ksize == 1 && a->stride == 1) return launch_gemm<1, 1>(p, s);
    return launch_gemm<1, 2>(p, s);
}

}  // extern "C"
