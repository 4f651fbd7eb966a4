// 3x3 / stride 1 / pad 1 convolution for gfx950 as a streaming implicit GEMM on MFMA.
//
// Mapping (M = output channels, N = pixels, K = 9 taps x input channels):
//   * v_mfma_f32_32x32x16_bf16 with A = weights (row = cout), B = pixels (col = pixel), so every
//     lane ends up holding, for ONE pixel, runs of 4 consecutive output channels.
//   * The whole weight tensor of the layer lives in REGISTERS for the life of the workgroup
//     (9 * KC * NCT fragments of 4 registers, pinned to AGPRs / VGPRs through asm operands; the
//     unified 512-entry file at one wave per SIMD holds the 288 registers of a 64->64 layer), so
//     LDS carries only activations.
//   * A workgroup walks a vertical strip of TW output columns top to bottom.  Input rows stream
//     HBM -> LDS through a ring of RING rows filled by LDS-DMA (buffer_load ... lds, 16 B per lane),
//     LA rows ahead of the row being computed; every input row is fetched once per strip.  Lanes
//     whose pixel lies outside the image fail the buffer bounds check and the DMA writes zeros for
//     them: that IS the zero padding, and no address arithmetic runs per row (the row enters
//     through the scalar offset of the instruction).  One wave = 32 consecutive pixels of a row.
//   * LDS row image: one pixel-major sub-image per group of 16/32/64 channels, [pixel][16-B chunk]
//     with the chunk index XOR-swizzled by the pixel index so that the ds_read_b128 of 32
//     consecutive pixels is conflict free; each 1-KiB DMA piece covers whole (sub-)pixels, i.e.
//     full lines of the source tensor.  LDS-DMA writes linearly, so the swizzle is applied to the
//     per-lane SOURCE offset.  The channel concat of two source tensors = two sub-images.
//   * All waits are COUNTED (s_waitcnt vmcnt(N) / lgkmcnt(N) written by hand): every vector-memory
//     operation is issued unconditionally (out-of-range lanes are dropped by buffer bounds checks),
//     so the number of younger operations at each wait is a compile-time constant.
//   * Epilogue: accumulators go AGPR -> LDS (fp32) right after the row's MFMAs; the rest (pixel-
//     major read back, bias, activation, scale, residual, bf16 pack, 16-byte whole-pixel stores)
//     and all per-row issue work (LDS-DMA pieces, residual loads) are micro-steps interleaved with
//     the MFMAs of the NEXT row, so they hide in the matrix pipe's shadow.
#include <atomic>
#include <cstdlib>

#include "conv_mfma_util.h"
#include "srhip_common.h"

namespace {

struct ConvParams {
    const char* src_a;
    const char* src_b;
    const bf16x8* wpack;
    const float* bias;
    const char* res;
    char* dst;
    int n, h, w;
    int ca, cb;  // padded channels of the two sources
    int ca_stride;  // channels per pixel of the tensor src_a points into (>= ca: a channel slice of a wider tensor)
    int cout, cout_stride;
    int cout_w;     // channels this launch writes per pixel (<= cout_stride: a channel slice of a wider tensor)
    float slope, alpha;  // slope: 1 (no activation), 0 (ReLU) or the LeakyReLU slope
    float mask_slope;    // MASK epilogue: factor where the mask tensor is <= 0 (0: ReLU backward, s: LeakyReLU backward)
    char* aux;           // sign-bit tensor (N, H, ceil(W/32), 64) uint32: written (EMIT) or read (MASKBITS / DUALBITS)
    char* dst2;          // DUALBITS: second output, dst masked by the bits
    int nstrips, nsegs, seg_h, total_items;
    // Two-class partition (one work item per workgroup): items [0, main_items) are the uniform (strip, segment) grid over
    // rows [0, main_h); items [main_items, total_items) are the strips' remaining rows [main_h, h), walked with stride
    // tail_stride by the workgroups the uniform grid leaves idle.  Uniform partition: main_items = total_items, main_h = h.
    int main_items, main_h, tail_stride;
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }

// Channel groups (sub-images) of the K dimension, by number of 16-channel planes.
//   KC = 3: one 48-channel tensor as 32 + 16;  KC = 6: concat of a 32-channel and a 64-channel tensor.
template <int KC> struct SubLayout;
template <> struct SubLayout<1> { static constexpr int NSUB = 1; static constexpr int CS0 = 16, CS1 = 0; };
template <> struct SubLayout<2> { static constexpr int NSUB = 1; static constexpr int CS0 = 32, CS1 = 0; };
template <> struct SubLayout<3> { static constexpr int NSUB = 2; static constexpr int CS0 = 32, CS1 = 16; };
template <> struct SubLayout<4> { static constexpr int NSUB = 1; static constexpr int CS0 = 64, CS1 = 0; };
template <> struct SubLayout<6> { static constexpr int NSUB = 2; static constexpr int CS0 = 32, CS1 = 64; };

// AUX_: sign bits of an activation as ONE dword per lane and row (bit 8k + e = element e of the lane's store group k):
//   1 EMIT      the epilogue also writes the bits of alpha * act(conv + bias) > 0 (before a residual is added)
//   2 MASKBITS  the result (after the residual add, if any) is zeroed where its bit is clear
//   3 DUALBITS  dst gets the result, dst2 the result zeroed where its bit is clear
// Lane roles at store time are those of the NCT = 2, CSPLIT = 1 configurations (8 lanes per pixel, 4 store groups).
template <int KC_, int CT_TOTAL_, int CSPLIT_, bool OUT_F32_, int RES_, int AUX_ = 0>
struct Cfg {
    static constexpr bool EMIT = AUX_ == 1, MASKBITS = AUX_ == 2, DUALBITS = AUX_ == 3;
    static constexpr bool AUX_LOAD = MASKBITS || DUALBITS;
    static_assert(AUX_ == 0 || (CT_TOTAL_ == 2 && CSPLIT_ == 1 && !OUT_F32_ && RES_ <= 1), "sign-bit forms: 64 output channels per wave, bf16 output");
    static constexpr bool OUT_F32 = OUT_F32_;   // epilogue writes fp32 NCHW instead of bf16 NHWC
    static constexpr bool HAS_RES = RES_ != 0;  // the epilogue reads a second tensor `res` (bf16 NHWC only):
    static constexpr bool RES_MASK = RES_ == 2; // 1: dst += res;  2: dst = res > 0 ? dst : 0 (ReLU backward)
    static constexpr bool RES_PRE = RES_ == 3;  // 3: res joins the sum BEFORE the activation (partial sums of wide layers)
    static constexpr int KC = KC_;              // 16-channel K planes
    static constexpr int CT_TOTAL = CT_TOTAL_;  // 32-wide output-channel tiles of the layer
    static constexpr int CSPLIT = CSPLIT_;      // waves that share a pixel group (split over cout tiles)
    static constexpr int NCT = CT_TOTAL / CSPLIT;
    static constexpr int NWAVES = 4;
    static constexpr int NPG = NWAVES / CSPLIT;  // pixel groups (32 px) per row
    static constexpr int TW = 32 * NPG;          // strip width
    // ---- LDS row image: sub-image s holds CS_s channels of every pixel, pixel-major ----
    using SL = SubLayout<KC>;
    static constexpr int NSUB = SL::NSUB;
    static constexpr int cs(int s) { return s == 0 ? SL::CS0 : SL::CS1; }       // channels
    static constexpr int cpp(int s) { return cs(s) / 8; }                        // 16-B chunks per pixel
    static constexpr int lcpp(int s) { return cpp(s) == 8 ? 3 : (cpp(s) == 4 ? 2 : 1); }
    static constexpr int pw(int s) { return round_up(TW + 2, 64 / cpp(s)); }     // pitch: whole 1-KiB pieces
    static constexpr int sub_bytes(int s) { return pw(s) * cs(s) * 2; }
    static constexpr int sub_off(int s) { return s == 0 ? 0 : sub_bytes(0); }
    static constexpr int sub_pieces(int s) { return sub_bytes(s) / 1024; }
    static constexpr int piece_base(int s) { return s == 0 ? 0 : sub_pieces(0); }
    static constexpr int sub_of(int kc) { return (NSUB == 2 && kc >= SL::CS0 / 16) ? 1 : 0; }
    static constexpr int kcl_of(int kc) { return sub_of(kc) == 1 ? kc - SL::CS0 / 16 : kc; }
    static constexpr int SLOT_BYTES = sub_bytes(0) + (NSUB == 2 ? sub_bytes(1) : 0);
    static constexpr int NPIECE = SLOT_BYTES / 1024;                // 1 KiB LDS-DMA pieces per row
    static constexpr int MAXP = (NPIECE + NWAVES - 1) / NWAVES;     // pieces per wave per row
    static constexpr int LA = KC_ == 3 ? 4 : 5;                     // rows issued ahead of the row being computed (48-channel rows: one fewer measured faster)
    static constexpr int RING = LA + 1;                             // LDS row slots
    static constexpr int RING_BYTES = RING * SLOT_BYTES;
    // ---- epilogue staging (bf16 NHWC output): the wave's 32 px x (32*NCT) fp32 results cross LDS
    // once so that every store instruction writes whole pixels (16 B per lane, 1 KiB per wave).
    static constexpr int NSLOT = 8 * NCT;                           // 16-B fp32 slots per staged pixel
    static constexpr int PXS = NSLOT * 16;                          // staged bytes per pixel
    static constexpr int LPP = NSLOT / 2;                           // lanes per pixel at store time (8 ch each)
    static constexpr int PPI = 64 / LPP;                            // pixels per store instruction
    static constexpr int NST = 2 * NCT;                             // 16-B stores per lane per row
    static constexpr int STAGE_BYTES = OUT_F32 ? 0 : 32 * PXS;      // per wave
    static constexpr int NSTORE = OUT_F32 ? 16 * NCT : NST;         // store instructions per wave per row
    static constexpr int NRES = HAS_RES ? NST : 0;                  // residual load instructions per wave per row
    // ---- vector-memory schedule of one row (per wave) and the counted waits that follow from it.
    // Tap t of iteration i issues, in this order,
    //   [store of group (t-2)/2 of row i-1: even t >= 2]  [LDS-DMA piece t of row i+LA: t < MAXP]
    //   [residual load t of row i: t < NRES]
    // one vector-memory operation of a kind per tap, placed in the late MFMA gaps of the tap (the early
    // gaps carry the LDS reads).  The fp32 NCHW form appends its 16*NCT stores after tap 8.
    static constexpr int clamp63(int v) { return v > 63 ? 63 : v; }  // vmcnt is a 6-bit field; smaller = safe
    static constexpr int store_in_tap(int t) { return (!OUT_F32 && t >= 2 && t % 2 == 0 && (t - 2) / 2 < NST) ? (DUALBITS ? 2 : 1) : 0; }
    static constexpr int dma_in_tap(int t) { return t < MAXP ? 1 : 0; }
    static constexpr int res_in_tap(int t) { return t < NRES ? 1 : 0; }
    // sign bits: one dword load in tap 0 (behind the tap's other operations) or one dword store closing tap 8
    static constexpr int aux_in_tap(int t) { return ((AUX_LOAD && t == 0) || (EMIT && t == 8)) ? 1 : 0; }
    static constexpr int vm_in_tap(int t) { return store_in_tap(t) + dma_in_tap(t) + res_in_tap(t) + aux_in_tap(t); }
    static constexpr int vm_before_tap(int t) { int n = 0; for (int k = 0; k < t; ++k) n += vm_in_tap(k); return n; }
    static constexpr int VM_ROW = vm_before_tap(9) + (OUT_F32 ? 16 * NCT : 0);  // operations per iteration
    static constexpr int POS_LAST_DMA = vm_before_tap(MAXP - 1) + store_in_tap(MAXP - 1);
    static constexpr int POS_LAST_RES = NRES > 0 ? vm_before_tap(NRES - 1) + store_in_tap(NRES - 1) + dma_in_tap(NRES - 1) : 0;
    // Top of iteration i >= LA-2 waits for input row i+2, issued in iteration i+2-LA: younger are the
    // rest of that iteration and LA-3 whole iterations.
    static constexpr int top_wait_steady() { return clamp63((VM_ROW - 1 - POS_LAST_DMA) + (LA - 3) * VM_ROW); }
    // Iterations 1 .. LA-3: row i+2 came from the prologue; younger are the prologue's later rows and i
    // whole iterations.  One constant (the smallest) serves them all.
    static constexpr int top_wait_early() {
        int best = 1 << 20;
        for (int i = 1; i <= LA - 3; ++i) {
            const int n = (LA - 3 - i) * MAXP + i * VM_ROW;
            if (n < best) best = n;
        }
        return clamp63(best);
    }
    // The residual of row i is waited for at the END of iteration i (behind the last store of the row) and
    // consumed by the epilogue micro-steps of iteration i+1 or by the drain after the loop: younger than its
    // last load is only the rest of its own iteration.
    static constexpr int res_wait() { return clamp63(VM_ROW - 1 - POS_LAST_RES); }
    // the sign-bit load of a row is the last operation of its tap 0
    static constexpr int POS_AUX_LOAD = vm_in_tap(0) - 1;
    static constexpr int aux_wait() { return clamp63(VM_ROW - 1 - POS_AUX_LOAD); }
    static_assert(LA >= 4, "the wait-count formulas assume at least 4 rows of look-ahead");
    static_assert(MAXP <= 9 && NRES <= 9, "one LDS-DMA piece and one residual load per tap");
    static constexpr int NFRAG = 9 * KC * NCT;                      // weight fragments per wave
    static constexpr int NFA_MAX = (256 - 16 * NCT) / 4;            // AGPR room beside the accumulators
    static constexpr int NFA = NFRAG < NFA_MAX ? NFRAG : NFA_MAX;   // fragments pinned to AGPRs
    static constexpr int DUMMY_OFF = RING_BYTES + NWAVES * STAGE_BYTES;  // 1 KiB sink of the DMA pieces a wave does not own
    static constexpr int LDS_BYTES = DUMMY_OFF + 1024;
    static constexpr int MPT = KC * NCT;                            // MFMAs per tap
    // With at least KC + 2 MFMAs per tap the LDS reads of a tap are spread, one per MFMA gap, over the
    // PREVIOUS tap (gap 0: staging read-back, gaps 1..KC: pixel fragments), and a tap starts with a plain
    // wait for everything issued at least MPT - KC - 1 MFMAs ago.
    static constexpr bool SPREAD = MPT >= KC + 2;
    // One MFMA per fragment (one cout tile per wave) and a wide K: the fragments of the next tap still go between the
    // MFMAs, two behind the first and one behind each of the next KC - 2, so that the last MFMA of the tap covers the
    // last read.  (Issued as one burst in front of the tap, the six reads of the four waves queue up at the LDS while
    // the matrix pipe runs dry: 18 % of the 32 + 64 -> 64 layer's time.)
    static constexpr bool TIGHT = !SPREAD && MPT == KC && KC >= 4;
    // micro-steps run between the MFMAs of tap t: first the share of the PREVIOUS row's epilogue
    //   tap 2k+1: channels 0..3 of store group k;  tap 2k+2: channels 4..7 + pack/store
    // then this tap's LDS-DMA piece and residual load.
    static constexpr int epi_steps(int t) {
        if (t == 0 || OUT_F32) return 0;
        const int k = (t - 1) / 2;
        if (k >= NST) return 0;
        return (t & 1) ? 4 : 5 + (DUALBITS ? 1 : 0) + ((EMIT && t == 8) ? 1 : 0);
    }
    static constexpr int tap_steps(int t) { return epi_steps(t) + dma_in_tap(t) + res_in_tap(t) + ((AUX_LOAD && t == 0) ? 1 : 0); }
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
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- weights -> registers (fragment order prepared by the pack kernel) ----
    bf16x8 wreg[9][C::KC][C::NCT];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int kc = 0; kc < C::KC; ++kc)
#pragma unroll
            for (int ct = 0; ct < C::NCT; ++ct)
                wreg[t][kc][ct] = p.wpack[((t * C::KC + kc) * C::CT_TOTAL + cs * C::NCT + ct) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);  // bound the registers the loads have in flight
    }

    // ---- epilogue lane roles (bf16 NHWC output).  Store time is pixel-major: lane = (pixel st_q of
    // a group of PPI pixels, 8-channel chunk st_c), so this lane always owns the same 8 channels.
    const int st_q = lane / C::LPP;
    const int st_c = lane % C::LPP;
    const int st_fw = C::NCT == 2 ? (pl & 15) : ((pl >> 1) & 7);  // slot swizzle of the pixel this lane stages
    const unsigned stage_lds = lds_base + C::RING_BYTES + wave * C::STAGE_BYTES;
    // staging write address of accumulator quad (ct, g): wst ^ ((ct*8 + 2g) << 4)
    const unsigned wst = stage_lds + pl * C::PXS + (((hl ^ st_fw) & (C::NSLOT - 1)) << 4);
    // staging read address of store group k: low half rdst[k], high half rdst[k] ^ 16
    unsigned rdst[C::NST];
#pragma unroll
    for (int k = 0; k < C::NST; ++k) {
        const int q = k * C::PPI + st_q;
        const int fq = C::NCT == 2 ? (q & 15) : ((q >> 1) & 7);
        rdst[k] = stage_lds + q * C::PXS + ((((2 * st_c) ^ fq) & (C::NSLOT - 1)) << 4);
    }

    // ---- bias -> registers.  bf16 NHWC: alpha * bias of this lane's 8 store-time channels;
    // fp32 NCHW: accumulator order, reg r <-> cout = tile*32 + 8*(r>>2) + 4*hl + (r&3).
    constexpr int NBIAS = C::OUT_F32 ? 16 * C::NCT : 8;
    float breg[NBIAS];
#pragma unroll
    for (int r = 0; r < NBIAS; ++r) {
        const int co = C::OUT_F32 ? (cs * C::NCT + r / 16) * 32 + 8 * ((r & 15) >> 2) + 4 * hl + (r & 3)
                                  : cs * C::NCT * 32 + st_c * 8 + r;
        const float b = (p.bias != nullptr && co < p.cout) ? p.bias[co] : 0.0f;
        breg[r] = C::OUT_F32 ? b : b * p.alpha;
    }

    // ---- per-lane LDS read offsets inside a row slot: rd[kx][s] addresses chunk (hl) of LDS pixel
    // 32pg + pl + kx in sub-image s; chunk pair kcl of the same pixel is at rd ^ (kcl << 5).
    unsigned rd[3][C::NSUB];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int s = 0; s < C::NSUB; ++s) {
            const int px = 32 * pg + pl + kx;
            const int f = (px >> (4 - C::lcpp(s))) & (C::cpp(s) - 1);
            rd[kx][s] = (unsigned)(C::sub_off(s) + px * C::cpp(s) * 16 + ((hl ^ f) << 4));
        }

    const int H = p.h, W = p.w;
    // source of sub-image s: KC == 6 is concat(A: 32 ch, B: 64 ch); KC == 3 is one 48-channel tensor
    const int csrc0 = p.ca_stride, csrc1 = (C::KC == 6) ? p.cb : p.ca_stride;
    const int choff1 = (C::KC == 3) ? 32 : 0;

    // pixel fragments, double buffered over taps; the buffer of (row i, tap t) is (i + t) & 1
    bf16x8 bq[2][C::KC];
    f32x16 acc[C::NCT];
    // residual of row i lives in resv[i & 1]
    u32x4 resv[2][C::NRES > 0 ? C::NRES : 1];
    // epilogue scratch carried between micro-steps
    f32x4 e_lo[2], e_hi[2];  // staging read-back of store group k lives in buffer k & 1
    float ev[8];
    float ev2[8];            // DUALBITS: the masked copy
    unsigned mbv[2] = {0u, 0u};  // sign bits of row i (loaded forms) live in mbv[i & 1]
    unsigned mb_out = 0u;        // EMIT: bits of the row whose epilogue is running

    const int item_stride = p.main_items == p.total_items ? (int)gridDim.x
                                                          : ((int)blockIdx.x < p.main_items ? p.total_items : p.tail_stride);
    for (int item = blockIdx.x; item < p.total_items; item += item_stride) {
        int strip, n, y0, y1;
        if (item < p.main_items) {
            strip = item % p.nstrips;
            const int tseg = item / p.nstrips;
            n = tseg / p.nsegs;
            y0 = (tseg % p.nsegs) * p.seg_h;
            y1 = min(y0 + p.seg_h, p.main_h);
        } else {   // a strip's rows below the uniform grid
            const int k = item - p.main_items;
            strip = k % p.nstrips;
            n = k / p.nstrips;
            y0 = p.main_h;
            y1 = H;
        }
        const int x0 = strip * C::TW;
        const int nout = y1 - y0;
        const int nin = nout + 2;  // input rows r = 0..nin-1  <->  y = y0 - 1 + r

        // ---- per-image buffer descriptors (bounds-checked: out-of-range lanes read zeros / are dropped)
        const unsigned img_a = (unsigned)(H * W) * (unsigned)csrc0 * 2u;
        const unsigned img_b = (unsigned)(H * W) * (unsigned)csrc1 * 2u;
        const __amdgpu_buffer_rsrc_t rsrc_a =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.src_a + (long long)n * img_a), 0, (int)img_a, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((C::KC == 6 ? p.src_b : p.src_a) + (long long)n * img_b), 0, (int)img_b, 0x00020000);
        const unsigned plane_bytes = (unsigned)(H * W) * 4u;
        const unsigned img_out = C::OUT_F32 ? (unsigned)p.cout * plane_bytes : (unsigned)(H * W) * (unsigned)p.cout_stride * 2u;
        const __amdgpu_buffer_rsrc_t dst_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(p.dst + (long long)n * img_out, 0, (int)img_out, 0x00020000);
        const i32x4 dst_srd = make_srd(p.dst + (long long)n * img_out, img_out);
        const i32x4 res_srd = make_srd(C::HAS_RES ? p.res + (long long)n * img_out : p.dst, C::HAS_RES ? img_out : 0u);
        // sign bits: one dword per lane, row and 32-pixel group; dst2 (DUALBITS) has dst's geometry
        const int wgroups = (W + 31) >> 5;
        const unsigned aux_pitch = (unsigned)wgroups * 256u;
        const unsigned img_aux = (unsigned)H * aux_pitch;
        const i32x4 aux_srd = make_srd(C::AUX_LOAD || C::EMIT ? p.aux + (long long)n * img_aux : p.dst, C::AUX_LOAD || C::EMIT ? img_aux : 0u);
        const i32x4 dst2_srd = make_srd(C::DUALBITS ? p.dst2 + (long long)n * img_out : p.dst, C::DUALBITS ? img_out : 0u);
        const unsigned aux_vo = (x0 + 32 * pg < W) ? (unsigned)((((x0 >> 5) + pg) * 64 + lane) * 4) : kOobOffset;
        const unsigned out_pitch = (unsigned)W * (unsigned)p.cout_stride * 2u;  // bytes per output row (NHWC)
        const int xpix = x0 + 32 * pg + pl;

        // ---- per-lane source offsets of this wave's LDS-DMA pieces (constant over the strip).
        // Piece q covers LDS chunks [64q, 64q+64) of the row slot = whole pixels of one sub-image.
        unsigned poff[C::MAXP];
        bool psub1[C::MAXP];  // wave-uniform: the piece belongs to sub-image 1
#pragma unroll
        for (int m = 0; m < C::MAXP; ++m) {
            const int q = wave + m * C::NWAVES;
            const bool s1 = (C::NSUB == 2) && (q >= C::piece_base(1));
            const int j = (q - (s1 ? C::piece_base(1) : 0)) * 64 + lane;
            const int lc = s1 ? C::lcpp(1) : C::lcpp(0);
            const int px = j >> lc;
            const int slot = j & ((1 << lc) - 1);
            const int c = slot ^ ((px >> (4 - lc)) & ((1 << lc) - 1));
            const int x = x0 - 1 + px;
            const bool ok = (x >= 0) && (x < W) && (px < C::TW + 2);
            const int csrc = s1 ? csrc1 : csrc0;
            const int choff = s1 ? choff1 : 0;
            poff[m] = ok ? (unsigned)((x * csrc + choff + c * 8) * 2) : kOobOffset;
            psub1[m] = s1;
        }
        // per-lane output offsets inside a row (constant over the strip)
        unsigned vox[C::NST];
#pragma unroll
        for (int k = 0; k < C::NST; ++k) {
            const int xq = x0 + 32 * pg + k * C::PPI + st_q;
            const int ch = cs * C::NCT * 32 + st_c * 8;
            vox[k] = (xq < W && ch < p.cout_w) ? (unsigned)((xq * p.cout_stride + ch) * 2) : kOobOffset;
        }

        int ring_pos = 0;  // ring slot of input row i
        // LDS-DMA piece m of one input row.  Everything that depends on the row (scalar offsets of the
        // two sources, LDS slot) is computed once per row by the caller; a piece this wave does not own
        // (only the last one can be) is issued all-out-of-range into a sink, so every wave issues exactly
        // MAXP pieces per row and no branch sits between the MFMAs.
        auto issue_piece = [&](auto m_c, const unsigned soff0, const unsigned soff1, const unsigned slot_off) {
            constexpr int m = decltype(m_c)::value;
            const int q = wave + m * C::NWAVES;
            const bool mine = (m < C::MAXP - 1) || (q < C::NPIECE);
            const unsigned off = mine ? slot_off + (unsigned)q * 1024u : (unsigned)C::DUMMY_OFF;
            auto* dst = (__attribute__((address_space(3))) void*)(smem + off);
            const unsigned soff = (C::NSUB == 2 && psub1[m]) ? soff1 : soff0;
            if constexpr (C::KC == 6) {
                if (psub1[m]) lds_dma_b128(rsrc_b, dst, poff[m], soff);
                else lds_dma_b128(rsrc_a, dst, poff[m], soff);
            } else {
                lds_dma_b128(rsrc_a, dst, poff[m], soff);
            }
        };
        // scalar row offsets of input row r (kOobOffset: the whole row reads as zeros)
        auto row_soff = [&](const int r, unsigned& soff0, unsigned& soff1) {
            const int y = y0 - 1 + r;
            const bool yin = (y >= 0) & (y < H) & (r < nin);
            const unsigned pix = (unsigned)(y * W);
            soff0 = yin ? pix * (unsigned)(csrc0 * 2) : kOobOffset;
            soff1 = yin ? pix * (unsigned)(csrc1 * 2) : kOobOffset;
        };

        // ---- micro-steps of the second epilogue half of output row yy (staging -> HBM), whose
        // residual is in resv[AP].  store_soff is the scalar row offset (kOobOffset drops the row).
        auto epi_res_wait = [&](auto ap_c) {
            SRHIP_CAPTURE(resv);  // asm operands alone do not make a generic lambda capture
            constexpr int AP = decltype(ap_c)::value;
            if constexpr (C::HAS_RES) {
                vm_wait<C::res_wait()>();
#pragma unroll
                for (int k = 0; k < C::NRES; ++k) SRHIP_LOAD_FENCE(resv[AP][k]);
            }
            if constexpr (C::AUX_LOAD) {
                SRHIP_CAPTURE(mbv);
                // the row's sign-bit dword was loaded in tap 0: older than the last residual load, so the wait above
                // covers it where there is a residual
                if constexpr (!C::HAS_RES) vm_wait<C::aux_wait()>();
                SRHIP_LOAD_FENCE(mbv[AP]);
            }
        };
        auto epi_read = [&](auto k_c) {  // staging reads of store group k (complete after the next LDS wait)
            constexpr int k = decltype(k_c)::value;
            e_lo[k & 1] = lds_read_f32x4(rdst[k]);
            e_hi[k & 1] = lds_read_f32x4(rdst[k] ^ 16u);
        };
        auto epi_elem = [&](auto k_c, auto e_c, auto ap_c) {  // channel e: alpha * act(x + bias) (+ residual)
            constexpr int k = decltype(k_c)::value;
            constexpr int e = decltype(e_c)::value;
            constexpr int AP = decltype(ap_c)::value;
            const float x = e < 4 ? e_lo[k & 1][e & 3] : e_hi[k & 1][e & 3];
            float t = fmaf(x, p.alpha, breg[e]);
            float rv = 0.0f;
            if constexpr (C::HAS_RES) {
                const unsigned rr = resv[AP][k][e >> 1];
                rv = (e & 1) ? bf16_hi(rr) : bf16_lo(rr);
            }
            if constexpr (C::RES_PRE) t += rv;
            float v = fmaxf(t, t * p.slope);  // act(t) = max(t, slope * t), slope in {1, 0, s}
            if constexpr (C::EMIT) {
                // v > 0 as an integer test (a negative zero is not positive), deposited at bit 8k + e
                const unsigned pos = __float_as_int(v) > 0 ? 1u : 0u;
                mb_out = (k == 0 && e == 0) ? pos : (mb_out | (pos << (8 * k + e)));
            }
            if constexpr (C::RES_MASK) v = rv > 0.0f ? v : v * p.mask_slope;
            else if constexpr (C::HAS_RES && !C::RES_PRE) v += rv;
            if constexpr (C::AUX_LOAD) {
                const unsigned keep = (unsigned)((int)(mbv[AP] << (31 - (8 * k + e))) >> 31);   // all ones where the bit is set
                const float vm = __uint_as_float(__float_as_uint(v) & keep);
                if constexpr (C::MASKBITS) v = vm;
                else ev2[e] = vm;
            }
            ev[e] = v;
        };
        auto epi_store = [&](auto k_c, const unsigned store_soff) {
            constexpr int k = decltype(k_c)::value;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(ev[2 * e], ev[2 * e + 1]);
            buffer_store_b128_guarded(o, dst_srd, vox[k], store_soff);
        };
        auto epi_store2 = [&](auto k_c, const unsigned store_soff) {   // DUALBITS: the masked copy
            constexpr int k = decltype(k_c)::value;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(ev2[2 * e], ev2[2 * e + 1]);
            buffer_store_b128_guarded(o, dst2_srd, vox[k], store_soff);
        };
        auto emit_store = [&](const unsigned aux_soff) {               // EMIT: the finished row's sign bits
            asm volatile("buffer_store_dword %0, %1, %2, %3 offen" ::"v"(mb_out), "v"(aux_vo), "s"(aux_srd), "s"(aux_soff) : "memory");
        };

        // One output row.  PAR = i & 1 selects the fragment buffer that holds tap 0 and the residual set.
        auto row_body = [&](auto par_c, const int i) {
            constexpr int PAR = decltype(par_c)::value;
            // Input rows i..i+2 must have landed: wait until only operations younger than this wave's
            // pieces of row i+2 are outstanding, then the barrier covers the other waves' pieces (and
            // tells everyone that row i-1's slot is free).
            if (i == 0) {
                vm_wait<0>();
            } else if (i < C::LA - 2) {
                vm_wait<C::top_wait_early()>();
            } else {
                vm_wait<C::top_wait_steady()>();
            }
            __builtin_amdgcn_s_barrier();

            const int y = y0 + i;
            // ring slots of input rows i, i+1, i+2 and of the row being fetched (i + LA = slot before i)
            unsigned dma_s0, dma_s1;
            row_soff(i + C::LA, dma_s0, dma_s1);
            const unsigned dma_slot = (unsigned)((ring_pos == 0 ? C::RING - 1 : ring_pos - 1) * C::SLOT_BYTES);
            const unsigned res_soff = (unsigned)y * out_pitch;                       // residual of row i
            const unsigned st_soff = i > 0 ? (unsigned)(y - 1) * out_pitch : kOobOffset;  // stores of row i-1
            const unsigned aux_ld_soff = (unsigned)y * aux_pitch;                    // sign bits of row i (loaded forms)
            const unsigned aux_st_soff = i > 0 ? (unsigned)(y - 1) * aux_pitch : kOobOffset;  // ... of row i-1 (EMIT)
            unsigned rowa[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int sl = ring_pos + ky;
                rowa[ky] = lds_base + (unsigned)((sl >= C::RING ? sl - C::RING : sl) * C::SLOT_BYTES);
            }
            ring_pos = (ring_pos + 1 == C::RING) ? 0 : ring_pos + 1;

            if constexpr (PAR == 0) {
                if (i == 0) {  // first row of the item: nobody fetched its tap 0 yet
                    static_for<0, C::KC>([&](auto kc_c) {
                        constexpr int kc = decltype(kc_c)::value;
                        bq[0][kc] = lds_read_b128((rowa[0] + rd[0][C::sub_of(kc)]) ^ (unsigned)(C::kcl_of(kc) << 5));
                    });
                    lds_wait<0>();  // landed before this path joins the one that carries the fragments from tap 8
                    static_for<0, C::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[0][decltype(kc_c)::value]); });
                }
            }

            // Pixel fragments run one tap ahead of the MFMAs: while tap t multiplies, the KC fragments
            // of tap t+1 (or of tap 0 of the NEXT row, whose LDS row is already resident) are in
            // flight.  Between the MFMAs run this tap's micro-steps.
            static_for<0, 9>([&](auto t_c) {
                SRHIP_CAPTURE(resv); SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(e_lo); SRHIP_CAPTURE(e_hi);
                SRHIP_CAPTURE(mbv); SRHIP_CAPTURE(mb_out); SRHIP_CAPTURE(ev2); SRHIP_CAPTURE(aux_vo); SRHIP_CAPTURE(aux_srd); SRHIP_CAPTURE(aux_ld_soff);
                constexpr int t = decltype(t_c)::value;
                constexpr int cur = (PAR + t) & 1, nxt = cur ^ 1;
                constexpr int tn = (t + 1) % 9;                  // tap being fetched
                constexpr int kyn = (t == 8) ? 1 : (t + 1) / 3;  // its LDS row relative to input row i
                constexpr int KG = (t - 1) / 2;                  // store group served by taps 2k+1, 2k+2
                constexpr bool EPI = !C::OUT_F32 && t >= 1 && KG < C::NST;
                if constexpr (C::SPREAD || C::TIGHT) {
                    lds_wait<0>();
                } else {
                    if constexpr (EPI && (t & 1)) epi_read(IntC<KG>{});
                    static_for<0, C::KC>([&](auto kc_c) {
                        constexpr int kc = decltype(kc_c)::value;
                        bq[nxt][kc] = lds_read_b128((rowa[kyn] + rd[tn % 3][C::sub_of(kc)]) ^ (unsigned)(C::kcl_of(kc) << 5));
                    });
                    lds_wait<C::KC>();
                }
                // everything issued before this tap's own reads has landed: staging read-back, pixel fragments
                if constexpr (EPI && (t & 1)) { SRHIP_LOAD_FENCE(e_lo[KG & 1]); SRHIP_LOAD_FENCE(e_hi[KG & 1]); }
                static_for<0, C::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[cur][decltype(kc_c)::value]); });

                constexpr int NS = C::tap_steps(t);
                static_for<0, C::MPT>([&](auto m_c) {
                    SRHIP_CAPTURE(resv); SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc);
                    SRHIP_CAPTURE(mbv); SRHIP_CAPTURE(mb_out); SRHIP_CAPTURE(ev2); SRHIP_CAPTURE(aux_vo); SRHIP_CAPTURE(aux_srd); SRHIP_CAPTURE(aux_ld_soff);
                    constexpr int ml = decltype(m_c)::value;
                    constexpr int kc = ml / C::NCT, ct = ml % C::NCT;
                    constexpr int f = (t * C::KC + kc) * C::NCT + ct;  // MFMA index within the row
                    mfma_pinned<(f < C::NFA), (t == 0 && kc == 0), ct>(acc[ct], wreg[t][kc][ct], bq[cur][kc]);
                    if constexpr (C::SPREAD) {
                        // gap 0: staging read-back of the store group the next two taps finish;
                        // gaps 1..KC: pixel fragments of the next tap
                        if constexpr (ml == 0 && !C::OUT_F32 && (t % 2 == 0) && (t / 2 < C::NST)) epi_read(IntC<t / 2 + 0 * ml>{});  // + 0 * ml: dependent, so the dead branch is not instantiated
                        if constexpr (ml >= 1 && ml <= C::KC) {
                            constexpr int kr = ml - 1;
                            bq[nxt][kr] = lds_read_b128((rowa[kyn] + rd[tn % 3][C::sub_of(kr)]) ^ (unsigned)(C::kcl_of(kr) << 5));
                        }
                    } else if constexpr (C::TIGHT) {
                        if constexpr (ml == 0) {
                            bq[nxt][0] = lds_read_b128((rowa[kyn] + rd[tn % 3][C::sub_of(0)]) ^ (unsigned)(C::kcl_of(0) << 5));
                            bq[nxt][1] = lds_read_b128((rowa[kyn] + rd[tn % 3][C::sub_of(1)]) ^ (unsigned)(C::kcl_of(1) << 5));
                        } else if constexpr (ml <= C::KC - 2) {
                            constexpr int kr = ml + 1;
                            bq[nxt][kr] = lds_read_b128((rowa[kyn] + rd[tn % 3][C::sub_of(kr)]) ^ (unsigned)(C::kcl_of(kr) << 5));
                        }
                        // staging read-back of the store group the next two taps finish, beside the third fragment
                        if constexpr (ml == 2 && !C::OUT_F32 && (t % 2 == 0) && (t / 2 < C::NST)) epi_read(IntC<t / 2 + 0 * ml>{});
                    }
                    // micro-steps [ml*NS/MPT, (ml+1)*NS/MPT) of this tap
                    static_for<ml * NS / C::MPT, (ml + 1) * NS / C::MPT>([&](auto s_c) {
                        SRHIP_CAPTURE(resv); SRHIP_CAPTURE(mbv); SRHIP_CAPTURE(mb_out); SRHIP_CAPTURE(ev2); SRHIP_CAPTURE(aux_vo); SRHIP_CAPTURE(aux_srd); SRHIP_CAPTURE(aux_ld_soff);
                        constexpr int S = decltype(s_c)::value;
                        constexpr int E = C::epi_steps(t);
                        if constexpr (S < E) {
                            constexpr int KD = KG + 0 * S;  // dependent on S: dead branches are not instantiated
                            if constexpr (t & 1) {
                                epi_elem(IntC<KD>{}, IntC<S + 0 * KD>{}, IntC<PAR ^ 1>{});
                            } else {
                                if constexpr (S < 4) epi_elem(IntC<KD>{}, IntC<4 + S>{}, IntC<PAR ^ 1>{});
                                else if constexpr (S == 4) epi_store(IntC<KD>{}, st_soff);
                                else if constexpr (C::DUALBITS) epi_store2(IntC<KD>{}, st_soff);   // S == 5
                                else emit_store(aux_st_soff);                                   // S == 5, tap 8 (EMIT)
                            }
                        } else if constexpr (S < E + C::dma_in_tap(t)) {
                            constexpr int TD = t + 0 * S;  // dependent on S: the dead branch is not instantiated
                            issue_piece(IntC<TD>{}, dma_s0, dma_s1, dma_slot);
                        } else if constexpr (S < E + C::dma_in_tap(t) + C::res_in_tap(t)) {
                            constexpr int TR = t + 0 * S;
                            resv[PAR][TR] = buffer_load_b128_async(res_srd, vox[TR], res_soff);
                        } else {   // tap 0 of the loaded sign-bit forms: this row's dword
                            asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(mbv[PAR]) : "v"(aux_vo), "s"(aux_srd), "s"(aux_ld_soff));
                        }
                    });
                    if constexpr (NS > 0) __builtin_amdgcn_sched_barrier(0);  // keep the interleave as written
                });
            });

            // MFMA result -> any other reader: the hazard is invisible to the compiler inside asm
            if constexpr (C::NCT == 2) asm volatile("s_nop 15\n\ts_nop 3" : "+" SRHIP_ACC_REG0(acc[0]), "+" SRHIP_ACC_REG1(acc[1]));
            else asm volatile("s_nop 15\n\ts_nop 3" : "+" SRHIP_ACC_REG0(acc[0]));
            if constexpr (!C::OUT_F32) {
                // first epilogue half: accumulator quads (ct, g) = channels 8g+4hl..+3 of pixel pl go
                // straight from AGPRs to this wave's staging area; the rest of the epilogue runs
                // between the MFMAs of the next row.
                static_for<0, 4 * C::NCT>([&](auto s_c) {
                    constexpr int S = decltype(s_c)::value;
                    constexpr int ct = S / 4, g = S % 4;
                    lds_write_acc_quad<ct, g>(wst ^ (unsigned)((ct * 8 + 2 * g) << 4), acc[ct]);
                });
                // Tap 8 fetched the fragments of the next row's tap 0 (older than the staging writes above).
                // No asm load stays in flight across the end of a row: where the two row parities' paths
                // merge the compiler moves registers, and it does not know that these are still being written.
                lds_wait<4 * C::NCT>();
                static_for<0, C::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[PAR ^ 1][decltype(kc_c)::value]); });
                // The row's residual must have landed before anything consumes it (next iteration or the drain
                // after the loop).  Waiting HERE, in the code of this row parity, keeps the wait in front of the
                // register moves the compiler places where the two parities' paths merge.
                epi_res_wait(IntC<PAR>{});
            } else {
                // fp32 NCHW output (network output / fp32-accumulate checks): plain epilogue.
                const unsigned pixoff = (unsigned)((y * W + xpix) * 4);
#pragma unroll
                for (int ct = 0; ct < C::NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = (cs * C::NCT + ct) * 32 + 8 * (r >> 2) + 4 * hl + (r & 3);
                        const float t = acc[ct][r] + breg[ct * 16 + r];
                        const float v = fmaxf(t, t * p.slope) * p.alpha;
                        const unsigned vo = (xpix < W && co < p.cout) ? (unsigned)co * plane_bytes + pixoff : kOobOffset;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dst_rsrc, vo, 0, 0);
                    }
                lds_wait<0>();  // tap 8's fetch of the next row's tap 0: nothing stays in flight across rows
                static_for<0, C::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[PAR ^ 1][decltype(kc_c)::value]); });
            }
        };

        __builtin_amdgcn_s_barrier();  // every wave is done reading the previous item's rows
#pragma unroll 1
        for (int r = 0; r < C::LA; ++r) {
            unsigned s0, s1;
            row_soff(r, s0, s1);
            static_for<0, C::MAXP>([&](auto m_c) { issue_piece(m_c, s0, s1, (unsigned)(r * C::SLOT_BYTES)); });
        }
        ring_pos = 0;

#pragma unroll 1
        for (int i = 0; i < nout; i += 2) {
            row_body(IntC<0>{}, i);
            if (i + 1 < nout) row_body(IntC<1>{}, i + 1);
        }

        if constexpr (!C::OUT_F32) {
            // drain: second epilogue half of the last row
            const unsigned st_soff = (unsigned)(y1 - 1) * out_pitch;
            auto tail = [&](auto ap_c) {
                static_for<0, C::NST>([&](auto k_c) {
                    epi_read(k_c);
                    lds_wait<0>();
                    SRHIP_LOAD_FENCE(e_lo[decltype(k_c)::value & 1]);
                    SRHIP_LOAD_FENCE(e_hi[decltype(k_c)::value & 1]);
                    static_for<0, 8>([&](auto e_c) { epi_elem(k_c, e_c, ap_c); });
                    epi_store(k_c, st_soff);
                    if constexpr (C::DUALBITS) epi_store2(k_c, st_soff);
                });
                if constexpr (C::EMIT) emit_store((unsigned)(y1 - 1) * aux_pitch);
            };
            if ((nout - 1) & 1) tail(IntC<1>{});
            else tail(IntC<0>{});
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// concat(32 ch, 64 ch) -> 64 with the REDUCTION split over the two waves of a pixel group (core/models/model.py:169-170).
//
// Cfg<6, 2, 2> above splits that layer over OUTPUT channels: a wave holds one 32-channel tile for all 96 input channels,
// so every pixel fragment it reads from LDS feeds a single MFMA -- four waves then ask the LDS for 128 bytes per clock
// at full matrix rate, its whole bandwidth, and the layer ran LDS-bound.  Here a wave holds BOTH output tiles for HALF
// of the input channels (k-chunks {A0, A1, B0} or {B1, B2, B3}: 216 registers of weights either way, all in AGPRs), every
// fragment feeds two MFMAs, and the partial accumulators meet in the fp32 staging area the epilogue crosses anyway:
// wave kh FINISHES output tile kh of its 32 pixels (bias, activation, alpha, bf16 pack, 16-byte stores) from its own
// partial of that tile and the partner's.
//   * No accumulator is staged outside the MFMA stream.  Taps 0 and 8 run tile-major.  In tap 8 the partner's tile comes
//     first and is staged (into the `give` area the partner reads after the next row's barrier, double buffered by row
//     parity) in the gaps behind the own tile's MFMAs; the own tile, complete with the row's last MFMA, is staged in the
//     first gaps of the NEXT row's tap 0, whose first three MFMAs belong to the partner's tile.
//   * Ring slots enter the fragment reads through the instruction's offset field: the row loop is unrolled over the six
//     slots, and two sets of lane addresses (slots 0..2 / 3..5) keep every offset under 64 KiB.
// ReLU / bias / alpha epilogue, bf16 NHWC output, no residual and no sign-bit forms: what the network's two concat
// layers need.
// ------------------------------------------------------------------------------------------------------------------
struct KsCfg {
    using L = Cfg<6, 2, 2, false, 0, 0>;                // LDS row image, LDS-DMA pieces and strip width of the concat layer
    static constexpr int KCW = 3, NCT = 2, MPT = KCW * NCT;
    static constexpr int NWAVES = 4, TW = L::TW;
    static constexpr int LA = L::LA, RING = L::RING, SLOT_BYTES = L::SLOT_BYTES, RING_BYTES = L::RING_BYTES;
    static constexpr int NPIECE = L::NPIECE, MAXP = L::MAXP;
    static constexpr int PXS = 128;                     // staged bytes per pixel of ONE 32-channel tile (fp32)
    static constexpr int TILE_BYTES = 32 * PXS;         // 4 KiB
    static constexpr int GIVE_OFF = RING_BYTES;         // [row parity][wave]: partial of the partner's tile
    static constexpr int OWN_OFF = GIVE_OFF + 2 * NWAVES * TILE_BYTES;   // [wave]: partial of the own tile
    static constexpr int DUMMY_OFF = OWN_OFF + NWAVES * TILE_BYTES;
    static constexpr int LDS_BYTES = DUMMY_OFF + 1024;
    // vector-memory operations of one row and wave, in issue order: LDS-DMA pieces in taps 0..MAXP-1, the two stores in
    // taps 4 and 8
    static constexpr int VM_ROW = MAXP + 2;
    static constexpr int WAIT_STEADY = (VM_ROW - MAXP) + (LA - 3) * VM_ROW;
    static constexpr int WAIT_EARLY = (LA - 4) * MAXP + VM_ROW < (LA - 3) * VM_ROW ? (LA - 4) * MAXP + VM_ROW : (LA - 3) * VM_ROW;
    static_assert(MAXP <= 4 && LA == 5 && WAIT_STEADY <= 63, "stores sit in taps 4 and 8 behind the row's last DMA piece");
    static_assert(RING == 6 && 2 * SLOT_BYTES < 65536, "two lane-address sets of three slots each");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int OFF>
__device__ __forceinline__ bf16x8 ks_lds_read(unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}

__global__ __launch_bounds__(KsCfg::NWAVES * 64, 1) void conv3x3_ksplit_kernel(const ConvParams p)
{
    using C = KsCfg;
    using L = KsCfg::L;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave & 1;          // which half of the reduction this wave owns = the output tile it finishes
    const int pg = wave >> 1;         // which 32-pixel group of the row
    const int hl = lane >> 5, pl = lane & 31;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // local tile 0 = the tile this wave finishes (global tile kh), local tile 1 = the partner's
    bf16x8 wreg[9][C::KCW][C::NCT];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int j = 0; j < C::KCW; ++j)
#pragma unroll
            for (int lt = 0; lt < C::NCT; ++lt) wreg[t][j][lt] = p.wpack[((t * 6 + 3 * kh + j) * 2 + (kh ^ lt)) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
    }
    static_for<0, 9 * C::KCW * C::NCT>([&](auto f_c) {   // the loads land before the row loop, not between its MFMAs
        SRHIP_CAPTURE(wreg);
        constexpr int f = decltype(f_c)::value;
        asm volatile("" : "+a"(wreg[f / (C::KCW * C::NCT)][(f / C::NCT) % C::KCW][f % C::NCT]));
    });

    // staging of one tile: pixel-major, 8 slots of 4 channels, slot index XOR-swizzled by the pixel
    const unsigned stage_px = (unsigned)(pl * C::PXS) + (unsigned)(((hl ^ ((pl >> 1) & 7)) & 7) << 4);   // quad g: ^ (2g << 4)
    const unsigned w_give = lds_base + (unsigned)C::GIVE_OFF + (unsigned)(wave * C::TILE_BYTES) + stage_px;          // + parity
    const unsigned w_own = lds_base + (unsigned)C::OWN_OFF + (unsigned)(wave * C::TILE_BYTES) + stage_px;
    // store-time lane roles: lane = (pixel st_q of a group of 16 pixels, 8-channel chunk st_c of the tile)
    const int st_q = lane >> 2, st_c = lane & 3;
    unsigned r_own[2], r_got[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = k * 16 + st_q;
        const unsigned off = (unsigned)(q * C::PXS) + (unsigned)((((2 * st_c) ^ ((q >> 1) & 7)) & 7) << 4);
        r_own[k] = lds_base + (unsigned)C::OWN_OFF + (unsigned)(wave * C::TILE_BYTES) + off;
        r_got[k] = lds_base + (unsigned)C::GIVE_OFF + (unsigned)((wave ^ 1) * C::TILE_BYTES) + off;                  // + parity
    }
    float breg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int co = kh * 32 + st_c * 8 + r;
        breg[r] = ((p.bias != nullptr && co < p.cout) ? p.bias[co] : 0.0f) * p.alpha;
    }
    // fragment j of tap column kx: k-chunk 3 kh + j = (sub-image, chunk pair) of LDS pixel 32 pg + pl + kx; set 0 is
    // based at ring slot 0, set 1 at slot 3
    unsigned rdk[2][3][C::KCW];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = 32 * pg + pl + kx;
#pragma unroll
        for (int j = 0; j < C::KCW; ++j) {
            const int c = 3 * kh + j;
            const int sub = c >= 2 ? 1 : 0, kcl = c - 2 * sub;
            const int cpp = sub ? L::cpp(1) : L::cpp(0), lc = sub ? L::lcpp(1) : L::lcpp(0);
            const int f = (px >> (4 - lc)) & (cpp - 1);
            const unsigned off = (unsigned)((sub ? L::sub_off(1) : 0) + px * cpp * 16 + ((hl ^ f) << 4));
            rdk[0][kx][j] = lds_base + (off ^ (unsigned)(kcl << 5));
            rdk[1][kx][j] = rdk[0][kx][j] + 3u * (unsigned)C::SLOT_BYTES;
        }
    }

    const int H = p.h, W = p.w;
    const int csrc0 = p.ca_stride, csrc1 = p.cb;
    bf16x8 bq[2][C::KCW];
    f32x16 acc[C::NCT];
    f32x4 ea[2][2], eb[2][2];   // [store group][lo / hi]: k half 0's and k half 1's partial (one summation order everywhere)
    float ev[8];

    const int item_stride = p.main_items == p.total_items ? (int)gridDim.x
                                                          : ((int)blockIdx.x < p.main_items ? p.total_items : p.tail_stride);
    for (int item = blockIdx.x; item < p.total_items; item += item_stride) {
        int strip, n, y0, y1;
        if (item < p.main_items) {
            strip = item % p.nstrips;
            const int tseg = item / p.nstrips;
            n = tseg / p.nsegs;
            y0 = (tseg % p.nsegs) * p.seg_h;
            y1 = min(y0 + p.seg_h, p.main_h);
        } else {
            const int k = item - p.main_items;
            strip = k % p.nstrips;
            n = k / p.nstrips;
            y0 = p.main_h;
            y1 = H;
        }
        const int x0 = strip * C::TW;
        const int nout = y1 - y0;
        const int nin = nout + 2;

        const unsigned img_a = (unsigned)(H * W) * (unsigned)csrc0 * 2u;
        const unsigned img_b = (unsigned)(H * W) * (unsigned)csrc1 * 2u;
        const __amdgpu_buffer_rsrc_t rsrc_a =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.src_a + (long long)n * img_a), 0, (int)img_a, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_b =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.src_b + (long long)n * img_b), 0, (int)img_b, 0x00020000);
        const unsigned img_out = (unsigned)(H * W) * (unsigned)p.cout_stride * 2u;
        const i32x4 dst_srd = make_srd(p.dst + (long long)n * img_out, img_out);
        const unsigned out_pitch = (unsigned)W * (unsigned)p.cout_stride * 2u;

        unsigned poff[C::MAXP];
        bool psub1[C::MAXP];
#pragma unroll
        for (int m = 0; m < C::MAXP; ++m) {
            const int q = wave + m * C::NWAVES;
            const bool s1 = q >= L::piece_base(1);
            const int j = (q - (s1 ? L::piece_base(1) : 0)) * 64 + lane;
            const int lc = s1 ? L::lcpp(1) : L::lcpp(0);
            const int px = j >> lc;
            const int slot = j & ((1 << lc) - 1);
            const int c = slot ^ ((px >> (4 - lc)) & ((1 << lc) - 1));
            const int x = x0 - 1 + px;
            const bool ok = (x >= 0) && (x < W) && (px < C::TW + 2);
            poff[m] = ok ? (unsigned)((x * (s1 ? csrc1 : csrc0) + c * 8) * 2) : kOobOffset;
            psub1[m] = s1;
        }
        unsigned vox[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int xq = x0 + 32 * pg + k * 16 + st_q;
            const int ch = kh * 32 + st_c * 8;
            vox[k] = (xq < W && ch < p.cout_w) ? (unsigned)((xq * p.cout_stride + ch) * 2) : kOobOffset;
        }

        auto issue_piece = [&](auto m_c, const unsigned soff0, const unsigned soff1, const unsigned slot_off) {
            constexpr int m = decltype(m_c)::value;
            const int q = wave + m * C::NWAVES;
            const bool mine = (m < C::MAXP - 1) || (q < C::NPIECE);
            const unsigned off = mine ? slot_off + (unsigned)q * 1024u : (unsigned)C::DUMMY_OFF;
            auto* dst = (__attribute__((address_space(3))) void*)(smem + off);
            if (psub1[m]) lds_dma_b128(rsrc_b, dst, poff[m], soff1);
            else lds_dma_b128(rsrc_a, dst, poff[m], soff0);
        };
        auto row_soff = [&](const int r, unsigned& soff0, unsigned& soff1) {
            const int y = y0 - 1 + r;
            const bool yin = (y >= 0) & (y < H) & (r < nin);
            const unsigned pix = (unsigned)(y * W);
            soff0 = yin ? pix * (unsigned)(csrc0 * 2) : kOobOffset;
            soff1 = yin ? pix * (unsigned)(csrc1 * 2) : kOobOffset;
        };
        // epilogue of store group k of the row whose partner partial was staged with parity offset par_off
        auto epi_read = [&](auto k_c, const unsigned par_off) {
            constexpr int k = decltype(k_c)::value;
            // [a] = k half 0's partial, [b] = k half 1's: the own one for kh == 0, the partner's otherwise
            const unsigned a0 = kh == 0 ? r_own[k] : r_got[k] + par_off;
            const unsigned b0 = kh == 0 ? r_got[k] + par_off : r_own[k];
            ea[k][0] = lds_read_f32x4(a0);
            ea[k][1] = lds_read_f32x4(a0 ^ 16u);
            eb[k][0] = lds_read_f32x4(b0);
            eb[k][1] = lds_read_f32x4(b0 ^ 16u);
        };
        auto epi_fence = [&](auto k_c) {
            SRHIP_CAPTURE(ea); SRHIP_CAPTURE(eb);
            constexpr int k = decltype(k_c)::value;
            SRHIP_LOAD_FENCE(ea[k][0]); SRHIP_LOAD_FENCE(ea[k][1]); SRHIP_LOAD_FENCE(eb[k][0]); SRHIP_LOAD_FENCE(eb[k][1]);
        };
        auto epi_elem = [&](auto k_c, auto e_c) {
            constexpr int k = decltype(k_c)::value;
            constexpr int e = decltype(e_c)::value;
            const float x = ea[k][e >> 2][e & 3] + eb[k][e >> 2][e & 3];
            const float t = fmaf(x, p.alpha, breg[e]);
            ev[e] = fmaxf(t, t * p.slope);
        };
        auto epi_store = [&](auto k_c, const unsigned store_soff) {
            constexpr int k = decltype(k_c)::value;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(ev[2 * e], ev[2 * e + 1]);
            buffer_store_b128_guarded(o, dst_srd, vox[k], store_soff);
        };

        // One output row; Q = i mod RING fixes the ring slots of input rows i .. i+2 and the row parity.
        auto row_body = [&](auto q_c, const int i) {
            constexpr int Q = decltype(q_c)::value;
            constexpr int PAR = Q & 1;
            if (i == 0) vm_wait<0>();
            else if (i < C::LA - 2) vm_wait<C::WAIT_EARLY>();
            else vm_wait<C::WAIT_STEADY>();
            lds_wait<0>();   // the partial this wave staged for its partner in the previous row is complete before the barrier says so
            __builtin_amdgcn_s_barrier();

            const int y = y0 + i;
            unsigned dma_s0, dma_s1;
            row_soff(i + C::LA, dma_s0, dma_s1);
            constexpr unsigned dma_slot = (unsigned)(((Q + C::LA) % C::RING) * C::SLOT_BYTES);
            const unsigned st_soff = i > 0 ? (unsigned)(y - 1) * out_pitch : kOobOffset;   // stores of row i-1
            constexpr unsigned prev_par = (unsigned)((PAR ^ 1) * C::NWAVES * C::TILE_BYTES);  // where the partner staged row i-1
            constexpr unsigned own_par = (unsigned)(PAR * C::NWAVES * C::TILE_BYTES);

            auto frag = [&](auto slot_c, auto kx_c, auto j_c) {
                constexpr int sl = decltype(slot_c)::value % C::RING;
                return ks_lds_read<(sl % 3) * C::SLOT_BYTES>(rdk[sl / 3][decltype(kx_c)::value][decltype(j_c)::value]);
            };
            if constexpr (Q == 0) {
                if (i == 0) {   // first row of the item: nobody fetched its tap 0 yet
                    static_for<0, C::KCW>([&](auto j_c) { bq[0][decltype(j_c)::value] = frag(IntC<0>{}, IntC<0>{}, j_c); });
                    lds_wait<0>();
                    static_for<0, C::KCW>([&](auto j_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[0][decltype(j_c)::value]); });
                }
            }

            // Schedule of the row (previous row = i - 1):
            //   tap 0   tile-major, partner's tile first; gaps 1, 2: the previous row's OWN-tile accumulator -> staging
            //           (before MFMA 3 starts to overwrite it)
            //   tap 1 / tap 4, gap 0    staging reads of store group 0 / 1 of the previous row (own partial + partner's)
            //   taps 2..4 / 5..7        its arithmetic (elements {0,1,2}, {3,4,5}, {6,7} in gaps 3..5); stores in taps 4 (gap 5) / 8
            //   tap 8   tile-major, partner's tile first; gaps 4, 5: that tile's accumulator -> the partner's staging area
            //   taps 0..MAXP-1, last gap   LDS-DMA piece of row i + LA
            static_for<0, 9>([&](auto t_c) {
                SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(ea); SRHIP_CAPTURE(eb); SRHIP_CAPTURE(ev); SRHIP_CAPTURE(wreg);
                constexpr int t = decltype(t_c)::value;
                constexpr int cur = (PAR + t) & 1, nxt = cur ^ 1;
                constexpr int tn = (t + 1) % 9;
                constexpr int kyn = (t == 8) ? 1 : (t + 1) / 3;
                constexpr bool TILE_MAJOR = t == 0 || t == 8;
                lds_wait<0>();
                if constexpr (t == 2) epi_fence(IntC<0>{});
                if constexpr (t == 5) epi_fence(IntC<1>{});
                static_for<0, C::KCW>([&](auto j_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[cur][decltype(j_c)::value]); });
                static_for<0, C::MPT>([&](auto m_c) {
                    SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(ea); SRHIP_CAPTURE(eb); SRHIP_CAPTURE(ev); SRHIP_CAPTURE(wreg);
                    constexpr int ml = decltype(m_c)::value;
                    constexpr int j = TILE_MAJOR ? ml % C::KCW : ml / C::NCT;
                    constexpr int lt = TILE_MAJOR ? 1 - ml / C::KCW : ml % C::NCT;
                    mfma_pinned<true, (t == 0 && j == 0), lt>(acc[lt], wreg[t][j][lt], bq[cur][j]);
                    if constexpr (ml == 0 && t == 1) epi_read(IntC<0 * ml>{}, prev_par);
                    if constexpr (ml == 0 && t == 4) epi_read(IntC<1 + 0 * ml>{}, prev_par);
                    if constexpr (ml >= 1 && ml <= C::KCW) {
                        constexpr int jr = ml - 1;
                        bq[nxt][jr] = frag(IntC<Q + kyn>{}, IntC<tn % 3>{}, IntC<jr>{});
                    }
                    if constexpr (t == 0 && (ml == 1 || ml == 2)) {   // own tile of the previous row (behind this tap's reads)
                        lds_write_acc_quad<0, 2 * (ml - 1)>(w_own ^ (unsigned)((4 * (ml - 1)) << 4), acc[0]);
                        lds_write_acc_quad<0, 2 * (ml - 1) + 1>(w_own ^ (unsigned)((4 * (ml - 1) + 2) << 4), acc[0]);
                    }
                    if constexpr (t == 8 && (ml == 4 || ml == 5)) {   // the partner's tile of this row
                        lds_write_acc_quad<1, 2 * (ml - 4)>((w_give + own_par) ^ (unsigned)((4 * (ml - 4)) << 4), acc[1]);
                        lds_write_acc_quad<1, 2 * (ml - 4) + 1>((w_give + own_par) ^ (unsigned)((4 * (ml - 4) + 2) << 4), acc[1]);
                    }
                    if constexpr ((t >= 2 && t <= 4) || (t >= 5 && t <= 7)) {
                        constexpr int G = t >= 5 ? 1 : 0;
                        constexpr int e0 = 3 * (t - (G ? 5 : 2));
                        constexpr int e = e0 + (ml - 3);
                        // group 0's last elements share tap 4 with group 1's reads; its store closes that tap
                        if constexpr (ml >= 3 && e < 8 && !(G == 0 && t == 4 && ml == 5)) epi_elem(IntC<G + 0 * ml>{}, IntC<e>{});
                    }
                    if constexpr (t == 4 && ml == 5) epi_store(IntC<0 * ml>{}, st_soff);
                    if constexpr (t == 8 && ml == 2) epi_store(IntC<1 + 0 * ml>{}, st_soff);
                    if constexpr (t < C::MAXP && ml == 5) issue_piece(IntC<t + 0 * ml>{}, dma_s0, dma_s1, dma_slot);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            // in flight across the end of the row: tap 8's four staging writes, issued behind its fragment reads
            lds_wait<4>();
            static_for<0, C::KCW>([&](auto j_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[PAR ^ 1][decltype(j_c)::value]); });
        };

        __builtin_amdgcn_s_barrier();  // every wave is done with the previous item's rows and staged partials
#pragma unroll 1
        for (int r = 0; r < C::LA; ++r) {
            unsigned s0, s1;
            row_soff(r, s0, s1);
            static_for<0, C::MAXP>([&](auto m_c) { issue_piece(m_c, s0, s1, (unsigned)(r * C::SLOT_BYTES)); });
        }
#pragma unroll 1
        for (int i = 0; i < nout; i += C::RING) {
            static_for<0, C::RING>([&](auto q_c) {
                constexpr int Q = decltype(q_c)::value;
                if (Q == 0 || i + Q < nout) row_body(q_c, i + Q);
            });
        }
        // drain: the last row's own tile goes to the staging area (no next row does it), the partials meet
        asm volatile("s_nop 15\n\ts_nop 3" : "+" SRHIP_ACC_REG0(acc[0]));
        static_for<0, 4>([&](auto g_c) {
            SRHIP_CAPTURE(acc);
            constexpr int g = decltype(g_c)::value;
            lds_write_acc_quad<0, g>(w_own ^ (unsigned)((2 * g) << 4), acc[0]);
        });
        lds_wait<0>();
        __builtin_amdgcn_s_barrier();
        {
            const unsigned st_soff = (unsigned)(y1 - 1) * out_pitch;
            const unsigned last_par = (unsigned)(((nout - 1) & 1) * C::NWAVES * C::TILE_BYTES);
            static_for<0, 2>([&](auto k_c) {
                epi_read(k_c, last_par);
                lds_wait<0>();
                epi_fence(k_c);
                static_for<0, 8>([&](auto e_c) { epi_elem(k_c, e_c); });
                epi_store(k_c, st_soff);
            });
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

template <class C, class Kernel>
int launch_conv_with(ConvParams p, hipStream_t stream, Kernel kernel)
{
    // per device, immutable after the first call on that device for this instantiation
    static std::atomic<int> blocks_per_cu_dev[kSrhipMaxDevices];
    const int dev = srhip_current_device();
    int blocks_per_cu = blocks_per_cu_dev[dev].load(std::memory_order_acquire);
    if (blocks_per_cu == 0) {
        if (C::LDS_BYTES > 48 * 1024) {
            if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    C::LDS_BYTES) != hipSuccess) {
                srhip_set_error("conv3x3: cannot raise the dynamic LDS limit to %d", C::LDS_BYTES);
                return SRHIP_ERR_LAUNCH;
            }
        }
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, C::NWAVES * 64, C::LDS_BYTES) !=
                hipSuccess || nb < 1)
            nb = 1;
        blocks_per_cu = nb > 2 ? 2 : nb;
        blocks_per_cu_dev[dev].store(blocks_per_cu, std::memory_order_release);
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
    long long total = (long long)p.nsegs * nstrips * p.n;
    p.main_items = (int)total; p.main_h = p.h; p.tail_stride = 0;
    // Two-class partition: when a whole number of segments per strip leaves workgroups idle (4K, 128-column strips:
    // 30 x 8 = 240 of 256), the main segments shrink a little and the idle workgroups walk the rows that remain at the
    // bottom of every strip, so that all of them finish together.
    {
        const double fill = 2 + 6.0;
        const long long columns = (long long)nstrips * p.n;
        const int segs = (int)(slots / columns);
        const int idle = (int)(slots - segs * columns);
        if (segs >= 1 && idle >= 1 && columns < (1 << 20)) {
            const int per_idle = srhip_div_up(columns, idle);           // tail pieces per idle workgroup
            // main cost h1 + fill, tail cost per_idle * (L + fill) with L = h - segs * h1: balance them
            int h1 = (int)(((double)per_idle * (p.h + fill) - fill) / (1.0 + (double)per_idle * segs)) + 1;
            if (h1 > p.h / segs) h1 = p.h / segs;
            const int tail_h = p.h - segs * h1;
            const double cost = (h1 + fill > per_idle * (tail_h + fill)) ? h1 + fill : per_idle * (tail_h + fill);
            if (h1 >= 1 && tail_h >= 1 && cost < 0.99 * best_cost) {
                p.seg_h = h1; p.nsegs = segs; p.main_h = segs * h1;
                p.main_items = (int)(segs * columns);
                p.tail_stride = idle;
                total = p.main_items + columns;
            }
        }
    }
    SRHIP_REQUIRE(total < (1ll << 30), "conv3x3: too many work items");
    p.total_items = (int)total;
    const int grid = p.main_items != p.total_items ? slots : (int)(total < slots ? total : slots);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(C::NWAVES * 64), C::LDS_BYTES, stream, p);
    return srhip_check_launch("conv3x3_kernel");
}
template <class C>
int launch_conv(ConvParams p, hipStream_t stream)
{
    return launch_conv_with<C>(p, stream, conv3x3_kernel<C>);
}

}  // namespace

extern "C" {

SRHIP_API size_t srhip_conv3x3_signbits_bytes(int n, int h, int w)
{
    if (n < 1 || h < 1 || w < 1) return 0;
    return (size_t)n * h * ((w + 31) / 32) * 256;
}

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
    SRHIP_REQUIRE(a->act == SRHIP_ACT_NONE || a->act == SRHIP_ACT_RELU || a->act == SRHIP_ACT_LRELU, "conv3x3: bad activation %d", a->act);
    SRHIP_REQUIRE((long long)a->h * a->w * 64 * 4 < (1ll << 31), "conv3x3: image too large for 32-bit offsets");
    const int ca_stride = a->ca_stride > 0 ? a->ca_stride : a->ca;
    const int cout_w = a->cout_w > 0 ? a->cout_w : a->cout_stride;
    SRHIP_REQUIRE(ca_stride >= a->ca && ca_stride % 8 == 0 && (a->cb == 0 || ca_stride == a->ca),
                  "conv3x3: bad ca_stride %d (multiple of 8, >= ca; the two-source form takes whole tensors)", ca_stride);
    SRHIP_REQUIRE((long long)a->w * (ca_stride > a->cb ? ca_stride : a->cb) * 2 < (1ll << 31), "conv3x3: row too long");
    SRHIP_REQUIRE((long long)a->h * a->w * ca_stride * 2 < (1ll << 31), "conv3x3: source image too large for 32-bit offsets");
    if (a->dst_nchw_f32) {
        SRHIP_REQUIRE(a->res == nullptr, "conv3x3: fp32 NCHW output takes no residual");
    } else {
        SRHIP_REQUIRE(a->cout_stride % 8 == 0 && a->cout_stride >= a->cout, "conv3x3: bad cout_stride %d", a->cout_stride);
        SRHIP_REQUIRE(cout_w % 8 == 0 && cout_w >= a->cout && cout_w <= 64 && cout_w <= a->cout_stride,
                      "conv3x3: bad cout_w %d (multiple of 8, cout <= cout_w <= min(64, cout_stride))", cout_w);
        SRHIP_REQUIRE((long long)a->h * a->w * a->cout_stride * 2 < (1ll << 31), "conv3x3: output image too large for 32-bit offsets");
    }
    ConvParams p;
    p.src_a = (const char*)a->src_a;
    p.src_b = a->src_b ? (const char*)a->src_b : (const char*)a->src_a;
    p.wpack = (const bf16x8*)a->wpack;
    p.bias = a->bias;
    p.res = (const char*)a->res;
    p.dst = (char*)a->dst;
    p.n = a->n; p.h = a->h; p.w = a->w;
    p.ca = a->ca; p.cb = a->cb; p.ca_stride = ca_stride;
    p.cout = a->cout; p.cout_stride = a->cout_stride; p.cout_w = cout_w;
    p.slope = a->act == SRHIP_ACT_NONE ? 1.0f : (a->act == SRHIP_ACT_RELU ? 0.0f : a->slope);
    p.alpha = a->alpha;
    // MASK form without an activation: the args' slope is the negative-side factor of the mask (LeakyReLU backward)
    p.mask_slope = (a->res_mode == SRHIP_RES_MASK && a->act == SRHIP_ACT_NONE) ? a->slope : 0.0f;
    SRHIP_REQUIRE(p.mask_slope >= 0.0f && p.mask_slope <= 1.0f, "conv3x3: the mask's slope must lie in [0, 1] (got %g)", (double)p.mask_slope);
    const int kc = (a->ca + a->cb) / 16;
    const int ct = (a->cout + 31) / 32;
    // supported source splits: one tensor of 16/32/48/64 channels, or concat(32 channels, 64 channels)
    if (a->cb != 0) SRHIP_REQUIRE(a->ca == 32 && a->cb == 64, "conv3x3: the only two-source form is concat(32 ch, 64 ch), got %d + %d", a->ca, a->cb);
    else SRHIP_REQUIRE(a->ca <= 64, "conv3x3: a single source carries at most 64 channels (got %d)", a->ca);
    SRHIP_REQUIRE(a->alpha > 0.0f, "conv3x3: alpha must be positive");
    SRHIP_REQUIRE(a->res_mode == SRHIP_RES_ADD || a->res_mode == SRHIP_RES_MASK || a->res_mode == SRHIP_RES_PRE, "conv3x3: bad res_mode %d", a->res_mode);
    hipStream_t s = (hipStream_t)stream;
    p.aux = (char*)a->aux;
    p.dst2 = (char*)a->dst2;
    if (a->aux_mode != SRHIP_AUX_NONE) {
        SRHIP_REQUIRE(a->aux != nullptr, "conv3x3: aux_mode %d needs the sign-bit tensor `aux`", a->aux_mode);
        SRHIP_REQUIRE(!a->dst_nchw_f32 && ct == 2 && a->cb == 0 && a->ca_stride == 0 && a->cout_w == 0 && a->cout_stride == 64,
                      "conv3x3: the sign-bit forms take one source and write whole 64-channel bf16 maps");
        SRHIP_REQUIRE(a->res == nullptr || a->res_mode == SRHIP_RES_ADD, "conv3x3: the sign-bit forms combine with the ADD residual only");
        SRHIP_REQUIRE((long long)a->h * ((a->w + 31) / 32) * 256 < (1ll << 31), "conv3x3: sign-bit tensor too large for 32-bit offsets");
        if (a->aux_mode == SRHIP_AUX_EMIT) {
            if (kc == 4) return a->res ? launch_conv<Cfg<4, 2, 1, false, 1, 1>>(p, s) : launch_conv<Cfg<4, 2, 1, false, 0, 1>>(p, s);
            if (kc == 1 && a->res == nullptr) return launch_conv<Cfg<1, 2, 1, false, 0, 1>>(p, s);
        } else if (a->aux_mode == SRHIP_AUX_MASKBITS) {
            if (kc == 4 && a->res != nullptr) return launch_conv<Cfg<4, 2, 1, false, 1, 2>>(p, s);
        } else if (a->aux_mode == SRHIP_AUX_DUALBITS) {
            SRHIP_REQUIRE(a->dst2 != nullptr && a->dst2 != a->dst, "conv3x3: the dual form needs a second output tensor `dst2`");
            if (kc == 4 && a->res == nullptr) return launch_conv<Cfg<4, 2, 1, false, 0, 3>>(p, s);
        }
        srhip_set_error("conv3x3: sign-bit form %d is not built for cin=%d with%s residual", a->aux_mode, a->ca, a->res ? "" : "out");
        return SRHIP_ERR_INVALID;
    }
    // concat(32, 64) -> 64 without a residual: the reduction-split kernel (SRHIP_NO_KSPLIT=1 keeps the output-channel split
    // for A/B timing)
    if (kc == 6 && ct == 2 && a->res == nullptr && !a->dst_nchw_f32) {
        static const bool ksplit = []() { const char* e = getenv("SRHIP_NO_KSPLIT"); return !(e && e[0] == '1'); }();
        if (ksplit) return launch_conv_with<KsCfg>(p, s, conv3x3_ksplit_kernel);
    }
#define SRHIP_CONV_CASE(KC, CT, CSPLIT)                                                                 \
    if (kc == KC && ct == CT) {                                                                         \
        if (a->dst_nchw_f32) return launch_conv<Cfg<KC, CT, CSPLIT, true, 0>>(p, s);                    \
        if (a->res != nullptr && a->res_mode == SRHIP_RES_MASK) return launch_conv<Cfg<KC, CT, CSPLIT, false, 2>>(p, s); \
        if (a->res != nullptr && a->res_mode == SRHIP_RES_PRE) return launch_conv<Cfg<KC, CT, CSPLIT, false, 3>>(p, s); \
        if (a->res != nullptr) return launch_conv<Cfg<KC, CT, CSPLIT, false, 1>>(p, s);                 \
        return launch_conv<Cfg<KC, CT, CSPLIT, false, 0>>(p, s);                                        \
    }
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
