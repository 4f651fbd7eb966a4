// Two chained 3x3 / stride 1 / pad 1 convolutions of a 64-channel residual block as ONE kernel for gfx950:
//
//     mid = ReLU(conv1(x) + b1)                       (never leaves the chip unless the caller asks for it)
//     dst = alpha * act2(conv2(mid) + b2) + x
//
// i.e. SRResidualBlock (act2 none, alpha 0.2) and Residual(ConvActivation, ConvActivation) (act2 ReLU, alpha 1) of the
// reference's default network.  As two launches the pair moves 128 + 128 (first layer) + 128 + 128 + 128 (second
// layer with its residual) = 640 bytes of HBM traffic per pixel; fused it reads x once and writes dst once: 256.
//
// Structure: the streaming implicit GEMM of conv3x3_mfma.hip (one wave per SIMD, a layer's whole weight tensor pinned
// in the wave's registers, input rows flowing HBM -> LDS by LDS-DMA a few rows ahead, v_mfma_f32_32x32x16_bf16 with
// A = weights and B = 32 pixels), with the four waves of a workgroup SPECIALISED: waves 0-1 hold the first layer's
// weights and produce rows of `mid` for 64 columns into an LDS ring (bias through the accumulator's initial value,
// ReLU and the zero padding of `mid` outside the image applied on packed bf16 pairs, written pixel-major with the same
// XOR swizzle the LDS-DMA gives the x rows); waves 2-3 hold the second layer's weights, read `mid` rows from that ring
// exactly as the first layer reads x rows, and run the single-layer kernel's epilogue (fp32 staging through LDS,
// bias / alpha / activation / residual, 16-byte whole-pixel stores).  The second layer runs four rows behind the first:
// one barrier per row step hands a finished `mid` row over.  A strip yields 62 output columns (the first layer
// computes the 64 `mid` columns they need, from 66 columns of x); a vertical segment of R rows costs R + 2 rows of
// first-layer work.
//
// Every vector-memory operation and every LDS read is asm with a COUNTED wait, as in the single-layer kernel; the same
// fence discipline applies (tools/isa_lint.py checks it).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "conv_mfma_util.h"
#include "srhip_common.h"

namespace {

struct PairParams {
    const char* src;
    const bf16x8* wpack1;
    const float* bias1;
    const bf16x8* wpack2;
    const float* bias2;
    char* mid;
    char* dst;
    int n, h, w;
    float slope2, alpha;
    int nstrips, nsegs, seg_h, total_items;
    // Two-class partition (one work item per workgroup): items [0, main_items) are the uniform (strip, segment) grid over
    // rows [0, main_h); items [main_items, total_items) are the strips' remaining rows [main_h, h), walked with stride
    // tail_stride by the workgroups the uniform grid leaves idle.  Uniform partition: main_items = total_items, main_h = h.
    int main_items, main_h, tail_stride;
    // Range partition (range_rows > 0, used when the strips outnumber the workgroups unevenly): the rows of all strips
    // form one sequence (strip-major) and workgroup b takes rows [b * range_rows, (b + 1) * range_rows) of it, i.e. the
    // end of one strip and the start of the next; every maximal run inside one strip is a work item.
    int range_rows;
    int cout2;   // head form: output channels (1..32) of the second layer, written as fp32 NCHW planes
    int two;     // output columns per strip (the geometry's TWO)
    char* midbits;   // sign bits of `mid` in first-layer lane order (BITS forms), (n, strip, row, wave, lane) dwords
};

struct PairCfg {
    static constexpr int CH = 64;                      // channels of x, mid and dst
    static constexpr int KC = 4, NCT = 2;              // 16-channel K planes, 32-wide output-channel tiles
    static constexpr int MPT = KC * NCT;               // MFMAs per tap
    static constexpr int TWO = 62;                     // output columns per strip
    static constexpr int XPX = 66;                     // x columns per row held in LDS (mid: 64)
    static constexpr int PITCH = 72;                   // pixels per LDS row image (whole 1-KiB pieces)
    static constexpr int SLOT_BYTES = PITCH * CH * 2;  // 9216
    static constexpr int NPIECE = SLOT_BYTES / 1024;   // 9 LDS-DMA pieces per x row
    static constexpr int NW1 = 2;                      // first-layer waves (32 columns of `mid` each)
    static constexpr int MAXP = 5;                     // pieces per first-layer wave per row (wave w: w, w+2, ..., w+8)
    static constexpr int LA = 5;                       // x rows fetched ahead of the row being computed
    static constexpr int RING = 8;                     // x row slots: rows s-2 (the second layer's residual) .. s+LA in step s
    static constexpr int NY = 4;                       // mid row slots
    static constexpr int LAG = 4;                      // the second layer computes output row s - LAG in step s
    static constexpr int XRING_BYTES = RING * SLOT_BYTES;
    static constexpr int YRING_OFF = XRING_BYTES;
    static constexpr int YRING_BYTES = NY * SLOT_BYTES;
    // second layer: epilogue staging as in the single-layer kernel (32 px x 64 fp32 per wave)
    static constexpr int NSLOT = 8 * NCT, PXS = NSLOT * 16, LPP = NSLOT / 2, PPI = 64 / LPP, NST = 2 * NCT;
    static constexpr int STAGE_OFF = YRING_OFF + YRING_BYTES;
    static constexpr int STAGE_BYTES = 32 * PXS;
    static constexpr int DUMMY_OFF = STAGE_OFF + 2 * STAGE_BYTES;  // 1 KiB sink of the one DMA piece nobody owns
    // sign-bit rows of `mid` (the backward form): four row slots of one dword per first-layer lane
    static constexpr int BITS_OFF = DUMMY_OFF + 1024;
    static constexpr int BITS_ROW = NW1 * 256;
    static constexpr int LDS_BYTES = BITS_OFF + 4 * BITS_ROW;
    static constexpr int NFA1 = 256 / 4;               // first layer: accumulators in VGPRs, 64 fragments in AGPRs
    static constexpr int NFA2 = (256 - 16 * NCT) / 4;  // second layer: accumulators a[0:31], 56 fragments in AGPRs
};

// Geometry of the HEAD3 form (the network's last two layers, 64 -> 64 -> cout2 <= 8): THREE first-layer waves and one wave
// for the narrow second layer.  In the two-plus-two split above a narrow second layer leaves its two waves idle half of
// the time and the launch costs what the first layer costs on two of the four SIMDs.  Here the second layer is
// reformulated so that one wave carries it for 96 columns: the matrix instruction's 32 output rows hold (kernel row dy,
// output channel co) pairs, so ONE pass over a `mid` row with the three column shifts (3 x 4 MFMAs per 32 pixels instead
// of 9 x 4) yields that row's contribution to the three output rows it touches, and the sum over dy is three vector
// adds per output in registers.  A strip is 92 output columns (94 `mid` columns from 96 columns of x: twelve 1-KiB
// pieces per row, four per first-layer wave); the x ring needs no residual rows, so six slots of 12 KiB keep every
// slot inside the 16-bit offset field of the LDS instructions.
struct HeadCfg {
    static constexpr int CH = 64;
    static constexpr int KC = 4, NCT = 2;
    static constexpr int MPT = KC * NCT;
    static constexpr int NW1 = 3;                      // first-layer waves
    static constexpr int TWO = 92;                     // output columns per strip
    static constexpr int XPX = 96;                     // x columns per row held in LDS (valid mid columns: 94)
    static constexpr int PITCH = 96;
    static constexpr int SLOT_BYTES = PITCH * CH * 2;  // 12288
    static constexpr int NPIECE = SLOT_BYTES / 1024;   // 12
    static constexpr int MAXP = 4;                     // wave w: pieces w, w+3, w+6, w+9
    static constexpr int LA = 5;
    static constexpr int RING = 6;                     // x rows s .. s+LA in step s
    static constexpr int NY = 2;                       // mid rows: s-1 being written, s-2 being read by the head wave
    static constexpr int LAG = 4;                      // steps per item beyond its output rows
    static constexpr int XRING_BYTES = RING * SLOT_BYTES;
    static constexpr int YRING_OFF = XRING_BYTES;
    static constexpr int YRING_BYTES = NY * SLOT_BYTES;
    static constexpr int DUMMY_OFF = YRING_OFF + YRING_BYTES;   // also absorbs the head wave's reads of mid pixels 96, 97
    static constexpr int LDS_BYTES = DUMMY_OFF + 1024;
    static constexpr int NFA1 = 256 / 4;
    static_assert(NPIECE == NW1 * MAXP && (RING - 1) * SLOT_BYTES < 65536, "piece ownership / LDS offset field");
};

// MFMA of the first layer: accumulator in VGPRs (the epilogue's vector ALU reads it directly); the first MFMA of a
// row's chain takes the bias (accumulator order) as C.
template <bool IN_AGPR, bool FIRST>
__device__ __forceinline__ void mfma_vacc(f32x16& acc, const bf16x8& w, const bf16x8& b, const f32x16& cinit)
{
    if constexpr (IN_AGPR) {
        if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(w), "v"(b), "v"(cinit));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
    } else {
        if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(w), "v"(b), "v"(cinit));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));
    }
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
// ReLU on a packed pair of bf16 values: a negative bf16 is a negative int16, so max(., 0) per half clears it.
__device__ __forceinline__ unsigned relu_bf16x2(unsigned v)
{
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}
template <int OFF>
__device__ __forceinline__ void lds_write_b64_off(unsigned addr, unsigned lo, unsigned hi)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    const unsigned long long d = (unsigned long long)lo | ((unsigned long long)hi << 32);
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(d), "n"(OFF) : "memory");
}
// addr ^ (C << 4) as an instruction the optimiser cannot hoist: with compile-time ring slots the 8 swizzled write
// addresses of a row are loop invariants, and 8 more live registers do not exist in the first-layer waves.
template <int C, bool OPAQUE>
__device__ __forceinline__ unsigned xor_slot16(unsigned addr)
{
    if constexpr (!OPAQUE) return addr ^ (unsigned)(C << 4);
    unsigned r;
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "n"(C << 4), "v"(addr));
    return r;
}
__device__ __forceinline__ u32x4 lds_read_u32x4(unsigned addr)
{
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
// Fragment read whose ring slot enters through the instruction's 16-bit offset field: the row loops are unrolled over
// the ring (8 x rows, 4 mid rows), so the slot of every read is a compile-time constant and no address arithmetic runs
// per read (one vector add per fragment otherwise: 36 per row and wave).
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_u32x4_off(unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_b128_off(unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// slot offset in the instruction (IMM) or added to the address
template <bool IMM, int OFF>
__device__ __forceinline__ bf16x8 lds_read_b128_sel(unsigned dyn_off, unsigned addr)
{
    if constexpr (IMM) return lds_read_b128_off<OFF>(addr);
    else return lds_read_b128(dyn_off + addr);
}

// `mid` written by the waves that read it: a fragment of the centre tap column is 8 consecutive channels of this lane's
// own output pixel, so the four fragments of a row are the row's 64 channels (32 contiguous bytes per lane pair).
template <int OFF>
__device__ __forceinline__ void store_fragment(const bf16x8& f, const i32x4 srd, unsigned voffset, unsigned soffset)
{
    static_assert(OFF >= 0 && OFF < 4096, "buffer instruction offset field is 12 bits");
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4\n\ts_nop 3" ::"v"(f), "v"(voffset), "s"(srd), "s"(soffset), "n"(OFF) : "memory");
}

struct PairItem {
    int n, x0, y0, nout;
};
__device__ __forceinline__ PairItem pair_decode_item(const PairParams& p, int item)
{
    PairItem it;
    if (item < p.main_items) {
        const int strip = item % p.nstrips;
        const int tseg = item / p.nstrips;
        const int seg = tseg % p.nsegs;
        it.n = tseg / p.nsegs;
        it.x0 = strip * p.two;
        it.y0 = seg * p.seg_h;
        it.nout = min(it.y0 + p.seg_h, p.main_h) - it.y0;
    } else {
        const int k = item - p.main_items;
        it.n = k / p.nstrips;
        it.x0 = (k % p.nstrips) * p.two;
        it.y0 = p.main_h;
        it.nout = p.h - p.main_h;
    }
    return it;
}
// The work items of this workgroup, in order; every wave of the workgroup walks the same sequence.
struct PairIter {
    int item, stride;   // item partitions: first, first + stride, ... (a workgroup that owns a main item owns nothing
                        // else when the partition has a tail class)
    int r, r1;          // range partition: rows [r, r1) of the strip-major row sequence
};
__device__ __forceinline__ PairIter pair_iter_begin(const PairParams& p)
{
    PairIter i;
    i.item = (int)blockIdx.x;
    i.stride = p.main_items == p.total_items ? (int)gridDim.x : ((int)blockIdx.x < p.main_items ? p.total_items : p.tail_stride);
    i.r = (int)blockIdx.x * p.range_rows;
    i.r1 = min(i.r + p.range_rows, p.nstrips * p.n * p.h);
    return i;
}
__device__ __forceinline__ bool pair_iter_next(const PairParams& p, PairIter& i, PairItem& it)
{
    if (p.range_rows > 0) {
        if (i.r >= i.r1) return false;
        const int col = i.r / p.h, y = i.r - col * p.h;
        it.n = col / p.nstrips;
        it.x0 = (col - it.n * p.nstrips) * p.two;
        it.y0 = y;
        it.nout = min(p.h - y, i.r1 - i.r);
        i.r += it.nout;
        return true;
    }
    if (i.item >= p.total_items) return false;
    it = pair_decode_item(p, i.item);
    i.item += i.stride;
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// Waves 0-1: first layer.  Step s computes mid row s (image row y0 - 1 + s) for the 32 columns x0 - 1 + 32 pg + pl from
// x rows s .. s+2 of the ring; the row's epilogue (accumulators -> bf16 -> LDS) runs between the MFMAs of step s + 1.
// ------------------------------------------------------------------------------------------------------------------
// BITS: how the activation between the two layers is applied.
//   0  ReLU (forward).
//   1  ReLU, and the sign pattern of `mid` is also WRITTEN, one dword per lane and row in this wave's accumulator order
//      (bit j / 16 + j: the even / odd channel of channel pair j of the lane's pixel): 8 bytes per pixel instead of the
//      128 of the bf16 map, for every row and column the wave computes (halo included; neighbours write equal values).
//   2  no ReLU: `mid` is zeroed where such a bit is clear.  This is the BACKWARD of the block: with transposed weights,
//      x = dL/d(out) and the bits of the forward's `mid`, the launch yields conv_a^T((conv_b^T x) * [mid > 0]) * alpha + x.
// (`mid` itself, when the caller asks for it, is written by the waves that READ it: the second layer's.)
template <class P, int BITS>
__device__ __forceinline__ void pair_first_layer(const PairParams& p, char* smem, const unsigned lds_base, const int pg,
                                                 const int lane)
{
    const int hl = lane >> 5, pl = lane & 31;
    constexpr bool IMM = true;                     // compile-time ring slots (see `step`)
    constexpr int UNR = P::RING;                   // copies of the row loop's body
    // vector-memory schedule of one step: [BITS 2: LDS-DMA of the sign-bit dwords of mid row s+2, first] taps 0..MAXP-1
    // [DMA piece t of x row s+LA] [BITS 1: the sign-bit dword of mid row s-1, tap 7].  The bits a step applies (row s-1)
    // were fetched three steps earlier, as that step's FIRST operation: MAXP + 2 VM_STEP younger ones, which the wait for
    // x row s+2 (2 VM_STEP younger ones) covers.  They travel through LDS like the x rows: a register that is the target
    // of a load in flight cannot be kept out of the compiler's copies across the unrolled loop's back edge.
    constexpr int VM_STEP = P::MAXP + (BITS != 0 ? 1 : 0);
    constexpr int WAIT_STEADY = (BITS == 1 ? 1 : 0) + (P::LA - 3) * VM_STEP;   // row s+2 was issued in step s+2-LA
    constexpr int WAIT_EARLY = P::MAXP + VM_STEP < 2 * VM_STEP ? P::MAXP + VM_STEP : 2 * VM_STEP;
    static_assert(WAIT_STEADY <= 63 && P::LA == 5, "vmcnt is a 6-bit field; the early-step waits assume LA = 5");
    static_assert(BITS == 0 || P::RING % 4 == 0, "the sign-bit rows rotate over four LDS slots");

    bf16x8 wreg[9][P::KC][P::NCT];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int kc = 0; kc < P::KC; ++kc)
#pragma unroll
            for (int ct = 0; ct < P::NCT; ++ct) wreg[t][kc][ct] = p.wpack1[((t * P::KC + kc) * P::NCT + ct) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
    }
    // The weight loads must have landed BEFORE the row loop: a wait the compiler leaves at their first use sits between
    // the MFMAs of every row and drains the hand-counted vector-memory queue there.  An empty asm that "uses" each
    // fragment (in the register file it is pinned to) makes the compiler place its wait here.
    static_for<0, 9 * P::KC * P::NCT>([&](auto f_c) {
        SRHIP_CAPTURE(wreg);
        constexpr int f = decltype(f_c)::value;
        if constexpr (f < P::NFA1) asm volatile("" : "+a"(wreg[f / (P::KC * P::NCT)][(f / P::NCT) % P::KC][f % P::NCT]));
        else asm volatile("" : "+v"(wreg[f / (P::KC * P::NCT)][(f / P::NCT) % P::KC][f % P::NCT]));
    });
    // bias in accumulator order: register r of tile ct <-> channel 32 ct + 8 (r >> 2) + 4 hl + (r & 3)
    f32x16 biasv[P::NCT];
#pragma unroll
    for (int ct = 0; ct < P::NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            biasv[ct][r] = p.bias1 != nullptr ? p.bias1[ct * 32 + 8 * (r >> 2) + 4 * hl + (r & 3)] : 0.0f;

    // fragment of k-chunk kr for tap column kx: chunk 2 kr + hl of LDS pixel 32 pg + pl + kx (chunk c of pixel px sits at
    // 16-byte slot c ^ ((px >> 1) & 7)); slot bases are multiples of 1 KiB, so one add per read forms the address
    unsigned rd[3][P::KC];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = 32 * pg + pl + kx;
#pragma unroll
        for (int kr = 0; kr < P::KC; ++kr) rd[kx][kr] = lds_base + (unsigned)(px * 128 + (((2 * kr + hl) ^ ((px >> 1) & 7)) << 4));
    }
    const int q = 32 * pg + pl;                                             // this lane's mid pixel in the row image
    // its quad (ct, g) goes to wy ^ ((4 ct + g) << 4) of the row's slot (slot offsets are multiples of 1 KiB: the XOR
    // of bits 4..6 commutes with adding them, so the slot enters through the write's offset field)
    const unsigned wy = lds_base + (unsigned)P::YRING_OFF + (unsigned)(q * 128 + 8 * hl + (((q >> 1) & 7) << 4));

    const int H = p.h, W = p.w;
    bf16x8 bq[2][P::KC];
    f32x16 acc[2][P::NCT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ct = 0; ct < P::NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][ct][r] = 0.0f;
    unsigned e_p0 = 0, e_p1 = 0, e_addr = 0;
    unsigned mb = 0u;   // BITS 2: sign bits of the mid row in the epilogue;  BITS 1: collects the bits of that row

    PairIter iter = pair_iter_begin(p);
    PairItem it;
    while (pair_iter_next(p, iter, it)) {
        const int x0 = it.x0, y0 = it.y0, nout = it.nout;
        const int nin = nout + 4;  // x rows rx = 0..nin-1 <-> image row y0 - 2 + rx
        const unsigned img = (unsigned)(H * W) * (unsigned)(P::CH * 2);
        const unsigned row_pitch = (unsigned)W * (unsigned)(P::CH * 2);
        const __amdgpu_buffer_rsrc_t rsrc_x =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + (long long)it.n * img), 0, (int)img, 0x00020000);
        // sign bits: (n, strip, image row, first-layer wave, lane) dwords
        const unsigned bits_img = (unsigned)H * (unsigned)(P::NW1 * 256);
        const i32x4 bits_srd = make_srd(BITS != 0 ? p.midbits + (long long)(it.n * p.nstrips + x0 / P::TWO) * bits_img : p.dst,
                                        BITS != 0 ? bits_img : 0u);
        const unsigned bits_vo = (unsigned)((pg * 64 + lane) * 4);
        auto bits_soff = [&](const int yimg) -> unsigned { return (yimg >= 0 && yimg < H) ? (unsigned)yimg * (unsigned)(P::NW1 * 256) : kOobOffset; };
        const __amdgpu_buffer_rsrc_t bits_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(BITS == 2 ? p.midbits + (long long)(it.n * p.nstrips + x0 / P::TWO) * bits_img : p.dst), 0,
            BITS == 2 ? (int)bits_img : 0, 0x00020000);
        // BITS 2: LDS-DMA of one row's dwords (this wave's 64) into row slot `slot`
        auto fetch_bits = [&](const int slot, const int yimg) {
            if constexpr (BITS == 2)
                lds_dma_b32(bits_rsrc, (__attribute__((address_space(3))) void*)(smem + P::BITS_OFF + slot * P::BITS_ROW + pg * 256),
                            (int)bits_vo, (int)bits_soff(yimg));
        };

        unsigned poff[P::MAXP];
#pragma unroll
        for (int m = 0; m < P::MAXP; ++m) {
            const int piece = pg + P::NW1 * m;
            const int j = piece * 64 + lane;
            const int px = j >> 3;
            const int c = (j & 7) ^ ((px >> 1) & 7);
            const int x = x0 - 2 + px;
            const bool ok = (x >= 0) && (x < W) && (px < P::XPX) && (piece < P::NPIECE);
            poff[m] = ok ? (unsigned)((x * P::CH + c * 8) * 2) : kOobOffset;
        }
        const int ycol = x0 - 1 + q;
        const unsigned colmask = (ycol >= 0 && ycol < W) ? 0xffffffffu : 0u;

        auto issue_piece = [&](auto m_c, const unsigned soff, const unsigned slot_off) {
            constexpr int m = decltype(m_c)::value;
            const int piece = pg + P::NW1 * m;
            const bool mine = (m < P::MAXP - 1) || (piece < P::NPIECE);
            const unsigned off = mine ? slot_off + (unsigned)piece * 1024u : (unsigned)P::DUMMY_OFF;
            lds_dma_b128(rsrc_x, (__attribute__((address_space(3))) void*)(smem + off), poff[m], soff);
        };
        auto row_soff = [&](const int rx) -> unsigned {
            const int y = y0 - 2 + rx;
            const bool yin = (y >= 0) & (y < H) & (rx < nin);
            return yin ? (unsigned)y * row_pitch : kOobOffset;
        };
        // epilogue of accumulator quad t = 4 ct + g: four consecutive channels of this lane's pixel (channel pairs 2t and
        // 2t + 1) -> bf16, ReLU or sign-bit mask, padding mask
        auto keep = [&](const float v, const unsigned bits, const int pos) -> float {   // v where bit `pos` is set, else 0
            return __uint_as_float(__float_as_uint(v) & (unsigned)__builtin_amdgcn_sbfe((int)bits, pos, 1));
        };
        auto pair_lo = [&](const f32x16& a, auto t_c, const unsigned bits) -> unsigned {
            constexpr int t = decltype(t_c)::value, g = t % 4;
            if constexpr (BITS == 2) return pack_bf16x2(keep(a[4 * g + 0], bits, 2 * t), keep(a[4 * g + 1], bits, 16 + 2 * t));
            else return pack_bf16x2(a[4 * g + 0], a[4 * g + 1]);
        };
        auto pair_hi = [&](const f32x16& a, auto t_c, const unsigned bits) -> unsigned {
            constexpr int t = decltype(t_c)::value, g = t % 4;
            if constexpr (BITS == 2) return pack_bf16x2(keep(a[4 * g + 2], bits, 2 * t + 1), keep(a[4 * g + 3], bits, 16 + 2 * t + 1));
            else return pack_bf16x2(a[4 * g + 2], a[4 * g + 3]);
        };
        auto act_mask = [&](const unsigned v, const unsigned ymask) -> unsigned {
            if constexpr (BITS == 2) return v & ymask;
            else return relu_bf16x2(v) & ymask;
        };
        // BITS 1: the two flags (value != 0, i.e. > 0 behind the ReLU) of a packed pair, at bits 0 and 16
        auto flags = [&](const unsigned v) -> unsigned {
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            const u16x2 one = {1, 1};
            return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, v), one));
        };

        // IMM (the form that does not store mid): the row loop is unrolled over the x ring, Q = s mod RING, so the slots
        // of rows s .. s+2 and of mid row s-1 are compile-time constants that enter through the offset field of the LDS
        // instructions.  The form that stores mid has no registers left for eight copies of the loop body's live ranges:
        // it is unrolled over the row parity only (Q = s mod 2) and adds the slot offsets per access.
        auto step = [&](auto q_c, const int s) {
            constexpr int Q = decltype(q_c)::value;
            constexpr int PAR = Q & 1;
            const bool compute = s < nout + 2;
            if (compute) {
                if (s == 0) vm_wait<0>();
                else if (s < P::LA - 2) vm_wait<WAIT_EARLY>();
                else vm_wait<WAIT_STEADY>();
            }
            __builtin_amdgcn_s_barrier();

            // mid row s-1 (accumulators of the other parity) goes to slot (s-1) & 3; rows outside the image are zeros
            const int yr = s - 1;
            const int yimg = y0 - 1 + yr;
            // offset of slot (s - 1) mod NY: in the instruction (IMM) or added to the address registers
            static_assert(P::RING % P::NY == 0 && P::RING % 2 == 0, "the step's ring phase fixes the mid slot and the row parity");
            constexpr int YSLOT = IMM ? ((Q + P::NY - 1) % P::NY) * P::SLOT_BYTES : 0;
            const unsigned yslot_dyn = IMM ? 0u : (unsigned)(((s + P::NY - 1) % P::NY) * P::SLOT_BYTES);
            const unsigned wys = wy + yslot_dyn;
            unsigned rowx[3];   // !IMM: x-ring slot offsets of rows s .. s+2
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) rowx[ky] = IMM ? 0u : (unsigned)(((s + ky) % P::RING) * P::SLOT_BYTES);
            const unsigned ymask = (s >= 1 && yimg >= 0 && yimg < H) ? colmask : 0u;
            if constexpr (BITS == 2) {
                // the sign bits of mid row s-1 (landed: see the schedule above; the last row's after a full drain)
                if (compute || s == nout + 2) {
                    if (!compute) vm_wait<0>();
                    mb = lds_read_b32_off<((Q + 3) & 3) * P::BITS_ROW>(lds_base + (unsigned)(P::BITS_OFF + pg * 256 + lane * 4));
                }
            }

            if (!compute) {
                if (s == nout + 2) {
                    // the last mid row's epilogue, with nothing to hide it behind
                    if constexpr (BITS == 2) {
                        lds_wait<0>();
                        SRHIP_LOAD_FENCE(mb);
                    }
                    static_for<0, 8>([&](auto t_c) {
                        SRHIP_CAPTURE(acc); SRHIP_CAPTURE(mb);
                        constexpr int t = decltype(t_c)::value;
                        const unsigned o0 = act_mask(pair_lo(acc[PAR ^ 1][t / 4], t_c, mb), ymask);
                        const unsigned o1 = act_mask(pair_hi(acc[PAR ^ 1][t / 4], t_c, mb), ymask);
                        if constexpr (BITS == 1) mb = (t == 0 ? 0u : mb) | (flags(o0) << (2 * t)) | (flags(o1) << (2 * t + 1));
                        lds_write_b64_off<YSLOT>(xor_slot16<4 * (t / 4) + (t % 4), IMM>(wys), o0, o1);
                    });
                    if constexpr (BITS == 1)
                        asm volatile("buffer_store_dword %0, %1, %2, %3 offen" ::"v"(mb), "v"(bits_vo), "s"(bits_srd), "s"(bits_soff(yimg)) : "memory");
                    lds_wait<0>();
                }
                return;
            }

            const unsigned dma_soff = row_soff(s + P::LA);
            const unsigned dma_slot = (unsigned)(((s + P::LA) % P::RING) * P::SLOT_BYTES);

            if constexpr (Q == 0) {
                if (s == 0) {  // first row of the item: nobody fetched its tap 0 yet
                    static_for<0, P::KC>([&](auto kc_c) {
                        constexpr int kc = decltype(kc_c)::value;
                        bq[0][kc] = lds_read_b128_sel<IMM, 0>(rowx[0], rd[0][kc]);
                    });
                    lds_wait<0>();
                    static_for<0, P::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[0][decltype(kc_c)::value]); });
                }
            }

            static_for<0, 9>([&](auto t_c) {
                SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(mb); SRHIP_CAPTURE(wreg); SRHIP_CAPTURE(biasv);
                SRHIP_CAPTURE(e_p0); SRHIP_CAPTURE(e_p1); SRHIP_CAPTURE(e_addr);
                constexpr int t = decltype(t_c)::value;
                constexpr int cur = (PAR + t) & 1, nxt = cur ^ 1;
                constexpr int tn = (t + 1) % 9;
                constexpr int kyn = (t == 8) ? 1 : (t + 1) / 3;
                // this tap's fragments (issued during the previous tap) have landed; the mid write issued BEHIND them may
                // still be in flight (LDS operations of a wave complete in order)
                if constexpr (t >= 1) lds_wait<1>();
                else lds_wait<0>();
                static_for<0, P::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[cur][decltype(kc_c)::value]); });
                if constexpr (BITS == 2 && t == 0) SRHIP_LOAD_FENCE(mb);
                static_for<0, P::MPT>([&](auto m_c) {
                    SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(mb); SRHIP_CAPTURE(wreg); SRHIP_CAPTURE(biasv);
                    SRHIP_CAPTURE(e_p0); SRHIP_CAPTURE(e_p1); SRHIP_CAPTURE(e_addr);
                    constexpr int ml = decltype(m_c)::value;
                    constexpr int kc = ml / P::NCT, ct = ml % P::NCT;
                    constexpr int f = (t * P::KC + kc) * P::NCT + ct;
                    mfma_vacc<(f < P::NFA1), (t == 0 && kc == 0)>(acc[PAR][ct], wreg[t][kc][ct], bq[cur][kc], biasv[ct]);
                    // gaps 1..KC: pixel fragments of the next tap (tap 8: tap 0 of the next row, whose x rows are resident)
                    if constexpr (ml >= 1 && ml <= P::KC) {
                        constexpr int kr = ml - 1;
                        bq[nxt][kr] = lds_read_b128_sel<IMM, ((Q + kyn) % P::RING) * P::SLOT_BYTES>(rowx[kyn], rd[tn % 3][kr]);
                    }
                    // BITS 2: this step's first vector-memory operation fetches the sign bits of mid row s+2 (used in step s+3)
                    if constexpr (BITS == 2 && t == 0 && ml == 0) fetch_bits((Q + 2) & 3, yimg + 3);
                    // taps 0..7: quad t of the previous row's accumulators -> mid ring
                    if constexpr (t < 8) {
                        constexpr int qct = t / 4, qg = t % 4;
                        if constexpr (ml == 0) e_p0 = pair_lo(acc[PAR ^ 1][qct], IntC<t + 0 * ml>{}, mb);
                        if constexpr (ml == 1) e_p1 = pair_hi(acc[PAR ^ 1][qct], IntC<t + 0 * ml>{}, mb);
                        if constexpr (ml == 2) e_p0 = act_mask(e_p0, ymask);
                        if constexpr (ml == 3) e_p1 = act_mask(e_p1, ymask);
                        if constexpr (BITS == 1 && ml == 4) mb = (t == 0 ? 0u : mb) | (flags(e_p0) << (2 * t)) | (flags(e_p1) << (2 * t + 1));
                        if constexpr (ml == 5) e_addr = xor_slot16<4 * qct + qg, IMM>(wys);
                        if constexpr (ml == 7) lds_write_b64_off<YSLOT>(e_addr, e_p0, e_p1);
                    }
                    if constexpr (t < P::MAXP && ml == 6) issue_piece(IntC<t + 0 * ml>{}, dma_soff, dma_slot);
                    // BITS 1: the finished row's sign bits (behind the step's last DMA piece: WAIT_STEADY counts it)
                    if constexpr (BITS == 1 && t == 7 && ml == 6)
                        asm volatile("buffer_store_dword %0, %1, %2, %3 offen" ::"v"(mb), "v"(bits_vo), "s"(bits_srd), "s"(bits_soff(s >= 1 ? yimg : -1)) : "memory");
                    __builtin_amdgcn_sched_barrier(0);  // keep the interleave as written
                });
            });
            // MFMA result -> vector ALU readers of the next step: the hazard is invisible to the compiler inside asm
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[PAR][0]), "+v"(acc[PAR][1]));
            // nothing stays in flight across the end of a step: next row's tap-0 fragments, the read-back, the mid writes
            lds_wait<0>();
            static_for<0, P::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[PAR ^ 1][decltype(kc_c)::value]); });
        };

        __builtin_amdgcn_s_barrier();  // every wave is done with the previous item's LDS rows
#pragma unroll 1
        for (int r = 0; r < P::LA; ++r) {
            const unsigned so = row_soff(r);
            static_for<0, P::MAXP>([&](auto m_c) { issue_piece(m_c, so, (unsigned)(r * P::SLOT_BYTES)); });
        }
        // the sign bits of mid rows 0 and 1 (image rows y0 - 1, y0): step 0 waits for everything issued so far
        fetch_bits(0, y0 - 1);
        fetch_bits(1, y0);
        const int nsteps = nout + P::LAG;
#pragma unroll 1
        for (int s = 0; s < nsteps; s += UNR) {
            static_for<0, UNR>([&](auto q_c) {
                constexpr int Q = decltype(q_c)::value;
                if (Q == 0 || s + Q < nsteps) step(q_c, s + Q);
            });
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Waves 2-3: second layer.  Step s computes output row o = s - LAG (image row y0 + o) for the columns x0 + 32 pg + pl
// from mid rows o .. o+2 of the ring; epilogue as in the single-layer kernel with the residual form.
// ------------------------------------------------------------------------------------------------------------------
// ACT2: activation of the second layer known at compile time (0 none, 1 ReLU, 2 LeakyReLU): its epilogue is the wave's
// critical vector-ALU work, and the common forms need fewer instructions than max(t, slope * t).
template <int ACT2, bool STORE_MID>
__device__ __forceinline__ void pair_second_layer(const PairParams& p, char* smem, const unsigned lds_base, const int pg,
                                                  const int lane)
{
    using P = PairCfg;
    const int hl = lane >> 5, pl = lane & 31;
    // The residual (x at the output pixel) is still in the x ring: four pixel-major LDS reads per row replace a second
    // trip to memory.  The only vector-memory operations of these waves are the output stores: nothing to wait for.
    constexpr int NRES = P::NST;

    bf16x8 wreg[9][P::KC][P::NCT];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int kc = 0; kc < P::KC; ++kc)
#pragma unroll
            for (int ct = 0; ct < P::NCT; ++ct) wreg[t][kc][ct] = p.wpack2[((t * P::KC + kc) * P::NCT + ct) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
    }
    static_for<0, 9 * P::KC * P::NCT>([&](auto f_c) {   // the loads land before the row loop: see the first layer
        SRHIP_CAPTURE(wreg);
        constexpr int f = decltype(f_c)::value;
        if constexpr (f < P::NFA2) asm volatile("" : "+a"(wreg[f / (P::KC * P::NCT)][(f / P::NCT) % P::KC][f % P::NCT]));
        else asm volatile("" : "+v"(wreg[f / (P::KC * P::NCT)][(f / P::NCT) % P::KC][f % P::NCT]));
    });
    // store-time lane roles: lane = (pixel st_q of a group of 8 pixels, 8-channel chunk st_c)
    const int st_q = lane / P::LPP, st_c = lane % P::LPP;
    const int st_fw = pl & 15;
    const unsigned stage_lds = lds_base + (unsigned)P::STAGE_OFF + (unsigned)(pg * P::STAGE_BYTES);
    const unsigned wst = stage_lds + (unsigned)(pl * P::PXS) + (unsigned)(((hl ^ st_fw) & (P::NSLOT - 1)) << 4);
    unsigned rdst[P::NST];
#pragma unroll
    for (int k = 0; k < P::NST; ++k) {
        const int qq = k * P::PPI + st_q;
        rdst[k] = stage_lds + (unsigned)(qq * P::PXS) + (unsigned)((((2 * st_c) ^ (qq & 15)) & (P::NSLOT - 1)) << 4);
    }
    float breg[8];  // alpha * bias of this lane's 8 store-time channels
#pragma unroll
    for (int r = 0; r < 8; ++r) breg[r] = (p.bias2 != nullptr ? p.bias2[st_c * 8 + r] : 0.0f) * p.alpha;
    unsigned rd[3][P::KC];  // as in the first layer, based at the mid ring
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = 32 * pg + pl + kx;
#pragma unroll
        for (int kr = 0; kr < P::KC; ++kr)
            rd[kx][kr] = lds_base + (unsigned)P::YRING_OFF + (unsigned)(px * 128 + (((2 * kr + hl) ^ ((px >> 1) & 7)) << 4));
    }
    unsigned rres[NRES];  // residual of store group k: x-ring pixel 2 + 32 pg + 8 k + st_q (output column + 2), chunk st_c
#pragma unroll
    for (int k = 0; k < NRES; ++k) {
        const int px = 2 + 32 * pg + k * P::PPI + st_q;
        rres[k] = (unsigned)(px * 128 + ((st_c ^ ((px >> 1) & 7)) << 4));
    }

    const int H = p.h, W = p.w;
    bf16x8 bq[2][P::KC];
    f32x16 acc[P::NCT];
    u32x4 resv[2][NRES];
    f32x4 e_lo[2], e_hi[2];
    float ev[8];
    u32x4 eo = {0u, 0u, 0u, 0u};   // packed results of the store group in progress

    PairIter iter = pair_iter_begin(p);
    PairItem it;
    while (pair_iter_next(p, iter, it)) {
        const int x0 = it.x0, y0 = it.y0, nout = it.nout;
        const unsigned img = (unsigned)(H * W) * (unsigned)(P::CH * 2);
        const unsigned row_pitch = (unsigned)W * (unsigned)(P::CH * 2);
        const i32x4 dst_srd = make_srd(p.dst + (long long)it.n * img, img);
        // STORE_MID: tap 4's fragments are mid row o + 1 = the output row's own image row, pixel = this lane's output column
        const i32x4 mid_srd = make_srd(STORE_MID ? p.mid + (long long)it.n * img : p.dst, STORE_MID ? img : 0u);
        const int mcol = x0 + 32 * pg + pl;
        const unsigned mvo = (STORE_MID && 32 * pg + pl < P::TWO && mcol < W) ? (unsigned)((mcol * P::CH + hl * 8) * 2) : kOobOffset;
        unsigned vox[P::NST];
#pragma unroll
        for (int k = 0; k < P::NST; ++k) {
            const int jq = 32 * pg + k * P::PPI + st_q;   // output column within the strip
            const int xq = x0 + jq;
            vox[k] = (jq < P::TWO && xq < W) ? (unsigned)((xq * P::CH + st_c * 8) * 2) : kOobOffset;
        }

        auto epi_read = [&](auto k_c) {
            constexpr int k = decltype(k_c)::value;
            e_lo[k & 1] = lds_read_f32x4(rdst[k]);
            e_hi[k & 1] = lds_read_f32x4(rdst[k] ^ 16u);
        };
        auto epi_elem = [&](auto k_c, auto e_c, auto ap_c) {
            constexpr int k = decltype(k_c)::value;
            constexpr int e = decltype(e_c)::value;
            constexpr int AP = decltype(ap_c)::value;
            const float x = e < 4 ? e_lo[k & 1][e & 3] : e_hi[k & 1][e & 3];
            const float t = fmaf(x, p.alpha, breg[e]);
            const unsigned rr = resv[AP][k][e >> 1];
            const float rv = (e & 1) ? bf16_hi(rr) : bf16_lo(rr);
            if constexpr (ACT2 == 0) ev[e] = t + rv;
            else if constexpr (ACT2 == 1) ev[e] = fmaxf(t, 0.0f) + rv;
            else ev[e] = fmaxf(t, t * p.slope2) + rv;
            // the pair's pack rides with its second element: the store step is the store alone (a step that also held
            // the four packs overflowed the issue slots of its MFMA gap)
            if constexpr (e & 1) eo[e >> 1] = pack_bf16x2(ev[e - 1], ev[e]);
        };
        auto epi_store = [&](auto k_c, const unsigned store_soff) {
            constexpr int k = decltype(k_c)::value;
            buffer_store_b128_guarded(eo, dst_srd, vox[k], store_soff);
        };

        // One output row.  Q = o mod NY fixes the mid-ring slots of rows o .. o+2 (compile-time fragment addresses);
        // PAR = o & 1 selects the fragment buffer that holds tap 0 and the residual set.
        auto row_body = [&](auto q_c, const int o) {
            constexpr int Q = decltype(q_c)::value;
            constexpr int PAR = Q & 1;
            const int y = y0 + o;
            const unsigned xrow = lds_base + (unsigned)(((o + 2) & (P::RING - 1)) * P::SLOT_BYTES);  // x row of this output row
            const unsigned st_soff = o > 0 ? (unsigned)(y - 1) * row_pitch : kOobOffset;

            if constexpr (Q == 0) {
                if (o == 0) {
                    static_for<0, P::KC>([&](auto kc_c) {
                        constexpr int kc = decltype(kc_c)::value;
                        bq[0][kc] = lds_read_b128_off<0>(rd[0][kc]);
                    });
                    lds_wait<0>();
                    static_for<0, P::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[0][decltype(kc_c)::value]); });
                }
            }

            // Schedule of the row (previous row = o - 1, whose accumulators sit in the staging buffer / in acc[1]):
            //   tap 0   MFMAs in tile-major order (tile 0's four, then tile 1's); the gaps behind the first four hand the
            //           previous row's tile-1 accumulator to the staging buffer before MFMA 4 starts to overwrite it
            //   tap 8   tile-major again; the gaps behind MFMAs 4..7 stage THIS row's tile-0 accumulator, complete since
            //           MFMA 3: no accumulator is staged outside the MFMA stream, a row ends without a drain
            //   odd taps 1..7, gap 0     staging read-back of store group (t - 1) / 2 of the previous row
            //   even taps 0..6, gap 0    residual read t / 2 of this row
            //   taps 2k+2, 2k+3 (k < 3)  channels 0..3 / 4..7 + store of group k;  tap 8: all of group 3
            static_for<0, 9>([&](auto t_c) {
                SRHIP_CAPTURE(resv); SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(e_lo); SRHIP_CAPTURE(e_hi);
                SRHIP_CAPTURE(wreg); SRHIP_CAPTURE(ev); SRHIP_CAPTURE(eo);
                constexpr int t = decltype(t_c)::value;
                constexpr int cur = (PAR + t) & 1, nxt = cur ^ 1;
                constexpr int tn = (t + 1) % 9;
                constexpr int kyn = (t == 8) ? 1 : (t + 1) / 3;
                constexpr int KG = t == 8 ? P::NST - 1 : (t - 2) / 2;   // store group whose arithmetic runs in this tap
                constexpr bool EPI = t >= 2;
                constexpr bool TILE_MAJOR = t == 0 || t == 8;
                // tap 0 needs only its fragments, fenced at the end of the previous row; the staging writes issued
                // in tap 8 may still be in flight
                if constexpr (t >= 1) lds_wait<0>();
                if constexpr (EPI && t % 2 == 0 && t < 8) { SRHIP_LOAD_FENCE(e_lo[KG & 1]); SRHIP_LOAD_FENCE(e_hi[KG & 1]); }
                if constexpr (t == 8) { SRHIP_LOAD_FENCE(e_lo[KG & 1]); SRHIP_LOAD_FENCE(e_hi[KG & 1]); }
                if constexpr (t % 2 == 1 && t / 2 < NRES) SRHIP_LOAD_FENCE(resv[PAR][t / 2]);   // read in gap 0 of the previous tap
                static_for<0, P::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[cur][decltype(kc_c)::value]); });

                constexpr int ESTEPS = !EPI ? 0 : (t == 8 ? 9 : ((t & 1) ? 5 : 4));
                constexpr int NS = ESTEPS;
                static_for<0, P::MPT>([&](auto m_c) {
                    SRHIP_CAPTURE(resv); SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(wreg); SRHIP_CAPTURE(ev);
                    SRHIP_CAPTURE(e_lo); SRHIP_CAPTURE(e_hi); SRHIP_CAPTURE(eo);
                    constexpr int ml = decltype(m_c)::value;
                    constexpr int kc = TILE_MAJOR ? ml % P::KC : ml / P::NCT;
                    constexpr int ct = TILE_MAJOR ? ml / P::KC : ml % P::NCT;
                    constexpr int f = (t * P::KC + kc) * P::NCT + ct;
                    mfma_pinned<(f < P::NFA2), (t == 0 && kc == 0), ct>(acc[ct], wreg[t][kc][ct], bq[cur][kc]);
                    // Every LDS read is issued early in its tap, so the wait at the top of the next tap finds it complete.
                    if constexpr (ml == 0 && (t % 2 == 1) && (t / 2 < P::NST)) epi_read(IntC<t / 2 + 0 * ml>{});
                    if constexpr (ml == 0 && (t % 2 == 0) && (t / 2 < NRES)) resv[PAR][t / 2] = lds_read_u32x4(xrow + rres[t / 2]);
                    if constexpr (ml >= 1 && ml <= P::KC) {
                        constexpr int kr = ml - 1;
                        bq[nxt][kr] = lds_read_b128_off<((Q + kyn) & (P::NY - 1)) * P::SLOT_BYTES>(rd[tn % 3][kr]);
                    }
                    // accumulator quads -> staging (behind the tap's LDS reads: the end-of-row wait counts on that)
                    if constexpr (STORE_MID && t == 4 && (ml & 1)) store_fragment<(ml / 2) * 32>(bq[cur][ml / 2], mid_srd, mvo, (unsigned)y * row_pitch);
                    if constexpr (t == 0 && ml < 4) lds_write_acc_quad<1, ml>(wst ^ (unsigned)((8 + 2 * ml) << 4), acc[1]);
                    if constexpr (t == 8 && ml >= 4) lds_write_acc_quad<0, ml - 4>(wst ^ (unsigned)((2 * (ml - 4)) << 4), acc[0]);
                    static_for<ml * NS / P::MPT, (ml + 1) * NS / P::MPT>([&](auto s_c) {
                        SRHIP_CAPTURE(resv); SRHIP_CAPTURE(ev); SRHIP_CAPTURE(e_lo); SRHIP_CAPTURE(e_hi); SRHIP_CAPTURE(eo);
                        constexpr int S = decltype(s_c)::value;
                        if constexpr (S < ESTEPS) {
                            constexpr int KD = KG + 0 * S;
                            if constexpr (t == 8) {
                                if constexpr (S < 8) epi_elem(IntC<KD>{}, IntC<S + 0 * KD>{}, IntC<PAR ^ 1>{});
                                else epi_store(IntC<KD>{}, st_soff);
                            } else if constexpr (t % 2 == 0) {
                                epi_elem(IntC<KD>{}, IntC<S + 0 * KD>{}, IntC<PAR ^ 1>{});
                            } else {
                                if constexpr (S < 4) epi_elem(IntC<KD>{}, IntC<4 + S>{}, IntC<PAR ^ 1>{});
                                else epi_store(IntC<KD>{}, st_soff);
                            }
                        }
                    });
                    if constexpr (NS > 0 || TILE_MAJOR) __builtin_amdgcn_sched_barrier(0);
                });
            });
            // in flight across the end of the row: only tap 8's four staging writes (issued behind its fragment reads)
            lds_wait<4>();
            static_for<0, P::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[PAR ^ 1][decltype(kc_c)::value]); });
        };

        __builtin_amdgcn_s_barrier();  // item start (matches the first layer's)
        const int nsteps = nout + P::LAG;
        static_assert(P::LAG % P::NY == 0, "the row's ring phase is the step's");
#pragma unroll 1
        for (int s = 0; s < nsteps; s += P::NY) {
            // step s + Q: output row s + Q - LAG
            static_for<0, P::NY>([&](auto q_c) {
                constexpr int Q = decltype(q_c)::value;
                if (Q == 0 || s + Q < nsteps) {
                    __builtin_amdgcn_s_barrier();
                    if (s + Q >= P::LAG) row_body(q_c, s + Q - P::LAG);
                }
            });
        }
        // drain: the last row's tile-1 accumulator goes to the staging buffer (no next row does it), then its epilogue
        {
            asm volatile("s_nop 15\n\ts_nop 3" : "+" SRHIP_ACC_REG1(acc[1]));   // MFMA result -> LDS data read
            static_for<0, 4>([&](auto g_c) {
                SRHIP_CAPTURE(acc);
                constexpr int g = decltype(g_c)::value;
                lds_write_acc_quad<1, g>(wst ^ (unsigned)((8 + 2 * g) << 4), acc[1]);
            });
            const unsigned st_soff = (unsigned)(y0 + nout - 1) * row_pitch;
            auto tail = [&](auto ap_c) {
                static_for<0, P::NST>([&](auto k_c) {
                    SRHIP_CAPTURE(e_lo); SRHIP_CAPTURE(e_hi);
                    epi_read(k_c);
                    lds_wait<0>();
                    SRHIP_LOAD_FENCE(e_lo[decltype(k_c)::value & 1]);
                    SRHIP_LOAD_FENCE(e_hi[decltype(k_c)::value & 1]);
                    static_for<0, 8>([&](auto e_c) { epi_elem(k_c, e_c, ap_c); });
                    epi_store(k_c, st_soff);
                });
            };
            if ((nout - 1) & 1) tail(IntC<1>{});
            else tail(IntC<0>{});
        }
    }
}

// Head form of the second layer: 64 -> cout2 <= 32 channels (one MFMA tile), no residual, output fp32 NCHW -- the last two
// layers of the network (core/models/model.py:111-115: Conv + ReLU -> Conv(64 -> 3)) as one launch.  Half the MFMAs of a
// 64-channel second layer, so these waves wait for the first layer's most of the time: the launch costs what the first
// layer alone costs, and the 64-channel map between the two layers never reaches memory.  Same barrier sequence, mid-ring
// addressing and fragment pipeline as pair_second_layer; the epilogue goes straight from the accumulator to 4-byte plane
// stores (lane = pixel: 128 contiguous bytes per half-wave and channel).
template <int ACT2, bool STORE_MID>
__device__ __forceinline__ void pair_head_layer(const PairParams& p, char* smem, const unsigned lds_base, const int pg,
                                                const int lane)
{
    using P = PairCfg;
    const int hl = lane >> 5, pl = lane & 31;
    bf16x8 wreg[9][P::KC];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int kc = 0; kc < P::KC; ++kc) wreg[t][kc] = p.wpack2[(t * P::KC + kc) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
    }
    static_for<0, 9 * P::KC>([&](auto f_c) {   // the loads land before the row loop: see the first layer
        SRHIP_CAPTURE(wreg);
        constexpr int f = decltype(f_c)::value;
        asm volatile("" : "+a"(wreg[f / P::KC][f % P::KC]));
    });
    // accumulator register r of this lane is output channel 8 (r >> 2) + 4 hl + (r & 3) of pixel pl
    float breg[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = 8 * (r >> 2) + 4 * hl + (r & 3);
        breg[r] = (p.bias2 != nullptr && co < p.cout2) ? p.bias2[co] : 0.0f;
    }
    const int ngroups = (p.cout2 + 7) >> 3;   // register quads that hold real channels
    unsigned rd[3][P::KC];  // as in the first layer, based at the mid ring
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = 32 * pg + pl + kx;
#pragma unroll
        for (int kr = 0; kr < P::KC; ++kr)
            rd[kx][kr] = lds_base + (unsigned)P::YRING_OFF + (unsigned)(px * 128 + (((2 * kr + hl) ^ ((px >> 1) & 7)) << 4));
    }
    const int H = p.h, W = p.w;
    bf16x8 bq[2][P::KC];
    f32x16 acc;

    PairIter iter = pair_iter_begin(p);
    PairItem it;
    while (pair_iter_next(p, iter, it)) {
        const int x0 = it.x0, y0 = it.y0, nout = it.nout;
        const unsigned plane = (unsigned)(H * W) * 4u;
        const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.dst + (long long)it.n * p.cout2 * plane), 0, (int)((unsigned)p.cout2 * plane), 0x00020000);
        const int jq = 32 * pg + pl, xq = x0 + jq;   // output column within the strip / the image
        const bool col_ok = jq < P::TWO && xq < W;
        const unsigned mimg = (unsigned)(H * W) * (unsigned)(P::CH * 2);
        const i32x4 mid_srd = make_srd(STORE_MID ? p.mid + (long long)it.n * mimg : p.dst, STORE_MID ? mimg : 0u);
        const unsigned mvo = (STORE_MID && col_ok) ? (unsigned)((xq * P::CH + hl * 8) * 2) : kOobOffset;

        auto row_body = [&](auto q_c, const int o) {
            constexpr int Q = decltype(q_c)::value;   // o mod NY: compile-time mid-ring slots, as in pair_second_layer
            constexpr int PAR = Q & 1;
            const int y = y0 + o;
            if constexpr (Q == 0) {
                if (o == 0) {
                    static_for<0, P::KC>([&](auto kc_c) {
                        constexpr int kc = decltype(kc_c)::value;
                        bq[0][kc] = lds_read_b128_off<0>(rd[0][kc]);
                    });
                    lds_wait<0>();
                    static_for<0, P::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[0][decltype(kc_c)::value]); });
                }
            }
            static_for<0, 9>([&](auto t_c) {
                SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(wreg);
                constexpr int t = decltype(t_c)::value;
                constexpr int cur = (PAR + t) & 1, nxt = cur ^ 1;
                constexpr int tn = (t + 1) % 9;
                constexpr int kyn = (t == 8) ? 1 : (t + 1) / 3;
                lds_wait<0>();
                static_for<0, P::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[cur][decltype(kc_c)::value]); });
                static_for<0, P::KC>([&](auto kc_c) {
                    SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(wreg);
                    constexpr int kc = decltype(kc_c)::value;
                    mfma_pinned<true, (t == 0 && kc == 0), 0>(acc, wreg[t][kc], bq[cur][kc]);
                    if constexpr (STORE_MID && t == 4) store_fragment<kc * 32>(bq[cur][kc], mid_srd, mvo, (unsigned)(y * W) * (unsigned)(P::CH * 2));
                    bq[nxt][kc] = lds_read_b128_off<((Q + kyn) & (P::NY - 1)) * P::SLOT_BYTES>(rd[tn % 3][kc]);   // fragment kc of the next tap
                });
            });
            asm volatile("s_nop 15\n\ts_nop 3" : "+" SRHIP_ACC_REG0(acc));   // MFMA result -> the reads below
            const unsigned soff = (unsigned)(y * W) * 4u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < ngroups) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int co = 8 * g + 4 * hl + e;
                        const float t = acc[4 * g + e] + breg[4 * g + e];
                        float v = t;
                        if constexpr (ACT2 == 1) v = fmaxf(t, 0.0f);
                        else if constexpr (ACT2 == 2) v = fmaxf(t, t * p.slope2);
                        const unsigned vo = (col_ok && co < p.cout2) ? (unsigned)co * plane + (unsigned)xq * 4u : kOobOffset;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v * p.alpha), dst_rsrc, vo, soff, 0);
                    }
                }
            }
            lds_wait<0>();  // tap 8's fetch of the next row's tap 0: nothing stays in flight across rows
            static_for<0, P::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[PAR ^ 1][decltype(kc_c)::value]); });
        };

        __builtin_amdgcn_s_barrier();  // item start (matches the first layer's)
        const int nsteps = nout + P::LAG;
#pragma unroll 1
        for (int s = 0; s < nsteps; s += P::NY) {
            static_for<0, P::NY>([&](auto q_c) {
                constexpr int Q = decltype(q_c)::value;
                if (Q == 0 || s + Q < nsteps) {
                    __builtin_amdgcn_s_barrier();
                    if (s + Q >= P::LAG) row_body(q_c, s + Q - P::LAG);
                }
            });
        }
    }
}

// MFMA with everything in VGPRs (the head wave has registers to spare); FIRST starts a chain from C = 0.
template <bool FIRST>
__device__ __forceinline__ void mfma_vv(f32x16& acc, const bf16x8& w, const bf16x8& b)
{
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));
}

// HEAD3: the narrow second layer (64 -> cout2 <= 8, fp32 NCHW out, no residual) on ONE wave for the strip's three pixel
// groups.  Step s handles mid row r = s - 2 (complete since the barrier of this step): for each 32-pixel group
//     R[8 dy + co][px] = sum over dx, k of W2[co][k][dy][dx] * mid[r][px + dx][k]           (3 x 4 MFMAs)
// and, with out[o] = R_o[dy 0] + R_{o+1}[dy 1] + R_{o+2}[dy 2], two running sums per group and channel:
//     out[r-2] = S1 + R[dy 2] (+ bias, activation, alpha -> stored),   S1 <- S0 + R[dy 1],   S0 <- R[dy 0].
// Rows of `mid` outside the image are zeros in the ring, so the vertical padding needs nothing here.  Accumulator
// register 4 dy + e of lane (pl, hl) is channel 4 hl + e of pixel pl.
template <int ACT2, bool STORE_MID, class P>
__device__ __forceinline__ void pair_head3_layer(const PairParams& p, char* smem, const unsigned lds_base, const int lane)
{
    constexpr int NG = P::NW1;
    const int hl = lane >> 5, pl = lane & 31;
    // weights from the standard 64 -> cout2 pack ([tap][kc][lane][8], lane = (k half, output channel)): this lane is
    // matrix row 8 dy + co
    bf16x8 wh[3][P::KC];
    {
        const int m = lane & 31, dyw = m >> 3, cow = m & 7;
        const bool wok = dyw < 3 && cow < p.cout2;
        const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int kc = 0; kc < P::KC; ++kc) {
                const bf16x8 v = p.wpack2[((3 * (wok ? dyw : 0) + dx) * P::KC + kc) * 64 + (wok ? cow : 0) + 32 * hl];
                wh[dx][kc] = wok ? v : zero;
            }
    }
    static_for<0, 3 * P::KC>([&](auto f_c) {   // the loads land before the row loop (their wait must not sit between the stores)
        SRHIP_CAPTURE(wh);
        constexpr int f = decltype(f_c)::value;
        asm volatile("" : "+v"(wh[f / P::KC][f % P::KC]));
    });
    float breg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) breg[e] = (p.bias2 != nullptr && 4 * hl + e < p.cout2) ? p.bias2[4 * hl + e] : 0.0f;
    const int ne = p.cout2 < 4 ? p.cout2 : 4;   // accumulator registers per dy that hold a real channel in some lane
    // fragment kc of mid pixel pl + dx of group 0; a group adds 32 pixels = 4096 bytes (the swizzle repeats every 16 pixels)
    unsigned rd[3][P::KC];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int px = pl + dx;
#pragma unroll
        for (int kc = 0; kc < P::KC; ++kc)
            rd[dx][kc] = lds_base + (unsigned)P::YRING_OFF + (unsigned)(px * 128 + (((2 * kc + hl) ^ ((px >> 1) & 7)) << 4));
    }
    const int H = p.h, W = p.w;
    bf16x8 bq[2][P::KC];
    f32x16 acc;
    float s0[NG][4], s1[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) s0[g][e] = s1[g][e] = 0.0f;

    PairIter iter = pair_iter_begin(p);
    PairItem it;
    while (pair_iter_next(p, iter, it)) {
        const int x0 = it.x0, y0 = it.y0, nout = it.nout;
        const unsigned plane = (unsigned)(H * W) * 4u;
        const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.dst + (long long)it.n * p.cout2 * plane), 0, (int)((unsigned)p.cout2 * plane), 0x00020000);
        const unsigned mimg = (unsigned)(H * W) * (unsigned)(P::CH * 2);
        const i32x4 mid_srd = make_srd(STORE_MID ? p.mid + (long long)it.n * mimg : p.dst, STORE_MID ? mimg : 0u);
        unsigned vo[NG][4], mvo[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int jq = 32 * g + pl, xq = x0 + jq;
            const bool col_ok = jq < P::TWO && xq < W;
            mvo[g] = (STORE_MID && col_ok) ? (unsigned)((xq * P::CH + hl * 8) * 2) : kOobOffset;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                vo[g][e] = (col_ok && 4 * hl + e < p.cout2) ? (unsigned)(4 * hl + e) * plane + (unsigned)xq * 4u : kOobOffset;
        }

        auto row_body = [&](auto q_c, const int r) {
            constexpr int Q = decltype(q_c)::value;                     // s mod NY; mid row r = s - 2 sits in slot Q
            constexpr int SLOT = ((Q + 2 * P::NY - 2) % P::NY) * P::SLOT_BYTES;
            const int o = r - 2;                                        // the output row this mid row completes
            const unsigned soff = (o >= 0 && o < nout) ? (unsigned)((y0 + o) * W) * 4u : kOobOffset;
            // STORE_MID: the centre column's fragments of mid row r are image row y0 - 1 + r, this lane's output column
            const unsigned msoff = (r >= 1 && r <= nout) ? (unsigned)((y0 - 1 + r) * W) * (unsigned)(P::CH * 2) : kOobOffset;
            static_for<0, P::KC>([&](auto kc_c) {
                constexpr int kc = decltype(kc_c)::value;
                bq[0][kc] = lds_read_b128_off<SLOT>(rd[0][kc]);
            });
            static_for<0, 3 * NG>([&](auto st_c) {
                SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(wh); SRHIP_CAPTURE(s0); SRHIP_CAPTURE(s1); SRHIP_CAPTURE(vo); SRHIP_CAPTURE(mvo);
                constexpr int st = decltype(st_c)::value;
                constexpr int g = st / 3, dx = st % 3;
                constexpr int cur = st & 1, nxt = cur ^ 1;
                constexpr int gn = (st + 1) / 3, dxn = (st + 1) % 3;
                lds_wait<0>();
                static_for<0, P::KC>([&](auto kc_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[cur][decltype(kc_c)::value]); });
                static_for<0, P::KC>([&](auto kc_c) {
                    SRHIP_CAPTURE(bq); SRHIP_CAPTURE(acc); SRHIP_CAPTURE(wh);
                    constexpr int kc = decltype(kc_c)::value;
                    mfma_vv<(dx == 0 && kc == 0)>(acc, wh[dx][kc], bq[cur][kc]);
                    if constexpr (STORE_MID && dx == 1) store_fragment<kc * 32>(bq[cur][kc], mid_srd, mvo[g], msoff);
                    if constexpr (st + 1 < 3 * NG) bq[nxt][kc] = lds_read_b128_off<SLOT + gn * 4096>(rd[dxn][kc]);
                });
                if constexpr (dx == 2) {
                    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));   // MFMA result -> the vector ALU reads below
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = s1[g][e] + acc[8 + e] + breg[e];
                        float v = t;
                        if constexpr (ACT2 == 1) v = fmaxf(t, 0.0f);
                        else if constexpr (ACT2 == 2) v = fmaxf(t, t * p.slope2);
                        if (e < ne) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v * p.alpha), dst_rsrc, vo[g][e], soff, 0);
                        s1[g][e] = s0[g][e] + acc[4 + e];
                        s0[g][e] = acc[e];
                    }
                }
            });
        };

        __builtin_amdgcn_s_barrier();  // item start (matches the first layer's)
        const int nsteps = nout + P::LAG;
#pragma unroll 1
        for (int s = 0; s < nsteps; s += P::NY) {
            static_for<0, P::NY>([&](auto q_c) {
                constexpr int Q = decltype(q_c)::value;
                if (Q == 0 || s + Q < nsteps) {
                    __builtin_amdgcn_s_barrier();
                    if (s + Q >= 2) row_body(q_c, s + Q - 2);
                }
            });
        }
    }
}

// HEAD: 0 the residual pair, 1 the head form with two waves per layer (cout2 <= 32), 2 HEAD3 (cout2 <= 8); BITS: sign bits
// of `mid` written (1) or applied instead of the ReLU (2), see pair_first_layer
template <bool STORE_MID, int ACT2, int HEAD = 0, int BITS = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_pair_kernel(const PairParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // (`mid`, when asked for, is written by the waves of the second layer)
    if constexpr (HEAD == 2) {
        if (wave < HeadCfg::NW1) pair_first_layer<HeadCfg, BITS>(p, smem, lds_base, wave, lane);
        else pair_head3_layer<ACT2, STORE_MID, HeadCfg>(p, smem, lds_base, lane);
    } else {
        if (wave < 2) pair_first_layer<PairCfg, BITS>(p, smem, lds_base, wave, lane);
        else if constexpr (HEAD == 1) pair_head_layer<ACT2, STORE_MID>(p, smem, lds_base, wave - 2, lane);
        else pair_second_layer<ACT2, STORE_MID>(p, smem, lds_base, wave - 2, lane);
    }
}

template <bool STORE_MID, int ACT2, int HEAD = 0, int BITS = 0>
int launch_pair(PairParams p, hipStream_t stream)
{
    using P = std::conditional_t<HEAD == 2, HeadCfg, PairCfg>;
    p.two = P::TWO;
    static std::atomic<bool> configured_dev[kSrhipMaxDevices];  // per device
    const int dev = srhip_current_device();
    if (!configured_dev[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)conv3x3_pair_kernel<STORE_MID, ACT2, HEAD, BITS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                P::LDS_BYTES) != hipSuccess) {
            srhip_set_error("conv3x3_pair: cannot raise the dynamic LDS limit to %d", P::LDS_BYTES);
            return SRHIP_ERR_LAUNCH;
        }
        configured_dev[dev].store(true, std::memory_order_release);
    }
    // work items = images x vertical segments x TWO-column strips, one workgroup per CU walking them; a segment of R
    // rows costs R + 2 first-layer rows + LAG steps of pipeline fill + the look-ahead's fill latency
    const int slots = srhip_num_cus();
    const int nstrips = srhip_div_up(p.w, P::TWO);
    const double fill = P::LAG + 4.0;
    int best_segs = 1;
    double best_cost = 1e30;
    for (int rounds = 1; rounds <= 32; ++rounds) {
        int segs = (int)(((long long)rounds * slots) / ((long long)nstrips * p.n));
        if (segs < 1) segs = 1;
        if (segs > p.h) segs = p.h;
        const int seg_h = srhip_div_up(p.h, segs);
        segs = srhip_div_up(p.h, seg_h);
        const long long items = (long long)segs * nstrips * p.n;
        const int r = srhip_div_up(items, slots);
        const double cost = (double)r * (seg_h + fill);
        if (cost < best_cost) { best_cost = cost; best_segs = segs; }
    }
    p.seg_h = srhip_div_up(p.h, best_segs);
    p.nsegs = srhip_div_up(p.h, p.seg_h);
    p.nstrips = nstrips;
    long long total = (long long)p.nsegs * nstrips * p.n;
    p.main_items = (int)total; p.main_h = p.h; p.tail_stride = 0;
    // Two-class partition: when a whole number of segments per strip leaves workgroups idle (4K: 62 strips x 4 = 248 of
    // 256), the main segments shrink a little and the idle workgroups walk the rows that remain at the bottom of every
    // strip, so that all of them finish together.
    {
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
    SRHIP_REQUIRE(total < (1ll << 30), "conv3x3_pair: too many work items");
    p.total_items = (int)total;
    int grid = p.main_items != p.total_items ? slots : (int)(total < slots ? total : slots);
    // Range partition: every workgroup the same number of rows, cut wherever that falls (16 x 1024^2: 272 strips for 256
    // workgroups, 1088 rows each = a strip and a piece of the next, against 14 rounds of 79-row items with their fills).
    p.range_rows = 0;
    {
        const long long total_rows = (long long)nstrips * p.n * p.h;
        const double now = p.main_items != p.total_items ? 0.0 : best_cost;   // a two-class partition is already balanced
        if (now > 0.0 && total_rows >= slots && total_rows < (1ll << 30)) {
            const int rr = (int)srhip_div_up(total_rows, slots);
            const double cost = rr + fill * (srhip_div_up(rr, p.h) + 1);
            if (cost < 0.97 * now) {
                p.range_rows = rr;
                grid = (int)srhip_div_up(total_rows, rr);
            }
        }
    }
    hipLaunchKernelGGL((conv3x3_pair_kernel<STORE_MID, ACT2, HEAD, BITS>), dim3(grid), dim3(256), P::LDS_BYTES, stream, p);
    return srhip_check_launch("conv3x3_pair_kernel");
}

}  // namespace

extern "C" {

SRHIP_API size_t srhip_conv3x3_pair_bits_bytes(int n, int h, int w)
{
    if (n < 1 || h < 1 || w < 1) return 0;
    return (size_t)n * srhip_div_up(w, PairCfg::TWO) * h * (PairCfg::NW1 * 256);
}

SRHIP_API int srhip_conv3x3_pair_fwd(const srhip_conv3x3_pair_args* a, void* stream)
{
    SRHIP_REQUIRE(a != nullptr, "conv3x3_pair: null args");
    SRHIP_REQUIRE(a->src && a->wpack1 && a->wpack2 && a->dst, "conv3x3_pair: null src / wpack1 / wpack2 / dst");
    SRHIP_REQUIRE(a->dst != a->src && a->mid != a->src && (a->mid == nullptr || a->mid != a->dst),
                  "conv3x3_pair: src, mid and dst must be distinct tensors (neighbouring strips read src while others write dst)");
    SRHIP_REQUIRE(a->n >= 1 && a->h >= 1 && a->w >= 1, "conv3x3_pair: empty tensor");
    SRHIP_REQUIRE(a->act2 == SRHIP_ACT_NONE || a->act2 == SRHIP_ACT_RELU || a->act2 == SRHIP_ACT_LRELU, "conv3x3_pair: bad activation %d", a->act2);
    SRHIP_REQUIRE(a->alpha > 0.0f, "conv3x3_pair: alpha must be positive");
    SRHIP_REQUIRE((long long)a->h * a->w * 64 * 2 < (1ll << 31), "conv3x3_pair: image too large for 32-bit offsets");
    PairParams p;
    p.src = (const char*)a->src;
    p.wpack1 = (const bf16x8*)a->wpack1;
    p.bias1 = a->bias1;
    p.wpack2 = (const bf16x8*)a->wpack2;
    p.bias2 = a->bias2;
    p.mid = (char*)a->mid;
    p.dst = (char*)a->dst;
    p.n = a->n; p.h = a->h; p.w = a->w;
    p.slope2 = a->act2 == SRHIP_ACT_NONE ? 1.0f : (a->act2 == SRHIP_ACT_RELU ? 0.0f : a->slope2);
    p.alpha = a->alpha;
    p.cout2 = a->cout2;
    p.midbits = (char*)a->mid_bits;
    hipStream_t s = (hipStream_t)stream;
    if (a->mid_bits_mode != 0) {
        SRHIP_REQUIRE(a->mid_bits_mode == 1 || a->mid_bits_mode == 2, "conv3x3_pair: mid_bits_mode must be 0, 1 (write) or 2 (apply)");
        SRHIP_REQUIRE(a->mid_bits != nullptr, "conv3x3_pair: mid_bits_mode %d needs the sign-bit tensor `mid_bits`", a->mid_bits_mode);
        SRHIP_REQUIRE((a->cout2 == 0 || a->cout2 == 64) && a->act2 == SRHIP_ACT_NONE && a->mid != nullptr,
                      "conv3x3_pair: the sign-bit forms are built for the SRResidualBlock pair that also stores `mid` (act2 none)");
        SRHIP_REQUIRE(a->mid_bits_mode == 1 || a->bias1 == nullptr, "conv3x3_pair: the form that applies sign bits takes no first-layer bias");
        SRHIP_REQUIRE((long long)a->h * 512 < (1ll << 31), "conv3x3_pair: sign-bit rows too long for 32-bit offsets");
        return a->mid_bits_mode == 1 ? launch_pair<true, 0, 0, 1>(p, s) : launch_pair<true, 0, 0, 2>(p, s);
    }
    if (a->cout2 != 0 && a->cout2 != 64) {
        SRHIP_REQUIRE(a->cout2 >= 1 && a->cout2 <= 32, "conv3x3_pair: cout2 must be 64 (or 0) or, for the head form, 1..32");
        SRHIP_REQUIRE((long long)a->h * a->w * a->cout2 * 4 < (1ll << 31), "conv3x3_pair: head output too large for 32-bit offsets");
        // up to 8 output channels: three first-layer waves and one wave for the narrow layer (SRHIP_PAIR_HEAD=2 in the
        // environment keeps the two-plus-two form for A/B timing)
        static const bool head3 = []() { const char* e = getenv("SRHIP_PAIR_HEAD"); return !(e && e[0] == '2'); }();
        if (a->cout2 <= 8 && head3) {
            if (a->act2 == SRHIP_ACT_NONE) return a->mid != nullptr ? launch_pair<true, 0, 2>(p, s) : launch_pair<false, 0, 2>(p, s);
            if (a->act2 == SRHIP_ACT_RELU) return a->mid != nullptr ? launch_pair<true, 1, 2>(p, s) : launch_pair<false, 1, 2>(p, s);
            return a->mid != nullptr ? launch_pair<true, 2, 2>(p, s) : launch_pair<false, 2, 2>(p, s);
        }
        if (a->act2 == SRHIP_ACT_NONE) return a->mid != nullptr ? launch_pair<true, 0, 1>(p, s) : launch_pair<false, 0, 1>(p, s);
        if (a->act2 == SRHIP_ACT_RELU) return a->mid != nullptr ? launch_pair<true, 1, 1>(p, s) : launch_pair<false, 1, 1>(p, s);
        return a->mid != nullptr ? launch_pair<true, 2, 1>(p, s) : launch_pair<false, 2, 1>(p, s);
    }
    if (a->act2 == SRHIP_ACT_NONE) return a->mid != nullptr ? launch_pair<true, 0>(p, s) : launch_pair<false, 0>(p, s);
    if (a->act2 == SRHIP_ACT_RELU) return a->mid != nullptr ? launch_pair<true, 1>(p, s) : launch_pair<false, 1>(p, s);
    return a->mid != nullptr ? launch_pair<true, 2>(p, s) : launch_pair<false, 2>(p, s);
}

}  // extern "C"
