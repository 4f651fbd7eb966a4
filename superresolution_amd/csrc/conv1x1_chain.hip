// A chain of up to three 1x1 convolutions (each + bias + activation) as ONE pass over the pixels: the temporal network's
// G-buffer encoder ends in 160 -> 160 -> 160 -> 64 and 112 -> 64 -> 64 -> 64 (core/models/model_final.py:237-246, the
// second one scaled and added to the first, :256-258).  Layer by layer each of those maps is written to memory and read
// back (0.9 GB for the first chain at 544 x 960, against 0.23 GB for its input and output), and a 1x1 layer has too
// little arithmetic per byte to hide that.  Here a tile of 32 pixels travels through all layers inside LDS:
//
//   * every wave keeps the weight fragments of its JOBS in registers for the whole launch.  A job = (layer, range of
//     32-wide output-channel tiles); the jobs of a chain are dealt to the four waves so that each wave issues the same
//     number of matrix instructions per tile (160-160-160-64: 30 each).  No weight is read from LDS or memory again.
//   * the layers form a pipeline over the tiles: in step s layer l works on tile s - l, reading the LDS image that layer
//     l - 1 wrote in step s - 1 (two slots per intermediate map), one barrier per step.  Input tiles arrive by LDS-DMA two
//     steps ahead (a tile is one contiguous piece of memory), the residual of the last layer the same way.
//   * LDS image of a 32-pixel tile with C channels: [C / 8 planes][32 pixels][16 B].  A lane's MFMA pixel fragment
//     (pixel l & 31, 8 channels) is one 16-byte read, a half-wave reads 512 contiguous bytes (no bank conflicts); the
//     accumulators (pixel, 4 consecutive channels) go back as 8-byte writes into the same image.  The DMA builds that
//     image straight from the pixel-major tensor by giving each lane the address of its (plane, pixel) piece.
//   * the last layer's wave transposes its tile through a padded pixel-major LDS image and stores 16 bytes per lane,
//     consecutive lanes = consecutive channels of a pixel.
#include <atomic>

#include "conv_mfma_util.h"

namespace {

constexpr int CH_TPX = 32;       // pixels per tile
constexpr int CH_PF = 2;         // tiles fetched ahead of the one layer 0 works on
constexpr int CH_XR = CH_PF + 1; // input ring slots

struct ChainParams {
    const char* src;
    const char* res;
    char* dst;
    const char* wpack[3];
    const float* bias[3];
    float slope[3];   // act(t) = max(t, slope * t): 0 ReLU, 1 none
    float alpha;
    long long npix;
    int ntiles, per;  // tiles in all, tiles per work-group
};

struct ChainJob { int layer, ct0, nct; };

// 160 -> 160 -> 160 -> 64: 50 + 50 + 20 matrix instructions per tile, 30 per wave
struct ChainMain {
    static constexpr int NL = 3;
    static constexpr int C[4] = {160, 160, 160, 64};
    static constexpr ChainJob JOBS[4][2] = {{{0, 0, 3}, {-1, 0, 0}}, {{0, 3, 2}, {1, 0, 1}}, {{1, 1, 3}, {-1, 0, 0}}, {{1, 4, 1}, {2, 0, 2}}};
    static constexpr bool RES = false;
    static constexpr int W_X = 0, W_R = 1;   // waves that issue the input / residual DMA (never the storing wave)
};
// 112 -> 64 -> 64 -> 64, the last layer scaled and added to a residual: 14 + 8 + 8 matrix instructions per tile
struct ChainEnhance {
    static constexpr int NL = 3;
    static constexpr int C[4] = {112, 64, 64, 64};
    static constexpr ChainJob JOBS[4][2] = {{{0, 0, 1}, {-1, 0, 0}}, {{0, 1, 1}, {-1, 0, 0}}, {{1, 0, 2}, {-1, 0, 0}}, {{2, 0, 2}, {-1, 0, 0}}};
    static constexpr bool RES = true;
    static constexpr int W_X = 0, W_R = 1;
};

template <class CH>
struct ChainLds {
    static constexpr int NL = CH::NL;
    static constexpr int CO = CH::C[NL];
    static constexpr int tile_bytes(int c) { return CH_TPX * c * 2; }
    static constexpr int X_OFF = 0;
    static constexpr int M0_OFF = X_OFF + CH_XR * tile_bytes(CH::C[0]);
    static constexpr int M1_OFF = M0_OFF + (NL >= 2 ? 2 * tile_bytes(CH::C[1]) : 0);
    static constexpr int RES_OFF = M1_OFF + (NL >= 3 ? 2 * tile_bytes(CH::C[2]) : 0);
    static constexpr int RES_SLOTS = CH_PF + NL;   // fetched with the input tile, used NL - 1 steps after it
    static constexpr int OUT_OFF = RES_OFF + (CH::RES ? RES_SLOTS * tile_bytes(CO) : 0);
    static constexpr int OUT_ROWB = CO * 2 + 16;   // padded pixel-major rows of the store staging
    static constexpr int BIAS_OFF = OUT_OFF + CH_TPX * OUT_ROWB;
    static constexpr int BIAS_FLOATS = CH::C[1] + (NL >= 2 ? CH::C[2] : 0) + (NL >= 3 ? CH::C[3] : 0);
    static constexpr int BYTES = BIAS_OFF + BIAS_FLOATS * 4;
    static_assert(BYTES <= 80 * 1024, "two work-groups per CU");
};

template <int OFF>
__device__ __forceinline__ bf16x8 ch_lds_read(unsigned addr)
{
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ void ch_lds_write_b64(unsigned addr, unsigned lo, unsigned hi)
{
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(OFF) : "memory");
}
__device__ __forceinline__ u32x4 ch_lds_read_u128(unsigned addr)
{
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
template <int OFF>
__device__ __forceinline__ f32x4 ch_lds_read_f128(unsigned addr)
{
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ void ch_lds_read_b64(unsigned addr, unsigned& lo, unsigned& hi)
{
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    lo = v.x; hi = v.y;
}

template <class CH>
__global__ __launch_bounds__(256, 2) void conv1x1_chain_kernel(const ChainParams p)
{
    using L = ChainLds<CH>;
    constexpr int NL = CH::NL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, hl = lane >> 5;

    // biases of all layers -> LDS (read back 16 bytes at a time in accumulator order)
    {
        float* bl = (float*)(smem + L::BIAS_OFF);
        int off = 0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            for (int i = tid; i < CH::C[l + 1]; i += 256) bl[off + i] = p.bias[l] != nullptr ? p.bias[l][i] : 0.0f;
            off += CH::C[l + 1];
        }
    }
    // this wave's weight fragments (packed [16-channel chunk][32-channel tile][lane][8]), in registers for the whole launch
    constexpr int MAXF = 30;
    bf16x8 wreg[MAXF];
    static_for<0, 2>([&](auto j_c) {
        SRHIP_CAPTURE(wreg);
        constexpr int j = decltype(j_c)::value;
        static_for<0, 4>([&](auto w_c) {
            SRHIP_CAPTURE(wreg);
            constexpr int wv = decltype(w_c)::value;
            constexpr ChainJob job = CH::JOBS[wv][j];
            if constexpr (job.layer >= 0) {
                if (wave == wv) {
                    constexpr int KC = CH::C[job.layer] / 16, CT = (CH::C[job.layer + 1] + 31) / 32;
                    constexpr int F0 = j == 0 ? 0 : (CH::JOBS[wv][0].nct * (CH::C[CH::JOBS[wv][0].layer] / 16));
                    static_assert(F0 + KC * job.nct <= MAXF, "weight fragments of a wave's jobs");
                    const uint4* wp = (const uint4*)p.wpack[job.layer];
                    static_for<0, KC * job.nct>([&](auto f_c) {
                        SRHIP_CAPTURE(wreg);
                        constexpr int f = decltype(f_c)::value, kc = f / job.nct, ct = job.ct0 + f % job.nct;
                        const uint4 v = wp[(kc * CT + ct) * 64 + lane];
                        __builtin_memcpy(&wreg[F0 + f], &v, 16);
                    });
                }
            }
        });
    });

    const int t0 = blockIdx.x * p.per;
    const int nt = min(p.per, p.ntiles - t0);
    if (nt <= 0) return;
    // DMA: piece i of a tile with C channels holds planes 2 i, 2 i + 1; lane l fetches (plane 2 i + (l >> 5), pixel l & 31)
    constexpr int C0 = CH::C[0], CO = L::CO;
    constexpr int NPX = C0 / 16, NPR = CO / 16;
    const long long src_bytes = p.npix * (long long)(C0 * 2), out_bytes = p.npix * (long long)(CO * 2);
    // a tensor may exceed the 4 GiB a buffer descriptor covers: the descriptor starts at this work-group's first tile
    const long long x_base = (long long)t0 * CH_TPX * C0 * 2, r_base = (long long)t0 * CH_TPX * CO * 2;
    const long long x_left = src_bytes - x_base, r_left = out_bytes - r_base;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + x_base), 0, (int)(x_left < 0x7fffffffll ? x_left : 0x7fffffffll), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc((void*)((CH::RES ? p.res : p.src) + (CH::RES ? r_base : 0)), 0,
                                                                         CH::RES ? (int)(r_left < 0x7fffffffll ? r_left : 0x7fffffffll) : 0, 0x00020000);
    // tile t of this work-group; beyond the last one the same instructions run with an offset that fails the bounds check
    // (zeros land in a free slot), so every wave's count of operations in flight stays the same in every step
    unsigned vo_x = (unsigned)((pl * C0 + hl * 8) * 2), vo_r = (unsigned)((pl * CO + hl * 8) * 2);
    auto fetch = [&](const int t) {
        asm volatile("" : "+v"(vo_x), "+v"(vo_r));   // piece offsets are formed per use, not kept in ten registers
        if (wave == CH::W_X) {
            const unsigned slot = (unsigned)(L::X_OFF + (t % CH_XR) * L::tile_bytes(C0));
            const unsigned so = t < nt ? (unsigned)t * (unsigned)(CH_TPX * C0 * 2) : kOobOffset;
            static_for<0, NPX>([&](auto i_c) {
                constexpr int i = decltype(i_c)::value;
                lds_dma_b128(xsrd, (__attribute__((address_space(3))) void*)(smem + slot + i * 1024), (int)(vo_x + (unsigned)(i * 32)), (int)so);
            });
        }
        if constexpr (CH::RES) {
            if (wave == CH::W_R) {
                const unsigned slot = (unsigned)(L::RES_OFF + (t % L::RES_SLOTS) * L::tile_bytes(CO));
                const unsigned so = t < nt ? (unsigned)t * (unsigned)(CH_TPX * CO * 2) : kOobOffset;
                static_for<0, NPR>([&](auto i_c) {
                    constexpr int i = decltype(i_c)::value;
                    lds_dma_b128(rsrd, (__attribute__((address_space(3))) void*)(smem + slot + i * 1024), (int)(vo_r + (unsigned)(i * 32)), (int)so);
                });
            }
        }
    };
#pragma unroll 1
    for (int t = 0; t < CH_PF; ++t) fetch(t);

    const unsigned frag_rd = lds_base + (unsigned)(hl * 512 + pl * 16);          // + plane pair kc: 1 KiB each
    const unsigned quad_wr = lds_base + (unsigned)(pl * 16 + hl * 8);            // + plane * 512
    const unsigned bias_rd = lds_base + (unsigned)(L::BIAS_OFF + hl * 16);       // + channel * 4
    const unsigned out_wr = lds_base + (unsigned)(L::OUT_OFF + pl * L::OUT_ROWB + hl * 8);

    const int nsteps = nt + NL - 1;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        // tile s has landed: the wave that fetched it waits for its own pieces, the barrier tells the others
        if (wave == CH::W_X) vm_wait<(CH_PF - 1) * NPX>();
        if (CH::RES && wave == CH::W_R) vm_wait<(CH_PF - 1) * NPR>();
        lds_wait<0>();
        __builtin_amdgcn_s_barrier();
        fetch(s + CH_PF);

        static_for<0, 2>([&](auto j_c) {
            SRHIP_CAPTURE(wreg);
            constexpr int j = decltype(j_c)::value;
            static_for<0, 4>([&](auto w_c) {
                SRHIP_CAPTURE(wreg);
                constexpr int wv = decltype(w_c)::value;
                constexpr ChainJob job = CH::JOBS[wv][j];
                if constexpr (job.layer >= 0) {
                    constexpr int l = job.layer;
                    const int tl = s - l;   // this layer's tile in this step
                    if (wave == wv && tl >= 0 && tl < nt) {
                        constexpr int CI = CH::C[l], KC = CI / 16, CN = CH::C[l + 1];
                        constexpr int F0 = j == 0 ? 0 : (CH::JOBS[wv][0].nct * (CH::C[CH::JOBS[wv][0].layer] / 16));
                        constexpr bool LAST = l == NL - 1;
                        constexpr int SRC_OFF = l == 0 ? L::X_OFF : (l == 1 ? L::M0_OFF : L::M1_OFF);
                        constexpr int DST_OFF = l == 0 ? L::M0_OFF : L::M1_OFF;
                        constexpr int BIAS0 = (l >= 1 ? CH::C[1] : 0) + (l >= 2 ? CH::C[2] : 0);
                        const unsigned src = frag_rd + (unsigned)(SRC_OFF + (l == 0 ? tl % CH_XR : (tl & 1)) * L::tile_bytes(CI));
                        bf16x8 bq[KC];
                        static_for<0, KC>([&](auto k_c) { SRHIP_CAPTURE(bq); constexpr int kc = decltype(k_c)::value; bq[kc] = ch_lds_read<kc * 1024>(src); });
                        lds_wait<0>();
                        static_for<0, KC>([&](auto k_c) { SRHIP_CAPTURE(bq); SRHIP_LOAD_FENCE(bq[decltype(k_c)::value]); });
                        const float slope = p.slope[l];
                        // this tile's slot of the map the layer writes / of the residual ring (+ plane * 512 in the instruction)
                        const unsigned dst_wr = quad_wr + (unsigned)(DST_OFF + (tl & 1) * L::tile_bytes(CN));
                        const unsigned res_rd = quad_wr + (unsigned)(L::RES_OFF + (tl % L::RES_SLOTS) * L::tile_bytes(CN));
                        (void)dst_wr; (void)res_rd;
                        static_for<0, job.nct>([&](auto c_c) {
                            SRHIP_CAPTURE(bq); SRHIP_CAPTURE(wreg);
                            constexpr int cl = decltype(c_c)::value, ct = job.ct0 + cl;
                            // bias (and residual) of the tile's four register quads: requested now, needed behind the MFMAs
                            f32x4 bv[4];
                            unsigned r0[4] = {0u, 0u, 0u, 0u}, r1[4] = {0u, 0u, 0u, 0u};
                            static_for<0, 4>([&](auto g_c) {
                                SRHIP_CAPTURE(bv); SRHIP_CAPTURE(r0); SRHIP_CAPTURE(r1);
                                constexpr int g = decltype(g_c)::value;
                                constexpr int c0 = 32 * ct + 8 * g;
                                if constexpr (c0 < CN) {
                                    bv[g] = ch_lds_read_f128<(BIAS0 + c0) * 4>(bias_rd);
                                    if constexpr (LAST && CH::RES) ch_lds_read_b64<(4 * ct + g) * 512>(res_rd, r0[g], r1[g]);
                                }
                            });
                            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            static_for<0, KC>([&](auto k_c) {
                                SRHIP_CAPTURE(bq); SRHIP_CAPTURE(wreg); SRHIP_CAPTURE(acc);
                                constexpr int kc = decltype(k_c)::value;
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[F0 + kc * job.nct + cl], bq[kc], acc, 0, 0, 0);
                            });
                            lds_wait<0>();
                            // register quad g: channels 32 ct + 8 g + 4 hl .. + 3 of pixel pl = plane 4 ct + g, bytes 8 hl .. 8 hl + 7
                            static_for<0, 4>([&](auto g_c) {
                                SRHIP_CAPTURE(bv); SRHIP_CAPTURE(r0); SRHIP_CAPTURE(r1); SRHIP_CAPTURE(acc);
                                constexpr int g = decltype(g_c)::value;
                                constexpr int c0 = 32 * ct + 8 * g;
                                if constexpr (c0 < CN) {
                                    SRHIP_LOAD_FENCE(bv[g]);
                                    float v[4];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float t = acc[4 * g + e] + bv[g][e];
                                        v[e] = fmaxf(t, t * slope);
                                    }
                                    if constexpr (LAST) {
                                        if constexpr (CH::RES) {
                                            SRHIP_LOAD_FENCE(r0[g]);
                                            SRHIP_LOAD_FENCE(r1[g]);
                                            const float rf[4] = {bf16_lo(r0[g]), bf16_hi(r0[g]), bf16_lo(r1[g]), bf16_hi(r1[g])};
#pragma unroll
                                            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], p.alpha, rf[e]);   // as the implicit-GEMM epilogue
                                        } else {
#pragma unroll
                                            for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
                                        }
                                        ch_lds_write_b64<c0 * 2>(out_wr, pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                                    } else {
                                        ch_lds_write_b64<(4 * ct + g) * 512>(dst_wr, pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                                    }
                                }
                            });
                            __builtin_amdgcn_sched_barrier(0);   // one tile's accumulators and biases live at a time
                        });
                        if constexpr (LAST) {
                            // the tile, pixel-major in LDS -> memory, 16 bytes per lane (written by this wave alone: no barrier)
                            lds_wait<0>();
                            constexpr int PPP = CN / 8;                       // 16-byte pieces per pixel
                            constexpr int NST = CH_TPX * PPP / 64;            // store instructions per tile
                            const long long pix0 = ((long long)t0 + tl) * CH_TPX;
                            static_for<0, NST>([&](auto i_c) {
                                constexpr int i = decltype(i_c)::value;
                                const int q = i * 64 + lane;
                                const int px = q / PPP, cg = q - px * PPP;
                                u32x4 v = ch_lds_read_u128(lds_base + (unsigned)(L::OUT_OFF + px * L::OUT_ROWB + cg * 16));
                                lds_wait<0>();
                                SRHIP_LOAD_FENCE(v);
                                if (pix0 + px < p.npix) *(u32x4*)(p.dst + ((pix0 + px) * CN + cg * 8) * 2) = v;
                            });
                        }
                    }
                }
            });
        });
    }
    // nothing of this work-group is in flight when its LDS is handed on
    vm_wait<0>();
}

template <class CH>
int launch_chain(ChainParams p, hipStream_t stream)
{
    using L = ChainLds<CH>;
    static std::atomic<bool> configured_dev[kSrhipMaxDevices];
    const int dev = srhip_current_device();
    if (!configured_dev[dev].load(std::memory_order_acquire)) {
        if (L::BYTES > 48 * 1024 &&
            hipFuncSetAttribute((const void*)conv1x1_chain_kernel<CH>, hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES) != hipSuccess) {
            srhip_set_error("conv1x1_chain: cannot raise the dynamic LDS limit to %d", L::BYTES);
            return SRHIP_ERR_LAUNCH;
        }
        configured_dev[dev].store(true, std::memory_order_release);
    }
    const long long ntiles = (p.npix + CH_TPX - 1) / CH_TPX;
    SRHIP_REQUIRE(ntiles < (1ll << 31), "conv1x1_chain: too many pixels");
    long long grid = 2ll * srhip_num_cus();
    if (grid > ntiles) grid = ntiles;
    const long long per = (ntiles + grid - 1) / grid;
    grid = (ntiles + per - 1) / per;
    SRHIP_REQUIRE(per * CH_TPX * 160 * 2 < 0x7fffffffll, "conv1x1_chain: a work-group's share exceeds a buffer descriptor");
    p.ntiles = (int)ntiles;
    p.per = (int)per;
    hipLaunchKernelGGL(conv1x1_chain_kernel<CH>, dim3((unsigned)grid), dim3(256), L::BYTES, stream, p);
    return srhip_check_launch("conv1x1_chain_kernel");
}

}  // namespace

extern "C" {

SRHIP_API int srhip_conv1x1_chain_fwd(const srhip_conv1x1_chain_args* a, void* stream)
{
    SRHIP_REQUIRE(a != nullptr, "conv1x1_chain: null argument block");
    SRHIP_REQUIRE(a->src && a->dst && a->nlayers == 3, "conv1x1_chain: null pointer, or a chain of %d layers (3)", a->nlayers);
    SRHIP_REQUIRE(a->npix >= 1, "conv1x1_chain: empty tensor");
    SRHIP_REQUIRE((size_t)a->src % 16 == 0 && (size_t)a->dst % 16 == 0 && (a->res == nullptr || (size_t)a->res % 16 == 0),
                  "conv1x1_chain: maps must be 16-byte aligned");
    SRHIP_REQUIRE(a->alpha > 0.0f, "conv1x1_chain: alpha must be positive");
    ChainParams p;
    p.src = (const char*)a->src; p.res = (const char*)a->res; p.dst = (char*)a->dst;
    for (int l = 0; l < 3; ++l) {
        SRHIP_REQUIRE(a->wpack[l] != nullptr, "conv1x1_chain: layer %d has no weights", l);
        SRHIP_REQUIRE(a->act[l] == SRHIP_ACT_NONE || a->act[l] == SRHIP_ACT_RELU || a->act[l] == SRHIP_ACT_LRELU, "conv1x1_chain: bad activation %d", a->act[l]);
        p.wpack[l] = (const char*)a->wpack[l];
        p.bias[l] = a->bias[l];
        p.slope[l] = a->act[l] == SRHIP_ACT_NONE ? 1.0f : (a->act[l] == SRHIP_ACT_RELU ? 0.0f : a->slope);
    }
    p.alpha = a->alpha;
    p.npix = a->npix;
    const int* c = a->c;
    if (c[0] == 160 && c[1] == 160 && c[2] == 160 && c[3] == 64 && a->res == nullptr) return launch_chain<ChainMain>(p, (hipStream_t)stream);
    if (c[0] == 112 && c[1] == 64 && c[2] == 64 && c[3] == 64 && a->res != nullptr) return launch_chain<ChainEnhance>(p, (hipStream_t)stream);
    srhip_set_error("conv1x1_chain: no kernel for %d -> %d -> %d -> %d %s residual (160-160-160-64 without, 112-64-64-64 with)", c[0], c[1], c[2], c[3],
                    a->res ? "with" : "without");
    return SRHIP_ERR_INVALID;
}

SRHIP_API int srhip_conv1x1_chain_supported(const int* c, int nlayers, int has_res)
{
    if (c == nullptr || nlayers != 3) return 0;
    if (c[0] == 160 && c[1] == 160 && c[2] == 160 && c[3] == 64 && !has_res) return 1;
    if (c[0] == 112 && c[1] == 64 && c[2] == 64 && c[3] == 64 && has_res) return 1;
    return 0;
}

}  // extern "C"
