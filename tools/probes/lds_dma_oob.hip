// Probe: does an LDS-DMA buffer load write ZEROS to LDS for lanes that fail the buffer bounds check?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void probe(const char* src, unsigned nbytes, unsigned* out, unsigned soff)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    // poison LDS
    ((u32x4*)smem)[lane] = u32x4{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)nbytes, 0x00020000);
    // even lanes in range, odd lanes out of range
    const unsigned voff = (lane & 1) ? 0x80000000u : (unsigned)lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u32x4 v = ((u32x4*)smem)[lane];
    out[lane * 4 + 0] = v[0]; out[lane * 4 + 1] = v[1]; out[lane * 4 + 2] = v[2]; out[lane * 4 + 3] = v[3];
}
int main()
{
    const int n = 64 * 16;
    std::vector<unsigned> h(n / 4);
    for (int i = 0; i < n / 4; ++i) h[i] = 0x1000 + i;
    char* d; unsigned* o;
    hipMalloc(&d, n); hipMalloc(&o, 64 * 16);
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    for (unsigned soff : {0u, 0x80000000u}) {
        hipMemset(o, 0xff, 64 * 16);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 1024, 0, d, (unsigned)n, o, soff);
        hipDeviceSynchronize();
        std::vector<unsigned> r(256);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        printf("soffset=%08x\n", soff);
        for (int l = 0; l < 6; ++l) printf("  lane %d: %08x %08x %08x %08x\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
        int zeros = 0, poison = 0, data = 0;
        for (int i = 0; i < 256; ++i) { if (r[i] == 0) zeros++; else if (r[i] == 0xdeadbeefu) poison++; else data++; }
        printf("  words: zeros=%d poison=%d data=%d\n", zeros, poison, data);
    }
    return 0;
}
