// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS image [64 rows][64 cols] of u16 with
// value = row*64+col.  Each 16-lane group g reads the block rows 4g..4g+3, cols 0..15; lane 4q+p
// supplies the address of (row 4g+q, col 4p).  Prints what every lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64*64];
    for (int i = threadIdx.x; i < 64*64; i += 64) lds[i] = in[i];
    __syncthreads();
    int l = threadIdx.x;
    int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    auto* a = (__attribute__((address_space(3))) s16x4*)(lds + (4*g+q)*64 + 4*p);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a);
    for (int e = 0; e < 4; ++e) out[l*4+e] = v[e];
}
int main() {
    unsigned short h[64*64], o[256];
    for (int i = 0; i < 64*64; ++i) h[i] = i;
    unsigned short *d, *r;
    hipMalloc(&d, sizeof h); hipMalloc(&r, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d, r);
    hipMemcpy(o, r, sizeof o, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", o[l*4+e] / 64, o[l*4+e] % 64);
        printf("\n");
    }
    return 0;
}
