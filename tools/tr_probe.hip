// probe: semantics of ds_read_b64_tr_b16 and permlane32_swap on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, unsigned* out2) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    int l = threadIdx.x;
    for (int i = l; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    // every lane supplies address of 4 contiguous shorts: element index 4*l  (lane l -> shorts 4l..4l+3)
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + 4 * l));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
    // second experiment: lanes supply row-major [key t>>2][channel 4*(t&3)] with row stride 100 shorts, group offset 1000
    int t = l & 15, g = l >> 4;
    s4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + 1000 * g + 100 * (t >> 2) + 4 * (t & 3)));
    for (int e = 0; e < 4; ++e) out[256 + l * 4 + e] = w[e];
    unsigned a = 100 + l, b = 200 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out2[l * 2] = r[0];
    out2[l * 2 + 1] = r[1];
}
int main() {
    short* d; unsigned* d2;
    hipMalloc(&d, 512 * 2); hipMalloc(&d2, 128 * 4);
    probe<<<1, 64>>>(d, d2);
    short h[512]; unsigned h2[128];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(h2, d2, sizeof(h2), hipMemcpyDeviceToHost);
    printf("tr16 (lane l supplies shorts 4l..4l+3):\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    printf("tr16 (group g: base 1000g, key=t>>2 stride 100, ch=4*(t&3)):\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[256+l*4], h[256+l*4+1], h[256+l*4+2], h[256+l*4+3]);
    printf("permlane32_swap(a=100+l, b=200+l):\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: r0=%u r1=%u\n", l, h2[l*2], h2[l*2+1]);
    return 0;
}
