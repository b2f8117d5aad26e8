// ds_read_b64_tr_b16 semantics probe: LDS image [pixel][16 channels] of f16 with value = pixel*16 + channel;
// lane l = 16g + 4q + p supplies the address of (pixel 4g + q, channels 4p..4p+3).  Expect lane 16g + i to receive
// channel i of pixels 4g .. 4g+3 in elements 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) __fp16 lds[64 * 16];
    for (int i = threadIdx.x; i < 64 * 16; i += 64) lds[i] = (__fp16)(float)i;
    __syncthreads();
    const int l = threadIdx.x;
    const int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    const int pixel = 4 * g + q;
    auto ptr = (__attribute__((address_space(3))) fp16x4*)(lds + pixel * 16 + p * 4);
    fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(ptr);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)v[e];
}
int main() {
    float* d; float h[256];
    (void)hipMalloc(&d, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const int g = l >> 4, i = l & 15;
            const float expect = (float)((4 * g + e) * 16 + i);
            if (h[l * 4 + e] != expect) ++bad;
        }
    printf("lane 0: %g %g %g %g | lane 17: %g %g %g %g | mismatches vs expectation: %d\n", h[0], h[1], h[2], h[3], h[68], h[69], h[70], h[71], bad);
    return 0;
}
