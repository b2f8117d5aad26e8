// what does v_permlane16_swap_b32 do?  prints, per 16-lane row, which (operand, row) each result row came from
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    const int l = threadIdx.x;
    const unsigned a = 100 + l, b = 200 + l;
    u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    (void)hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int r = 0; r < 2; ++r) { printf("result %d rows:", r); for (int row = 0; row < 4; ++row) printf("  [%u..]", h[r * 64 + row * 16]); printf("\n"); }
    return 0;
}
