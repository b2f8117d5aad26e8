// hipcc waitcnt experiment (compile with -S --cuda-device-only -DMODE=0|1|2 and grep vmcnt): a plain global load that is
// pending together with LDS-DMA (global_load_lds) is waited for with vmcnt(0) -- the compiler treats the two as
// unordered -- even when 6 younger DMA instructions would allow vmcnt(6).  fused_h3.hip therefore keeps ordinary
// global loads out of its tile loop.
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma(const char* src, char* __restrict__ dst, int wave, int i) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
        (__attribute__((address_space(3))) void*)(dst + wave * 1024 + i * 8192), 16, 0, 0);
}
#if MODE == 2
__shared__ __attribute__((aligned(16))) char la[32768];
__shared__ __attribute__((aligned(16))) char lb[65536];
#endif
__global__ void k(const char* g, const char* r, char* out, int n) {
#if MODE == 2
    char* a = la; char* b = lb;
#else
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* a = lds; char* b = lds + 65536;
#endif
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = 0; i < n; ++i) {
        h4 res = *reinterpret_cast<const h4*>(r + i * 4096 + threadIdx.x * 8);
        __builtin_amdgcn_sched_barrier(0);
        for (int j = 0; j < 6; ++j) dma(g + (i * 6 + j) * 4096 + threadIdx.x * 16, b, wave, j);
        __builtin_amdgcn_sched_barrier(0);
#if MODE == 0
        h4 o = res;
#else
        h8 v = *reinterpret_cast<const h8*>(a + threadIdx.x * 16);
        h4 o = res + (h4){v[0], v[1], v[2], v[3]};
#endif
        *reinterpret_cast<h4*>(out + i * 4096 + threadIdx.x * 8) = o * (_Float16)2;
        __builtin_amdgcn_s_waitcnt((7 & 15) | 0x0F70);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        asm volatile("s_barrier" ::: "memory");
    }
}
