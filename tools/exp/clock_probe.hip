// What do s_memtime / s_memrealtime tick at, and what clock does an MFMA loop hold?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int LDSREAD>
__global__ __launch_bounds__(256, 2) void probe(float* out, unsigned long long* t, int iters)
{
    __shared__ __attribute__((aligned(16))) float tile[18 * 36 * 16];
    for (int i = threadIdx.x; i < 18 * 36 * 16; i += 256) tile[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = lane * 0.001f;
    unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long c0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            f32x4 b[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                b[j] = LDSREAD ? *reinterpret_cast<const f32x4*>(tile + ((lane & 15) + j * 40 + tap * 36 + (it & 7)) * 16 + (lane >> 4) * 4)
                               : (f32x4){a, a + 1, a + 2, a + 3};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[j] = MFMA(a, b[j][kk], acc[j]);
        }
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long c1 = wall_clock64();
    if (threadIdx.x == 0) {
        t[blockIdx.x * 3 + 0] = m1 - m0;
        t[blockIdx.x * 3 + 1] = r1 - r0;
        t[blockIdx.x * 3 + 2] = c1 - c0;
    }
    if (acc[0].x + acc[1].x + acc[2].x == 12345.f) out[0] = 1.f;
}

template <int LDSREAD>
void run(const char* name)
{
    float* out; unsigned long long* t;
    hipMalloc(&out, 4096); hipMalloc(&t, 512 * 3 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<LDSREAD>, dim3(512), dim3(256), 0, 0, out, t, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<LDSREAD>, dim3(512), dim3(256), 0, 0, out, t, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512 * 3];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double mfma = (double)iters * 108;     // per wave
    printf("%-10s %8.3f ms  memtime %llu  memrealtime %llu  wall_clock64 %llu | memtime/ms %.0f  realtime/ms %.0f | cycles/MFMA(2 waves/SIMD) %.1f if memtime=cycles\n",
           name, ms, h[0], h[1], h[2], h[0] / ms, h[1] / ms, h[0] / mfma / 2.0);
    const double flop = 512.0 * 4 * mfma * 2048;
    printf("           %.1f TF\n", flop / ms / 1e9);
}

int main() { run<0>("regs"); run<1>("lds"); return 0; }
