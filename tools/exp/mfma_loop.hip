// Micro-benchmark: how fast can the conv inner loop (ds_read_b128 -> 4x v_mfma_f32_16x16x4_f32 per tap)
// run in steady state?  Build: hipcc -O3 --offload-arch=gfx950 mfma_loop.hip -o mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int IW = 36, IH = 18;

// MODE 0: registers only. MODE 1: compiler-scheduled LDS reads (as the product kernel).
// MODE 2: explicit one-tap-ahead LDS prefetch.  NG groups per pass.
template <int MODE, int NG>
__global__ __launch_bounds__(256, 2) void loop_kernel(const float* __restrict__ wpack, float* __restrict__ out, int passes)
{
    extern __shared__ __attribute__((aligned(16))) float tin[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < IH * IW * 16; i += blockDim.x) tin[i] = (float)((i * 7 + 3) % 13) * 0.01f;
    float w[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) w[i] = wpack[i * 64 + lane];
    __syncthreads();
    const int p = lane & 15, q = lane >> 4;
    f32x4 total = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < passes; ++it) {
        int base[NG];
        f32x4 acc[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int f = ((wave + 4 * j + it) % 30) * 16 + p;
            const int my = f / 34, mx = f - my * 34;
            base[j] = (my * IW + mx) * 16 + q * 4;
            acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (MODE == 0) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < NG; ++j) acc[j] = MFMA(w[tap * 4 + kk], w[(tap * 4 + kk + j + 1) % 36], acc[j]);
        } else if (MODE == 1) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int off = ((tap / 3) * IW + (tap % 3)) * 16;
                f32x4 bv[NG];
#pragma unroll
                for (int j = 0; j < NG; ++j) bv[j] = *reinterpret_cast<const f32x4*>(tin + base[j] + off);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < NG; ++j) acc[j] = MFMA(w[tap * 4 + kk], bv[j][kk], acc[j]);
            }
        } else {
            f32x4 cur[NG], nxt[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) cur[j] = *reinterpret_cast<const f32x4*>(tin + base[j]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap < 8) {
                    const int off = (((tap + 1) / 3) * IW + ((tap + 1) % 3)) * 16;
#pragma unroll
                    for (int j = 0; j < NG; ++j) nxt[j] = *reinterpret_cast<const f32x4*>(tin + base[j] + off);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < NG; ++j) acc[j] = MFMA(w[tap * 4 + kk], cur[j][kk], acc[j]);
#pragma unroll
                for (int j = 0; j < NG; ++j) cur[j] = nxt[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) total += acc[j];
    }
    if (total.x == 12345.f) out[tid] = total.y + total.z + total.w;
}

// MODE 3: the product kernel's conv1 pass structure (address prologue, NG groups, relu + zero-check
// epilogue, ds_write of the intermediate), no barriers, no global traffic.
template <int NG, int EPI>
__global__ __launch_bounds__(256, 2) void pass_kernel(const float* __restrict__ wpack, float* __restrict__ out, int passes, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tin = lds;
    float* tmid = lds + IH * IW * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < IH * IW * 16; i += blockDim.x) tin[i] = (float)((i * 7 + 3) % 13) * 0.01f;
    float w[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) w[i] = wpack[i * 64 + lane];
    __syncthreads();
    const int p = lane & 15, q = lane >> 4;
    for (int it = 0; it < passes; ++it) {
        int base[NG], f[NG];
        f32x4 acc[NG];
        const int y0 = (it % 18) * 14, x0 = ((it >> 2) % 8) * 32;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int g = (wave + 4 * j + it * 4 * NG) % 34;
            f[j] = g * 16 + p;
            const int my = f[j] / 34, mx = f[j] - my * 34;
            base[j] = (my * IW + mx) * 16 + q * 4;
            acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int off = ((tap / 3) * IW + (tap % 3)) * 16;
            f32x4 bv[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) bv[j] = *reinterpret_cast<const f32x4*>(tin + base[j] + off);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < NG; ++j) acc[j] = MFMA(w[tap * 4 + kk], bv[j][kk], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            f32x4 v = acc[j];
            if (EPI) {
                const int my = f[j] / 34, mx = f[j] - my * 34;
                const int gy = y0 - 1 + my, gx = x0 - 1 + mx;
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                if (gy < 0 || gy >= H || gx < 0 || gx >= W) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            *reinterpret_cast<f32x4*>(tmid + f[j] * 16 + q * 4) = v;
        }
    }
    __syncthreads();
    if (tmid[tid] == 12345.f) out[tid] = tmid[tid + 1];
}

template <int NG, int EPI>
void run_pass(const char* name, const float* w, float* out, int wgs_per_cu)
{
    const int passes = 2000;
    const int grid = 256 * wgs_per_cu;
    const size_t lds = (IH * IW + 16 * 34) * 16 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(pass_kernel<NG, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((pass_kernel<NG, EPI>), dim3(grid), dim3(256), lds, 0, w, out, 10, 256, 256);
    hipEventRecord(e0);
    hipLaunchKernelGGL((pass_kernel<NG, EPI>), dim3(grid), dim3(256), lds, 0, w, out, passes, 256, 256);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * passes * NG * 36 * 2048.0;
    printf("%-34s wg/cu %d  %8.3f ms  %7.1f TF  (%.1f%% of 157.3)  %s\n", name, wgs_per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100,
           hipGetErrorString(hipGetLastError()));
}

template <int MODE, int NG>
void run(const char* name, const float* w, float* out, int wgs_per_cu)
{
    const int passes = 2000;
    const int grid = 256 * wgs_per_cu;
    const size_t lds = IH * IW * 16 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((loop_kernel<MODE, NG>), dim3(grid), dim3(256), lds, 0, w, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((loop_kernel<MODE, NG>), dim3(grid), dim3(256), lds, 0, w, out, passes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * passes * NG * 36 * 2048.0;
    printf("%-34s wg/cu %d  %8.3f ms  %7.1f TF  (%.1f%% of 157.3)\n", name, wgs_per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

int main()
{
    float *w, *out;
    hipMalloc(&w, 36 * 64 * 4); hipMalloc(&out, 4096 * 4);
    hipMemset(w, 0, 36 * 64 * 4);
    run<0, 3>("regs only NG=3", w, out, 1);
    run<0, 3>("regs only NG=3", w, out, 2);
    run<1, 3>("lds compiler-sched NG=3", w, out, 1);
    run<1, 3>("lds compiler-sched NG=3", w, out, 2);
    run<1, 4>("lds compiler-sched NG=4", w, out, 2);
    run<2, 3>("lds 1-tap-ahead NG=3", w, out, 1);
    run<2, 3>("lds 1-tap-ahead NG=3", w, out, 2);
    run<2, 4>("lds 1-tap-ahead NG=4", w, out, 2);
    run<2, 2>("lds 1-tap-ahead NG=2", w, out, 2);
    run<1, 2>("lds compiler-sched NG=2", w, out, 2);
    run<1, 1>("lds compiler-sched NG=1", w, out, 2);
    run_pass<3, 1>("conv1 passes NG=3 epi", w, out, 2);
    run_pass<3, 0>("conv1 passes NG=3 plain store", w, out, 2);
    run_pass<4, 1>("conv1 passes NG=4 epi", w, out, 2);
    run_pass<2, 1>("conv1 passes NG=2 epi", w, out, 2);
    run_pass<6, 1>("conv1 passes NG=6 epi", w, out, 2);
    run_pass<3, 1>("conv1 passes NG=3 epi", w, out, 1);
    return 0;
}
