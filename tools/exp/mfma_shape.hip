// Micro-benchmark: sustained rate of the two dense f16 MFMA shapes under the package power cap, with random (non-zero) operands:
//   v_mfma_f32_16x16x32_f16 (8 192 MAC, 16 cycles)   against   v_mfma_f32_32x32x16_f16 (16 384 MAC, 32 cycles: half the operand
//   register reads per MAC).  256 workgroups x 12 waves (3 per SIMD), ~50 ms per launch, wall-clock TFLOP/s from HIP events.
//   hipcc -O3 --offload-arch=gfx950 mfma_shape.hip -o mfma_shape && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(768, 3) void shape_kernel(const h8* __restrict__ src, float* __restrict__ out, int iters)
{
    const int tid = threadIdx.x, lane = tid & 63;
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(i * 64 + lane)]; b[i] = src[((4 + i) * 64 + lane)]; }
    float r = 0.f;
    if (SHAPE == 0) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k], b[0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k], b[1], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k], b[2], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k], b[3], c3, 0, 0, 0);
            }
        }
        const f32x4 t = c0 + c1 + c2 + c3;
        r = t[0] + t[1] + t[2] + t[3];
    } else {
        f32x16 c0, c1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[(k + 1) & 3], c1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) r += c0[i] + c1[i];
    }
    out[blockIdx.x * blockDim.x + tid] = r;
}

int main()
{
    h8* src; float* out;
    std::vector<_Float16> h(8 * 64 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
    hipMalloc(&src, h.size() * 2); hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 768 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 2; ++shape) {
            // same MACs per launch: 16 x 8192 per iteration of shape 0, 8 x 16384 of shape 1
            const int iters = 150000;
            auto launch = [&]() {
                if (shape == 0) hipLaunchKernelGGL(shape_kernel<0>, dim3(256), dim3(768), 0, 0, src, out, iters);
                else hipLaunchKernelGGL(shape_kernel<1>, dim3(256), dim3(768), 0, 0, src, out, iters);
            };
            launch();
            hipEventRecord(e0);
            for (int i = 0; i < 4; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 4.0 * 256 * 12 * (double)iters * 16 * 8192 * 2;
            printf("%s: %.1f ms per launch, %.0f TFLOP/s\n", shape == 0 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_32x32x16_f16", ms / 4, flop / (ms * 1e-3) * 1e-12);
        }
    return 0;
}
