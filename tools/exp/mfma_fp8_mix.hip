// Micro-benchmark for VERDICT r03 item 4 (cheaper arithmetic for the two 2^-11 products of the split-f16 convolutions): cycles per
// 16-pixel group of ONE 3x3 16 -> 16 convolution row step on a SIMD with three waves, for
//   A  the shipped mix: 15 x v_mfma_f32_16x16x32_f16 (3 split products), 3 x ds_read_b128 operand fetches, 14 VALU micro-ops (the
//      hi / lo split epilogue), 2 x ds_write_b64
//   B  the fp8 scheme sized in tools/exp/emulate_f16x3.py ("f16+fp8": 9.6e-6 normalised MAE through 1x18, bar 1e-4): hi . hi stays
//      f16 -- 6 MFMAs per group (taps (dy,0)|(dy,1) paired, tap (dy,2) alone in a half-empty K) -- and BOTH lo products of a (row, dy)
//      pair ride in ONE v_mfma_scale_f32_16x16x128_f8f6f4 (K = 128 = [x_lo . w_hi : 3 dx x 16 ci | x_hi . w_lo : 48 | 32 unused]):
//      3 fp8 MFMAs per group, 4 x ds_read_b128 (the fp8 B operand is 32 bytes per lane), 20 VALU (the split epilogue + four
//      conversions to fp8), 3 x ds_write_b64 (hi, lo, fp8 copies)
// Prints cycles per group and SIMD.   hipcc -O3 --offload-arch=gfx950 mfma_fp8_mix.hip -o mfma_fp8_mix && ./mfma_fp8_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define MFMA_8(a, b, c) __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4((a), (b), (c), 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f)

template <int SCHEME>            // 0 = A (f16 x 3), 1 = B (f16 + fp8), 2 = B without its extra VALU / LDS (matrix work alone), 3 = A's MFMAs alone
__global__ __launch_bounds__(768, 3) void mix_kernel(const h8* __restrict__ wg, float* __restrict__ out, int iters, unsigned long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)((i * 7 + 3) % 13) * 0.01f;
    h8 w[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) w[i] = wg[i * 64 + lane];
    i32x8 w8[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) w8[i] = (i32x8){lane + i, 1, 2, 3, 4, 5, 6, 7};
    __syncthreads();
    const char* rp = lds + wave * 4096 + lane * 16;
    char* wp = lds + 49152 + wave * 1024 + lane * 8;
    h8 f0 = *reinterpret_cast<const h8*>(rp), f1 = *reinterpret_cast<const h8*>(rp + 1024), f2 = *reinterpret_cast<const h8*>(rp + 2048);
    i32x4 g0 = *reinterpret_cast<const i32x4*>(rp + 3072), g1 = *reinterpret_cast<const i32x4*>(rp + 3072 + 16);
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
    float v0 = 1.f, v1 = 2.f, sc = 0.5f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        h8 n0, n1, n2;
        i32x4 m0, m1;
        const int o = (it & 1) * 16;
        if (SCHEME < 2) {
            n0 = *reinterpret_cast<const h8*>(rp + o);
            n1 = *reinterpret_cast<const h8*>(rp + 1024 + o);
            if (SCHEME == 0) n2 = *reinterpret_cast<const h8*>(rp + 2048 + o);
            else { m0 = *reinterpret_cast<const i32x4*>(rp + 3072 + o); m1 = *reinterpret_cast<const i32x4*>(rp + 3072 + 32 + o); }
        }
        __builtin_amdgcn_sched_barrier(0);
#define VALU2 if (SCHEME < 2) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(sc)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(sc)); __builtin_amdgcn_sched_barrier(0); }
#define ONE(ACC, W, F) ACC = MFMA_H(w[W], F, ACC); __builtin_amdgcn_sched_barrier(0);
        if (SCHEME == 0 || SCHEME == 3) {
            // 15 MFMAs, 14 VALU in their shadows
            ONE(a0, 0, f0) ONE(a1, 1, f0) ONE(a2, 2, f0) VALU2 ONE(a0, 3, f1) VALU2 ONE(a1, 4, f1) VALU2 ONE(a2, 5, f1) VALU2 ONE(a0, 6, f2) VALU2
            ONE(a1, 7, f2) VALU2 ONE(a2, 8, f2) VALU2 ONE(a0, 9, f0) ONE(a1, 10, f1) ONE(a2, 11, f2) ONE(a0, 12, f0) ONE(a1, 0, f1) ONE(a2, 1, f2)
        } else {
            const i32x8 b8 = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
            // 6 f16 MFMAs + 3 fp8 MFMAs, 20 VALU
            ONE(a0, 0, f0) ONE(a1, 1, f0) VALU2 ONE(a2, 2, f0) VALU2 ONE(a0, 3, f1) VALU2 ONE(a1, 4, f1) VALU2 ONE(a2, 5, f1) VALU2
            a0 = MFMA_8(w8[0], b8, a0); __builtin_amdgcn_sched_barrier(0); VALU2 VALU2
            a1 = MFMA_8(w8[1], b8, a1); __builtin_amdgcn_sched_barrier(0); VALU2 VALU2
            a2 = MFMA_8(w8[2], b8, a2); __builtin_amdgcn_sched_barrier(0); VALU2
        }
#undef ONE
        if (SCHEME < 2) {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            const unsigned b0 = __builtin_bit_cast(unsigned, v0), b1 = __builtin_bit_cast(unsigned, v1);
            *reinterpret_cast<u2*>(wp) = (u2){b0, b1};
            *reinterpret_cast<u2*>(wp + 512) = (u2){b1, b0};
            if (SCHEME == 1) *reinterpret_cast<u2*>(wp + 1024 - 512 * 3) = (u2){b0, b0};
            __builtin_amdgcn_sched_barrier(0);
            f0 = n0; f1 = n1;
            if (SCHEME == 0) f2 = n2; else { g0 = m0; g1 = m1; }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    const f32x4 total = a0 + a1 + a2;
    out[blockIdx.x * blockDim.x + tid] = total[0] + total[1] + total[2] + total[3] + v0 + v1;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int SCHEME>
static void run(const char* name, const h8* w, float* out, unsigned long long* cyc)
{
    const int iters = 4000;
    for (int waves = 4; waves <= 12; waves += 4) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel<SCHEME>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(mix_kernel<SCHEME>, dim3(256), dim3(waves * 64), 65536, 0, w, out, iters, cyc);
        hipLaunchKernelGGL(mix_kernel<SCHEME>, dim3(256), dim3(waves * 64), 65536, 0, w, out, iters, cyc);
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double mx = 0;
        for (int i = 0; i < waves; ++i) mx = h[i] > mx ? (double)h[i] : mx;
        printf("%-74s %d wave(s)/SIMD: %7.1f cycles per group and SIMD\n", name, waves / 4, mx / ((double)iters * (waves / 4)));
    }
}

int main()
{
    h8* w; float* out; unsigned long long* cyc;
    hipMalloc(&w, 13 * 64 * 16); hipMemset(w, 0, 13 * 64 * 16);
    hipMalloc(&out, 256 * 768 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    run<3>("A: 15 f16 MFMAs alone", w, out, cyc);
    run<2>("B: 6 f16 + 3 fp8 (16x16x128) MFMAs alone", w, out, cyc);
    run<0>("A: 15 f16 MFMAs + 3 ds_read_b128 + 14 VALU + 2 ds_write_b64 (shipped mix)", w, out, cyc);
    run<1>("B: 6 f16 + 3 fp8 MFMAs + 4 ds_read_b128 + 20 VALU + 3 ds_write_b64", w, out, cyc);
    return 0;
}
