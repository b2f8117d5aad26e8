// Micro-benchmark: sustained issue rate of v_mfma_f32_16x16x32_f16 on one SIMD with 1 / 2 / 3 waves, alone and with the
// instruction mix of the row-streaming kernels around it (VALU micro-ops in the MFMA shadows, ds_read_b128 operand fetches,
// ds_write_b64 results).  Prints shader cycles per MFMA and SIMD (16 = the matrix pipe's rate for this instruction).
//   hipcc -O3 --offload-arch=gfx950 mfma_mix.hip -o mfma_mix && ./mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// MODE bits: 1 = two VALU micro-ops behind every MFMA, 2 = operand fragments re-read from LDS (4 x ds_read_b128 per 16 MFMAs),
// 4 = 2 x ds_write_b64 per 16 MFMAs, 8 = one VALU op (instead of two) per MFMA, 16 = the four reads spread over the iteration (one
// behind every fourth MFMA) instead of issued together in front of it, 32 = one ds_write_b128 instead of the two ds_write_b64,
// 64 = the waves of a SIMD run at different priorities (wave / 4)
template <int MODE>
__global__ __launch_bounds__(768, 3) void mix_kernel(const h8* __restrict__ wg, float* __restrict__ out, int iters, unsigned long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)((i * 7 + 3) % 13) * 0.01f;
    h8 w[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) w[i] = wg[i * 64 + lane];
    __syncthreads();
    const char* rp = lds + wave * 4096 + lane * 16;
    char* wp = lds + 49152 + wave * 1024 + lane * 8;
    h8 f0 = *reinterpret_cast<const h8*>(rp), f1 = *reinterpret_cast<const h8*>(rp + 1024), f2 = *reinterpret_cast<const h8*>(rp + 2048),
       f3 = *reinterpret_cast<const h8*>(rp + 3072);
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
    float v0 = 1.f, v1 = 2.f, sc = 0.5f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        h8 n0, n1, n2, n3;
        if (MODE & 64) {
            if ((wave >> 2) == 0) __builtin_amdgcn_s_setprio(3);
            else if ((wave >> 2) == 1) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(1);
        }
        if ((MODE & 2) && !(MODE & 16)) {
            n0 = *reinterpret_cast<const h8*>(rp + (it & 1) * 16);
            n1 = *reinterpret_cast<const h8*>(rp + 1024 + (it & 1) * 16);
            n2 = *reinterpret_cast<const h8*>(rp + 2048 + (it & 1) * 16);
            n3 = *reinterpret_cast<const h8*>(rp + 3072 + (it & 1) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#define RD(N, OFF) if ((MODE & 2) && (MODE & 16)) { N = *reinterpret_cast<const h8*>(rp + OFF + (it & 1) * 16); __builtin_amdgcn_sched_barrier(0); }
#define ONE(ACC, W, F)                                                                                      \
        ACC = MFMA_H(w[W], F, ACC);                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (MODE & 1) {                                                                                     \
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(sc));                                  \
            if (!(MODE & 8)) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(sc));                 \
            __builtin_amdgcn_sched_barrier(0);                                                              \
        }
        ONE(a0, 0, f0) RD(n0, 0) ONE(a1, 1, f0) ONE(a2, 2, f0) ONE(a0, 3, f1) ONE(a1, 4, f1) RD(n1, 1024) ONE(a2, 5, f1) ONE(a0, 6, f2) ONE(a1, 7, f2)
        ONE(a2, 8, f2) RD(n2, 2048) ONE(a0, 9, f3) ONE(a1, 10, f3) ONE(a2, 11, f3) ONE(a0, 12, f0) RD(n3, 3072) ONE(a1, 0, f1) ONE(a2, 1, f2) ONE(a0, 2, f3)
#undef ONE
#undef RD
        if (MODE & 4) {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const unsigned b0 = __builtin_bit_cast(unsigned, v0), b1 = __builtin_bit_cast(unsigned, v1);
            if (MODE & 32) *reinterpret_cast<u4*>(lds + 49152 + wave * 1024 + lane * 16) = (u4){b0, b1, b1, b0};
            else {
                *reinterpret_cast<u2*>(wp) = (u2){b0, b1};
                *reinterpret_cast<u2*>(wp + 512) = (u2){b1, b0};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 2) { f0 = n0; f1 = n1; f2 = n2; f3 = n3; }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    const f32x4 total = a0 + a1 + a2;
    out[blockIdx.x * blockDim.x + tid] = total[0] + total[1] + total[2] + total[3] + v0 + v1;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char* name, const h8* w, float* out, unsigned long long* cyc)
{
    const int iters = 4000;
    for (int waves = 4; waves <= 12; waves += 4) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(mix_kernel<MODE>, dim3(256), dim3(waves * 64), 65536, 0, w, out, iters, cyc);
        hipLaunchKernelGGL(mix_kernel<MODE>, dim3(256), dim3(waves * 64), 65536, 0, w, out, iters, cyc);
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double mx = 0;
        for (int i = 0; i < waves; ++i) mx = h[i] > mx ? (double)h[i] : mx;
        printf("%-52s %d wave(s)/SIMD: %6.2f cycles per MFMA and SIMD\n", name, waves / 4, mx / ((double)iters * 16 * (waves / 4)));
    }
}

int main(int argc, char** argv)
{
    h8* w; float* out; unsigned long long* cyc;
    hipMalloc(&w, 13 * 64 * 16); hipMemset(w, 0, 13 * 64 * 16);
    hipMalloc(&out, 256 * 768 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    if (argc > 2) {
        // power probe: ./mfma_mix <mode 0|7> <launches>: 12 waves per CU, 200 000 iterations per launch (~25 ms each)
        const int mode = atoi(argv[1]), launches = atoi(argv[2]);
        hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        for (int i = 0; i < launches; ++i) {
            if (mode == 0) hipLaunchKernelGGL(mix_kernel<0>, dim3(256), dim3(768), 65536, 0, w, out, 200000, cyc);
            else hipLaunchKernelGGL(mix_kernel<7>, dim3(256), dim3(768), 65536, 0, w, out, 200000, cyc);
        }
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d: %.2f cycles per MFMA and SIMD\n", mode, (double)h[11] / (200000.0 * 16 * 3));
        return 0;
    }
    run<0>("MFMA only", w, out, cyc);
    run<9>("+ 1 VALU per MFMA", w, out, cyc);
    run<1>("+ 2 VALU per MFMA", w, out, cyc);
    run<2>("+ 4 ds_read_b128 per 16", w, out, cyc);
    run<3>("+ 2 VALU per MFMA + 4 ds_read_b128 per 16", w, out, cyc);
    run<7>("+ 2 VALU + 4 ds_read_b128 + 2 ds_write_b64", w, out, cyc);
    run<2 + 16>("+ 4 ds_read_b128, spread", w, out, cyc);
    run<3 + 16>("+ 2 VALU + 4 ds_read_b128, spread", w, out, cyc);
    run<7 + 16>("+ 2 VALU + 4 reads spread + 2 ds_write_b64", w, out, cyc);
    run<7 + 16 + 32>("+ 2 VALU + 4 reads spread + 1 ds_write_b128", w, out, cyc);
    run<7 + 64>("+ 2 VALU + 4 reads + 2 writes, staggered prio", w, out, cyc);
    run<11 + 16 + 32>("+ 1 VALU + 4 reads spread + 1 ds_write_b128", w, out, cyc);
    return 0;
}
