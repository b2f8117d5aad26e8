// Micro-benchmark for the split-f16 row step: how fast does [4 ds_read_b128 prefetch -> 15 x v_mfma_f32_16x16x32_f16 on 3
// accumulators -> split epilogue + 2 ds_write_b64] run with 1 or 2 waves per SIMD?
//   hipcc -O3 --offload-arch=gfx950 mfma16_loop.hip -o mfma16_loop && ./mfma16_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float sub_half(const float v, const unsigned hh, const bool high)
{
    float r;
    if (high) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(v));
    return r;
}
__device__ __forceinline__ void split(const f32x4 v, h4& hi, h4& lo)
{
    hi = __builtin_convertvector(v, h4);
    const unsigned a = __builtin_bit_cast(unsigned, (h2){hi[0], hi[1]}), b = __builtin_bit_cast(unsigned, (h2){hi[2], hi[3]});
    const f32x4 d = {sub_half(v[0], a, false), sub_half(v[1], a, true), sub_half(v[2], b, false), sub_half(v[3], b, true)};
    lo = __builtin_convertvector(d, h4);
}

struct Frag { h8 ph, pl, sh, sl; };

// MODE bit 0: LDS prefetch reads (distance 2), bit 1: split epilogue + LDS writes, bit 2: no MFMA
template <int MODE>
__global__ __launch_bounds__(512, 2) void step_kernel(const h8* __restrict__ wg, float* __restrict__ out, int steps, unsigned long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 40960 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)((i * 7 + 3) % 13) * 0.01f;
    h8 w[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) w[i] = wg[i * 64 + lane];
    __syncthreads();
    const int n = lane & 15, q = lane >> 4;
    const int vp = (q & 1) * 11520 + (wave * 2 * 36 + n) * 16 + (q >> 1) * 16, vs = (q & 1) * 11520 + (wave * 2 * 36 + n) * 16 + 32;
    const int wr = 23040 + (q >> 1) * 8192 + (wave * 34 + n) * 16 + (q & 1) * 8;
    Frag f0, f1, f2;
    auto load = [&](Frag& f, int row) {
        const int r = (row & 1) * 576;
        f.ph = *reinterpret_cast<const h8*>(lds + vp + r);
        f.pl = *reinterpret_cast<const h8*>(lds + vp + r + 5760);
        f.sh = *reinterpret_cast<const h8*>(lds + vs + r);
        f.sl = *reinterpret_cast<const h8*>(lds + vs + r + 5760);
    };
    load(f0, 0); load(f1, 1); f2 = f0;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, total = a0;
    // one step: prefetch into PF, 15 MFMAs on CUR with accumulators (OLD finishes, MID, NEW starts), epilogue of OLD
#define STEP(CUR, PF, OLD, MID, NEW, IT)                                                                    \
    do {                                                                                                    \
        if (MODE & 1) load(PF, (IT) + 2);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        NEW = (f32x4){0, 0, 0, 0};                                                                          \
        if (!(MODE & 4)) {                                                                                  \
            _Pragma("unroll") for (int m = 0; m < 5; ++m) {                                                 \
                const h8& x = m < 2 ? CUR.ph : (m == 2 ? CUR.pl : (m == 3 ? CUR.sh : CUR.sl));              \
                const int wi = m == 0 ? 0 : (m == 1 ? 1 : (m == 2 ? 0 : (m == 3 ? 2 : 3)));                 \
                OLD = MFMA_H(w[8 + wi], x, OLD);                                                            \
                MID = MFMA_H(w[4 + wi], x, MID);                                                            \
                NEW = MFMA_H(w[0 + wi], x, NEW);                                                            \
            }                                                                                               \
        } else {                                                                                            \
            OLD[0] += (float)CUR.ph[0] + (float)CUR.pl[1] + (float)CUR.sh[2] + (float)CUR.sl[3];            \
        }                                                                                                   \
        if (MODE & 2) {                                                                                     \
            h4 hi, lo;                                                                                      \
            f32x4 v = OLD * 0.001f;                                                                         \
            v.x = __builtin_amdgcn_fmed3f(v.x, 0.f, __builtin_inff()); v.y = __builtin_amdgcn_fmed3f(v.y, 0.f, __builtin_inff()); \
            v.z = __builtin_amdgcn_fmed3f(v.z, 0.f, __builtin_inff()); v.w = __builtin_amdgcn_fmed3f(v.w, 0.f, __builtin_inff()); \
            split(v, hi, lo);                                                                               \
            *reinterpret_cast<h4*>(lds + wr + ((IT) & 1) * 544) = hi;                                       \
            *reinterpret_cast<h4*>(lds + wr + ((IT) & 1) * 544 + 16384) = lo;                               \
        } else {                                                                                            \
            total += OLD;                                                                                   \
        }                                                                                                   \
    } while (0)
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < steps; it += 3) {      // three steps per iteration: no register rotation copies
        STEP(f0, f2, a2, a1, a0, it);
        STEP(f1, f0, a1, a0, a2, it + 1);
        STEP(f2, f1, a0, a2, a1, it + 2);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    total += a1 + a2;
    if (total.x == 12345.f) out[tid] = total.y + total.z + total.w;
    if (lane == 0 && cyc) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int MODE>
static void run(const char* name, const h8* w, float* out, unsigned long long* cyc, int threads, int steps)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(step_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(step_kernel<MODE>, dim3(256), dim3(threads), 65536, 0, w, out, steps, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(step_kernel<MODE>, dim3(256), dim3(threads), 65536, 0, w, out, steps, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const int nw = 256 * threads / 64;
    unsigned long long* h = (unsigned long long*)malloc(nw * 8);
    hipMemcpy(h, cyc, nw * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < nw; ++i) avg += (double)h[i];
    avg /= nw;
    free(h);
    const double per_step = avg / steps;         // s_memtime-domain cycles (100 MHz ref? see ratio to wall)
    printf("%-44s %2d waves/SIMD: %8.3f ms  %8.1f counter-ticks/step/wave  wall %7.1f ns/step  (MFMA floor %d x 15 x 16 cyc)\n",
           name, threads / 256, ms, per_step, ms * 1e6 / steps, threads / 256);
}

int main()
{
    h8* w; float* out; unsigned long long* cyc;
    hipMalloc(&w, 13 * 64 * 16); hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 4096 * 8);
    hipMemset(w, 0x3c, 13 * 64 * 16);
    const int steps = 3999;
    for (int threads : {256, 512}) {
        run<0>("MFMA only (registers)", w, out, cyc, threads, steps);
        run<1>("MFMA + LDS prefetch", w, out, cyc, threads, steps);
        run<3>("MFMA + LDS prefetch + split epilogue/write", w, out, cyc, threads, steps);
        run<5>("no MFMA: LDS prefetch only", w, out, cyc, threads, steps);
        run<7>("no MFMA: LDS prefetch + epilogue/write", w, out, cyc, threads, steps);
    }
    return 0;
}
