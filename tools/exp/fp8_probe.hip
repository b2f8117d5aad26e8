// semantics of v_cvt_scalef32_pk_fp8_f16 / v_cvt_scalef32_pk_f16_fp8 on gfx950: direction of the scale, rounding, saturation,
// which half of the destination / source the selector picks.   hipcc --offload-arch=gfx950 fp8_probe.hip -o fp8_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float scale, unsigned* enc, float* dec)
{
    const int l = threadIdx.x;
    h2 v = {(_Float16)in[2 * l], (_Float16)in[2 * l + 1]};
    s2 old = {(short)0x1111, (short)0x2222};
    s2 lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, v, scale, false);
    s2 hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, v, scale, true);
    enc[2 * l] = ((unsigned)(unsigned short)lo[0]) | ((unsigned)(unsigned short)lo[1] << 16);
    enc[2 * l + 1] = ((unsigned)(unsigned short)hi[0]) | ((unsigned)(unsigned short)hi[1] << 16);
    const unsigned packed = enc[2 * l];
    h2 d0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(packed, scale, false);
    h2 d1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(packed, scale, true);
    dec[4 * l] = (float)d0[0]; dec[4 * l + 1] = (float)d0[1]; dec[4 * l + 2] = (float)d1[0]; dec[4 * l + 3] = (float)d1[1];
}
int main()
{
    const float vals[16] = {1.0f, -1.0f, 0.3f, 448.f, 1000.f, 0.001f, 1.0f / 4096, 3.7f / 4096, -0.9f / 4096, 1e-6f, 0.0625f, 17.f, 0.00195f, 0.0029f, 2.5e-4f, -6e-5f};
    float* din; unsigned* denc; float* ddec;
    (void)hipMalloc(&din, 64); (void)hipMalloc(&denc, 64); (void)hipMalloc(&ddec, 128);
    (void)hipMemcpy(din, vals, 64, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 1.0f / 4096, 4096.0f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, din, scale, denc, ddec);
        unsigned e[16]; float d[32];
        (void)hipMemcpy(e, denc, 64, hipMemcpyDeviceToHost); (void)hipMemcpy(d, ddec, 128, hipMemcpyDeviceToHost);
        printf("scale %g\n", scale);
        for (int l = 0; l < 8; ++l)
            printf("  in %12.6g %12.6g | sel0 %08x sel1 %08x | dec(sel0) %12.6g %12.6g dec(sel1) %12.6g %12.6g\n", vals[2 * l], vals[2 * l + 1],
                   e[2 * l], e[2 * l + 1], d[4 * l], d[4 * l + 1], d[4 * l + 2], d[4 * l + 3]);
    }
    return 0;
}
