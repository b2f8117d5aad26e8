// On-device data corruption for training: prepare_data_fn of bfcnn/dataset.py:126-239 without tf.data.
//   geometric_augmentation_fn (dataset.py:131-159): whole-batch flip left-right / up-down (the host draws the two
//     coin flips, as the reference draws them once per batch), then tf.round (dataset.py:234)
//   noise_augmentation_fn (dataset.py:161-230): x * truncated_normal(mean 1, std m) then + truncated_normal(mean 0,
//     std a), each applied or not per batch (host coin flips; m, a ~ U[min,max] drawn by the host), then tf.round.
//     No clipping after the noise (the reference has none).
// tf.random.truncated_normal: values more than 2 standard deviations from the mean are dropped and re-picked.
// Random numbers: Philox4x32-10, counter = (element index lo, hi, attempt, stream), key = seed -- counter based, so
// the result does not depend on the launch geometry and a numpy restatement (oracle/bfcnn_oracle.py) can follow it.
// HBM-bound: reads 4 B, writes 8 B per element.
#include "bf_common.h"

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// first standard normal with |z| <= 2 of the element's stream (Box-Muller on successive Philox outputs)
__device__ __forceinline__ float truncated_normal(uint32_t idx_lo, uint32_t idx_hi, uint32_t stream, uint32_t k0, uint32_t k1)
{
    for (uint32_t attempt = 0; attempt < 16; ++attempt) {
        uint32_t r[4];
        philox4x32_10(idx_lo, idx_hi, attempt, stream, k0, k1, r);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(r[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);       // (0, 1]
            const float u2 = (float)(r[2 * h + 1] >> 8) * (1.0f / 16777216.0f);            // [0, 1)
            const float rad = sqrtf(-2.0f * logf(u1));
            const float z0 = rad * cosf(6.28318530717958647692f * u2), z1 = rad * sinf(6.28318530717958647692f * u2);
            if (fabsf(z0) <= 2.0f) return z0;
            if (fabsf(z1) <= 2.0f) return z1;
        }
    }
    return 0.0f;      // probability 0.0455^64
}

__global__ __launch_bounds__(256) void noise_augment_kernel(const float* __restrict__ in, float* __restrict__ out_clean,
                                                            float* __restrict__ out_noisy, int B, int H, int W, int C, int flip_mask,
                                                            float mult_std, float add_std, uint32_t k0, uint32_t k1)
{
    const int64_t n = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int64_t b = t / H;
        const int sx = (flip_mask & 1) ? W - 1 - x : x, sy = (flip_mask & 2) ? H - 1 - y : y;
        const float clean = rintf(in[((b * H + sy) * W + sx) * C + c]);      // tf.round = half to even
        float v = clean;
        // the stream is indexed by the OUTPUT element, so the noise field does not move with the flips
        if (mult_std > 0.f) v *= 1.0f + mult_std * truncated_normal((uint32_t)i, (uint32_t)(i >> 32), 0u, k0, k1);
        if (add_std > 0.f) v += add_std * truncated_normal((uint32_t)i, (uint32_t)(i >> 32), 1u, k0, k1);
        if (out_clean) out_clean[i] = clean;
        out_noisy[i] = rintf(v);
    }
}

extern "C" int bf_noise_augment(const float* in, float* out_clean, float* out_noisy, int B, int H, int W, int C, int flip_mask,
                                float mult_std, float add_std, uint64_t seed, void* stream)
{
    if (!in || !out_noisy || B <= 0 || H <= 0 || W <= 0 || C <= 0) return BF_EINVAL;
    if (mult_std < 0.f || add_std < 0.f || (flip_mask & ~3)) return BF_EINVAL;
    if (in == out_clean || in == out_noisy) return BF_EINVAL;               // flips read other elements: not in place
    const int64_t n = (int64_t)B * H * W * C;
    int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(noise_augment_kernel, dim3((unsigned)(g < 16384 ? g : 16384)), dim3(256), 0, (hipStream_t)stream, in,
                       out_clean, out_noisy, B, H, W, C, flip_mask, mult_std, add_std, (uint32_t)seed, (uint32_t)(seed >> 32));
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}
