// Base convolution of the metric's configuration on the f16 matrix cores, row streaming (base_conv_rows_kernel):
//     uint8 [B,Hs,Ws,3] -> [virtual pad_to_power_of_2] -> normalise -> conv 3x3, 3 -> 16, SAME, no bias [-> relu] -> split-planar f16 hi / lo
//   cast / pad : bfcnn/module_denoiser.py:53-56, bfcnn/utilities.py:736-751
//   normalise  : bfcnn/model.py:100-102 -> bfcnn/utilities.py:449-461  clip(x)/(max-min) - 0.5
//   conv       : bfcnn/backbone_resnet.py:137-147,258-262 (no BN on the base layer)
// The vector kernel (edge_layers.hip base_conv_kernel: 432 FMAs per pixel, one 16x16 tile per workgroup) takes 168-186 us for
// 128 x 256 x 256 where writing its 537 MB takes ~80: measured with its loads AND stores removed it still took 134 us -- it is bound by
// the launch rate of its 32 768 workgroups and by the vector ALU, not by bytes.  This form:
//   * a uint8 value is EXACT in f16, so the input needs no lo part: out = sum w (u / 255 - 0.5 [inside the frame]) is computed as
//     sum (w / 255) u + (-0.5 sum_c w) 1_inside with the "inside" indicator as a FOURTH input channel -- zero padding, the power-of-two
//     band (u = 0 but inside) and the normalisation offset all come out of the same matrix product, no border cases in the code;
//   * one v_mfma_f32_16x16x32_f16 per vertical tap and 16 pixels: K = 32 = {w_hi, w_lo} x {(dx 0, dx 1) | (dx 2, -)} x 4 channels,
//     both split products of the weights (pre-scaled by a power of two, like pack_h3_kernel) in one instruction: 3 MFMAs per 16 pixels;
//   * a workgroup walks down a band of rows of a 256-column chunk: the [r, g, b, 1] f16 records of four input rows live in an LDS
//     ring (8 bytes per pixel), row y+2 is requested from global memory while row y is multiplied; ONE barrier per row.
// Takes u8 input, cin = 3, k = 3, value range [0, 255], split-planar output; everything else stays on the vector kernel.
#include "bf_common.h"
#include "h3_core.h"
#include <cstdlib>

constexpr int BR_NT = 256, BR_CW = 256, BR_RING = 4, BR_ROWPX = BR_CW + 4;      // ring pixels: image columns x0 - 1 .. x0 + 258

__global__ __launch_bounds__(BR_NT) void base_conv_rows_kernel(BaseConvArgs a, int nchunks, int rows_per_band, int nbands, int abl)
{
    // two copies of every ring row, the second shifted by one pixel: the 16 bytes of a pixel PAIR (p, p + 1) are 16-byte aligned in
    // copy p & 1, so that a B fragment is ONE ds_read_b128 whatever the lane's column
    __shared__ __attribute__((aligned(16))) h4 ring[BR_RING][2][BR_ROWPX];
    __shared__ float red[BR_NT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    if (a.status && blockIdx.x == 0 && tid == 0) *a.status = 0;               // first kernel of a forward
    const float* __restrict__ w = a.w;                                         // [3][3][3][16] HWIO

    // ---- A operands: (w / 255 | -0.5 sum_c w) * s split into hi / lo; s = the power of two that puts the largest entry in [2^13, 2^14)
    auto wval = [&](const int tap, const int c, const int m) -> float {
        if (c < 3) return w[(tap * 3 + c) * 16 + m] / 255.0f;
        return -0.5f * (w[(tap * 3 + 0) * 16 + m] + w[(tap * 3 + 1) * 16 + m] + w[(tap * 3 + 2) * 16 + m]);
    };
    float mx = 0.f;
    for (int i = tid; i < 9 * 4 * 16; i += BR_NT) mx = fmaxf(mx, fabsf(wval(i / 64, (i >> 4) & 3, i & 15)));
    red[tid] = mx;
    __syncthreads();
    for (int st = BR_NT / 2; st > 0; st >>= 1) {
        if (tid < st) red[tid] = fmaxf(red[tid], red[tid + st]);
        __syncthreads();
    }
    float s = 1.f;
    {
        const float m0 = red[0];
        if (m0 > 0.f && m0 < 3.0e38f) {
            int ex;
            (void)frexpf(m0, &ex);
            ex = max(-100, min(100, ex));
            s = ldexpf(1.f, 14 - ex);
        }
    }
    const float inv_s = 1.0f / s;
    h8 wa[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // k-slot 8 (q & 1) + j of the half (q >> 1): slots 0..3 = tap dx 0, 4..7 = dx 1, 8..11 = dx 2, 12..15 = unused
            const int slot = 8 * (q & 1) + j, dx = slot >> 2, c = slot & 3;
            float v = 0.f;
            if (dx < 3) v = wval(dy * 3 + dx, c, n) * s;
            const _Float16 hi = (_Float16)v;
            wa[dy][j] = (q >> 1) ? (_Float16)(v - (float)hi) : hi;
        }
    }

    int bx = blockIdx.x;
    const int ch = bx % nchunks; bx /= nchunks;
    const int band = bx % nbands;
    const int b = bx / nbands;
    const int x0 = ch * BR_CW;
    const int y0 = band * rows_per_band, y1 = min(y0 + rows_per_band, a.H);
    const uint8_t* __restrict__ src = reinterpret_cast<const uint8_t*>(a.in) + (int64_t)b * a.Hs * a.Ws * 3;

    // thread t fetches ring pixel j = t (and j = t + 256 for t < 2): image column x0 - 1 + j
    auto fetch = [&](const int yy, const int j) -> h4 {
        const int x = x0 - 1 + j;
        h4 r = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (yy >= 0 && yy < a.H && x >= 0 && x < a.W) {
            r[3] = (_Float16)1.f;                                  // inside the (padded) frame: carries the -0.5 of the normalisation
            if (!(abl & 2) && yy < a.Hs && x < a.Ws) {
                // three byte loads: measured, they cost 41 of the kernel's 128 us (87 us without loads = the time of its 537 MB of
                // stores, `tools/exp/base_rows_abl.sh`), but every other form tried was slower still: one unaligned 4-byte load per
                // pixel 146 us, one aligned 12-byte load per four pixels on 66 threads 139 us, the row requested two steps ahead
                // with its raw bytes held in registers 145 us
                const uint8_t* p = src + ((int64_t)yy * a.Ws + x) * 3;
                r[0] = (_Float16)(float)p[0];
                r[1] = (_Float16)(float)p[1];
                r[2] = (_Float16)(float)p[2];
            }
        }
        return r;
    };
    // ring pixel j of a row (image column x0 - 1 + j) goes to copy 0 at j and to copy 1 at j - 1; pixels 256 .. 259 by threads 0 .. 3
    auto store_px = [&](const int slot, const int j, const h4 v) {
        ring[slot][0][j] = v;
        if (j > 0) ring[slot][1][j - 1] = v;
    };
    auto put = [&](const int yy) {
        const int slot = ((yy % BR_RING) + BR_RING) % BR_RING;
        store_px(slot, tid, fetch(yy, tid));
        if (tid < 4) store_px(slot, BR_CW + tid, fetch(yy, BR_CW + tid));
    };
    put(y0 - 1);
    put(y0);
    put(y0 + 1);
    __syncthreads();

    const int64_t hw = (int64_t)a.H * a.W;
    // after the row exchange of h3_split_record lane (n, q) holds ONE 16-byte record of pixel n: plane (q >> 1) + 2 (q & 1)
    char* outb = reinterpret_cast<char*>(a.out) + (int64_t)b * hw * 64 + (int64_t)((q >> 1) + 2 * (q & 1)) * hw * 16;
    // B fragment of lane (n, q), group g, ring row slot: pixel pair starting at ring pixel p = xc + 2 (q & 1) (q even: taps dx 0, 1;
    // q odd: tap dx 2 and a pixel that meets zero weights), read from copy p & 1 at its aligned position
    int frag_off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int p = 64 * wave + 16 * g + n + 2 * (q & 1);
        frag_off[g] = ((p & 1) * BR_ROWPX + (p & ~1)) * 8;
    }
    const char* ring_b = reinterpret_cast<const char*>(&ring[0][0][0]);
    constexpr int SLOT_BYTES = 2 * BR_ROWPX * 8;
    for (int y = y0; y < y1; ++y) {
        // row y+2: requested now, written to the ring at the end of the step
        const h4 nx = fetch(y + 2, tid);
        h4 nx2 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (tid < 4) nx2 = fetch(y + 2, BR_CW + tid);
        int so[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) so[dy] = ((((y + dy - 1) % BR_RING) + BR_RING) % BR_RING) * SLOT_BYTES;
        char* orow = outb + (int64_t)y * a.W * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int xc = 64 * wave + 16 * g + n;                 // chunk-relative output column of this lane
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
                acc = MFMA_H(wa[dy], *reinterpret_cast<const h8*>(ring_b + so[dy] + frag_off[g]), acc);
            acc = bf_acc_ready(acc) * inv_s;
            if (a.act_relu) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaxf(acc[c], 0.f);
            }
            const int x = x0 + xc;
            const h8 rec = h3_split_record(acc);                   // (all lanes: the exchange needs EXEC all ones)
            if ((abl & 1) ? acc[0] == 12345.678f : x < a.W) *reinterpret_cast<h8*>(orow + (int64_t)x * 16) = rec;
        }
        const int slot = (((y + 2) % BR_RING) + BR_RING) % BR_RING;
        store_px(slot, tid, nx);
        if (tid < 4) store_px(slot, BR_CW + tid, nx2);
        __syncthreads();
    }
}

bool bf_base_conv_rows_supports(const BaseConvArgs& a)
{
    return a.in_is_u8 && a.cin == 3 && a.k == 3 && a.out_split == 1 && a.v_min == 0.f && a.v_max == 255.f && a.H >= 1 && a.W >= 1;
}

hipError_t bf_launch_base_conv_rows(const BaseConvArgs& a, hipStream_t s)
{
    if (!bf_base_conv_rows_supports(a)) return hipErrorInvalidValue;
    const int nchunks = (a.W + BR_CW - 1) / BR_CW;
    // rows per band: enough workgroups to fill the chip a few times, bands tall enough to amortise the three prologue rows
    // (128 x 256 x 256: 32 rows = 1 024 workgroups 117 us, 16 rows 128 us, 8 rows 134 us, 64 rows 158 us)
    int rows = 32;
    int ablate = 0;
#ifdef BF_ABLATE            // timing builds only (tools/exp/base_rows_abl.sh builds with -DBF_ABLATE): the shipped library ignores the environment
    if (const char* e = getenv("BF_BASE_ROWS_ABL")) ablate = atoi(e);                          // 1 = no stores, 2 = no loads
    if (const char* e = getenv("BF_BASE_ROWS_BAND")) rows = atoi(e) > 0 ? atoi(e) : rows;
    else
#endif
    while (rows > 8 && (int64_t)a.B * nchunks * ((a.H + rows - 1) / rows) < 1024) rows /= 2;
    const int nbands = (a.H + rows - 1) / rows;
    const int64_t grid = (int64_t)a.B * nchunks * nbands;
    if (grid > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(base_conv_rows_kernel, dim3((unsigned)grid), dim3(BR_NT), 0, s, a, nchunks, rows, nbands, ablate);
    return hipGetLastError();
}
