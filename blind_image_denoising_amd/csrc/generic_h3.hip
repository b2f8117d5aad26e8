// The bottleneck block of the shipped resnet config (resnet_color_1x6_bn_32x128x32_1x3x1: backbone_resnet.py:149-178,
// backbone_blocks.py:160-243) as ONE kernel with split-f16 GEMMs:
//   out = x + act2(act1(depthwise3x3_x4(act0(x . W0 + shift0)) + shift1) . W2 + shift2)
// x, out [B][H][W][32] fp32; W0 [32][32]; depthwise [3][3][32][4] (BatchNorm scale folded in); W2 [128][32] (the grouped 1x1 as a
// block-diagonal dense matrix, BatchNorm scale folded in); shifts = the folded BatchNorm offsets (or NULL).
// Before: bf_op_pointwise (233 us) + bf_op_dwmult_pointwise (1 220 us, its fp32 MFMA chain behind 72 LDS weight reads per 16 pixels)
// per block at batch 64 x 256 x 256; the 32-channel map made two round trips and the matrix work ran on v_mfma_f32_16x16x4_f32.
//
// A workgroup (4 waves) walks tiles of 8 x 32 pixels:
//   phase A  the leading 1x1 on the tile + 1 pixel of halo (340 pixels = 22 groups of 16): lane (q, n) loads channels 8q .. 8q+7 of
//            pixel n, splits them into hi / lo f16, 3 MFMAs per output tile (w_hi x_hi + w_lo x_hi + w_hi x_lo), activation, and the
//            fp32 result goes to LDS ([pixel][36]: 32 channels + 4 of padding; zero outside the image = the depthwise's padding).
//   phase B  wave w takes 16 columns x 4 rows.  Per chunk c of 32 hidden channels lane (q, n) forms hidden channels 32c + 8q .. + 7
//            (= input channels 8c + 2q, + 1 times the 4 multipliers) of its 4 pixels: 18 two-channel LDS reads (6 rows x 3 columns)
//            feed 4 x 72 FMAs, the 9 x 8 depthwise weights are read once per chunk for all four rows.  Those 8 values, activated
//            and split, ARE the B fragment of K chunk c of the closing 1x1: 6 MFMAs per pixel group.  The 128-channel tensor exists
//            only as 32 registers.
#include "unet_h3_core.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int UG_TH = 8, UG_TW = 32, UG_IH = UG_TH + 2, UG_IW = UG_TW + 2, UG_NPX = UG_IH * UG_IW, UG_PITCH = 36;
constexpr int UG_W0_BYTES = 2 * 2 * 1024, UG_W2_BYTES = 4 * 2 * 2 * 1024, UG_DW_FLOATS = 9 * 128;
constexpr int UG_PACK_BYTES = UG_W0_BYTES + UG_W2_BYTES + UG_DW_FLOATS * 4;           // + 16: {1 / s0, 1 / s2, 0, 0}
constexpr int UG_SHIFT_OFF = UG_PACK_BYTES, UG_TILE_OFF = UG_SHIFT_OFF + (32 + 128 + 32) * 4;
constexpr int UG_LDS_BYTES = UG_TILE_OFF + UG_NPX * UG_PITCH * 4;
static_assert(UG_TILE_OFF % 16 == 0 && 2 * UG_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

// linear / relu / leaky relu without a branch or a template instance per combination: max(v, slope v) with slope 1 / 0 / alpha
// (0 <= alpha <= 1).  relu of a negative number comes out as -0 instead of +0.
// RELU: all three activations are relu (the shipped config): one instruction instead of two
template <bool RELU>
__device__ __forceinline__ float ug_act(float v, float slope) { return RELU ? fmaxf(v, 0.f) : fmaxf(v, slope * v); }

// power of two s with max |w| s in [2^13, 2^14): the lo halves of the scaled weights stay normal f16 numbers
__device__ float ug_block_scale(const float* __restrict__ w, int n, float* red)
{
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    const float mx = red[0];
    __syncthreads();
    if (!(mx > 0.f) || !isfinite(mx)) return 1.f;
    int ex;
    (void)frexpf(mx, &ex);
    ex = max(-100, min(100, ex));
    return ldexpf(1.f, 14 - ex);
}

// packed operand: [W0 fragments 4 KB | W2 fragments 16 KB | depthwise [9][128] fp32 | 1/s0, 1/s2, 0, 0]
// fragment f = (chunk * 2 + tile) * 2 + (0 hi | 1 lo); element (f * 64 + lane) * 8 + i, lane = 16 q + m: W[32 chunk + 8 q + i][16 tile + m] * s
__global__ __launch_bounds__(256) void ug_pack_bneck_kernel(const float* __restrict__ w0, const float* __restrict__ wd,
                                                            const float* __restrict__ w2, char* __restrict__ dst)
{
    __shared__ float red[256];
    const float s0 = ug_block_scale(w0, 32 * 32, red);
    const float s2 = ug_block_scale(w2, 128 * 32, red);
    _Float16* d0 = reinterpret_cast<_Float16*>(dst);
    _Float16* d2 = reinterpret_cast<_Float16*>(dst + UG_W0_BYTES);
    for (int e = threadIdx.x; e < UG_W0_BYTES / 2 + UG_W2_BYTES / 2; e += 256) {
        const bool second = e >= UG_W0_BYTES / 2;
        const int el = second ? e - UG_W0_BYTES / 2 : e;
        const int i = el & 7, lane = (el >> 3) & 63, f = el >> 9;
        const int hl = f & 1, t = (f >> 1) & 1, c = f >> 2, q = lane >> 4, m = lane & 15;
        const float v = second ? w2[(32 * c + 8 * q + i) * 32 + 16 * t + m] * s2 : w0[(8 * q + i) * 32 + 16 * t + m] * s0;
        const _Float16 hi = (_Float16)v;
        (second ? d2 : d0)[el] = hl ? (_Float16)(v - (float)hi) : hi;
    }
    float* dw = reinterpret_cast<float*>(dst + UG_W0_BYTES + UG_W2_BYTES);
    for (int e = threadIdx.x; e < UG_DW_FLOATS; e += 256) dw[e] = wd[e];
    if (threadIdx.x == 0) {
        float* aux = reinterpret_cast<float*>(dst + UG_PACK_BYTES);
        aux[0] = 1.f / s0;
        aux[1] = 1.f / s2;
        aux[2] = 0.f;
        aux[3] = 0.f;
    }
}

extern "C" int64_t bf_op_bneck_h3_pack_bytes(void) { return UG_PACK_BYTES + 16; }

extern "C" int bf_op_pack_bneck_h3(const float* w0, const float* wd, const float* w2, void* packed, void* stream)
{
    if (!w0 || !wd || !w2 || !packed || (uintptr_t)packed % 16) return BF_EINVAL;
    hipLaunchKernelGGL(ug_pack_bneck_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w0, wd, w2, (char*)packed);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

#ifndef UG_ABLATE
#define UG_ABLATE 0       // timing builds (results wrong): 1 no phase A, 2 no depthwise FMAs, 4 no closing MFMAs, 8 no residual / store
#endif

// RELU: all three activations are ReLU; FAST: every tile is full (height % 8 == 0, width % 32 == 0) and the skip is on
template <bool RELU, bool FAST>
__global__ __launch_bounds__(256, 2) void ug_bneck_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          const char* __restrict__ packed, const float* __restrict__ shift0,
                                                          const float* __restrict__ shift1, const float* __restrict__ shift2,
                                                          float slope0, float slope1, float slope2, int add_res, int B, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    {
        const int4* src = reinterpret_cast<const int4*>(packed);
        int4* dstv = reinterpret_cast<int4*>(lds);
        for (int i = threadIdx.x; i < UG_PACK_BYTES / 16; i += 256) dstv[i] = src[i];
        float* sh = reinterpret_cast<float*>(lds + UG_SHIFT_OFF);
        if (threadIdx.x < 32) sh[threadIdx.x] = shift0 ? shift0[threadIdx.x] : 0.f;
        else if (threadIdx.x < 160) sh[threadIdx.x] = shift1 ? shift1[threadIdx.x - 32] : 0.f;
        else if (threadIdx.x < 192) sh[threadIdx.x] = shift2 ? shift2[threadIdx.x - 160] : 0.f;
    }
    const float* aux = reinterpret_cast<const float*>(packed + UG_PACK_BYTES);
    const float inv0 = aux[0], inv2 = aux[1];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
    const char* w0l = lds + lane * 16;
    const char* w2l = lds + UG_W0_BYTES + lane * 16;
    const float* dwl = reinterpret_cast<const float*>(lds + UG_W0_BYTES + UG_W2_BYTES);
    const float* sh0 = reinterpret_cast<const float*>(lds + UG_SHIFT_OFF);
    const float* sh1 = sh0 + 32;
    const float* sh2 = sh0 + 160;
    float* tile = reinterpret_cast<float*>(lds + UG_TILE_OFF);
    const int tiles_x = (W + UG_TW - 1) / UG_TW, tiles_y = (H + UG_TH - 1) / UG_TH;
    const int ntiles = B * tiles_y * tiles_x;
    // each XCD (workgroup id mod 8) walks its own contiguous eighth of the tiles: a tile's halo is its neighbours' interior, and only a
    // neighbour on the same XCD finds it in that XCD's L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3, per_xcd = (ntiles + 7) >> 3;
    const int cw = wave & 1, rq = wave >> 1;                       // phase B: columns 16 cw .. + 15, rows 4 rq .. + 3 of the tile

    // phase A's pixels are requested one tile ahead (behind the barrier that ends the previous phase A, so that phase B's matrix and
    // vector work covers the round trip): 48 registers per lane; the loads are unconditional from clamped addresses (a branch around
    // them makes hipcc drain the queue where the paths join)
    constexpr int NG = 6;                                          // groups wave + 4 k of the 22 (340 pixels) of a haloed tile
    f32x4 xa[NG][2];
    bool inimg[NG];
    int pyk[NG], pxk[NG];                                          // the lane's pixel of group k inside the haloed tile: the same for every tile
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int e = 16 * (wave + 4 * k) + n;
        pyk[k] = e < UG_NPX ? e / UG_IW - 1 : -(1 << 20);         // past the 340th pixel: never inside an image
        pxk[k] = e - UG_IW * (e / UG_IW) - 1;
    }
    auto request = [&](const int tile_id) {
        const int tx = tile_id % tiles_x, rest = tile_id / tiles_x, ty = rest % tiles_y;
        const float* ib = x + (int64_t)(rest / tiles_y) * H * W * 32 + 8 * q;
        const int x0 = tx * UG_TW, y0 = ty * UG_TH;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int yy = y0 + pyk[k], xx = x0 + pxk[k];
            inimg[k] = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const float* src = ib + (min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)) * 32;
            xa[k][0] = *reinterpret_cast<const f32x4*>(src);
            xa[k][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
    };
    // ---- phase A: the leading 1x1 on the haloed tile whose pixels `request` fetched
    auto phase_a = [&]() {
        if (UG_ABLATE & 1) return;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            if (16 * (wave + 4 * k) >= UG_NPX) continue;           // wave-uniform
            uh8 xh, xl;
            uh_split8(xa[k][0], xa[k][1], xh, xl);
            const int e = 16 * (wave + 4 * k) + n;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uh8 ah = *reinterpret_cast<const uh8*>(w0l + (2 * t) * 1024);
                const uh8 al = *reinterpret_cast<const uh8*>(w0l + (2 * t + 1) * 1024);
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                a = UH_MFMA_REAL(ah, xh, a);
                a = UH_MFMA_REAL(al, xh, a);
                a = UH_MFMA_REAL(ah, xl, a);
                f32x4 v = a * inv0 + *reinterpret_cast<const f32x4*>(sh0 + 16 * t + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = inimg[k] ? ug_act<RELU>(v[r], slope0) : 0.f;
                if (e < UG_NPX) *reinterpret_cast<f32x4*>(tile + e * UG_PITCH + 16 * t + 4 * q) = v;
            }
        }
    };
    // One chunk of 32 hidden channels of phase B (no memory instruction inside: the chunk loops below do not disturb hipcc's count of
    // what is in flight)
    auto chunk = [&](const int c, const float* tb, f32x4 (&acc)[4][2]) {
        f32x2 px[6][3];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) px[r][kx] = *reinterpret_cast<const f32x2*>(tb + (r * UG_IW + kx) * UG_PITCH + 8 * c);
        const int h0 = 32 * c + 8 * q;
        f32x4 hv[4][2];
        {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sh1 + h0), b1 = *reinterpret_cast<const f32x4*>(sh1 + h0 + 4);
#pragma unroll
            for (int o = 0; o < 4; ++o) { hv[o][0] = b0; hv[o][1] = b1; }
        }
        if (!(UG_ABLATE & 2)) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 wa = *reinterpret_cast<const f32x4*>(dwl + (ky * 3 + kx) * 128 + h0);
                    const f32x4 wb = *reinterpret_cast<const f32x4*>(dwl + (ky * 3 + kx) * 128 + h0 + 4);
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        hv[o][0] += wa * px[o + ky][kx][0];
                        hv[o][1] += wb * px[o + ky][kx][1];
                    }
                }
        }
        uh8 bh[4], bl[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hv[o][0][r] = ug_act<RELU>(hv[o][0][r], slope1);
                hv[o][1][r] = ug_act<RELU>(hv[o][1][r], slope1);
            }
            uh_split8(hv[o][0], hv[o][1], bh[o], bl[o]);
        }
        if (!(UG_ABLATE & 4)) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uh8 ah = *reinterpret_cast<const uh8*>(w2l + ((c * 2 + t) * 2) * 1024);
                const uh8 al = *reinterpret_cast<const uh8*>(w2l + ((c * 2 + t) * 2 + 1) * 1024);
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o][t] = UH_MFMA_REAL(ah, bh[o], acc[o][t]);
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o][t] = UH_MFMA_REAL(al, bh[o], acc[o][t]);
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o][t] = UH_MFMA_REAL(ah, bl[o], acc[o][t]);
            }
        } else {
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o][0] += __builtin_bit_cast(f32x4, bh[o]) + __builtin_bit_cast(f32x4, bl[o]);
        }
    };

    // The tile loop is rotated: [request tile k+1] [phase B of tile k] [barrier] [phase A of tile k+1] [barrier].  gfx950 counts loads and
    // stores in one queue (vmcnt): with the request and the phase A that consumes it in ONE straight-line iteration -- and, in the FAST
    // instance (full tiles, skip on), no branch around the skip's loads and the stores in between -- hipcc knows that 8 loads and 8 stores
    // were issued behind the request and waits with vmcnt(26) .. vmcnt(16); in the first form (request consumed across the loop's back
    // edge, conditional stores) it waited with vmcnt(10) .. (0): every tile waited for its own stores to land before phase A could start.
    int tl = slot, tile_id = xcd * per_xcd + tl;
    bool valid = tl < per_xcd && tile_id < ntiles;
    if (valid) {
        request(tile_id);
        phase_a();
    }
    __syncthreads();
    while (valid) {                                                // workgroup-uniform
        const int tx = tile_id % tiles_x, rest = tile_id / tiles_x, ty = rest % tiles_y;
        const int64_t img = (int64_t)(rest / tiles_y) * H * W;
        const int x0 = tx * UG_TW, y0 = ty * UG_TH;
        const int tn = tl + nslots, idn = xcd * per_xcd + tn;
        const bool has_next = tn < per_xcd && idn < ntiles;
        request(has_next ? idn : tile_id);

        // ---- phase B: depthwise x4 -> activation -> closing 1x1 on 4 rows x 16 columns per wave
        {
            const int gx = x0 + 16 * cw + n;
            f32x4 res[4][2];
            f32x4 acc[4][2];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[o][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* tb = tile + ((4 * rq) * UG_IW + 16 * cw + n) * UG_PITCH + 2 * q;
#pragma unroll 1
            for (int c = 0; c < 2; ++c) chunk(c, tb, acc);
            if ((FAST || add_res) && !(UG_ABLATE & 8)) {           // the skip: two chunks of work ahead of its use, out of the way before
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int gy = y0 + 4 * rq + o;
                    const float* src = x + (img + (int64_t)min(gy, H - 1) * W + min(gx, W - 1)) * 32 + 4 * q;
#pragma unroll
                    for (int t = 0; t < 2; ++t) res[o][t] = *reinterpret_cast<const f32x4*>(src + 16 * t);
                }
            }
#pragma unroll 1
            for (int c = 2; c < 4; ++c) chunk(c, tb, acc);
            if (!(UG_ABLATE & 8)) {
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int gy = y0 + 4 * rq + o;
                    if (!FAST && (gy >= H || gx >= W)) continue;
                    float* dst = out + (img + (int64_t)gy * W + gx) * 32 + 4 * q;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        f32x4 v = acc[o][t] * inv2 + *reinterpret_cast<const f32x4*>(sh2 + 16 * t + 4 * q);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = ug_act<RELU>(v[r], slope2);
                        if (FAST || add_res) v += res[o][t];
                        *reinterpret_cast<f32x4*>(dst + 16 * t) = v;
                    }
                }
            } else if (acc[0][0][0] == 12345.678f) {
                out[lane] = acc[1][0][0] + acc[2][1][1] + acc[3][0][2];
            }
        }
        __syncthreads();                                           // phase A of the next tile overwrites the staged map
        if (has_next) phase_a();
        __syncthreads();
        tl = tn;
        tile_id = idn;
        valid = has_next;
    }
}

// out = [x +] act2(act1(depthwise3x3_x4(act0(x . W0 + shift0)) + shift1) . W2 + shift2); 32 -> 32 -> 128 -> 32 channels; packed from
// bf_op_pack_bneck_h3; shift0 [32], shift1 [128], shift2 [32] or NULL; out != x (a tile's halo is read after its neighbours are written).
extern "C" int bf_op_bneck_block_h3(const float* x, float* out, const void* packed, const float* shift0, int act0, float alpha0,
                                    const float* shift1, int act1, float alpha1, const float* shift2, int act2, float alpha2, int add_res,
                                    int B, int H, int W, void* stream)
{
    if (!x || !out || !packed || x == out || B <= 0 || H <= 0 || W <= 0) return BF_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)packed) % 16) return BF_EINVAL;
    float slope[3];
    const int acts[3] = {act0, act1, act2};
    const float alphas[3] = {alpha0, alpha1, alpha2};
    for (int i = 0; i < 3; ++i) {
        if (acts[i] < 0 || acts[i] > 2) return BF_EUNSUPPORTED;    // gelu / tanh: the two-kernel path
        if (acts[i] == 2 && !(alphas[i] >= 0.f && alphas[i] <= 1.f)) return BF_EINVAL;
        slope[i] = acts[i] == 0 ? 1.f : (acts[i] == 1 ? 0.f : alphas[i]);
    }
    const int64_t ntiles = (int64_t)B * ((H + UG_TH - 1) / UG_TH) * ((W + UG_TW - 1) / UG_TW);
    if (ntiles >= ((int64_t)1 << 31) || (int64_t)B * H * W >= ((int64_t)1 << 31) || (int64_t)H * W * 32 >= ((int64_t)1 << 31))
        return BF_EUNSUPPORTED;                                    // 32-bit element offsets inside an image
    const bool relu = act0 == 1 && act1 == 1 && act2 == 1;
    const bool fast = add_res && H % UG_TH == 0 && W % UG_TW == 0;
    const int per_xcd = (int)((ntiles + 7) / 8);
    const int grid = 8 * (int)std::min<int64_t>(per_xcd, 64);      // 512 workgroups: two per CU, a multiple of the 8 XCDs
#define UG_LAUNCH(R, F)                                                                                                        \
    do {                                                                                                                       \
        if (bf_set_max_lds(reinterpret_cast<const void*>(ug_bneck_kernel<R, F>), UG_LDS_BYTES) != hipSuccess) return BF_EHIP;  \
        hipLaunchKernelGGL((ug_bneck_kernel<R, F>), dim3(grid), dim3(256), UG_LDS_BYTES, (hipStream_t)stream, x, out,          \
                           (const char*)packed, shift0, shift1, shift2, slope[0], slope[1], slope[2], add_res, B, H, W);       \
    } while (0)
    if (relu && fast) UG_LAUNCH(true, true);
    else if (relu) UG_LAUNCH(true, false);
    else if (fast) UG_LAUNCH(false, true);
    else UG_LAUNCH(false, false);
#undef UG_LAUNCH
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}
