// Fused ENCODER ConvNext block kernels of the unet_laplacian backbone (split-f16 MLP of unet_h3.hip behind a k x k
// depthwise convolution + LayerNorm); see unet_h3.hip for the arithmetic and the fragment layouts.
#include "unet_h3_core.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------
// Whole ENCODER ConvNextBlock (k x k depthwise, 32 channels) + residual Add in one kernel:
//   out = x + mult * (act(LayerNorm(dw_kxk(x)) * gamma . W1) . W2)
// A wave owns an 8-pixel-wide column strip and walks down it as uo_dwconv_ln_rows_kernel does (4 channels per lane,
// 8 lanes per pixel, k rotating accumulators, DPP LayerNorm).  Every 8 finished rows (64 pixels) are handed over through a
// wave-private LDS buffer to the matrix-core layout (lane (q, n): channels 8q..8q+7 of pixel n; a 16-pixel group = 2 rows
// x 8 columns) and go through the split-f16 MLP; the skip comes from the rows just read (cache hits).  The LayerNorm
// output and the hidden layer never reach HBM: x is read once (+ halo), out written once.
// ------------------------------------------------------------------------------------------
constexpr int UH_ENC_ROWS = 16;        // rows per tile
constexpr int UH_ROWBUF_F4 = 96;       // 16-byte elements of a strip row incl. halo (12 pixels x 8)
constexpr int UH_STG_PITCH = 36;       // floats per staged pixel (32 + 4: the 16 lanes of a fragment read spread over the banks)
template <int K, int ACT>
__global__ __launch_bounds__(256, 2) void uh_enc32_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          const float* __restrict__ dww, const float* __restrict__ gamma, float eps,
                                                          const void* __restrict__ packed, const float* __restrict__ mult, int B, int H,
                                                          int W, float alpha)
{
    constexpr int C = 32, NP = 4, RAD = K / 2, T2 = 2, RB = 8;
    constexpr int W_BYTES = 32 * C * C;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    {
        const int4* src = reinterpret_cast<const int4*>(packed);
        int4* dstv = reinterpret_cast<int4*>(lds);
        for (int i = threadIdx.x; i < W_BYTES / 16; i += 256) dstv[i] = src[i];
    }
    const float* aux = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + W_BYTES);
    const float inv1 = aux[0], inv2 = aux[1];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;            // matrix-core layout
    const int cl = lane & 7, pl = lane >> 3;           // depthwise layout: channels 4cl..4cl+3 of strip column pl
    float* stg = reinterpret_cast<float*>(lds + W_BYTES) + wave * (RB * 8 * UH_STG_PITCH);
    f32x4 m4[T2];
#pragma unroll
    for (int t = 0; t < T2; ++t) {
        m4[t] = (f32x4){inv2, inv2, inv2, inv2};
        if (mult) m4[t] *= *reinterpret_cast<const f32x4*>(mult + 16 * t + 4 * q);
    }
    f32x4 gm = {1.f, 1.f, 1.f, 1.f};
    if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + 4 * cl);
    const int tiles_x = (W + 31) / 32, tiles_y = (H + UH_ENC_ROWS - 1) / UH_ENC_ROWS;
    const int64_t ntiles = (int64_t)B * tiles_y * tiles_x;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = (int)(tile % tiles_x);
        const int ty = (int)((tile / tiles_x) % tiles_y);
        const int64_t img = (tile / ((int64_t)tiles_x * tiles_y)) * H * W;
        const int x0 = tx * 32 + wave * 8, y0 = ty * UH_ENC_ROWS;
        if (x0 >= W) continue;                           // wave-uniform; no workgroup barrier inside the tile loop
        // ---- depthwise column walk state
        const int xd = x0 + pl;
        int xo[K];
        float xm[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int xx = xd + kx - RAD;
            xm[kx] = (xx >= 0 && xx < W) ? 1.f : 0.f;
            xo[kx] = min(max(xx, 0), W - 1) * C + 4 * cl;
        }
        auto load_row = [&](int yi, f32x4 (&v)[K]) {
            const float* row = x + (img + (int64_t)min(max(yi, 0), H - 1) * W) * C;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) v[kx] = *reinterpret_cast<const f32x4*>(row + xo[kx]);
        };
        f32x4 acc[K], vn[K];
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        load_row(y0 - RAD, vn);
        int yi = y0 - RAD;
        for (int batch = 0; batch < UH_ENC_ROWS / RB; ++batch) {
            const int yb = y0 + batch * RB;              // first output row of the batch
            if (yb >= H) break;
            {
                // depthwise weights: reloaded per batch (opaque pointer) so that they are not live across the matrix phase
                const float* wp = dww;
                asm volatile("" : "+s"(wp));
                f32x4 wk[K * K];
#pragma unroll
                for (int i = 0; i < K * K; ++i) wk[i] = *reinterpret_cast<const f32x4*>(wp + i * C + 4 * cl);
                for (; yi < yb + RB + RAD; ++yi) {
                    f32x4 v[K];
                    const float ym = (yi >= 0 && yi < H) ? 1.f : 0.f;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) v[kx] = vn[kx] * (xm[kx] * ym);
                    load_row(yi + 1, vn);
#pragma unroll
                    for (int ky = 0; ky < K; ++ky)
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) acc[ky] += wk[ky * K + kx] * v[kx];
                    const int yo = yi - RAD;
                    if (yo >= yb) {
                        f32x4 r = acc[K - 1];
                        if (gamma) {
                            const float mean = uh_pixel_sum8(r[0] + r[1] + r[2] + r[3]) * (1.f / C);
                            const f32x4 d = r - mean;
                            const float var = uh_pixel_sum8(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / C);
                            r = d * (gm * rsqrtf(var + eps));
                        }
                        *reinterpret_cast<f32x4*>(stg + ((yo - yb) * 8 + pl) * UH_STG_PITCH + 4 * cl) = r;
                    }
#pragma unroll
                    for (int j = K - 1; j > 0; --j) acc[j] = acc[j - 1];
                    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            // ---- hand-over inside the wave: the staged rows are read by other lanes than wrote them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int wl = lane * 16;
            asm volatile("" : "+v"(wl));
            const char* w1l = lds + wl;
            const char* w2l = lds + 16 * C * C + wl;
            uh8 xh[1][NP], xl[1][NP];
            f32x4 sk[T2][NP];
            int64_t pix[NP];
            bool ok[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int s = 16 * i + n;                 // staged pixel: row 2i + n / 8, column n % 8
                const float* sp = stg + s * UH_STG_PITCH + 8 * q;
                uh_split8(*reinterpret_cast<const f32x4*>(sp), *reinterpret_cast<const f32x4*>(sp + 4), xh[0][i], xl[0][i]);
                const int py = yb + 2 * i + (n >> 3), px = x0 + (n & 7);
                ok[i] = py < H && px < W;
                pix[i] = img + (int64_t)min(py, H - 1) * W + min(px, W - 1);
#pragma unroll
                for (int t = 0; t < T2; ++t) sk[t][i] = *reinterpret_cast<const f32x4*>(x + pix[i] * C + 16 * t + 4 * q);
            }
            f32x4 acc2[T2][NP];
            uh_mlp_core<C, NP, ACT>(xh, xl, w1l, w2l, inv1, alpha, acc2);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (!ok[i]) continue;
#pragma unroll
                for (int t = 0; t < T2; ++t)
                    *reinterpret_cast<f32x4*>(out + pix[i] * C + 16 * t + 4 * q) = bf_acc_ready(acc2[t][i]) * m4[t] + sk[t][i];
            }
            // the next batch overwrites the staging buffer
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// workgroup barrier that publishes LDS writes (staging) but leaves vector-memory operations in flight: hipcc puts
// "s_waitcnt vmcnt(0)" in front of every barrier it can see, which here would drain the producers' row ring and make the
// consumers wait for their stores to reach memory at every step (measured: producers alone 606 us, consumers alone 789 us,
// together 1076 us with __syncthreads()).  The wait goes through the builtin so that hipcc's own bookkeeping sees it.
__device__ __forceinline__ void uh_step_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0); vmcnt / expcnt untouched (gfx9 encoding)
    asm volatile("s_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// Wave-specialised form of the kernel above: waves 0-3 of a 512-thread workgroup only PRODUCE (depthwise walk + LayerNorm
// of their 8-pixel strip into a staging buffer), waves 4-7 only CONSUME (split-f16 MLP, skip, store) the batch of 8 rows
// the producers finished one step earlier; one workgroup barrier per step flips the double-buffered staging area.
// A SIMD hosts one wave of each kind, so the vector-ALU stream of the depthwise phase and the matrix stream of the MLP
// overlap instead of alternating inside one wave (ablations of uh_enc32_kernel: data movement 696 us + depthwise 265 us
// + matrix 280 us, additive), and the depthwise weights stay in registers.
// ------------------------------------------------------------------------------------------
template <int K, int ACT>
__global__ __launch_bounds__(512, 1) void uh_enc32s_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           const float* __restrict__ dww, const float* __restrict__ gamma, float eps,
                                                           const void* __restrict__ packed, const float* __restrict__ mult, int B, int H,
                                                           int W, float alpha)
{
    constexpr int C = 32, NP = 4, RAD = K / 2, T2 = 2, RB = 8;
    constexpr int W_BYTES = 32 * C * C, STG_FLOATS = RB * 8 * UH_STG_PITCH;       // one strip of one batch
    extern __shared__ __attribute__((aligned(16))) char lds[];
    {
        const int4* src = reinterpret_cast<const int4*>(packed);
        int4* dstv = reinterpret_cast<int4*>(lds);
        for (int i = threadIdx.x; i < W_BYTES / 16; i += 512) dstv[i] = src[i];
    }
    const float* aux = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + W_BYTES);
    const float inv1 = aux[0], inv2 = aux[1];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool producer = wave < 4;
    const int strip = wave & 3;
    float* stg_base = reinterpret_cast<float*>(lds + W_BYTES) + strip * STG_FLOATS;     // + (step & 1) * 4 * STG_FLOATS
    const int tiles_x = (W + 31) / 32, tiles_y = (H + UH_ENC_ROWS - 1) / UH_ENC_ROWS;
    const int64_t ntiles = (int64_t)B * tiles_y * tiles_x;
    const int64_t my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int64_t nsteps = 2 * my_tiles;                   // batches of RB rows; UH_ENC_ROWS / RB = 2 per tile
    static_assert(UH_ENC_ROWS == 2 * RB, "two batches per tile");

    // The two roles are two separate loops (wave-uniform branch) with the same number of barrier arrivals per wave: as one
    // loop hipcc kept the producers' 100 weight registers AND the consumers' matrix state live together (spills).
    if (producer) {
        // ---- depthwise layout: channels 4cl..4cl+3 of strip column pl
        const int cl = lane & 7, pl = lane >> 3;
        // A strip row (8 + 2 RAD pixels x 32 channels, contiguous in memory) is fetched by the whole wave with one 16-byte
        // and one 8-byte load per lane, PD rows ahead of its use (register ring with static slots: the row loop is
        // unrolled by PD), passed through a wave-private LDS row buffer and read back as the k taps of every lane.
        // A SIMD hosts ONE producer wave, so nothing else hides its load latency, and what counts is UNIQUE bytes in
        // flight: with the k taps loaded directly one row ahead (5 KB per wave in flight, 1.5 KB of it unique) every row
        // cost a memory round trip (~0.9 us) and the kernel was no faster than uh_enc32_kernel.
        constexpr int PD = 4, ROWF4 = (8 + 2 * RAD) * 8, HALVES = 2 * (ROWF4 - 64);
        static_assert((RB + 2 * RAD) % PD == 0 || K == 3, "row groups");
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        struct RowRegs { f32x4 a; f32x2 b; };
        f32x4 wk[K * K], gm = {1.f, 1.f, 1.f, 1.f}, acc[K];
        RowRegs ring[PD];
        int go0 = 0, go1 = 0;
        float gm0 = 0.f, gm1 = 0.f;
        int yi = 0, py0 = 0;
        int64_t pimg = 0;
        bool plive = false;
        const int h1 = lane & (HALVES - 1);                  // half-element (8 bytes) of the row tail this lane fetches
        f32x4* rowbuf = reinterpret_cast<f32x4*>(lds + W_BYTES + 2 * 4 * STG_FLOATS * 4) + strip * (2 * UH_ROWBUF_F4);
#pragma unroll
        for (int i = 0; i < K * K; ++i) wk[i] = *reinterpret_cast<const f32x4*>(dww + i * C + 4 * cl);
        if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + 4 * cl);
        auto issue = [&](int yy, RowRegs& r) {
            const float* row = x + (pimg + (int64_t)min(max(yy, 0), H - 1) * W) * C;
            r.a = *reinterpret_cast<const f32x4*>(row + go0);
            r.b = *reinterpret_cast<const f32x2*>(row + go1);
        };
        for (int64_t step = 0; step <= nsteps; ++step) {
            if (step < nsteps) {
                const int batch = (int)(step & 1);
                float* stg = stg_base + (step & 1) * 4 * STG_FLOATS;
                if (batch == 0) {                           // new tile: coordinates, column masks, first rows
                    const int64_t tile = blockIdx.x + (step >> 1) * gridDim.x;
                    const int tx = (int)(tile % tiles_x);
                    const int ty = (int)((tile / tiles_x) % tiles_y);
                    pimg = (tile / ((int64_t)tiles_x * tiles_y)) * H * W;
                    const int x0 = tx * 32 + strip * 8;
                    py0 = ty * UH_ENC_ROWS;
                    plive = x0 < W;
                    const int xg0 = x0 - RAD + (lane >> 3), xg1 = x0 - RAD + 8 + (h1 >> 4);
                    gm0 = (xg0 >= 0 && xg0 < W) ? 1.f : 0.f;
                    gm1 = (xg1 >= 0 && xg1 < W) ? 1.f : 0.f;
                    go0 = min(max(xg0, 0), W - 1) * C + 4 * cl;
                    go1 = min(max(xg1, 0), W - 1) * C + 2 * (h1 & 15);
#pragma unroll
                    for (int j = 0; j < K; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    yi = py0 - RAD;
                    if (plive) {
#pragma unroll
                        for (int d = 0; d < PD; ++d) issue(yi + d, ring[d]);
                    }
                }
                const int yb = py0 + batch * RB;
                if (plive && yb < H && !(UH_ROLE_ABLATE & 2)) {
                    auto row_body = [&](RowRegs& slot, f32x4* rb) {
                        // row yi: registers -> row buffer (column mask applied here), then request row yi + PD
                        rb[lane] = slot.a * gm0;
                        if (lane < HALVES) reinterpret_cast<f32x2*>(rb + 64)[lane] = slot.b * gm1;
                        issue(yi + PD, slot);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const float ym = (yi >= 0 && yi < H) ? 1.f : 0.f;
                        f32x4 v[K];
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) v[kx] = rb[(pl + kx) * 8 + cl] * ym;
#pragma unroll
                        for (int ky = 0; ky < K; ++ky)
#pragma unroll
                            for (int kx = 0; kx < K; ++kx) acc[ky] += wk[ky * K + kx] * v[kx];
                        const int yo = yi - RAD;
                        if (yo >= yb) {
                            f32x4 r = acc[K - 1];
                            if (gamma) {
                                const float mean = uh_pixel_sum8(r[0] + r[1] + r[2] + r[3]) * (1.f / C);
                                const f32x4 d = r - mean;
                                const float var = uh_pixel_sum8(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / C);
                                r = d * (gm * rsqrtf(var + eps));
                            }
                            *reinterpret_cast<f32x4*>(stg + ((yo - yb) * 8 + pl) * UH_STG_PITCH + 4 * cl) = r;
                        }
#pragma unroll
                        for (int j = K - 1; j > 0; --j) acc[j] = acc[j - 1];
                        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        ++yi;
                    };
                    // rows per batch: RB + 2 RAD, then RB; K = 3 (10 rows) ends its first batch on a half group
                    while (yi < yb + RB + RAD) {
                        row_body(ring[0], rowbuf);
                        row_body(ring[1], rowbuf + UH_ROWBUF_F4);
                        if (K == 3 && yi >= yb + RB + RAD) {          // rotate the ring by two so that slot 0 is the next row
                            const RowRegs t0 = ring[0], t1 = ring[1];
                            ring[0] = ring[2]; ring[1] = ring[3]; ring[2] = t0; ring[3] = t1;
                            break;
                        }
                        row_body(ring[2], rowbuf);
                        row_body(ring[3], rowbuf + UH_ROWBUF_F4);
                    }
                }
            }
            uh_step_barrier();
        }
    } else {
        // ---- matrix-core layout
        const int q = lane >> 4, n = lane & 15;
        f32x4 m4[T2];
#pragma unroll
        for (int t = 0; t < T2; ++t) {
            m4[t] = (f32x4){inv2, inv2, inv2, inv2};
            if (mult) m4[t] *= *reinterpret_cast<const f32x4*>(mult + 16 * t + 4 * q);
        }
        for (int64_t step = 0; step <= nsteps; ++step) {
            if (step >= 1) {
            const int64_t cs = step - 1;                    // the batch the producers finished in the previous step
            const int batch = (int)(cs & 1);
            const float* stg = stg_base + (cs & 1) * 4 * STG_FLOATS;
            const int64_t tile = blockIdx.x + (cs >> 1) * gridDim.x;
            const int tx = (int)(tile % tiles_x);
            const int ty = (int)((tile / tiles_x) % tiles_y);
            const int64_t img = (tile / ((int64_t)tiles_x * tiles_y)) * H * W;
            const int x0 = tx * 32 + strip * 8, yb = ty * UH_ENC_ROWS + batch * RB;
            if (x0 < W && yb < H && !(UH_ROLE_ABLATE & 1)) {
                int wl = lane * 16;
                asm volatile("" : "+v"(wl));
                const char* w1l = lds + wl;
                const char* w2l = lds + 16 * C * C + wl;
                uh8 xh[1][NP], xl[1][NP];
                f32x4 sk[T2][NP];
                int64_t pix[NP];
                bool ok[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int s = 16 * i + n;             // staged pixel: row 2i + n / 8, column n % 8
                    const float* sp = stg + s * UH_STG_PITCH + 8 * q;
                    uh_split8(*reinterpret_cast<const f32x4*>(sp), *reinterpret_cast<const f32x4*>(sp + 4), xh[0][i], xl[0][i]);
                    const int py = yb + 2 * i + (n >> 3), px = x0 + (n & 7);
                    ok[i] = py < H && px < W;
                    pix[i] = img + (int64_t)min(py, H - 1) * W + min(px, W - 1);
#pragma unroll
                    for (int t = 0; t < T2; ++t)
                        sk[t][i] = (UH_ROLE_ABLATE & 4) ? (f32x4){0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(x + pix[i] * C + 16 * t + 4 * q);
                }
                f32x4 acc2[T2][NP];
                uh_mlp_core<C, NP, ACT>(xh, xl, w1l, w2l, inv1, alpha, acc2);
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    if (!ok[i] || ((UH_ROLE_ABLATE & 4) && acc2[0][i][0] != 12345.678f)) continue;
#pragma unroll
                    for (int t = 0; t < T2; ++t)
                        *reinterpret_cast<f32x4*>(out + pix[i] * C + 16 * t + 4 * q) = bf_acc_ready(acc2[t][i]) * m4[t] + sk[t][i];
                }
            }
            }
            uh_step_barrier();
        }
    }
}

#ifndef UH_ENC_ROLE_MAP
#define UH_ENC_ROLE_MAP 0
#endif
#ifndef UH_ENC_STAMP
#define UH_ENC_STAMP 0                // 1: timing build -- s_memtime stamps per phase and wave of uh_enc32u_kernel (tools/exp/enc_stamps.py)
#endif
#if UH_ENC_STAMP
__device__ unsigned long long uh_enc_stamps[256 * 12 * 8];
extern "C" int bf_debug_enc_stamps(unsigned long long* host_dst, int clear)
{
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(uh_enc_stamps)) != hipSuccess) return BF_EHIP;
        return hipMemset(p, 0, sizeof(unsigned long long) * 256 * 12 * 8) == hipSuccess ? BF_OK : BF_EHIP;
    }
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(uh_enc_stamps), sizeof(unsigned long long) * 256 * 12 * 8) == hipSuccess ? BF_OK : BF_EHIP;
}
#define UH_STAMP(k)                                                                                      \
    do {                                                                                                 \
        unsigned long long now_;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        stamp_sum[k] += now_ - stamp_prev;                                                               \
        stamp_prev = now_;                                                                               \
    } while (0)
#define UH_STAMP_BEGIN()                                                                                 \
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory")
#define UH_STAMP_END()                                                                                   \
    do {                                                                                                 \
        if (lane == 0 && blockIdx.x < 256)                                                               \
            for (int k_ = 0; k_ < 8; ++k_) uh_enc_stamps[((int)blockIdx.x * 12 + wave) * 8 + k_] += stamp_sum[k_]; \
    } while (0)
#else
#define UH_STAMP(k) do { } while (0)
#define UH_STAMP_BEGIN() do { } while (0)
#define UH_STAMP_END() do { } while (0)
#endif
#ifndef UH_ENC_XCD_ORDER
#define UH_ENC_XCD_ORDER 1
#endif
#ifndef UH_ENC_PD
#define UH_ENC_PD 4                 // input rows a producer requests ahead of the one it multiplies
#endif
// uh_enc32s_kernel with the producer's row walk fully unrolled per tile (see the comment in the producer branch).
// NCW = consumer waves: 4 (512 threads: one producer + one consumer per SIMD) or 8 (768 threads: one producer + TWO consumers
// per SIMD, each taking half of a strip's batch -- a consumer wave alone on its SIMD runs its chain staging read -> split ->
// GEMM1 -> activation / split -> GEMM2 -> store without anything to overlap it with: 25 % of the matrix pipe when timed alone).
template <int K, int ACT, int NCW = 4>
__global__ __launch_bounds__(256 + 64 * NCW, 1) void uh_enc32u_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           const float* __restrict__ dww, const float* __restrict__ gamma, float eps,
                                                           const void* __restrict__ packed, const float* __restrict__ mult, int B, int H,
                                                           int W, float alpha)
{
    constexpr int C = 32, NP = 16 / NCW, RAD = K / 2, T2 = 2, RB = 8, NT = 256 + 64 * NCW;
    static_assert(NCW == 4 || (NCW == 8 && !UH_ENC_ROLE_MAP), "4 or 8 consumer waves");
    constexpr int W_BYTES = 32 * C * C, STG_FLOATS = RB * 8 * UH_STG_PITCH;       // one strip of one batch
    extern __shared__ __attribute__((aligned(16))) char lds[];
    {
        const int4* src = reinterpret_cast<const int4*>(packed);
        int4* dstv = reinterpret_cast<int4*>(lds);
        for (int i = threadIdx.x; i < W_BYTES / 16; i += NT) dstv[i] = src[i];
    }
    const float* aux = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + W_BYTES);
    const float inv1 = aux[0], inv2 = aux[1];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Waves go to the four SIMDs round-robin (wave w -> SIMD w & 3).  Map 0 (default): one wave of each role per SIMD -- the
    // producer's vector work and the consumer's matrix work share a SIMD; map 1 (producers = even waves: a SIMD hosts two waves
    // of ONE role) measured 4.5 % slower (1 002 vs 958 us): complementary pipes beat latency hiding within a role.
    const bool producer = UH_ENC_ROLE_MAP ? !(wave & 1) : wave < 4;
    const int strip = UH_ENC_ROLE_MAP ? wave >> 1 : wave & 3;
    float* stg_base = reinterpret_cast<float*>(lds + W_BYTES) + strip * STG_FLOATS;     // + (step & 1) * 4 * STG_FLOATS
    const int tiles_x = (W + 31) / 32, tiles_y = (H + UH_ENC_ROWS - 1) / UH_ENC_ROWS;
    const int64_t ntiles = (int64_t)B * tiles_y * tiles_x;
    // Tile order (round 4): each XCD (workgroup id mod 8) walks its own contiguous eighth of the tiles, so that the 2-pixel / 2-row halo a
    // tile reads from its neighbours is found in that XCD's L2 (round-robin order: every neighbour lives on another XCD and the halo --
    // 41 % of a tile's reads -- comes from the Infinity Cache / HBM: 1.42 x the algorithmic traffic by the counters)
    const bool xmap = UH_ENC_XCD_ORDER && (gridDim.x & 7) == 0 && ntiles >= (int64_t)gridDim.x;
    const int64_t per_xcd = (ntiles + 7) >> 3, nslots = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int64_t xlim = ntiles - xcd * per_xcd < per_xcd ? (ntiles - xcd * per_xcd > 0 ? ntiles - xcd * per_xcd : 0) : per_xcd;
    const int64_t my_tiles = xmap ? (slot < xlim ? (xlim - slot + nslots - 1) / nslots : 0)
                                  : (blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0);
    auto tile_index = [&](const int64_t ti) -> int64_t { return xmap ? xcd * per_xcd + slot + ti * nslots : blockIdx.x + ti * gridDim.x; };
    const int64_t nsteps = 2 * my_tiles;                   // batches of RB rows; UH_ENC_ROWS / RB = 2 per tile
    static_assert(UH_ENC_ROWS == 2 * RB, "two batches per tile");

    // The two roles are two separate loops (wave-uniform branch) with the same number of barrier arrivals per wave: as one
    // loop hipcc kept the producers' 100 weight registers AND the consumers' matrix state live together (spills).
    if (producer) {
        // ---- depthwise layout: channels 4cl..4cl+3 of strip column pl.  The 16 + 2 RAD input rows of a tile are FULLY
        // UNROLLED: the ring slot of a row (R % PD), the accumulator of an output row (o % K) and which (row, ky) pairs
        // exist at the tile's top / bottom are compile-time facts -- no register rotation (16 moves per row in
        // uh_enc32s_kernel), no arithmetic for the output rows a halo row does not reach (-20 % of the FMAs), no row-mask
        // multiplies, and no loads for rows past the tile's last one (4 of 24 issued there).
        const int cl = lane & 7, pl = lane >> 3;
        constexpr int PD = UH_ENC_PD, ROWF4 = (8 + 2 * RAD) * 8, HALVES = 2 * (ROWF4 - 64), NROWS = UH_ENC_ROWS + 2 * RAD;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        struct RowRegs { f32x4 a; f32x2 b; };
        f32x4 wk[K * K], gm = {1.f, 1.f, 1.f, 1.f}, acc[K];
        RowRegs ring[PD];
        const int h1 = lane & (HALVES - 1);
        f32x4* rowbuf = reinterpret_cast<f32x4*>(lds + W_BYTES + 2 * 4 * STG_FLOATS * 4) + strip * (2 * UH_ROWBUF_F4);
#pragma unroll
        for (int i = 0; i < K * K; ++i) wk[i] = *reinterpret_cast<const f32x4*>(dww + i * C + 4 * cl);
        if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + 4 * cl);
        UH_STAMP_BEGIN();
        for (int64_t ti = 0; ti < my_tiles; ++ti) {
            const int64_t tile = tile_index(ti);
            const int tx = (int)(tile % tiles_x);
            const int ty = (int)((tile / tiles_x) % tiles_y);
            const int64_t pimg = (tile / ((int64_t)tiles_x * tiles_y)) * H * W;
            const int x0 = tx * 32 + strip * 8, py0 = ty * UH_ENC_ROWS;
            const bool plive = x0 < W && !(UH_ROLE_ABLATE & 2);
            const int xg0 = x0 - RAD + (lane >> 3), xg1 = x0 - RAD + 8 + (h1 >> 4);
            const float gm0 = (xg0 >= 0 && xg0 < W) ? 1.f : 0.f, gm1 = (xg1 >= 0 && xg1 < W) ? 1.f : 0.f;
            const int go0 = min(max(xg0, 0), W - 1) * C + 4 * cl, go1 = min(max(xg1, 0), W - 1) * C + 2 * (h1 & 15);
            auto issue = [&](int yy, RowRegs& r) {
                const float* row = x + (pimg + (int64_t)min(max(yy, 0), H - 1) * W) * C;
                r.a = *reinterpret_cast<const f32x4*>(row + go0);
                r.b = *reinterpret_cast<const f32x2*>(row + go1);
            };
            float* stg0 = stg_base + ((2 * ti) & 1) * 4 * STG_FLOATS;           // == stg_base: two batches per tile
            float* stg1 = stg_base + ((2 * ti + 1) & 1) * 4 * STG_FLOATS;
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (plive) {
#pragma unroll
                for (int d = 0; d < PD; ++d) issue(py0 - RAD + d, ring[d]);
            }
            auto rows = [&](auto lo_c, auto hi_c, float* stg, const int out0) {
                constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#pragma unroll
                for (int R = LO; R < HI; ++R) {
                    RowRegs& slot = ring[R % PD];
                    f32x4* rb = rowbuf + (R & 1) * UH_ROWBUF_F4;
                    rb[lane] = slot.a * gm0;
                    if (lane < HALVES) reinterpret_cast<f32x2*>(rb + 64)[lane] = slot.b * gm1;
                    if (R + PD < NROWS) issue(py0 - RAD + R + PD, slot);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int yi = py0 - RAD + R;
                    if (yi >= 0 && yi < H) {                   // wave-uniform: a row outside the image contributes nothing
                        f32x4 v[K];
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) v[kx] = rb[(pl + kx) * 8 + cl];
#pragma unroll
                        for (int ky = 0; ky < K; ++ky) {
                            const int o = R - ky;              // output row (of the tile) this input row feeds through tap row ky
                            if (o >= 0 && o < UH_ENC_ROWS) {
#pragma unroll
                                for (int kx = 0; kx < K; ++kx) acc[o % K] += wk[ky * K + kx] * v[kx];
                            }
                        }
                    }
                    if (R >= K - 1) {
                        const int o = R - (K - 1);
                        f32x4 r = acc[o % K];
                        acc[o % K] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (gamma) {
                            const float mean = uh_pixel_sum8(r[0] + r[1] + r[2] + r[3]) * (1.f / C);
                            const f32x4 d = r - mean;
                            const float var = uh_pixel_sum8(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / C);
                            r = d * (gm * rsqrtf(var + eps));
                        }
                        *reinterpret_cast<f32x4*>(stg + ((o - out0) * 8 + pl) * UH_STG_PITCH + 4 * cl) = r;
                    }
                }
            };
            if (plive && py0 < H) rows(std::integral_constant<int, 0>{}, std::integral_constant<int, RB + 2 * RAD>{}, stg0, 0);
            UH_STAMP(0);
            uh_step_barrier();
            UH_STAMP(1);
            if (plive && py0 + RB < H) rows(std::integral_constant<int, RB + 2 * RAD>{}, std::integral_constant<int, NROWS>{}, stg1, RB);
            UH_STAMP(0);
            uh_step_barrier();
            UH_STAMP(1);
        }
        uh_step_barrier();                                   // the consumers' last step
        UH_STAMP_END();
    } else {
        // ---- matrix-core layout
        const int q = lane >> 4, n = lane & 15;
        const int g0 = NCW == 8 ? NP * ((wave - 4) >> 2) : 0;              // first 16-pixel group of the batch this wave takes
        f32x4 m4[T2];
#pragma unroll
        for (int t = 0; t < T2; ++t) {
            m4[t] = (f32x4){inv2, inv2, inv2, inv2};
            if (mult) m4[t] *= *reinterpret_cast<const f32x4*>(mult + 16 * t + 4 * q);
        }
        // The skip (x at the batch's own pixels) is fetched ONE STEP AHEAD, while the producers are still reading those rows:
        // loaded at the top of the step that consumes it, it cost the consumers 295 of their 791 us (one wave per SIMD keeps
        // too few bytes in flight to hide a memory round trip inside a 5 us step).
        struct TileAt { int64_t img; int x0, y0; };
        auto tile_at = [&](const int64_t ti) {
            const int tile = (int)tile_index(ti);                         // ntiles < 2^31 (checked by the launcher)
            const int tx = tile % tiles_x, rest = tile / tiles_x;
            return TileAt{(int64_t)(rest / tiles_y) * H * W, tx * 32 + strip * 8, (rest % tiles_y) * UH_ENC_ROWS};
        };
        auto pixel_of = [&](const TileAt& t, const int yb, const int i, bool& ok) {
            const int py = yb + 2 * (g0 + i) + (n >> 3), px = t.x0 + (n & 7);
            ok = py < H && px < W;
            return t.img + (int64_t)min(py, H - 1) * W + min(px, W - 1);
        };
        f32x4 skn[T2][NP];
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int i = 0; i < NP; ++i) skn[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        TileAt cur = tile_at(0), prev = cur;
        UH_STAMP_BEGIN();
        for (int64_t step = 0; step <= nsteps; ++step) {
            const int b = (int)(step & 1);
            if (b == 0 && step > 0) { prev = cur; cur = tile_at(step >> 1); }
            f32x4 sk[T2][NP];
#pragma unroll
            for (int t = 0; t < T2; ++t)
#pragma unroll
                for (int i = 0; i < NP; ++i) sk[t][i] = skn[t][i];
            if (step < nsteps && !(UH_ROLE_ABLATE & 5)) {          // the batch the producers work on now: consumed in the next step
                const int ybn = cur.y0 + b * RB;
                if (cur.x0 < W && ybn < H) {
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        bool okn;
                        const int64_t pn = pixel_of(cur, ybn, i, okn);
#pragma unroll
                        for (int t = 0; t < T2; ++t) skn[t][i] = *reinterpret_cast<const f32x4*>(x + pn * C + 16 * t + 4 * q);
                    }
                }
            }
            UH_STAMP(0);
            if (step >= 1) {
                const int64_t cs = step - 1;                    // the batch the producers finished in the previous step
                const float* stg = stg_base + (cs & 1) * 4 * STG_FLOATS;
                const TileAt& tl = b == 0 ? prev : cur;         // cs odd: second batch of the previous tile; even: first of this one
                const int yb = tl.y0 + (int)(cs & 1) * RB;
                if (tl.x0 < W && yb < H && !(UH_ROLE_ABLATE & 1)) {
                    int wl = lane * 16;
                    asm volatile("" : "+v"(wl));
                    const char* w1l = lds + wl;
                    const char* w2l = lds + 16 * C * C + wl;
                    uh8 xh[1][NP], xl[1][NP];
                    int64_t pix[NP];
                    bool ok[NP];
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        const int s = 16 * (g0 + i) + n;      // staged pixel: row 2 (g0 + i) + n / 8, column n % 8
                        const float* sp = stg + s * UH_STG_PITCH + 8 * q;
                        uh_split8(*reinterpret_cast<const f32x4*>(sp), *reinterpret_cast<const f32x4*>(sp + 4), xh[0][i], xl[0][i]);
                        pix[i] = pixel_of(tl, yb, i, ok[i]);
                    }
                    UH_STAMP(1);
                    f32x4 acc2[T2][NP];
                    uh_mlp_core<C, NP, ACT>(xh, xl, w1l, w2l, inv1, alpha, acc2);
                    UH_STAMP(2);
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        if (!ok[i] || ((UH_ROLE_ABLATE & 4) && acc2[0][i][0] != 12345.678f)) continue;
#pragma unroll
                        for (int t = 0; t < T2; ++t)
                            *reinterpret_cast<f32x4*>(out + pix[i] * C + 16 * t + 4 * q) = bf_acc_ready(acc2[t][i]) * m4[t] + sk[t][i];
                    }
                    UH_STAMP(3);
                }
            }
            uh_step_barrier();
            UH_STAMP(4);
        }
        UH_STAMP_END();
    }
}

// uh_enc32u_kernel reshaped for TWO workgroups per CU (four waves per SIMD: two producers + two consumers, so that a stalled
// wave of either role has a sibling to issue): batches of 4 rows instead of 8 (half the staging area), one wave-private row
// buffer per strip (a wave's LDS operations execute in order: the next row's writes queue behind this row's reads), two
// 16-pixel groups per consumer step, and the 25 depthwise weight vectors read from LDS every row instead of living in 100
// registers: 78 976 bytes of LDS and <= 128 registers per wave.
template <int K, int ACT>
__global__ __launch_bounds__(512, 4) void uh_enc32w_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           const float* __restrict__ dww, const float* __restrict__ gamma, float eps,
                                                           const void* __restrict__ packed, const float* __restrict__ mult, int B, int H,
                                                           int W, float alpha)
{
    constexpr int C = 32, NP = 2, RAD = K / 2, T2 = 2, RB = 4, NB = UH_ENC_ROWS / RB;
    constexpr int W_BYTES = 32 * C * C, STG_FLOATS = RB * 8 * UH_STG_PITCH;       // one strip of one batch
    extern __shared__ __attribute__((aligned(16))) char lds[];
    {
        const int4* src = reinterpret_cast<const int4*>(packed);
        int4* dstv = reinterpret_cast<int4*>(lds);
        for (int i = threadIdx.x; i < W_BYTES / 16; i += 512) dstv[i] = src[i];
    }
    constexpr int DW_OFF = W_BYTES + 2 * 4 * STG_FLOATS * 4 + 4 * UH_ROWBUF_F4 * 16;       // depthwise weights [K*K][32] behind the row buffers
    for (int i = threadIdx.x; i < K * K * C; i += 512) reinterpret_cast<float*>(lds + DW_OFF)[i] = dww[i];
    const float* aux = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + W_BYTES);
    const float inv1 = aux[0], inv2 = aux[1];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifndef UH_ENC_ROLE_MAP
#define UH_ENC_ROLE_MAP 0
#endif
    // Waves go to the four SIMDs round-robin (wave w -> SIMD w & 3).  Map 0 (default): one wave of each role per SIMD -- the
    // producer's vector work and the consumer's matrix work share a SIMD; map 1 (producers = even waves: a SIMD hosts two waves
    // of ONE role) measured 4.5 % slower (1 002 vs 958 us): complementary pipes beat latency hiding within a role.
    const bool producer = UH_ENC_ROLE_MAP ? !(wave & 1) : wave < 4;
    const int strip = UH_ENC_ROLE_MAP ? wave >> 1 : wave & 3;
    float* stg_base = reinterpret_cast<float*>(lds + W_BYTES) + strip * STG_FLOATS;     // + (step & 1) * 4 * STG_FLOATS
    const int tiles_x = (W + 31) / 32, tiles_y = (H + UH_ENC_ROWS - 1) / UH_ENC_ROWS;
    const int64_t ntiles = (int64_t)B * tiles_y * tiles_x;
    const int64_t my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int64_t nsteps = NB * my_tiles;                  // batches of RB rows
    static_assert(NB == 4, "four batches per tile below");
    static_assert(NB % 2 == 0, "the staging parity of a batch is its index inside the tile");

    // The two roles are two separate loops (wave-uniform branch) with the same number of barrier arrivals per wave: as one
    // loop hipcc kept the producers' 100 weight registers AND the consumers' matrix state live together (spills).
    if (producer) {
        // ---- depthwise layout: channels 4cl..4cl+3 of strip column pl.  The 16 + 2 RAD input rows of a tile are FULLY
        // UNROLLED: the ring slot of a row (R % PD), the accumulator of an output row (o % K) and which (row, ky) pairs
        // exist at the tile's top / bottom are compile-time facts -- no register rotation (16 moves per row in
        // uh_enc32s_kernel), no arithmetic for the output rows a halo row does not reach (-20 % of the FMAs), no row-mask
        // multiplies, and no loads for rows past the tile's last one (4 of 24 issued there).
        const int cl = lane & 7, pl = lane >> 3;
        constexpr int PD = 4, ROWF4 = (8 + 2 * RAD) * 8, HALVES = 2 * (ROWF4 - 64), NROWS = UH_ENC_ROWS + 2 * RAD;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        struct RowRegs { f32x4 a; f32x2 b; };
        f32x4 gm = {1.f, 1.f, 1.f, 1.f}, acc[K];
        const f32x4* wl = reinterpret_cast<const f32x4*>(lds + DW_OFF) + cl;          // tap t: wl[t * 8]
        RowRegs ring[PD];
        const int h1 = lane & (HALVES - 1);
        f32x4* rowbuf = reinterpret_cast<f32x4*>(lds + W_BYTES + 2 * 4 * STG_FLOATS * 4) + strip * UH_ROWBUF_F4;
        if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + 4 * cl);
        for (int64_t ti = 0; ti < my_tiles; ++ti) {
            const int64_t tile = blockIdx.x + ti * gridDim.x;
            const int tx = (int)(tile % tiles_x);
            const int ty = (int)((tile / tiles_x) % tiles_y);
            const int64_t pimg = (tile / ((int64_t)tiles_x * tiles_y)) * H * W;
            const int x0 = tx * 32 + strip * 8, py0 = ty * UH_ENC_ROWS;
            const bool plive = x0 < W && !(UH_ROLE_ABLATE & 2);
            const int xg0 = x0 - RAD + (lane >> 3), xg1 = x0 - RAD + 8 + (h1 >> 4);
            const float gm0 = (xg0 >= 0 && xg0 < W) ? 1.f : 0.f, gm1 = (xg1 >= 0 && xg1 < W) ? 1.f : 0.f;
            const int go0 = min(max(xg0, 0), W - 1) * C + 4 * cl, go1 = min(max(xg1, 0), W - 1) * C + 2 * (h1 & 15);
            auto issue = [&](int yy, RowRegs& r) {
                const float* row = x + (pimg + (int64_t)min(max(yy, 0), H - 1) * W) * C;
                r.a = *reinterpret_cast<const f32x4*>(row + go0);
                r.b = *reinterpret_cast<const f32x2*>(row + go1);
            };
            float* stg0 = stg_base;                                           // batch b of a tile: parity b & 1 (NB is even)
            float* stg1 = stg_base + 4 * STG_FLOATS;
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (plive) {
#pragma unroll
                for (int d = 0; d < PD; ++d) issue(py0 - RAD + d, ring[d]);
            }
            auto rows = [&](auto lo_c, auto hi_c, float* stg, const int out0) {
                constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#pragma unroll
                for (int R = LO; R < HI; ++R) {
                    RowRegs& slot = ring[R % PD];
                    f32x4* rb = rowbuf;
                    rb[lane] = slot.a * gm0;
                    if (lane < HALVES) reinterpret_cast<f32x2*>(rb + 64)[lane] = slot.b * gm1;
                    if (R + PD < NROWS) issue(py0 - RAD + R + PD, slot);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int yi = py0 - RAD + R;
                    if (yi >= 0 && yi < H) {                   // wave-uniform: a row outside the image contributes nothing
                        f32x4 v[K];
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) v[kx] = rb[(pl + kx) * 8 + cl];
#pragma unroll
                        for (int ky = 0; ky < K; ++ky) {
                            const int o = R - ky;              // output row (of the tile) this input row feeds through tap row ky
                            if (o >= 0 && o < UH_ENC_ROWS) {
#pragma unroll
                                for (int kx = 0; kx < K; ++kx) acc[o % K] += wl[(ky * K + kx) * 8] * v[kx];
                            }
                        }
                    }
                    if (R >= K - 1) {
                        const int o = R - (K - 1);
                        f32x4 r = acc[o % K];
                        acc[o % K] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (gamma) {
                            const float mean = uh_pixel_sum8(r[0] + r[1] + r[2] + r[3]) * (1.f / C);
                            const f32x4 d = r - mean;
                            const float var = uh_pixel_sum8(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / C);
                            r = d * (gm * rsqrtf(var + eps));
                        }
                        *reinterpret_cast<f32x4*>(stg + ((o - out0) * 8 + pl) * UH_STG_PITCH + 4 * cl) = r;
                    }
                }
            };
            constexpr int R0 = RB + 2 * RAD;                                  // input rows of the first batch
            if (plive && py0 < H) rows(std::integral_constant<int, 0>{}, std::integral_constant<int, R0>{}, stg0, 0);
            uh_step_barrier();
            if (plive && py0 + RB < H) rows(std::integral_constant<int, R0>{}, std::integral_constant<int, R0 + RB>{}, stg1, RB);
            uh_step_barrier();
            if (plive && py0 + 2 * RB < H) rows(std::integral_constant<int, R0 + RB>{}, std::integral_constant<int, R0 + 2 * RB>{}, stg0, 2 * RB);
            uh_step_barrier();
            if (plive && py0 + 3 * RB < H) rows(std::integral_constant<int, R0 + 2 * RB>{}, std::integral_constant<int, NROWS>{}, stg1, 3 * RB);
            uh_step_barrier();
        }
        uh_step_barrier();                                   // the consumers' last step
    } else {
        // ---- matrix-core layout
        const int q = lane >> 4, n = lane & 15;
        f32x4 m4[T2];
#pragma unroll
        for (int t = 0; t < T2; ++t) {
            m4[t] = (f32x4){inv2, inv2, inv2, inv2};
            if (mult) m4[t] *= *reinterpret_cast<const f32x4*>(mult + 16 * t + 4 * q);
        }
        // The skip (x at the batch's own pixels) is fetched ONE STEP AHEAD, while the producers are still reading those rows:
        // loaded at the top of the step that consumes it, it cost the consumers 295 of their 791 us (one wave per SIMD keeps
        // too few bytes in flight to hide a memory round trip inside a 5 us step).
        struct TileAt { int64_t img; int x0, y0; };
        auto tile_at = [&](const int64_t ti) {
            const int tile = (int)(blockIdx.x + ti * gridDim.x);          // ntiles < 2^31 (checked by the launcher)
            const int tx = tile % tiles_x, rest = tile / tiles_x;
            return TileAt{(int64_t)(rest / tiles_y) * H * W, tx * 32 + strip * 8, (rest % tiles_y) * UH_ENC_ROWS};
        };
        auto pixel_of = [&](const TileAt& t, const int yb, const int i, bool& ok) {
            const int py = yb + 2 * i + (n >> 3), px = t.x0 + (n & 7);
            ok = py < H && px < W;
            return t.img + (int64_t)min(py, H - 1) * W + min(px, W - 1);
        };
        f32x4 skn[T2][NP];
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int i = 0; i < NP; ++i) skn[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        TileAt cur = tile_at(0), prev = cur;
        for (int64_t step = 0; step <= nsteps; ++step) {
            const int b = (int)(step % NB);
            if (b == 0 && step > 0) { prev = cur; cur = tile_at(step / NB); }
            f32x4 sk[T2][NP];
#pragma unroll
            for (int t = 0; t < T2; ++t)
#pragma unroll
                for (int i = 0; i < NP; ++i) sk[t][i] = skn[t][i];
            if (step < nsteps && !(UH_ROLE_ABLATE & 5)) {          // the batch the producers work on now: consumed in the next step
                const int ybn = cur.y0 + b * RB;
                if (cur.x0 < W && ybn < H) {
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        bool okn;
                        const int64_t pn = pixel_of(cur, ybn, i, okn);
#pragma unroll
                        for (int t = 0; t < T2; ++t) skn[t][i] = *reinterpret_cast<const f32x4*>(x + pn * C + 16 * t + 4 * q);
                    }
                }
            }
            if (step >= 1) {
                const int64_t cs = step - 1;                    // the batch the producers finished in the previous step
                const float* stg = stg_base + (cs & 1) * 4 * STG_FLOATS;
                const TileAt& tl = b == 0 ? prev : cur;         // b == 0: the last batch of the previous tile
                const int yb = tl.y0 + (int)(cs % NB) * RB;
                if (tl.x0 < W && yb < H && !(UH_ROLE_ABLATE & 1)) {
                    int wl = lane * 16;
                    asm volatile("" : "+v"(wl));
                    const char* w1l = lds + wl;
                    const char* w2l = lds + 16 * C * C + wl;
                    uh8 xh[1][NP], xl[1][NP];
                    int64_t pix[NP];
                    bool ok[NP];
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        const int s = 16 * i + n;             // staged pixel: row 2i + n / 8, column n % 8
                        const float* sp = stg + s * UH_STG_PITCH + 8 * q;
                        uh_split8(*reinterpret_cast<const f32x4*>(sp), *reinterpret_cast<const f32x4*>(sp + 4), xh[0][i], xl[0][i]);
                        pix[i] = pixel_of(tl, yb, i, ok[i]);
                    }
                    f32x4 acc2[T2][NP];
                    uh_mlp_core<C, NP, ACT>(xh, xl, w1l, w2l, inv1, alpha, acc2);
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        if (!ok[i] || ((UH_ROLE_ABLATE & 4) && acc2[0][i][0] != 12345.678f)) continue;
#pragma unroll
                        for (int t = 0; t < T2; ++t)
                            *reinterpret_cast<f32x4*>(out + pix[i] * C + 16 * t + 4 * q) = bf_acc_ready(acc2[t][i]) * m4[t] + sk[t][i];
                    }
                }
            }
            uh_step_barrier();
        }
    }
}

static int g_uh_enc_variant = 2;     // 4 = 2 with two consumer waves per SIMD (768 threads), 3 two workgroups per CU, 2 wave-specialised + unrolled producer (default), 1 wave-specialised, 0 one kind of wave
extern "C" int bf_op_set_variant(const char* key, int value)
{
    if (key && !strcmp(key, "enc32")) { g_uh_enc_variant = value < 0 ? 2 : (value > 4 ? 2 : value); return BF_OK; }
    return BF_EINVAL;
}

extern "C" int bf_op_convnext_block_h3(const float* x, float* out, const float* dw, int k, const float* ln_gamma, float eps,
                                       const void* packed, const float* mult, int B, int H, int W, int C, int act, float alpha,
                                       void* stream)
{
    if (!x || !out || !dw || !packed || B <= 0 || H <= 0 || W <= 0) return BF_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)packed | (uintptr_t)mult | (uintptr_t)dw | (uintptr_t)ln_gamma) % 16) return BF_EINVAL;
    if (act == 2 && !(alpha >= 0.f && alpha <= 1.f)) return BF_EINVAL;
    if (C != 32 || (k != 3 && k != 5) || act < 0 || act > 3) return BF_EUNSUPPORTED;
    if (x == out) return BF_EINVAL;                      // neighbouring strips read the halo of this one
    hipStream_t s = (hipStream_t)stream;
    const int64_t ntiles = (int64_t)B * ((H + UH_ENC_ROWS - 1) / UH_ENC_ROWS) * ((W + 31) / 32);
    if (ntiles >= 0x7fffffff) return BF_EUNSUPPORTED;      // 32-bit tile arithmetic in the kernels
    if (g_uh_enc_variant >= 1) {
        constexpr int LDS_S = 32 * 32 * 32 + 2 * 4 * 8 * 8 * UH_STG_PITCH * 4 + 4 * 2 * UH_ROWBUF_F4 * 16;
        const int grid_s = (int)(ntiles < 256 ? ntiles : 256);
#define UH_ENCS(KK, A)                                                                                                         \
    {                                                                                                                          \
        if (g_uh_enc_variant == 3) {                                                                                           \
            constexpr int LDS_W = 32 * 32 * 32 + 2 * 4 * 4 * 8 * UH_STG_PITCH * 4 + 4 * UH_ROWBUF_F4 * 16 + KK * KK * 32 * 4;  \
            const int grid_w = (int)(ntiles < 512 ? ntiles : 512);                                                             \
            if (bf_set_max_lds(reinterpret_cast<const void*>(uh_enc32w_kernel<KK, A>), LDS_W) != hipSuccess) return BF_EHIP;   \
            hipLaunchKernelGGL((uh_enc32w_kernel<KK, A>), dim3(grid_w), dim3(512), LDS_W, s, x, out, dw, ln_gamma, eps, packed, mult, B, \
                               H, W, alpha);                                                                                   \
        } else if (g_uh_enc_variant == 4 && KK == 3) {   /* k = 5 needs 250 registers for its producers: 84 spills at the 168 of 768 threads */ \
            if (bf_set_max_lds(reinterpret_cast<const void*>(uh_enc32u_kernel<3, A, 8>), LDS_S) != hipSuccess) return BF_EHIP;  \
            hipLaunchKernelGGL((uh_enc32u_kernel<3, A, 8>), dim3(grid_s), dim3(768), LDS_S, s, x, out, dw, ln_gamma, eps, packed, mult, B, \
                               H, W, alpha);                                                                                   \
        } else if (g_uh_enc_variant == 2 || g_uh_enc_variant == 4) {                                                                                    \
            if (bf_set_max_lds(reinterpret_cast<const void*>(uh_enc32u_kernel<KK, A>), LDS_S) != hipSuccess) return BF_EHIP;   \
            hipLaunchKernelGGL((uh_enc32u_kernel<KK, A>), dim3(grid_s), dim3(512), LDS_S, s, x, out, dw, ln_gamma, eps, packed, mult, B, \
                               H, W, alpha);                                                                                   \
        } else {                                                                                                               \
            if (bf_set_max_lds(reinterpret_cast<const void*>(uh_enc32s_kernel<KK, A>), LDS_S) != hipSuccess) return BF_EHIP;   \
            hipLaunchKernelGGL((uh_enc32s_kernel<KK, A>), dim3(grid_s), dim3(512), LDS_S, s, x, out, dw, ln_gamma, eps, packed, mult, B, \
                               H, W, alpha);                                                                                   \
        }                                                                                                                      \
    }
#define UH_ENCS_K(KK)                                                                                                          \
    switch (act) {                                                                                                             \
    case 0: UH_ENCS(KK, 0) break;                                                                                              \
    case 1: UH_ENCS(KK, 1) break;                                                                                              \
    case 2: UH_ENCS(KK, 2) break;                                                                                              \
    default: UH_ENCS(KK, 3) break;                                                                                             \
    }
        if (k == 5) { UH_ENCS_K(5) } else { UH_ENCS_K(3) }
#undef UH_ENCS_K
#undef UH_ENCS
        return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
    }
    constexpr int LDS = 32 * 32 * 32 + 4 * 8 * 8 * UH_STG_PITCH * 4;
    const int grid = (int)(ntiles < 512 ? ntiles : 512);
#define UH_ENC(KK, A)                                                                                                          \
    {                                                                                                                          \
        if (bf_set_max_lds(reinterpret_cast<const void*>(uh_enc32_kernel<KK, A>), LDS) != hipSuccess) return BF_EHIP;      /* once per device */ \
        hipLaunchKernelGGL((uh_enc32_kernel<KK, A>), dim3(grid), dim3(256), LDS, s, x, out, dw, ln_gamma, eps, packed, mult, B, H, W, \
                           alpha);                                                                                             \
    }
#define UH_ENC_K(KK)                                                                                                           \
    switch (act) {                                                                                                             \
    case 0: UH_ENC(KK, 0) break;                                                                                               \
    case 1: UH_ENC(KK, 1) break;                                                                                               \
    case 2: UH_ENC(KK, 2) break;                                                                                               \
    default: UH_ENC(KK, 3) break;                                                                                              \
    }
    if (k == 5) { UH_ENC_K(5) } else { UH_ENC_K(3) }
#undef UH_ENC_K
#undef UH_ENC
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

