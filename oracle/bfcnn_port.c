/*
 * CPU port (fp32, OpenMP) of the bfcnn resnet-denoiser inference path -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of DenoiserModule.__call__ (bfcnn/module_denoiser.py:46-75) for the
 * canonical resnet (bfcnn/backbone_resnet.py:250-298, bfcnn/backbone_blocks.py:167-246,
 * bfcnn/model.py:297-342, bfcnn/utilities.py:435-461,736-764).  It exists to time the algorithm
 * on the GPU box's host cores (bench.py `cpu_baseline`, kind "port": the reference's TF/Keras
 * path cannot be run, TensorFlow is not installed) and is itself checked against the fp64
 * NumPy oracle in tests/test_port_vs_oracle.py.  Parity status: as oracle/bfcnn_oracle.py
 * ("parity unpinned" by the reference for conv/BN numerics).  The product path never links it.
 *
 * Flat parameter layout = the engine's / oracle's: base [k,k,cin,16], per block
 * conv0 [3,3,16,16], conv1 [3,3,16,16], gamma[16]; head conv0 [16,hf], conv1 [hf,cout].
 * State: per block moving_mean[16], moving_variance[16].
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define F 16

static int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

/* 3x3 16->16 SAME conv, NHWC, HWIO; epilogue: relu / (scale,shift,+res).
 * Register-blocked over 4 consecutive pixels x 16 output channels (two 8-float vectors per pixel,
 * GCC vector extensions -> AVX2/AVX-512 FMAs under -march=native), rows split over OpenMP threads,
 * so the baseline is a fair multi-core SIMD implementation, not a scalar strawman. */
typedef float v8 __attribute__((vector_size(32), aligned(4)));

static inline void conv_px_block(const float* in, const float* w, int H, int W, int y, int x0, int nx, v8 acc[4][2])
{
    for (int j = 0; j < 4; ++j) { acc[j][0] = (v8){0}; acc[j][1] = (v8){0}; }
    for (int ky = 0; ky < 3; ++ky) {
        const int gy = y + ky - 1;
        if (gy < 0 || gy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const float* wp = w + (size_t)(ky * 3 + kx) * F * F;
            const float* ip[4];
            int ok[4];
            for (int j = 0; j < 4; ++j) {
                const int gx = x0 + j + kx - 1;
                ok[j] = j < nx && gx >= 0 && gx < W;
                ip[j] = in + ((size_t)gy * W + (ok[j] ? gx : 0)) * F;
            }
            if (ok[0] && ok[1] && ok[2] && ok[3]) {
                for (int ci = 0; ci < F; ++ci) {
                    const v8 w0 = *(const v8*)(wp + ci * F), w1 = *(const v8*)(wp + ci * F + 8);
                    for (int j = 0; j < 4; ++j) {
                        const float v = ip[j][ci];
                        acc[j][0] += v * w0;
                        acc[j][1] += v * w1;
                    }
                }
            } else {
                for (int j = 0; j < 4; ++j) {
                    if (!ok[j]) continue;
                    for (int ci = 0; ci < F; ++ci) {
                        const float v = ip[j][ci];
                        acc[j][0] += v * *(const v8*)(wp + ci * F);
                        acc[j][1] += v * *(const v8*)(wp + ci * F + 8);
                    }
                }
            }
        }
    }
}

static void conv3x3(const float* in, const float* w, float* out, int H, int W,
                    int relu, const float* scale, const float* shift, const float* res)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x0 = 0; x0 < W; x0 += 4) {
            const int nx = W - x0 < 4 ? W - x0 : 4;
            v8 acc[4][2];
            conv_px_block(in, w, H, W, y, x0, nx, acc);
            for (int j = 0; j < nx; ++j) {
                float a[F];
                memcpy(a, &acc[j][0], 32);
                memcpy(a + 8, &acc[j][1], 32);
                float* op = out + ((size_t)y * W + x0 + j) * F;
                if (relu) for (int c = 0; c < F; ++c) a[c] = a[c] > 0.f ? a[c] : 0.f;
                if (scale) {
                    const float* rp = res + ((size_t)y * W + x0 + j) * F;
                    for (int c = 0; c < F; ++c) a[c] = rp[c] + a[c] * scale[c] + shift[c];
                }
                memcpy(op, a, sizeof(a));
            }
        }
    }
}

/* one image: u8 [H,W,cin] -> u8 [H,W,cout]; returns 0 on success */
static int forward_image(const float* params, const float* state, int no_layers, int k, int cin, int hf, int cout,
                         float eps, const uint8_t* in, uint8_t* out, int Hs, int Ws, float* b0, float* b1, float* b2)
{
    const int H = next_pow2(Hs), W = next_pow2(Ws), R = k / 2;
    const float* wb = params;
    /* base conv on the normalised, zero-padded (value 0 -> -0.5) image */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float acc[F];
            for (int c = 0; c < F; ++c) acc[c] = 0.f;
            for (int ky = 0; ky < k; ++ky) {
                const int gy = y + ky - R;
                if (gy < 0 || gy >= H) continue;
                for (int kx = 0; kx < k; ++kx) {
                    const int gx = x + kx - R;
                    if (gx < 0 || gx >= W) continue;
                    for (int ci = 0; ci < cin; ++ci) {
                        const float raw = (gy < Hs && gx < Ws) ? (float)in[((size_t)gy * Ws + gx) * cin + ci] : 0.f;
                        const float v = raw / 255.0f - 0.5f;
                        const float* wr = wb + ((size_t)(ky * k + kx) * cin + ci) * F;
                        for (int c = 0; c < F; ++c) acc[c] += v * wr[c];
                    }
                }
            }
            memcpy(b0 + ((size_t)y * W + x) * F, acc, sizeof(acc));
        }
    const float* p = params + (size_t)k * k * cin * F;
    float *cur = b0, *tmp = b1, *nxt = b2;
    for (int i = 0; i < no_layers; ++i) {
        const float *w1 = p, *w2 = p + 2304, *gamma = p + 4608;
        const float *mean = state + i * 32, *var = mean + 16;
        float scale[F], shift[F];
        for (int c = 0; c < F; ++c) {
            scale[c] = gamma[c] / sqrtf(var[c] + eps);
            shift[c] = -scale[c] * mean[c];
        }
        conv3x3(cur, w1, tmp, H, W, 1, NULL, NULL, NULL);
        conv3x3(tmp, w2, nxt, H, W, 0, scale, shift, cur);
        float* t = cur; cur = nxt; nxt = t;
        p += 4608 + 16;
    }
    const float *w0 = p, *w1h = p + (size_t)F * hf;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Hs; ++y)
        for (int x = 0; x < Ws; ++x) {
            const float* f = cur + ((size_t)y * W + x) * F;
            float h0[64];
            for (int j = 0; j < hf; ++j) {
                float s = 0.f;
                for (int c = 0; c < F; ++c) s += f[c] * w0[c * hf + j];
                h0[j] = s;
            }
            for (int o = 0; o < cout; ++o) {
                float s = 0.f;
                for (int j = 0; j < hf; ++j) s += h0[j] * w1h[j * cout + o];
                float v = tanhf(2.0f * s) * 0.51f;
                v = v < -0.5f ? -0.5f : (v > 0.5f ? 0.5f : v);
                v = (v + 0.5f) * 255.0f;
                v = nearbyintf(v);                           /* round-half-even */
                v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
                out[((size_t)y * Ws + x) * cout + o] = (uint8_t)v;
            }
        }
    return 0;
}

void bfcnn_port_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int bfcnn_port_max_threads(void) { return omp_get_max_threads(); }

int bfcnn_port_forward_u8(const float* params, const float* state, int no_layers, int kernel_size, int cin, int hf,
                          int cout, float bn_eps, const uint8_t* in, uint8_t* out, int B, int H, int W)
{
    if (hf > 64 || cin > 4 || cout > 4) return -1;
    const int Hp = next_pow2(H), Wp = next_pow2(W);
    const size_t n = (size_t)Hp * Wp * F;
    float* buf = (float*)malloc(3 * n * sizeof(float));
    if (!buf) return -2;
    for (int b = 0; b < B; ++b)
        forward_image(params, state, no_layers, kernel_size, cin, hf, cout, bn_eps, in + (size_t)b * H * W * cin,
                      out + (size_t)b * H * W * cout, H, W, buf, buf + n, buf + 2 * n);
    free(buf);
    return 0;
}
