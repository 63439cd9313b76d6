// Rows of the information-bottleneck layer (models/temporal.py:104-116, _add_info), one wave per row: shared by the launches that evaluate it
// beside their own work (sea_silu_outer_ib, sea_cond_mlp) and by sea_ib_add.
#pragma once
#include "sea_common.hpp"

// 'linear' / 'fourier' info-bottleneck layers (SeaIbParams.mode 1 / 2): 4 consecutive output columns of one row
__device__ __forceinline__ void ib_simple4(const SeaIbParams& P, float cv, int e0, float (&o)[4]) {
    if (P.mode == 1) {
        float w[4], b[4];
        load4(P.w1 + e0, w);
        load4(P.b1 + e0, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = w[e] * cv + b[e];
    } else {
        const int half = P.E >> 1;   // E % 8 == 0: the 4 columns are all sines or all cosines
        const bool is_cos = e0 >= half;
        float w[4];
        load4(P.w1 + (is_cos ? e0 - half : e0), w);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = cv * w[e] * 2.0f * 3.14159274101257324f;   // ((x @ W) * 2) * pi in fp32, as the reference forms it
            o[e] = is_cos ? cosf(a) : sinf(a);
        }
    }
}

// one row of the information-bottleneck MLP, stored (the wave-per-row body of ib_add_kernel without the add)
__device__ __forceinline__ void ib_store_row(const SeaIbParams& P, float cv, int row, int lane) {
    if (P.mode != 0) {   // block-uniform
        for (int e0 = lane * 4; e0 < P.E; e0 += 256) {
            float o[4];
            ib_simple4(P, cv, e0, o);
            store4(P.X[0] + (int64_t)row * P.ldx + e0, o[0], o[1], o[2], o[3]);
        }
        return;
    }
    const int h = P.h;
    const bool act = lane < h;
    const float pre = act ? P.w1[lane] * cv + P.b1[lane] : 0.f;
    const float mean = wave_sum(pre) / (float)h;
    const float cen = act ? pre - mean : 0.f;
    const float var = wave_sum(cen * cen) / (float)h;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float hid = act ? gelu_erf(cen * rstd * P.lnw[lane] + P.lnb[lane]) : 0.f;
    for (int e0 = lane * 4; e0 < P.E; e0 += 256) {
        float o[4];
        load4(P.b2 + e0, o);
        if ((h & 3) == 0) {   // rows of w2 are whole 16-byte chunks: all of them requested before the first use (one memory round trip, not h)
            for (int k0 = 0; k0 < h; k0 += 4) {
                float w[4][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) load4(P.w2 + (int64_t)(e0 + e) * h + k0, w[e]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float hk = __shfl(hid, k0 + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += w[e][k] * hk;
                }
            }
        } else {
            for (int k = 0; k < h; ++k) {
                const float hk = __shfl(hid, k);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += P.w2[(int64_t)(e0 + e) * h + k] * hk;
            }
        }
        store4(P.X[0] + (int64_t)row * P.ldx + e0, o[0], o[1], o[2], o[3]);
    }
}

