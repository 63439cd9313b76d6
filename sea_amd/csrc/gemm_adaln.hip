// AdaLN without its modulation matrix (gfx950, bf16): sea_gemm_adaln.
//
//     [w | b] = h . W2^T + b2,   h = silu(cond_mlp.0(c))                       (models/base_blocks.py:337-339, 344)
//     y       = xhat * (gamma + 1 + w) + (beta + b),   xhat = (x - mean) / sqrt(var_biased + eps)      (:345-350)
//
// The reference — and rounds 1-3 here — evaluate the condition MLP for every module into a [M, 2 d] modulation matrix and read it back in the normalisation:
// at 8 trajectories that is 174 MB written and read again.  Here the normalisation IS the epilogue of cond_mlp.2's GEMM: a 128-row tile
// takes 64 columns of the scale half and the SAME 64 columns of the shift half of W2 (GemmMainloop::nsplit), the four waves are stacked 4 x 1 so that a lane holds w
// and b of its (row, column) in two accumulator blocks of its own, the row statistics of the tile's 128 rows are computed by the tile itself while its first
// operand tiles are in flight (a wave per 32 rows: one 16-byte load per lane and row, two wave reductions), and what leaves the launch is y.  Groups without X
// are plain cond_mlp.2 GEMMs (the modulation matrix is stored: ln_cross, whose normalisation is another launch's epilogue) — one launch carries both kinds.
#include "gemm_core.hpp"
#include <stdlib.h>

struct AdalnLaunch {
    SeaAdalnGroup g[SEA_MAX_ADALN_GROUPS];
    int tile_start[SEA_MAX_ADALN_GROUPS + 1];
    int n_groups;
    float eps;
};

// mean / rstd of ROWS (32 or 16) rows by one wave (two-pass, fp32: the arithmetic of rownorm_kernel): P passes of 256 columns per row, RB rows requested together — the
// launch is short (a round or two of tiles), so what this costs is its dependent memory round trips: 32 / RB of them (16 rows at d <= 256: two).
template <int P, int RB, int ROWS>
__device__ __forceinline__ void adaln_row_stats(const float* X0, int ldx, int rows_left, int d, float eps, float* st, int lane) {
    const float inv_d = 1.0f / (float)d;
    for (int rr = 0; rr < ROWS; rr += RB) {
        float xv[RB][P][4], s[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            int row = rr + u;
            row = row < rows_left ? row : rows_left - 1;
            const float* xp = X0 + (int64_t)row * ldx;
            s[u] = 0.f;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int c = p * 256 + lane * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[u][p][e] = 0.f;
                if (c < d) load4(xp + c, xv[u][p]);
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
#pragma unroll
            for (int p = 0; p < P; ++p) s[u] += (xv[u][p][0] + xv[u][p][1]) + (xv[u][p][2] + xv[u][p][3]);
            const float mean = wave_sum_xor(s[u], lane) * inv_d;
            float q = 0.f;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (p * 256 + lane * 4 < d) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float cc = xv[u][p][e] - mean;
                        q = fma1(cc, cc, q);
                    }
                }
            }
            const float var = wave_sum_xor(q, lane) * inv_d;
            if (lane == 0) {
                st[(rr + u) * 2] = mean;
                st[(rr + u) * 2 + 1] = 1.0f / sqrtf(var + eps);
            }
        }
    }
}

// BM = 128: long launches (several rounds of tiles); BM = 64: short ones — at one trajectory 288 tiles of 128 rows are one workgroup on most CUs and two on 32 of
// them, 576 tiles of 64 rows (three resident per CU) even out
template <typename T, int BM>
__global__ __launch_bounds__(256) void gemm_adaln_kernel(const AdalnLaunch L) {
    constexpr int BN = 128, WR = BM / 4;   // WR: rows of a wave
    using C = GemmCfg<T, BM, BN, 4>;
    static_assert(C::MI == WR / 16 && C::NI == 8, "a wave owns WR rows x the tile's 128 columns");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int gi = 0;
    while (gi + 1 < L.n_groups && bid >= L.tile_start[gi + 1]) ++gi;
    const SeaAdalnGroup& G = L.g[gi];
    const bool norm = G.X != nullptr;                    // block-uniform
    const int d = G.d, N = 2 * d;
    const int t = bid - L.tile_start[gi];
    const int tiles_n = norm ? (d + 63) / 64 : (N + BN - 1) / BN;
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;

    GemmMainloop<T, BM, BN, 4> ml;
    ml.A = static_cast<const T*>(G.A);
    ml.W = static_cast<const T*>(G.W);
    ml.a_seg_stride = 0;
    ml.lda = G.lda; ml.ldw = G.ldw; ml.M = G.M; ml.N = N; ml.K = G.K; ml.n_seg = 1;
    ml.m0 = tm * BM;
    ml.n0 = norm ? tn * 64 : tn * BN;
    ml.nsplit = norm ? d : 0;
    // ---- row statistics of this wave's 32 rows (two-pass, fp32: the arithmetic of rownorm_kernel), requested before the main loop
    float* stats = reinterpret_cast<float*>(smem + C::LDS_BYTES);   // [BM rows][mean, rstd]
    if (norm) {   // block-uniform
        const int w0 = ml.m0 + wave * WR;                                   // (a wave past the last row of the group re-reads that row; its statistics are never used)
        const float* X0 = G.X + (int64_t)(w0 < G.M ? w0 : G.M - 1) * G.ldx;
        const int rows_left = w0 < G.M ? G.M - w0 : 1;
        if (d <= 256) adaln_row_stats<1, 16, WR>(X0, G.ldx, rows_left, d, L.eps, stats + wave * WR * 2, lane);
        else if (d <= 512) adaln_row_stats<2, 8, WR>(X0, G.ldx, rows_left, d, L.eps, stats + wave * WR * 2, lane);
        else adaln_row_stats<4, 4, WR>(X0, G.ldx, rows_left, d, L.eps, stats + wave * WR * 2, lane);
    }
    // epilogue operands of a normalising tile, requested before the main loop (a short launch: behind it they are one more exposed round trip)
    float psc[4][4], psh[4][4], pxv[C::MI][4][4];   // gamma + 1 + bias_w, beta + bias_b (folded as they arrive: two register blocks instead of four), x
    if (norm) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ml.n0 + j * 16 + g * 4;
            const bool nok = n < d;
#pragma unroll
            for (int q = 0; q < 4; ++q) psc[j][q] = psh[j][q] = 0.f;
            if (nok) {
                float t4[4];
                load4(G.gamma + n, psc[j]);
                if (G.beta != nullptr) load4(G.beta + n, psh[j]);
                if (G.bias != nullptr) {
                    load4(G.bias + n, t4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) psc[j][q] += t4[q];
                    load4(G.bias + d + n, t4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) psh[j][q] += t4[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) psc[j][q] += 1.0f;
            }
#pragma unroll
            for (int i = 0; i < C::MI; ++i) {
                int m = ml.m0 + wave * WR + i * 16 + r;
                m = m < G.M ? m : G.M - 1;
#pragma unroll
                for (int q = 0; q < 4; ++q) pxv[i][j][q] = 0.f;
                if (nok) load4(G.X + (int64_t)m * G.ldx + n, pxv[i][j]);
            }
        }
    }
    f32x4 acc[C::MI][C::NI];
    ml.run(smem, acc);
    // (each wave reads only the statistics it wrote itself: rows wave * 32 ..: no barrier needed beyond the main loop's)
    const float* bias = G.bias;
    if (!norm) {
        T* Y = static_cast<T*>(G.Yact);
#pragma unroll
        for (int j = 0; j < C::NI; ++j) {
            const int n = ml.n0 + j * 16 + g * 4;
            if (n >= N) continue;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr) load4(bias + n, bv);
#pragma unroll
            for (int i = 0; i < C::MI; ++i) {
                const int m = ml.m0 + wave * WR + i * 16 + r;
                if (m < G.M) store4(Y + (int64_t)m * G.ldyact + n, acc[i][j][0] + bv[0], acc[i][j][1] + bv[1], acc[i][j][2] + bv[2], acc[i][j][3] + bv[3]);
            }
        }
        return;
    }
    float mean[C::MI], rstd[C::MI];
#pragma unroll
    for (int i = 0; i < C::MI; ++i) {
        mean[i] = stats[(wave * WR + i * 16 + r) * 2];
        rstd[i] = stats[(wave * WR + i * 16 + r) * 2 + 1];
    }
#pragma unroll
    for (int j = 0; j < C::NI / 2; ++j) {
        const int n = ml.n0 + j * 16 + g * 4;   // output column; the scale accumulators are blocks j, the shift accumulators blocks j + 4
        if (n >= d) continue;
#pragma unroll
        for (int i = 0; i < C::MI; ++i) {
            const int m = ml.m0 + wave * WR + i * 16 + r;
            if (m >= G.M) continue;
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (pxv[i][j][q] - mean[i]) * rstd[i] * (psc[j][q] + acc[i][j][q]) + (psh[j][q] + acc[i][j + 4][q]);
            if (G.Yact != nullptr) store4(static_cast<T*>(G.Yact) + (int64_t)m * G.ldyact + n, o[0], o[1], o[2], o[3]);
            if (G.Y32 != nullptr) store4(G.Y32 + (int64_t)m * G.ldy32 + n, o[0], o[1], o[2], o[3]);
        }
    }
    if (tn == 0 && g == 0) {
#pragma unroll
        for (int i = 0; i < C::MI; ++i) {
            const int m = ml.m0 + wave * WR + i * 16 + r;
            if (m < G.M) {
                if (G.mean != nullptr) G.mean[m] = mean[i];
                if (G.rstd != nullptr) G.rstd[m] = rstd[i];
            }
        }
    }
}

extern "C" int sea_gemm_adaln(const SeaAdalnGroup* groups, int n_groups, float eps, int dtype, void* stream) {
    SEA_REQUIRE(groups != nullptr && n_groups >= 1 && n_groups <= SEA_MAX_ADALN_GROUPS, "sea_gemm_adaln: n_groups=%d out of range", n_groups);
    if (dtype != SEA_BF16) {
        sea_set_error("sea_gemm_adaln: bf16 only (dtype=%d)", dtype);
        return SEA_EUNSUPPORTED;
    }
    AdalnLaunch L;
    memset(&L, 0, sizeof(L));
    for (int i = 0; i < n_groups; ++i) {
        const SeaAdalnGroup& G = groups[i];
        SEA_REQUIRE(G.A && G.W && G.M >= 1 && G.d >= 16 && G.d % 8 == 0 && G.d <= 1024 && G.K >= 8 && G.K % 8 == 0 && G.lda % 8 == 0 && G.lda >= G.K && G.ldw % 8 == 0 && G.ldw >= G.K,
                    "sea_gemm_adaln[%d]: null operand or bad shape M=%d d=%d K=%d lda=%d ldw=%d", i, G.M, G.d, G.K, G.lda, G.ldw);
        SEA_REQUIRE(sea_aligned16(G.A) && sea_aligned16(G.W) && sea_aligned16(G.bias) && sea_aligned16(G.X) && sea_aligned16(G.gamma) && sea_aligned16(G.beta) && sea_aligned16(G.Yact) &&
                        sea_aligned16(G.Y32), "sea_gemm_adaln[%d]: pointers must be 16-byte aligned", i);
        if (G.X != nullptr)
            SEA_REQUIRE(G.gamma && (G.Yact || G.Y32) && G.ldx % 4 == 0 && G.ldx >= G.d && (!G.Yact || (G.ldyact % 4 == 0 && G.ldyact >= G.d)) && (!G.Y32 || (G.ldy32 % 4 == 0 && G.ldy32 >= G.d)),
                        "sea_gemm_adaln[%d]: normalising group: gamma, an output and row strides >= d", i);
        else
            SEA_REQUIRE(G.Yact && G.ldyact % 4 == 0 && G.ldyact >= 2 * G.d, "sea_gemm_adaln[%d]: plain group: Yact [M, 2 d]", i);
        L.g[i] = G;
    }
    // tile height: 128 rows unless that leaves the chip under two rounds of tiles
    int t128 = 0;
    for (int i = 0; i < n_groups; ++i) t128 += ((groups[i].M + 127) / 128) * (groups[i].X != nullptr ? (groups[i].d + 63) / 64 : (2 * groups[i].d + 127) / 128);
    const int bm = t128 >= 2 * 256 /* CUs of an MI355X */ ? 128 : 64;
    int total = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaAdalnGroup& G = groups[i];
        L.tile_start[i] = total;
        total += ((G.M + bm - 1) / bm) * (G.X != nullptr ? (G.d + 63) / 64 : (2 * G.d + 127) / 128);
    }
    L.tile_start[n_groups] = total;
    L.n_groups = n_groups;
    L.eps = eps;
    if (bm == 128) {
        constexpr int lds = GemmCfg<__bf16, 128, 128, 4>::LDS_BYTES + 128 * 2 * 4;
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_adaln_kernel<__bf16, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)once;
        gemm_adaln_kernel<__bf16, 128><<<dim3(total), dim3(256), lds, static_cast<hipStream_t>(stream)>>>(L);
    } else {
        constexpr int lds = GemmCfg<__bf16, 64, 128, 4>::LDS_BYTES + 64 * 2 * 4;
        gemm_adaln_kernel<__bf16, 64><<<dim3(total), dim3(256), lds, static_cast<hipStream_t>(stream)>>>(L);
    }
    SEA_CHECK_LAUNCH("sea_gemm_adaln");
    return SEA_OK;
}
