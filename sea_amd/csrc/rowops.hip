// Row-wise kernels of the SEA temporal path (gfx950): sea_rownorm, sea_silu_outer, sea_ib_add, sea_convert_f32_to_act.
// All of them are HBM/L2-bandwidth-bound element/row passes: 16-byte vector accesses, one wave per row, statistics in
// fp32 with the two-pass (mean, then centred variance) form the reference uses.
#include "sea_common.hpp"
#include "ib_rows.hpp"

#define SEA_MAX_NORM_GROUPS 16

struct NormLaunch {
    SeaNormGroup g[SEA_MAX_NORM_GROUPS];
    int M, d, gelu;
    float eps;
};

// One wave per row, 4 rows per workgroup, grid = (ceil(M/4), n_groups).
// KMAX > 0 (d <= 256*KMAX): the row is read from memory ONCE and kept in registers (lane owns columns lane*4 + 256k); for
// KMAX <= 2 the gain/shift/modulation loads are issued together with it, so the whole row costs one memory round trip instead of
// three dependent ones (these launches are latency-bound at M = 2024).  KMAX == 0: any d, three passes over the (cache-hot) row.
template <typename T, bool X_IS_ACT, int KMAX>
__global__ __launch_bounds__(256) void rownorm_kernel(const NormLaunch L) {
    const SeaNormGroup& G = L.g[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= L.M) return;
    const int d = L.d;
    using XT = typename std::conditional<X_IS_ACT, T, float>::type;
    const XT* x = static_cast<const XT*>(G.X) + (int64_t)row * G.ldx;
    const T* mod = G.mod != nullptr ? static_cast<const T*>(G.mod) + (int64_t)row * G.ldmod : nullptr;
    float* y32 = G.Y32 != nullptr ? G.Y32 + (int64_t)row * G.ldy32 : nullptr;
    T* yact = G.Yact != nullptr ? static_cast<T*>(G.Yact) + (int64_t)row * G.ldyact : nullptr;
    const float inv_d = 1.0f / (float)d;

    auto finish = [&](int i, const float (&v)[4], float (&gm)[4], float (&bt)[4], const float (&mw)[4], const float (&mb)[4], float mean, float rstd) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gq = mod != nullptr ? gm[e] + 1.0f + mw[e] : gm[e];
            const float bq = mod != nullptr ? bt[e] + mb[e] : bt[e];
            o[e] = (v[e] - mean) * rstd * gq + bq;
            if (L.gelu) o[e] = y32 != nullptr ? gelu_erf(o[e]) : gelu_for<T>(o[e]);   // (an output that only exists in bf16 takes the polynomial form)
        }
        if (y32 != nullptr) store4(y32 + i, o[0], o[1], o[2], o[3]);
        if (yact != nullptr) store4(yact + i, o[0], o[1], o[2], o[3]);
    };

    if constexpr (KMAX > 0) {
        constexpr bool PRE = KMAX <= 2;
        constexpr int KP = PRE ? KMAX : 1;
        float xv[KMAX][4], gm[KP][4], bt[KP][4], mw[KP][4], mb[KP][4];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = lane * 4 + 256 * k;
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[k][e] = 0.f;
            if (i < d) load4(x + i, xv[k]);
            if constexpr (!X_IS_ACT) {
                if (G.addend != nullptr && i < d) {   // x' = x + addend (info-bottleneck term), written back for the residual consumers
                    float av[4];
                    load4(G.addend + (int64_t)row * G.ldadd + i, av);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[k][e] += av[e];
                    if (G.Xout != nullptr) store4(G.Xout + (int64_t)row * G.ldxout + i, xv[k][0], xv[k][1], xv[k][2], xv[k][3]);
                }
            }
            if constexpr (PRE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gm[k][e] = bt[k][e] = mw[k][e] = mb[k][e] = 0.f;
                if (i < d) {
                    load4(G.gamma + i, gm[k]);
                    if (G.beta != nullptr) load4(G.beta + i, bt[k]);
                    if (mod != nullptr) {
                        load4(mod + i, mw[k]);
                        load4(mod + d + i, mb[k]);
                    }
                }
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) sum += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);  // columns >= d hold zeros
        const float mean = wave_sum(sum) * inv_d;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (lane * 4 + 256 * k < d) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = xv[k][e] - mean;
                    sq += c * c;
                }
            }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + L.eps);
        if (lane == 0) {
            if (G.mean != nullptr) G.mean[row] = mean;
            if (G.rstd != nullptr) G.rstd[row] = rstd;
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = lane * 4 + 256 * k;
            if (i < d) {
                if constexpr (PRE) {
                    finish(i, xv[k], gm[k], bt[k], mw[k], mb[k], mean, rstd);
                } else {
                    float g1[4], b1[4] = {0.f, 0.f, 0.f, 0.f}, w1[4] = {0.f, 0.f, 0.f, 0.f}, m1[4] = {0.f, 0.f, 0.f, 0.f};
                    load4(G.gamma + i, g1);
                    if (G.beta != nullptr) load4(G.beta + i, b1);
                    if (mod != nullptr) {
                        load4(mod + i, w1);
                        load4(mod + d + i, m1);
                    }
                    finish(i, xv[k], g1, b1, w1, m1, mean, rstd);
                }
            }
        }
    } else {
        float sum = 0.f;
        for (int i = lane * 4; i < d; i += 256) {
            float v[4];
            load4(x + i, v);
            sum += (v[0] + v[1]) + (v[2] + v[3]);
        }
        const float mean = wave_sum(sum) * inv_d;
        float sq = 0.f;
        for (int i = lane * 4; i < d; i += 256) {
            float v[4];
            load4(x + i, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = v[e] - mean;
                sq += c * c;
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + L.eps);
        if (lane == 0) {
            if (G.mean != nullptr) G.mean[row] = mean;
            if (G.rstd != nullptr) G.rstd[row] = rstd;
        }
        for (int i = lane * 4; i < d; i += 256) {
            float v[4], g1[4], b1[4] = {0.f, 0.f, 0.f, 0.f}, w1[4] = {0.f, 0.f, 0.f, 0.f}, m1[4] = {0.f, 0.f, 0.f, 0.f};
            load4(x + i, v);
            load4(G.gamma + i, g1);
            if (G.beta != nullptr) load4(G.beta + i, b1);
            if (mod != nullptr) {
                load4(mod + i, w1);
                load4(mod + d + i, m1);
            }
            finish(i, v, g1, b1, w1, m1, mean, rstd);
        }
    }
}

// A FEW LONG rows (the KV-cache step of the shipped widths: one row of 8192 / 16384 hidden values per field): a wave per row walks such a row in three
// dependent passes of d / 256 iterations (78 us for 2 rows of 16384 at the multiphase width).  Here a workgroup of 1024 threads owns a row: every thread
// keeps its d / 4096 pieces of 4 columns in registers (one memory round trip), the two statistics cross the 16 waves through LDS.  d <= 32768.
// KM <= 4 (d <= 16384: a KV-cache step's rows at the shipped widths): the gains, shifts and modulations of the thread's pieces are requested WITH the row, so the launch is
// one memory round trip, not two (the wave-per-row kernel below takes 8 us for 2 rows of 2048, three dependent passes of 8 pieces).
template <typename T, bool X_IS_ACT, int KM>
__global__ __launch_bounds__(1024) void rownorm_fewrows_kernel(const NormLaunch L) {
    constexpr int NT = 1024;
    __shared__ float red[2][NT / 64];
    const SeaNormGroup& G = L.g[blockIdx.y];
    const int row = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = L.d;
    using XT = typename std::conditional<X_IS_ACT, T, float>::type;
    const XT* x = static_cast<const XT*>(G.X) + (int64_t)row * G.ldx;
    const T* mod = G.mod != nullptr ? static_cast<const T*>(G.mod) + (int64_t)row * G.ldmod : nullptr;
    float* y32 = G.Y32 != nullptr ? G.Y32 + (int64_t)row * G.ldy32 : nullptr;
    T* yact = G.Yact != nullptr ? static_cast<T*>(G.Yact) + (int64_t)row * G.ldyact : nullptr;
    const float inv_d = 1.0f / (float)d;
    float xv[KM][4];
    constexpr bool PRE = KM <= 4;            // gains / shifts / modulations of the thread's pieces requested up front, with the row
    constexpr int KP = PRE ? KM : 1;
    float pg[KP][4], pb[KP][4], pw[KP][4], pm[KP][4];
    if constexpr (PRE) {
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int i = tid * 4 + NT * 4 * k;
#pragma unroll
            for (int e = 0; e < 4; ++e) pg[k][e] = pb[k][e] = pw[k][e] = pm[k][e] = 0.f;
            if (i < d) {
                load4(G.gamma + i, pg[k]);
                if (G.beta != nullptr) load4(G.beta + i, pb[k]);
                if (mod != nullptr) {
                    load4(mod + i, pw[k]);
                    load4(mod + d + i, pm[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        const int i = tid * 4 + NT * 4 * k;
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[k][e] = 0.f;
        if (i < d) {
            load4(x + i, xv[k]);
            if constexpr (!X_IS_ACT) {
                if (G.addend != nullptr) {
                    float av[4];
                    load4(G.addend + (int64_t)row * G.ldadd + i, av);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[k][e] += av[e];
                    if (G.Xout != nullptr) store4(G.Xout + (int64_t)row * G.ldxout + i, xv[k][0], xv[k][1], xv[k][2], xv[k][3]);
                }
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k) sum += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);   // columns >= d hold zeros
    sum = wave_sum_xor(sum, lane);
    if (lane == 0) red[0][wave] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) tot += red[0][w];
    const float mean = tot * inv_d;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k)
        if (tid * 4 + NT * 4 * k < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = xv[k][e] - mean;
                sq += c * c;
            }
        }
    sq = wave_sum_xor(sq, lane);
    if (lane == 0) red[1][wave] = sq;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) tot += red[1][w];
    const float rstd = 1.0f / sqrtf(tot * inv_d + L.eps);
    if (tid == 0) {
        if (G.mean != nullptr) G.mean[row] = mean;
        if (G.rstd != nullptr) G.rstd[row] = rstd;
    }
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        const int i = tid * 4 + NT * 4 * k;
        if (i < d) {
            float g1[4], b1[4] = {0.f, 0.f, 0.f, 0.f}, w1[4] = {0.f, 0.f, 0.f, 0.f}, m1[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
            if constexpr (PRE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) g1[e] = pg[k][e], b1[e] = pb[k][e], w1[e] = pw[k][e], m1[e] = pm[k][e];
            } else {
                load4(G.gamma + i, g1);
                if (G.beta != nullptr) load4(G.beta + i, b1);
                if (mod != nullptr) {
                    load4(mod + i, w1);
                    load4(mod + d + i, m1);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gq = mod != nullptr ? g1[e] + 1.0f + w1[e] : g1[e];
                const float bq = mod != nullptr ? b1[e] + m1[e] : b1[e];
                o[e] = (xv[k][e] - mean) * rstd * gq + bq;
                if (L.gelu) o[e] = y32 != nullptr ? gelu_erf(o[e]) : gelu_for<T>(o[e]);
            }
            if (y32 != nullptr) store4(y32 + i, o[0], o[1], o[2], o[3]);
            if (yact != nullptr) store4(yact + i, o[0], o[1], o[2], o[3]);
        }
    }
}

// MANY wide bf16 rows -> bf16 rows (the MLP's nn.LayerNorm(S) + GELU over the [M, S] hidden matrix: models/base_blocks.py:23-24; no modulation, no
// addend): a workgroup of NT threads per row, a thread owns KCH pieces of 8 consecutive columns — ONE 16-byte load and ONE 16-byte store per piece (the
// wave-per-row kernel above moves 8 bytes per lane per access and ran this pass at 3.3 TB/s) —, gain and shift of its columns stay in registers for all
// its rows, and the NEXT row's pieces are requested before the current row's statistics (two barriers) and activations.  S = 2048 / 4096 / 8192 / 16384 on
// 256 / 512 / 1024 / 1024 threads with 1 / 1 / 1 / 2 pieces per thread.
template <int NT, int KCH, bool GELU>
__global__ __launch_bounds__(NT) void ln_rows_bf16_kernel(const NormLaunch L) {
    using T = __bf16;
    constexpr int NW = NT / 64;
    __shared__ float red[2][2][NW];   // [parity of the row][sum | sum of squares][wave]: two rows' statistics never share a slot
    const SeaNormGroup& G = L.g[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = L.d;
    const float inv_d = 1.0f / (float)d;
    float gm[KCH][8], bt[KCH][8], xv[KCH][8], xn[KCH][8];
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
        const int i = tid * 8 + NT * 8 * k;
#pragma unroll
        for (int e = 0; e < 8; ++e) gm[k][e] = bt[k][e] = xv[k][e] = xn[k][e] = 0.f;
        if (i < d) {
            load8(G.gamma + i, gm[k]);
            if (G.beta != nullptr) load8(G.beta + i, bt[k]);
        }
    }
    const T* X = static_cast<const T*>(G.X);
    T* Y = static_cast<T*>(G.Yact);
    int row = blockIdx.x;
    if (row < L.M) {
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int i = tid * 8 + NT * 8 * k;
            if (i < d) load8(X + (int64_t)row * G.ldx + i, xv[k]);
        }
    }
    int par = 0;
    for (; row < L.M; row += gridDim.x, par ^= 1) {
        const int nxt = row + gridDim.x;
        if (nxt < L.M) {
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                const int i = tid * 8 + NT * 8 * k;
                if (i < d) load8(X + (int64_t)nxt * G.ldx + i, xn[k]);
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < KCH; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += xv[k][e];   // columns >= d hold zeros
        sum = wave_sum(sum);
        if (lane == 0) red[par][0][wave] = sum;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[par][0][w];
        const float mean = tot * inv_d;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < KCH; ++k)
            if (tid * 8 + NT * 8 * k < d) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float c = xv[k][e] - mean;
                    sq += c * c;
                }
            }
        sq = wave_sum(sq);
        if (lane == 0) red[par][1][wave] = sq;
        __syncthreads();
        tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[par][1][w];
        const float rstd = 1.0f / sqrtf(tot * inv_d + L.eps);
        if (tid == 0) {
            if (G.mean != nullptr) G.mean[row] = mean;
            if (G.rstd != nullptr) G.rstd[row] = rstd;
        }
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int i = tid * 8 + NT * 8 * k;
            if (i < d) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = (xv[k][e] - mean) * rstd * gm[k][e] + bt[k][e];
                    if (GELU) o[e] = gelu_for<T>(o[e]);
                }
                store8(Y + (int64_t)row * G.ldyact + i, o);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[k][e] = xn[k][e];
        }
    }
}

extern "C" int sea_rownorm(const SeaNormGroup* groups, int n_groups, int M, int d, int x_is_act, int gelu, float eps,
                           int dtype, void* stream) {
    SEA_REQUIRE(groups != nullptr && n_groups >= 1 && n_groups <= SEA_MAX_NORM_GROUPS, "sea_rownorm: n_groups=%d", n_groups);
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_rownorm: bad dtype %d", dtype);
    SEA_REQUIRE(M >= 1 && d >= 4 && d % 4 == 0, "sea_rownorm: bad M=%d d=%d (d must be a multiple of 4)", M, d);
    NormLaunch L;
    memset(&L, 0, sizeof(L));
    for (int i = 0; i < n_groups; ++i) {
        const SeaNormGroup& G = groups[i];
        SEA_REQUIRE(G.X && G.gamma && (G.Y32 || G.Yact), "sea_rownorm[%d]: null pointer", i);
        SEA_REQUIRE(!G.addend || (!x_is_act && d <= 2048 && G.ldadd % 4 == 0 && G.ldadd >= d && sea_aligned16(G.addend) && sea_aligned16(G.Xout) &&
                                  (!G.Xout || (G.ldxout % 4 == 0 && G.ldxout >= d))), "sea_rownorm[%d]: addend needs f32 x, d <= 2048, aligned rows", i);
        SEA_REQUIRE(G.addend || !G.Xout, "sea_rownorm[%d]: Xout without addend", i);
        SEA_REQUIRE(G.ldx % 4 == 0 && G.ldx >= d, "sea_rownorm[%d]: bad ldx=%d", i, G.ldx);
        SEA_REQUIRE(!G.mod || (G.ldmod % 4 == 0 && G.ldmod >= 2 * d), "sea_rownorm[%d]: bad ldmod=%d", i, G.ldmod);
        SEA_REQUIRE(!G.Y32 || (G.ldy32 % 4 == 0 && G.ldy32 >= d), "sea_rownorm[%d]: bad ldy32=%d", i, G.ldy32);
        SEA_REQUIRE(!G.Yact || (G.ldyact % 4 == 0 && G.ldyact >= d), "sea_rownorm[%d]: bad ldyact=%d", i, G.ldyact);
        SEA_REQUIRE(sea_aligned16(G.X) && sea_aligned16(G.gamma) && sea_aligned16(G.beta) && sea_aligned16(G.mod) &&
                        sea_aligned16(G.Y32) && sea_aligned16(G.Yact), "sea_rownorm[%d]: pointers must be 16-byte aligned", i);
        L.g[i] = G;
    }
    L.M = M; L.d = d; L.gelu = gelu; L.eps = eps;
    const dim3 grid((M + 3) / 4, n_groups), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (M <= 32 && d >= 1024 && d <= 32768) {   // a few long rows: a workgroup per row
        const dim3 gridw(M, n_groups), blockw(1024);
        if (d <= 4096) {
            if (dtype == SEA_BF16) {
                if (x_is_act) rownorm_fewrows_kernel<__bf16, true, 1><<<gridw, blockw, 0, s>>>(L);
                else rownorm_fewrows_kernel<__bf16, false, 1><<<gridw, blockw, 0, s>>>(L);
            } else {
                rownorm_fewrows_kernel<float, false, 1><<<gridw, blockw, 0, s>>>(L);
            }
        } else if (d <= 16384 && dtype == SEA_BF16 && x_is_act) {   // (the MLP's hidden rows of a KV-cache step at the shipped widths: 8192 / 16384 columns)
            rownorm_fewrows_kernel<__bf16, true, 4><<<gridw, blockw, 0, s>>>(L);
        } else if (dtype == SEA_BF16) {
            if (x_is_act) rownorm_fewrows_kernel<__bf16, true, 8><<<gridw, blockw, 0, s>>>(L);
            else rownorm_fewrows_kernel<__bf16, false, 8><<<gridw, blockw, 0, s>>>(L);
        } else {
            rownorm_fewrows_kernel<float, false, 8><<<gridw, blockw, 0, s>>>(L);
        }
        SEA_CHECK_LAUNCH("sea_rownorm");
        return SEA_OK;
    }
    {   // many wide bf16 -> bf16 rows without modulation / addend / f32 output (the MLP's LayerNorm + GELU pass): 16-byte pieces, a workgroup per row
        bool wide = dtype == SEA_BF16 && x_is_act && d >= 1024 && d <= 16384 && d % 8 == 0 && M > 32 && sea_tune("ln_rows", 1) != 0;
        for (int i = 0; i < n_groups && wide; ++i) {
            const SeaNormGroup& G = groups[i];
            wide = !G.mod && !G.addend && !G.Y32 && G.Yact && G.ldx % 8 == 0 && G.ldyact % 8 == 0;
        }
        if (wide) {
            const int nt = d <= 2048 ? 256 : (d <= 4096 ? 512 : 1024);
            static const int per_cu_t = sea_tune("ln_rows_per_cu", 0);   // tuning aid
            // resident workgroups per CU: all that the 64-register kernel's waves fill (8 x 256 threads at d = 2048: 94 -> 88 us at cfg3 against 6; a second row of
            // prefetch in registers costs the eighth wave per SIMD and loses: 107 us)
            const int per_cu = per_cu_t > 0 ? per_cu_t : (d <= 2048 ? 8 : (d <= 4096 ? 3 : 1));
            int nblk = (per_cu * 256 / n_groups) / 32 * 32;
            nblk = nblk < 32 ? 32 : nblk;
            nblk = nblk > M ? M : nblk;
            const dim3 gridw(nblk, n_groups);
#define LAUNCH_LNR(NTV, KCV)                                                                 \
    do {                                                                                      \
        if (gelu) ln_rows_bf16_kernel<NTV, KCV, true><<<gridw, dim3(NTV), 0, s>>>(L);         \
        else ln_rows_bf16_kernel<NTV, KCV, false><<<gridw, dim3(NTV), 0, s>>>(L);             \
    } while (0)
            if (nt == 256) LAUNCH_LNR(256, 1);
            else if (nt == 512) LAUNCH_LNR(512, 1);
            else if (d <= 8192) LAUNCH_LNR(1024, 1);
            else LAUNCH_LNR(1024, 2);
#undef LAUNCH_LNR
            SEA_CHECK_LAUNCH("sea_rownorm");
            return SEA_OK;
        }
    }
#define LAUNCH_RN(TT, XA)                                                     \
    do {                                                                       \
        if (d <= 256) rownorm_kernel<TT, XA, 1><<<grid, block, 0, s>>>(L);     \
        else if (d <= 512) rownorm_kernel<TT, XA, 2><<<grid, block, 0, s>>>(L);\
        else if (d <= 1024) rownorm_kernel<TT, XA, 4><<<grid, block, 0, s>>>(L);\
        else if (d <= 2048) rownorm_kernel<TT, XA, 8><<<grid, block, 0, s>>>(L);\
        else rownorm_kernel<TT, XA, 0><<<grid, block, 0, s>>>(L);              \
    } while (0)
    if (dtype == SEA_BF16) {
        if (x_is_act) LAUNCH_RN(__bf16, true);
        else LAUNCH_RN(__bf16, false);
    } else {
        LAUNCH_RN(float, false);  // f32 activations: x is float either way
    }
#undef LAUNCH_RN
    SEA_CHECK_LAUNCH("sea_rownorm");
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ silu outer product
#define SEA_MAX_SILU_GROUPS 24
struct SiluLaunch {
    SeaSiluGroup g[SEA_MAX_SILU_GROUPS];
    const float* c;
    int M;
    int n_groups;
};
struct SiluIbLaunch {
    SeaIbParams ib[SEA_MAX_SILU_IB];
};

// 1-D grid: first n_ib * ceil(M / 4) workgroups of information-bottleneck rows (one wave per row: the longer passes — a dependent LayerNorm + GELU + h-term dot per
// element — dispatched first), then n_groups * ceil(M / 32) workgroups of silu rows: a workgroup = 32 rows of one group, a wave takes 8 of them with its slice of
// w1 / b1 (4 columns per lane and 256-column pass) in registers, so a row is one scalar load, 4 silu per pass and one 512-byte store per wave.  (Round 4: the
// one-wave-per-row form was 6578 workgroups of a few hundred cycles each at cfg2 — 11 us for 24 MB of stores, bound by workgroup dispatch.)
template <typename T, bool WITH_IB>
__global__ __launch_bounds__(256) void silu_outer_kernel(const SiluLaunch L, const SiluIbLaunch I, int n_ib) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    if constexpr (WITH_IB) {
        const int per_ib = (L.M + 3) >> 2;
        if (bid < n_ib * per_ib) {   // block-uniform
            const int k = bid / per_ib;
            const int row = (bid - k * per_ib) * 4 + wave;
            if (row < L.M) ib_store_row(I.ib[k], L.c[row], row, lane);
            return;
        }
        bid -= n_ib * per_ib;
    }
    const int per_g = (L.M + 31) >> 5;
    const int gy = bid / per_g;
    const SeaSiluGroup& G = L.g[gy];
    const int row0 = (bid - gy * per_g) * 32 + wave * 8;
    if (row0 >= L.M) return;
    float cv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cv[i] = L.c[row0 + i < L.M ? row0 + i : L.M - 1];
    T* out = static_cast<T*>(G.Hid) + (int64_t)row0 * G.ld;
    for (int k = lane * 4; k < G.K2; k += 256) {
        float w[4], bb[4];
        load4(G.w1 + k, w);
        load4(G.b1 + k, bb);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (row0 + i < L.M)
                store4(out + (int64_t)i * G.ld + k, silu_f(w[0] * cv[i] + bb[0]), silu_f(w[1] * cv[i] + bb[1]), silu_f(w[2] * cv[i] + bb[2]), silu_f(w[3] * cv[i] + bb[3]));
        }
    }
}

extern "C" int sea_silu_outer(const SeaSiluGroup* groups, int n_groups, const float* c, int M, int dtype, void* stream) {
    return sea_silu_outer_ib(groups, n_groups, c, M, dtype, nullptr, 0, stream);
}

extern "C" int sea_silu_outer_ib(const SeaSiluGroup* groups, int n_groups, const float* c, int M, int dtype, const SeaIbParams* ibs, int n_ib, void* stream) {
    SEA_REQUIRE(n_ib >= 0 && n_ib <= SEA_MAX_SILU_IB && (n_ib == 0 || ibs != nullptr), "sea_silu_outer_ib: n_ib=%d out of range", n_ib);
    // (n_groups = 0 with n_ib > 0: the information-bottleneck rows alone — plans whose condition GEMM generates its operand have no silu rows to store)
    SEA_REQUIRE(c != nullptr && n_groups >= (n_ib > 0 ? 0 : 1) && (n_groups == 0 || groups != nullptr) && n_groups <= SEA_MAX_SILU_GROUPS && M >= 1, "sea_silu_outer: bad arguments (n_groups=%d, M=%d)", n_groups, M);
    SiluIbLaunch I;
    memset(&I, 0, sizeof(I));
    for (int k = 0; k < n_ib; ++k) {
        const SeaIbParams& P = ibs[k];
        if (P.mode == 0)
            SEA_REQUIRE(P.X[0] && sea_aligned16(P.X[0]) && P.E >= 4 && P.E % 4 == 0 && P.h >= 1 && P.h <= 64 && P.ldx >= P.E && P.ldx % 4 == 0 && P.w1 && P.b1 && P.lnw && P.lnb &&
                            P.w2 && P.b2 && sea_aligned16(P.b2), "sea_silu_outer_ib: ib[%d]: bad sizes / null / misaligned pointer", k);
        else
            SEA_REQUIRE((P.mode == 1 || P.mode == 2) && P.X[0] && sea_aligned16(P.X[0]) && P.E >= 8 && P.E % 8 == 0 && P.ldx >= P.E && P.ldx % 4 == 0 && P.w1 && sea_aligned16(P.w1) &&
                            (P.mode == 2 || (P.b1 && sea_aligned16(P.b1))), "sea_silu_outer_ib: ib[%d]: bad sizes / null / misaligned pointer (mode %d)", k, P.mode);
        I.ib[k] = P;
    }
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_silu_outer: bad dtype %d", dtype);
    SiluLaunch L;
    memset(&L, 0, sizeof(L));
    for (int i = 0; i < n_groups; ++i) {
        const SeaSiluGroup& G = groups[i];
        SEA_REQUIRE(G.w1 && G.b1 && G.Hid && G.K2 >= 4 && G.K2 % 4 == 0 && G.ld >= G.K2 && G.ld % 4 == 0, "sea_silu_outer[%d]: bad group", i);
        SEA_REQUIRE(sea_aligned16(G.w1) && sea_aligned16(G.b1) && sea_aligned16(G.Hid), "sea_silu_outer[%d]: pointers must be 16-byte aligned", i);
        L.g[i] = G;
    }
    L.c = c; L.M = M; L.n_groups = n_groups;
    const dim3 grid(n_ib * ((M + 3) / 4) + n_groups * ((M + 31) / 32)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_ib > 0) {
        if (dtype == SEA_BF16) silu_outer_kernel<__bf16, true><<<grid, block, 0, s>>>(L, I, n_ib);
        else silu_outer_kernel<float, true><<<grid, block, 0, s>>>(L, I, n_ib);
    } else {
        if (dtype == SEA_BF16) silu_outer_kernel<__bf16, false><<<grid, block, 0, s>>>(L, I, 0);
        else silu_outer_kernel<float, false><<<grid, block, 0, s>>>(L, I, 0);
    }
    SEA_CHECK_LAUNCH("sea_silu_outer");
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ information bottleneck add
// One wave per row.  Lanes 0..h-1 compute the hidden vector gelu(LN_h(w1 c + b1)) (h <= 64), the wave shares it by
// __shfl, then every lane produces 4 consecutive output columns at a time and adds them to all fields.
// SPLIT (a few rows, e.g. a KV-cache step): the four waves of a workgroup share ONE row (wave w the columns 256 w + 1024 k ..), so a wide row is four short
// chains instead of one long one (22 us for one row of 2048 at the multiphase width).
template <bool SPLIT>
__global__ __launch_bounds__(256) void ib_add_kernel(const SeaIbParams P) {
    const int lane = threadIdx.x & 63;
    const int row = SPLIT ? (int)blockIdx.x : (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= P.M) return;
    const int e_first = SPLIT ? (int)(threadIdx.x >> 6) * 256 + lane * 4 : lane * 4, e_step = SPLIT ? 1024 : 256;
    const int h = P.h;
    const float cv = P.c[row];
    if (P.mode != 0) {   // 'linear' / 'fourier' layers: no hidden vector (block-uniform)
        for (int e0 = e_first; e0 < P.E; e0 += e_step) {
            float o[4];
            ib_simple4(P, cv, e0, o);
            for (int f = 0; f < P.n_fields; ++f) {
                float* x = P.X[f] + (int64_t)row * P.ldx + e0;
                float v[4];
                load4(x, v);
                store4(x, v[0] + o[0], v[1] + o[1], v[2] + o[2], v[3] + o[3]);
            }
        }
        return;
    }
    const bool act = lane < h;
    const float pre = act ? P.w1[lane] * cv + P.b1[lane] : 0.f;
    const float mean = wave_sum(pre) / (float)h;
    const float cen = act ? pre - mean : 0.f;
    const float var = wave_sum(cen * cen) / (float)h;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float hid = act ? gelu_erf(cen * rstd * P.lnw[lane] + P.lnb[lane]) : 0.f;
    for (int e0 = e_first; e0 < P.E; e0 += e_step) {
        float o[4];
        load4(P.b2 + e0, o);
        if ((h & 3) == 0) {
            // rows of w2 are whole 16-byte chunks: request all of them before the first use (one memory round trip, not h of them)
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
        for (int f = 0; f < P.n_fields; ++f) {
            float* x = P.X[f] + (int64_t)row * P.ldx + e0;
            float v[4], df[4] = {1.f, 1.f, 1.f, 1.f};
            load4(x, v);
            if (P.drop.thr > 0) {  // the reference evaluates the MLP (and its dropout) once per field: independent masks
                const uint32_t w = drop_word(P.drop.seed, P.drop.stream + f, (uint32_t)row, (uint32_t)(e0 >> 2));
                const float sc = drop_scale(P.drop.thr);
#pragma unroll
                for (int e = 0; e < 4; ++e) df[e] = drop_factor(w, e, P.drop.thr, sc);
            }
            store4(x, v[0] + o[0] * df[0], v[1] + o[1] * df[1], v[2] + o[2] * df[2], v[3] + o[3] * df[3]);
        }
    }
}

// A few rows of a wide model ('mlp' layers, h = 8, no dropout: a KV-cache step at the shipped widths): a workgroup per row, a thread per 4 output columns,
// and EVERYTHING the thread needs — its rows of w2, b2, the fields' x — requested before the first wait, beside the condition and the hidden layer's
// parameters: one memory round trip (the wave-per-quarter-row form above walks two column passes behind the hidden vector: 7.6 us at embed_dim 2048).
__global__ __launch_bounds__(1024) void ib_add_fewrows_kernel(const SeaIbParams P) {
    constexpr int H8 = 8, MAXF = 4;
    const int lane = threadIdx.x & 63, row = (int)blockIdx.x;
    const int e0 = (int)threadIdx.x * 4;
    const bool live = e0 < P.E;
    const int ec = live ? e0 : 0;
    float w[4][H8], o[4], xv[MAXF][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        load4(P.w2 + (int64_t)(ec + e) * H8, *reinterpret_cast<float(*)[4]>(w[e]));
        load4(P.w2 + (int64_t)(ec + e) * H8 + 4, *reinterpret_cast<float(*)[4]>(w[e] + 4));
    }
    load4(P.b2 + ec, o);
#pragma unroll
    for (int f = 0; f < MAXF; ++f)
        if (f < P.n_fields) load4(P.X[f] + (int64_t)row * P.ldx + ec, xv[f]);
    const float cv = P.c[row];
    const bool act = lane < H8;
    const float pre = act ? P.w1[lane] * cv + P.b1[lane] : 0.f;
    const float lw = act ? P.lnw[lane] : 0.f, lb = act ? P.lnb[lane] : 0.f;
    const float mean = wave_sum(pre) / (float)H8;
    const float cen = act ? pre - mean : 0.f;
    const float var = wave_sum(cen * cen) / (float)H8;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float hid = act ? gelu_erf(cen * rstd * lw + lb) : 0.f;
#pragma unroll
    for (int k = 0; k < H8; ++k) {
        const float hk = __shfl(hid, k);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += w[e][k] * hk;
    }
    if (!live) return;
#pragma unroll
    for (int f = 0; f < MAXF; ++f)
        if (f < P.n_fields) store4(P.X[f] + (int64_t)row * P.ldx + e0, xv[f][0] + o[0], xv[f][1] + o[1], xv[f][2] + o[2], xv[f][3] + o[3]);
}

extern "C" int sea_ib_add(const SeaIbParams* params, void* stream) {
    SEA_REQUIRE(params != nullptr, "sea_ib_add: null params");
    const SeaIbParams& P = *params;
    SEA_REQUIRE(P.mode >= 0 && P.mode <= 2, "sea_ib_add: bad mode %d", P.mode);
    if (P.mode == 0) {
        SEA_REQUIRE(P.n_fields >= 1 && P.n_fields <= 8 && P.M >= 1 && P.E >= 4 && P.E % 4 == 0 && P.h >= 1 && P.h <= 64 && P.ldx >= P.E && P.ldx % 4 == 0,
                    "sea_ib_add: bad sizes n_fields=%d M=%d E=%d h=%d ldx=%d", P.n_fields, P.M, P.E, P.h, P.ldx);
        SEA_REQUIRE(P.c && P.w1 && P.b1 && P.lnw && P.lnb && P.w2 && P.b2, "sea_ib_add: null parameter pointer");
        SEA_REQUIRE(sea_aligned16(P.b2) && ((P.h & 3) != 0 || sea_aligned16(P.w2)), "sea_ib_add: b2 (and w2 when h %% 4 == 0) must be 16-byte aligned");
    } else {
        SEA_REQUIRE(P.n_fields >= 1 && P.n_fields <= 8 && P.M >= 1 && P.E >= 8 && P.E % 8 == 0 && P.ldx >= P.E && P.ldx % 4 == 0 && P.drop.thr == 0,
                    "sea_ib_add: bad sizes n_fields=%d M=%d E=%d ldx=%d (linear / fourier: E a multiple of 8, no dropout)", P.n_fields, P.M, P.E, P.ldx);
        SEA_REQUIRE(P.c && P.w1 && sea_aligned16(P.w1) && (P.mode == 2 || (P.b1 && sea_aligned16(P.b1))), "sea_ib_add: null / misaligned parameter pointer");
    }
    for (int f = 0; f < P.n_fields; ++f) SEA_REQUIRE(P.X[f] && sea_aligned16(P.X[f]), "sea_ib_add: X[%d] null or misaligned", f);
    if (P.mode == 0 && P.M <= 16 && P.E >= 1024 && P.E <= 4096 && P.h == 8 && P.n_fields <= 4 && P.drop.thr == 0)
        ib_add_fewrows_kernel<<<dim3(P.M), dim3((P.E / 4 + 63) / 64 * 64), 0, static_cast<hipStream_t>(stream)>>>(P);
    else if (P.M <= 16 && P.E > 256) ib_add_kernel<true><<<dim3(P.M), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(P);
    else ib_add_kernel<false><<<dim3((P.M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(P);
    SEA_CHECK_LAUNCH("sea_ib_add");
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ dropout mask (tests)
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* out, int64_t rows, int64_t cols, uint32_t seed, uint32_t stream, int thr) {
    const int64_t total = rows * cols;
    const float sc = thr > 0 ? drop_scale(thr) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols, c = i - r * cols;
        out[i] = thr > 0 ? drop_factor(drop_word(seed, stream, (uint32_t)r, (uint32_t)(c >> 2)), (int)(c & 3), thr, sc) : 1.f;
    }
}

extern "C" int sea_dropout_mask(float* out, int64_t rows, int64_t cols, uint32_t seed, uint32_t stream, int32_t thr, void* stream_handle) {
    SEA_REQUIRE(out && rows >= 1 && cols >= 1 && thr >= 0 && thr <= 255, "sea_dropout_mask: bad arguments");
    int64_t blocks = (rows * cols + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dropout_mask_kernel<<<dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream_handle)>>>(out, rows, cols, seed, stream, thr);
    SEA_CHECK_LAUNCH("sea_dropout_mask");
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ conversion
template <typename T>
__global__ __launch_bounds__(256) void convert_kernel(const float* __restrict__ src, int64_t lds, T* __restrict__ dst, int64_t ldd,
                                                      int64_t rows, int64_t cols4) {
    const int64_t total = rows * cols4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols4, c = (i - r * cols4) * 4;
        float v[4];
        load4(src + r * lds + c, v);
        store4(dst + r * ldd + c, v[0], v[1], v[2], v[3]);
    }
}

extern "C" int sea_convert_f32_to_act(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int dtype,
                                      void* stream) {
    SEA_REQUIRE(src && dst && rows >= 1 && cols >= 4 && cols % 4 == 0 && lds >= cols && ldd >= cols && lds % 4 == 0 && ldd % 4 == 0,
                "sea_convert_f32_to_act: bad arguments rows=%lld cols=%lld lds=%lld ldd=%lld", (long long)rows, (long long)cols, (long long)lds, (long long)ldd);
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_convert_f32_to_act: bad dtype %d", dtype);
    SEA_REQUIRE(sea_aligned16(src) && sea_aligned16(dst), "sea_convert_f32_to_act: pointers must be 16-byte aligned");
    const int64_t total = rows * (cols / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == SEA_BF16) convert_kernel<__bf16><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(src, lds, static_cast<__bf16*>(dst), ldd, rows, cols / 4);
    else convert_kernel<float><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(src, lds, static_cast<float*>(dst), ldd, rows, cols / 4);
    SEA_CHECK_LAUNCH("sea_convert_f32_to_act");
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ un-patchify
// One thread per (b, p, c) visits the F fields: the index-map lookup is shared by the fields and the F outputs of a mesh point are
// contiguous ([B, n_points, F]).  Reads walk c fastest (contiguous in the decoder's [B,P,F,C] layout), writes are a scatter by design.
__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ in, int64_t sb, int64_t sp, int64_t sf, int64_t sc,
                                                         const int32_t* __restrict__ imap, const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float* __restrict__ out, int B, int P, int F, int C, int n_points) {
    const int64_t total = (int64_t)B * P * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t bp = i / C;
        const int p = (int)(bp % P);
        const int b = (int)(bp / P);
        const int idx = imap[(int64_t)p * C + c];
        if (idx < 0) continue;
        const float* src = in + b * sb + p * sp + c * sc;
        float* dst = out + ((int64_t)b * n_points + idx) * F;
        for (int f = 0; f < F; ++f) dst[f] = src[f * sf] * scale[f] + shift[f];
    }
}

// Gather form: grid = (point blocks, snapshots); a workgroup's 256 points of one snapshot write 256 * F contiguous floats; the reads
// hit the snapshot's decoded cells in L2.
__global__ __launch_bounds__(256) void unpatchify_gather_kernel(const float* __restrict__ in, int64_t sb, int64_t sp, int64_t sf, int64_t sc,
                                                                const int32_t* __restrict__ slot, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, float* __restrict__ out, int F, int C, int n_points) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_points) return;
    const int sl = slot[n];
    const int p = sl / C, c = sl - p * C;
    const int b = blockIdx.y;
    const float* src = in + b * sb + p * sp + c * sc;
    float* dst = out + ((int64_t)b * n_points + n) * F;
    for (int f = 0; f < F; ++f) dst[f] = src[f * sf] * scale[f] + shift[f];
}

extern "C" int sea_unpatchify(const float* in, int64_t sb, int64_t sp, int64_t sf, int64_t sc, const int32_t* index_map, const int32_t* point_slot,
                              const float* scale, const float* shift, float* out, int B, int P, int F, int C, int n_points, void* stream) {
    SEA_REQUIRE(in && index_map && scale && shift && out, "sea_unpatchify: null pointer");
    SEA_REQUIRE(B >= 1 && P >= 1 && F >= 1 && C >= 1 && n_points >= 1, "sea_unpatchify: bad sizes B=%d P=%d F=%d C=%d n_points=%d", B, P, F, C, n_points);
    if (point_slot != nullptr) {
        SEA_REQUIRE(B <= 65535, "sea_unpatchify: B=%d too large for grid.y", B);
        unpatchify_gather_kernel<<<dim3((n_points + 255) / 256, B, 1), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(in, sb, sp, sf, sc, point_slot, scale, shift,
                                                                                                                         out, F, C, n_points);
        SEA_CHECK_LAUNCH("sea_unpatchify");
        return SEA_OK;
    }
    const int64_t total = (int64_t)B * P * C;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    unpatchify_kernel<<<dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(in, sb, sp, sf, sc, index_map, scale, shift, out, B, P, F, C, n_points);
    SEA_CHECK_LAUNCH("sea_unpatchify");
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ patchify (forward direction of the mesh partition)
// grid = (slot blocks, snapshots): thread = one (cell, slot) of one snapshot; the F values of its mesh point are read together (12 contiguous
// bytes at F = 3, served from L2: a snapshot is n_points * F * 4 bytes) and written at stride sf — with the encoder's [B, P, F, C] layout the
// writes of a wavefront are contiguous in c.  Slots >= C_map (the row padding to n_inp) and empty slots (index -1) get pad_value.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ in, const int32_t* __restrict__ imap, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ out, int64_t sb, int64_t sp, int64_t sf, int64_t sc,
                                                       int P, int F, int C_map, int C_out, int n_points, float pad_value) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)P * C_out) return;
    const int p = (int)(i / C_out), c = (int)(i - (int64_t)p * C_out);
    const int b = blockIdx.y;
    const int idx = c < C_map ? imap[(int64_t)p * C_map + c] : -1;
    float* dst = out + b * sb + p * sp + c * sc;
    if (idx < 0) {
        for (int f = 0; f < F; ++f) dst[f * sf] = pad_value;
        return;
    }
    const float* src = in + ((int64_t)b * n_points + idx) * F;
    for (int f = 0; f < F; ++f) dst[f * sf] = src[f] * scale[f] + shift[f];
}

extern "C" int sea_patchify(const float* in, const int32_t* index_map, const float* scale, const float* shift, float* out, int64_t sb, int64_t sp, int64_t sf,
                            int64_t sc, int B, int P, int F, int C_map, int C_out, int n_points, float pad_value, void* stream) {
    SEA_REQUIRE(in && index_map && scale && shift && out, "sea_patchify: null pointer");
    SEA_REQUIRE(B >= 1 && B <= 65535 && P >= 1 && F >= 1 && C_map >= 1 && C_out >= C_map && n_points >= 1, "sea_patchify: bad sizes B=%d P=%d F=%d C_map=%d C_out=%d n_points=%d",
                B, P, F, C_map, C_out, n_points);
    const int64_t per = (int64_t)P * C_out;
    patchify_kernel<<<dim3((unsigned)((per + 255) / 256), B, 1), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(in, index_map, scale, shift, out, sb, sp, sf, sc, P, F,
                                                                                                                  C_map, C_out, n_points, pad_value);
    SEA_CHECK_LAUNCH("sea_patchify");
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ split-K finish
// out = sum_s P[s] + bias * bias_scale + R  (fp32 sums in split order: deterministic), as fp32 and / or in the activation dtype: the pass behind a
// sea_gemm_grouped launch whose groups are K-slices of one product writing fp32 partial matrices (skinny M against K = 16384: the reference's own widths).
#define SEA_MAX_SPLITK_GROUPS 8
struct SplitkLaunch {
    SeaSplitkGroup g[SEA_MAX_SPLITK_GROUPS];
};

template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const SplitkLaunch L) {
    const SeaSplitkGroup& G = L.g[blockIdx.y];
    const int n4 = G.N >> 2;
    const long total = (long)G.M * n4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int m = (int)(idx / n4), n = (int)(idx - (long)m * n4) * 4;
        float v[4];
        load4(G.P + (int64_t)m * G.ldp + n, v);
        for (int s_ = 1; s_ < G.S; ++s_) {
            float p[4];
            load4(G.P + (int64_t)s_ * G.p_stride + (int64_t)m * G.ldp + n, p);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += p[q];
        }
        if (G.bias != nullptr) {
            float b[4];
            load4(G.bias + n, b);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += b[q] * G.bias_scale;
        }
        if (G.R != nullptr) {
            float r[4];
            load4(G.R + (int64_t)m * G.ldr + n, r);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += r[q];
        }
        if (G.C32 != nullptr) store4(G.C32 + (int64_t)m * G.ldc32 + n, v[0], v[1], v[2], v[3]);
        if (G.Cact != nullptr) store4(static_cast<T*>(G.Cact) + (int64_t)m * G.ldcact + n, v[0], v[1], v[2], v[3]);
    }
}

extern "C" int sea_splitk_finish(const SeaSplitkGroup* groups, int n_groups, int dtype, void* stream) {
    SEA_REQUIRE(groups != nullptr && n_groups >= 1 && n_groups <= SEA_MAX_SPLITK_GROUPS, "sea_splitk_finish: n_groups=%d out of range", n_groups);
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_splitk_finish: bad dtype %d", dtype);
    SplitkLaunch L;
    memset(&L, 0, sizeof(L));
    long most = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaSplitkGroup& G = groups[i];
        SEA_REQUIRE(G.P && G.S >= 1 && G.M >= 1 && G.N >= 4 && G.N % 4 == 0 && G.ldp % 4 == 0 && G.ldp >= G.N && G.p_stride % 4 == 0 && (G.C32 || G.Cact),
                    "sea_splitk_finish[%d]: null pointer or bad shape S=%d M=%d N=%d ldp=%d", i, G.S, G.M, G.N, G.ldp);
        SEA_REQUIRE((!G.R || (G.ldr % 4 == 0 && G.ldr >= G.N)) && (!G.C32 || (G.ldc32 % 4 == 0 && G.ldc32 >= G.N)) && (!G.Cact || (G.ldcact % 4 == 0 && G.ldcact >= G.N)),
                    "sea_splitk_finish[%d]: bad strides", i);
        SEA_REQUIRE(sea_aligned16(G.P) && sea_aligned16(G.bias) && sea_aligned16(G.R) && sea_aligned16(G.C32) && sea_aligned16(G.Cact), "sea_splitk_finish[%d]: pointers must be 16-byte aligned", i);
        L.g[i] = G;
        const long items = (long)G.M * (G.N / 4);
        most = items > most ? items : most;
    }
    long bx = (most + 255) / 256;
    bx = bx > 2048 ? 2048 : bx;
    const dim3 grid((unsigned)bx, (unsigned)n_groups);
    if (dtype == SEA_BF16) splitk_finish_kernel<__bf16><<<grid, dim3(256), 0, static_cast<hipStream_t>(stream)>>>(L);
    else splitk_finish_kernel<float><<<grid, dim3(256), 0, static_cast<hipStream_t>(stream)>>>(L);
    SEA_CHECK_LAUNCH("sea_splitk_finish");
    return SEA_OK;
}
