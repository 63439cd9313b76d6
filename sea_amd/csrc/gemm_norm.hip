// Linear layer + row normalisation in one launch (gfx950): sea_gemm_rownorm.
//
// A workgroup tile is 64 rows x the WHOLE output row (BN = 64, 128 or 256 >= N); its 4 waves sit in a 4 x 1 grid, so a wave owns
// 16 complete output rows: in the transposed accumulator map of gemm_core.hpp a row is then spread over the 4 lane groups of ONE
// wave (lane & 15 = row, lane >> 4 = column quad), and the LayerNorm statistics are two cross-lane adds per pass — no LDS, no
// second launch, and the [M, N] pre-normalisation matrix never goes to HBM unless the caller asks for it (training).
// The arithmetic is the one of rownorm_kernel (rowops.hip): two-pass fp32 mean / centred biased variance, modulation
// y = xhat * (gamma + 1 + w) + (beta + b).
#include "gemm_core.hpp"
#include <stdlib.h>

struct GemmNormLaunch {
    SeaGemmNormGroup g[SEA_MAX_GEMM_NORM_GROUPS];
    int tile_start[SEA_MAX_GEMM_NORM_GROUPS + 1];
    int n_groups;
    float eps;
};

// sum over the 4 lane groups {l, l^16, l^32, l^48} (the 4 column quads of one output row)
__device__ __forceinline__ float group_sum4(float x) {
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

template <typename T, int BN>
__global__ __launch_bounds__(256) void gemm_rownorm_kernel(const GemmNormLaunch L) {
    constexpr int BM = 64;
    using C = GemmCfg<T, BM, BN, 4>;
    static_assert(C::MI == 1, "a wave owns 16 rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int gi = 0;
    while (gi + 1 < L.n_groups && (int)blockIdx.x >= L.tile_start[gi + 1]) ++gi;
    const SeaGemmNormGroup& G = L.g[gi];
    const int tm = blockIdx.x - L.tile_start[gi];

    GemmMainloop<T, BM, BN, 4> ml;
    ml.A = static_cast<const T*>(G.A);
    ml.W = static_cast<const T*>(G.W);
    ml.a_seg_stride = 0;
    ml.lda = G.lda; ml.ldw = G.ldw; ml.M = G.M; ml.N = G.N; ml.K = G.K; ml.n_seg = 1;
    ml.m0 = tm * BM; ml.n0 = 0;
    f32x4 acc[C::MI][C::NI];
    ml.run_single(smem, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int m = ml.m0 + wave * 16 + r;
    const bool mok = m < G.M;
    const int N = G.N;
    const float inv_n = 1.0f / (float)N;

    float v[C::NI][4];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
        const int n = j * 16 + g * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = 0.f;
        if (n < N) {   // N % 16 == 0: whole 16-column blocks are valid or not
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (G.bias != nullptr) load4(G.bias + n, bv);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[j][q] = acc[0][j][q] + bv[q];
            if (G.R != nullptr && mok) {
                float rv[4];
                load4(G.R + (int64_t)m * G.ldr + n, rv);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[j][q] += rv[q];
            }
            if (G.C32 != nullptr && mok) store4(G.C32 + (int64_t)m * G.ldc32 + n, v[j][0], v[j][1], v[j][2], v[j][3]);
            sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
    }
    const float mean = group_sum4(sum) * inv_n;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
        if (j * 16 < N) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float c = v[j][q] - mean;
                sq += c * c;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(group_sum4(sq) * inv_n + L.eps);
    if (!mok) return;
    if (g == 0) {
        if (G.mean != nullptr) G.mean[m] = mean;
        if (G.rstd != nullptr) G.rstd[m] = rstd;
    }
    const T* mod = G.mod != nullptr ? static_cast<const T*>(G.mod) + (int64_t)m * G.ldmod : nullptr;
    float* y32 = G.Y32 != nullptr ? G.Y32 + (int64_t)m * G.ldy32 : nullptr;
    T* yact = G.Yact != nullptr ? static_cast<T*>(G.Yact) + (int64_t)m * G.ldyact : nullptr;
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
        const int n = j * 16 + g * 4;
        if (n >= N) continue;
        float gm[4], bt[4] = {0.f, 0.f, 0.f, 0.f}, mw[4] = {0.f, 0.f, 0.f, 0.f}, mb[4] = {0.f, 0.f, 0.f, 0.f};
        load4(G.gamma + n, gm);
        if (G.beta != nullptr) load4(G.beta + n, bt);
        if (mod != nullptr) {
            load4(mod + n, mw);
            load4(mod + N + n, mb);
        }
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gq = mod != nullptr ? gm[q] + 1.0f + mw[q] : gm[q];
            const float bq = mod != nullptr ? bt[q] + mb[q] : bt[q];
            o[q] = (v[j][q] - mean) * rstd * gq + bq;
        }
        if (y32 != nullptr) store4(y32 + n, o[0], o[1], o[2], o[3]);
        if (yact != nullptr) store4(yact + n, o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------------------------- 16-row tiles (short launches)
// At M = 2024 the 64-row form is 32 workgroups: the launch is one workgroup's serial latency and its epilogue has 16 column blocks per
// lane.  Here a workgroup owns 16 rows x the whole output row with its 4 waves side by side (1 x 4: wave w owns columns
// [w BN/4, (w+1) BN/4)): 4x the workgroups, a quarter of the epilogue per lane, the modulation / gain loads issued before the
// statistics; the row statistics cross the waves through 2 x 64 floats of LDS.  The W tile is staged per 16 rows instead of per 64
// (L2 -> LDS traffic x4), which is why the long launches keep the 64-row form.
template <typename T, int BN>
__global__ __launch_bounds__(256) void gemm_rownorm16_kernel(const GemmNormLaunch L) {
    constexpr int BM = 16, BKB = 128;
    constexpr int EPC = ActTraits<T>::EPC, BK = BKB / (int)sizeof(T);
    constexpr int WTN = BN / 4, NI = WTN / 16;
    constexpr int ROWS = BM + BN, CHUNKS = ROWS * 8, CH = (CHUNKS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int gi = 0;
    while (gi + 1 < L.n_groups && (int)blockIdx.x >= L.tile_start[gi + 1]) ++gi;
    const SeaGemmNormGroup& G = L.g[gi];
    const int m0 = (blockIdx.x - L.tile_start[gi]) * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int N = G.N, K = G.K, M = G.M;
    const T* A = static_cast<const T*>(G.A);
    const T* W = static_cast<const T*>(G.W);

    // staging: chunk id = row * 8 + c over the ROWS = 16 + BN tile rows (A rows first), 16 bytes each
    uint4 rg[CH];
    const T* src[CH];
    int dst[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int id = tid + 256 * i;
        const int row = id >> 3, c = id & 7;
        src[i] = nullptr;
        dst[i] = row * BKB + ((c ^ (row & 7)) << 4);
        if (id < CHUNKS) {
            if (row < BM) {
                int mr = m0 + row;
                mr = mr < M ? mr : M - 1;
                src[i] = A + (int64_t)mr * G.lda + c * EPC;
            } else {
                int nr = row - BM;
                nr = nr < N ? nr : N - 1;
                src[i] = W + (int64_t)nr * G.ldw + c * EPC;
            }
        }
    }
    auto load_tile = [&](int kt) {
        const int c = tid & 7;
        const bool kvalid = kt * BK + c * EPC < K;
#pragma unroll
        for (int i = 0; i < CH; ++i)
            rg[i] = (src[i] != nullptr && kvalid) ? *reinterpret_cast<const uint4*>(src[i] + kt * BK) : make_uint4(0, 0, 0, 0);
    };
    f32x4 acc[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = (K + BK - 1) / BK;
    load_tile(0);

    // epilogue operands that do not depend on the accumulators: requested now, they arrive under the main loop
    const int m = m0 + r;
    const bool mok = m < M;
    const int mc = mok ? m : M - 1;
    const T* mod = G.mod != nullptr ? static_cast<const T*>(G.mod) + (int64_t)mc * G.ldmod : nullptr;
    float bv[NI][4], gm[NI][4], bt[NI][4], mw[NI][4], mb[NI][4], rv[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = wave * WTN + j * 16 + g * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[j][q] = gm[j][q] = bt[j][q] = mw[j][q] = mb[j][q] = rv[j][q] = 0.f;
        if (n < N) {
            if (G.bias != nullptr) load4(G.bias + n, bv[j]);
            load4(G.gamma + n, gm[j]);
            if (G.beta != nullptr) load4(G.beta + n, bt[j]);
            if (mod != nullptr) {
                load4(mod + n, mw[j]);
                load4(mod + N + n, mb[j]);
            }
            if (G.R != nullptr) load4(G.R + (int64_t)mc * G.ldr + n, rv[j]);
        }
    }

    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (tid + 256 * i < CHUNKS) *reinterpret_cast<uint4*>(smem + dst[i]) = rg[i];
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
        const char* sA = smem + r * BKB;
        const char* sB = smem + (BM + wave * WTN + r) * BKB;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
            const uint4 af = *reinterpret_cast<const uint4*>(sA + off);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const uint4 bf = *reinterpret_cast<const uint4*>(sB + j * 16 * BKB + off);
                mma16<T>(bf, af, acc[j]);   // transposed tile: lane & 15 = output row, 4 (lane >> 4) + q = column inside the 16-block
            }
        }
        __syncthreads();
    }

    float* red = reinterpret_cast<float*>(smem);   // [2][4 waves][16 rows]
    const float inv_n = 1.0f / (float)N;
    float v[NI][4];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = wave * WTN + j * 16 + g * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = n < N ? acc[j][q] + bv[j][q] + rv[j][q] : 0.f;
        sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        if (G.C32 != nullptr && mok && n < N) store4(G.C32 + (int64_t)m * G.ldc32 + n, v[j][0], v[j][1], v[j][2], v[j][3]);
    }
    sum = group_sum4(sum);
    if (g == 0) red[wave * 16 + r] = sum;
    __syncthreads();
    const float mean = ((red[r] + red[16 + r]) + (red[32 + r] + red[48 + r])) * inv_n;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        if (wave * WTN + j * 16 < N) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float c = v[j][q] - mean;
                sq += c * c;
            }
        }
    }
    sq = group_sum4(sq);
    if (g == 0) red[64 + wave * 16 + r] = sq;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[64 + r] + red[80 + r]) + (red[96 + r] + red[112 + r])) * inv_n + L.eps);
    if (!mok) return;
    if (g == 0 && wave == 0) {
        if (G.mean != nullptr) G.mean[m] = mean;
        if (G.rstd != nullptr) G.rstd[m] = rstd;
    }
    float* y32 = G.Y32 != nullptr ? G.Y32 + (int64_t)m * G.ldy32 : nullptr;
    T* yact = G.Yact != nullptr ? static_cast<T*>(G.Yact) + (int64_t)m * G.ldyact : nullptr;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = wave * WTN + j * 16 + g * 4;
        if (n >= N) continue;
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gq = mod != nullptr ? gm[j][q] + 1.0f + mw[j][q] : gm[j][q];
            const float bq = mod != nullptr ? bt[j][q] + mb[j][q] : bt[j][q];
            o[q] = (v[j][q] - mean) * rstd * gq + bq;
        }
        if (y32 != nullptr) store4(y32 + n, o[0], o[1], o[2], o[3]);
        if (yact != nullptr) store4(yact + n, o[0], o[1], o[2], o[3]);
    }
}

template <typename K>
static int set_lds_gn(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : -1;
}

extern "C" int sea_gemm_rownorm(const SeaGemmNormGroup* groups, int n_groups, float eps, int dtype, void* stream) {
    SEA_REQUIRE(groups != nullptr && n_groups >= 1 && n_groups <= SEA_MAX_GEMM_NORM_GROUPS, "sea_gemm_rownorm: n_groups=%d out of range", n_groups);
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_gemm_rownorm: bad dtype %d", dtype);
    const int epc = dtype == SEA_BF16 ? 8 : 4;
    int nmax = 0;
    GemmNormLaunch L;
    memset(&L, 0, sizeof(L));
    int total = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaGemmNormGroup& G = groups[i];
        SEA_REQUIRE(G.A && G.W && G.gamma && (G.Y32 || G.Yact), "sea_gemm_rownorm[%d]: null pointer", i);
        SEA_REQUIRE(G.M >= 1 && G.N >= 16 && G.N % 16 == 0 && G.N <= 256 && G.K >= 8 && G.K % 8 == 0, "sea_gemm_rownorm[%d]: bad shape M=%d N=%d K=%d (N a multiple of 16 up to 256)", i, G.M, G.N, G.K);
        SEA_REQUIRE(G.lda % epc == 0 && G.ldw % epc == 0 && G.lda >= G.K && G.ldw >= G.K, "sea_gemm_rownorm[%d]: bad operand strides lda=%d ldw=%d", i, G.lda, G.ldw);
        SEA_REQUIRE((!G.R || (G.ldr % 4 == 0 && G.ldr >= G.N)) && (!G.C32 || (G.ldc32 % 4 == 0 && G.ldc32 >= G.N)) && (!G.Y32 || (G.ldy32 % 4 == 0 && G.ldy32 >= G.N)) &&
                        (!G.Yact || (G.ldyact % 4 == 0 && G.ldyact >= G.N)) && (!G.mod || (G.ldmod % 4 == 0 && G.ldmod >= 2 * G.N)),
                    "sea_gemm_rownorm[%d]: bad output / modulation strides", i);
        SEA_REQUIRE(sea_aligned16(G.A) && sea_aligned16(G.W) && sea_aligned16(G.bias) && sea_aligned16(G.R) && sea_aligned16(G.C32) && sea_aligned16(G.mod) &&
                        sea_aligned16(G.gamma) && sea_aligned16(G.beta) && sea_aligned16(G.Y32) && sea_aligned16(G.Yact),
                    "sea_gemm_rownorm[%d]: pointers must be 16-byte aligned", i);
        nmax = G.N > nmax ? G.N : nmax;
        L.g[i] = G;
        L.tile_start[i] = total;
        total += (G.M + 63) / 64;
    }
    // 16-row tiles while the 64-row launch would leave most CUs without a workgroup (measured at cfg2: see DESIGN.md); SEA_GEMM_NORM_ROWS=16|64 forces
    static const int forced = []() { const char* e = getenv("SEA_GEMM_NORM_ROWS"); return e ? atoi(e) : 0; }();
    const bool small = forced == 16 || (forced != 64 && total <= 512);
    if (small) {
        total = 0;
        for (int i = 0; i < n_groups; ++i) {
            L.tile_start[i] = total;
            total += (groups[i].M + 15) / 16;
        }
    }
    L.tile_start[n_groups] = total;
    L.n_groups = n_groups;
    L.eps = eps;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define LAUNCH_GN(TT, BNN)                                                                      \
    do {                                                                                        \
        constexpr int lds_ = GemmCfg<TT, 64, BNN, 4>::BUF_BYTES;                                \
        static int once = set_lds_gn(gemm_rownorm_kernel<TT, BNN>, lds_);                       \
        (void)once;                                                                             \
        gemm_rownorm_kernel<TT, BNN><<<dim3(total), dim3(256), lds_, s>>>(L);                   \
    } while (0)
#define LAUNCH_GN16(TT, BNN)                                                                    \
    do {                                                                                        \
        constexpr int lds_ = (16 + BNN) * 128;                                                  \
        static int once = set_lds_gn(gemm_rownorm16_kernel<TT, BNN>, lds_);                     \
        (void)once;                                                                             \
        gemm_rownorm16_kernel<TT, BNN><<<dim3(total), dim3(256), lds_, s>>>(L);                 \
    } while (0)
#define LAUNCH_GN_T(TT)                                                                          \
    do {                                                                                        \
        if (small) {                                                                            \
            if (nmax <= 64) LAUNCH_GN16(TT, 64);                                                \
            else if (nmax <= 128) LAUNCH_GN16(TT, 128);                                         \
            else LAUNCH_GN16(TT, 256);                                                          \
        } else {                                                                                \
            if (nmax <= 64) LAUNCH_GN(TT, 64);                                                  \
            else if (nmax <= 128) LAUNCH_GN(TT, 128);                                           \
            else LAUNCH_GN(TT, 256);                                                            \
        }                                                                                       \
    } while (0)
    if (dtype == SEA_BF16) LAUNCH_GN_T(__bf16); else LAUNCH_GN_T(float);
#undef LAUNCH_GN_T
#undef LAUNCH_GN16
#undef LAUNCH_GN
    SEA_CHECK_LAUNCH("sea_gemm_rownorm");
    return SEA_OK;
}
