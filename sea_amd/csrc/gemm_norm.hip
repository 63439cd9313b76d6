// Linear layer + row normalisation in one launch (gfx950): sea_gemm_rownorm.
//
// A workgroup tile spans the WHOLE output row (BN = 64, 128 or 256 >= N), so the LayerNorm statistics of a row are available in
// the GEMM epilogue: no second launch, and the [M, N] pre-normalisation matrix goes to HBM only when the caller asks for it.
// The arithmetic is the one of rownorm_kernel (rowops.hip): two-pass fp32 mean / centred biased variance, modulation
// y = xhat * (gamma + 1 + w) + (beta + b); the optional info-bottleneck addend is the one of ib_add_kernel (rowops.hip).
//
// Two tile shapes:
//   * 64 rows, waves 4 x 1: a wave owns 16 complete output rows — in the transposed accumulator map of gemm_core.hpp a row is spread
//     over the 4 lane groups of ONE wave (lane & 15 = row, lane >> 4 = column quad): statistics = two cross-lane adds per pass.
//   * 16 rows, waves 1 x 4 (short launches): at M = 2024 the 64-row form is 32 workgroups — one workgroup's serial latency with 16
//     column blocks per lane in the epilogue.  16-row tiles are 4x the workgroups with a quarter of the epilogue per lane, the
//     epilogue operands requested before the main loop; the statistics cross the waves through 2 x 64 floats of LDS.  The W tile is
//     staged per 16 rows instead of per 64 (L2 -> LDS traffic x4), which is why long launches keep the 64-row form.
#include "norm_epilogue.hpp"
#include <stdlib.h>

struct GemmNormLaunch {
    SeaGemmNormGroup g[SEA_MAX_GEMM_NORM_GROUPS];
    int tile_start[SEA_MAX_GEMM_NORM_GROUPS + 1];
    int n_groups;
    float eps;
};

// ---------------------------------------------------------------------------------------------- 64-row tiles
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
    ml.a_seg_stride = G.a_seg_stride;
    ml.lda = G.lda; ml.ldw = G.ldw; ml.M = G.M; ml.N = G.N; ml.K = G.K; ml.n_seg = G.n_seg;
    ml.m0 = tm * BM; ml.n0 = 0;
    f32x4 acc[C::MI][C::NI];
    ml.run_single(smem, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    NormEpilogue<T, C::NI, false, false> epi;
    epi.finish(G, acc[0], ml.m0 + wave * 16 + (lane & 15), 0, lane & 15, lane >> 4, wave, L.eps, nullptr);
}

// ---------------------------------------------------------------------------------------------- 16-row tiles (short launches)
// DMA: the whole contraction (at most 4 K-tiles) of the A rows and of W goes HBM/L2 -> LDS in ONE burst of global_load_lds (no VGPRs, no
// ds_write, one memory round trip instead of one per K-tile), then the K-tiles are computed back to back.  LDS = nk (16 + BN) 128 B <= 136 KiB.
template <typename T, int BN, bool DMA>
__global__ __launch_bounds__(256) void gemm_rownorm16_kernel(const GemmNormLaunch L) {
    constexpr int BM = 16, BKB = 128;
    constexpr int EPC = ActTraits<T>::EPC, BK = BKB / (int)sizeof(T);
    constexpr int WTN = BN / 4, NI = WTN / 16;
    constexpr int ROWS = BM + BN, CHUNKS = ROWS * 8, CH = (CHUNKS + 255) / 256;
    // DMA form: nk <= 4 stages of ROWS * 128 bytes, ROWS / 8 slabs of 1 KiB each: the host requests nk_max * ROWS * 128 bytes (sea_gemm_rownorm)
    static_assert(!DMA || (ROWS % 8 == 0 && 4 * ROWS * BKB <= 160 * 1024), "LDS request of the one-burst Linear + row norm");
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

    if constexpr (DMA) {
        const int k_total = K * G.n_seg;
        const int nk = k_total / BK;
        const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int rl = lane >> 3;
        const int chunk = (lane & 7) ^ (rl & 7);   // swizzle on the source side: LDS position p of a row holds chunk p ^ (row & 7)
        constexpr int SLABS = ROWS / 8, STAGE = ROWS * BKB;
        for (int kt = 0; kt < nk; ++kt) {
            const int kk = kt * BK;
            int seg = 0, kin = kk;
            if (G.n_seg > 1) {
                seg = kk / K;
                kin = kk - seg * K;
            }
            const T* a_base = A + seg * G.a_seg_stride + kin + chunk * EPC;
            const T* w_base = W + kin + chunk * EPC;
            for (int u = wv; u < SLABS; u += 4) {   // slab = 8 tile rows = 1 KiB; wave-uniform
                const int row = u * 8 + rl;
                const T* p;
                if (u < BM / 8) {
                    int mr = m0 + row;
                    mr = mr < M ? mr : M - 1;
                    p = a_base + (int64_t)mr * G.lda;
                } else {
                    int nr = row - BM;
                    nr = nr < N ? nr : N - 1;
                    p = w_base + (int64_t)nr * G.ldw;
                }
                glds16_gn(p, lds_base + (unsigned)(kt * STAGE + u * 8 * BKB));
            }
        }
        const int m = m0 + r;
        NormEpilogue<T, NI, true, true> epi;
        epi.prefetch(G, m < M ? m : M - 1, wave * WTN, g);
        f32x4 acc[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const char* sA = smem + kt * STAGE + r * BKB;
            const char* sB = smem + kt * STAGE + (BM + wave * WTN + r) * BKB;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
                const uint4 af = *reinterpret_cast<const uint4*>(sA + off);
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const uint4 bf = *reinterpret_cast<const uint4*>(sB + j * 16 * BKB + off);
                    mma16<T>(bf, af, acc[j]);
                }
            }
        }
        __syncthreads();
        epi.finish(G, acc, m, wave * WTN, r, g, wave, L.eps, reinterpret_cast<float*>(smem));
        return;
    }

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
                src[i] = A + (int64_t)mr * G.lda;
            } else {
                int nr = row - BM;
                nr = nr < N ? nr : N - 1;
                src[i] = W + (int64_t)nr * G.ldw;
            }
        }
    }
    const int k_total = K * G.n_seg;
    auto load_tile = [&](int kt) {
        const int kk = kt * BK + (tid & 7) * EPC;   // a 16-byte chunk never straddles two segments (K % 8 == 0)
        const bool kvalid = kk < k_total;
        int seg = 0, kin = kk;
        if (G.n_seg > 1) {
            seg = kk / K;
            kin = kk - seg * K;
        }
        const int64_t a_off = seg * G.a_seg_stride + kin;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const bool is_a = (tid + 256 * i) < BM * 8;
            rg[i] = (src[i] != nullptr && kvalid) ? *reinterpret_cast<const uint4*>(src[i] + (is_a ? a_off : (int64_t)kin)) : make_uint4(0, 0, 0, 0);
        }
    };
    f32x4 acc[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = (k_total + BK - 1) / BK;
    load_tile(0);

    // epilogue operands that do not depend on the accumulators: requested now, they arrive under the main loop
    const int m = m0 + r;
    NormEpilogue<T, NI, true, true> epi;
    epi.prefetch(G, m < M ? m : M - 1, wave * WTN, g);

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
    epi.finish(G, acc, m, wave * WTN, r, g, wave, L.eps, reinterpret_cast<float*>(smem));
}

// ---------------------------------------------------------------------------------------------- exchange tail (three Linear layers, 16 rows)
// One field's exchange stage after its cross-attention launch, sea_exchange_tail (include/sea_hip.h):
//   stage 1  g = sum_s gelu(att_s . Wp_s^T)     per segment a [16 x D] x [D x D] product; the GELU outputs are summed in fp32 (cross_up is linear and
//                                               shared by the segments: sum_s g_s Wup^T = (sum_s g_s) Wup^T) and packed to bf16 once, as the A rows of stage 2
//   stage 2  x += g . Wup^T + S bup             x (fp32) updated in place; its bf16 copy goes to LDS as the A tile of stage 3
//   stage 3  y = AdaLN(x . Wdown^T + bdown)     rows normalised by NormEpilogue
// EVERY operand matrix goes L2 -> LDS by global_load_lds bursts (full 128-byte lines, no VGPRs): a probe build showed that MFMA fragments loaded
// straight from global memory — 16 rows x 64 B per wave-instruction — cost ~4x their bytes in the CU's memory pipeline (Wdown alone +1.6 us, the
// kernel 11.9 us against 7.7 without stage 3).  att_s, Wp_s and Wup are requested at once up front; Wdown takes over the Wp region as soon as
// stage 1 has read it and lands under stage 2.
// A workgroup = 4 waves owns 16 rows; wave w owns the column quarter w of every stage's output.
// LDS: R1 = S (D/64) K-tiles of (16 + D) rows (att_s | Wp_s; later D/.. Wdown K-tiles), R2 = (D/64) K-tiles of (16 + E) rows (g | Wup), the bf16 x
// tile [16, E], 512 B for the statistics: 152 KB at D = 128, E = 256, S = 2.
struct XTailLaunch {
    SeaExchangeTail p[SEA_XTAIL_MAX_GROUPS];   // grid.y = group (field)
    float eps;
};

template <int D, int E>
struct XTailCfg {
    static constexpr int BM = 16, BKB = 128, BK = 64;
    static constexpr int KT1 = D / BK;                       // K-tiles of stage 1 (per segment) and of stage 2
    static constexpr int KT3 = E / BK;                       // K-tiles of stage 3
    static constexpr int ST1 = (BM + D) * BKB;               // stage-1 tile: att rows, then Wp rows
    static constexpr int ST2 = (BM + E) * BKB;               // stage-2 tile: g rows, then Wup rows
    static constexpr int ST3 = D * BKB;                      // stage-3 tile: Wdown rows (the A rows live in the x tile)
    static constexpr int SMAX = D == 128 ? 2 : 4;
    static __host__ __device__ constexpr int r1_bytes(int S) { return S * KT1 * ST1 > KT3 * ST3 ? S * KT1 * ST1 : KT3 * ST3; }
    static __host__ __device__ constexpr int lds_bytes(int S) { return r1_bytes(S) + KT1 * ST2 + BM * E * 2 + 512; }
    // LDS-DMA destinations: burst 1 fills R1 up to S KT1 ST1 <= r1_bytes(S) and R2 (from r1_bytes(S)) up to KT1 ST2; burst 2 (Wdown) the first KT3 ST3
    // <= r1_bytes(S) bytes: all inside lds_bytes(S), which the host requests for the launch's S (every group of a launch has the same n_seg)
    static_assert(lds_bytes(SMAX) <= 160 * 1024 && KT3 * ST3 <= r1_bytes(1), "LDS request of sea_exchange_tail");
};

template <int D, int E>
__global__ __launch_bounds__(256) void exchange_tail_kernel(const XTailLaunch L) {
    using T = __bf16;
    using C = XTailCfg<D, E>;
    constexpr int BM = C::BM, BKB = C::BKB, BK = C::BK, KT1 = C::KT1, KT3 = C::KT3, ST1 = C::ST1, ST2 = C::ST2, ST3 = C::ST3, SMAX = C::SMAX;
    constexpr int KT2 = KT1;                                       // K-tiles of stage 2
    constexpr int NI1 = D / 64, NI2 = E / 64, NI3 = D / 64;       // 16-column blocks per wave in each stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SeaExchangeTail& P = L.p[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * BM, M = P.M, S = P.n_seg;
    if (m0 >= M) return;   // block-uniform (groups of different row counts share the grid)
    const int r1b = C::r1_bytes(S);
    const int r2b = KT2 * ST2;
    char* R1 = smem;
    char* R2 = smem + r1b;
    char* x3 = R2 + r2b;
    float* red = reinterpret_cast<float*>(x3 + BM * E * 2);
    const int m = m0 + r;
    const bool mok = m < M;
    const int mc = mok ? m : M - 1;

    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const unsigned r2_base = lds_base + (unsigned)r1b;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int rl = lane >> 3;
    const int chunk = (lane & 7) ^ (rl & 7);   // swizzle on the source side: LDS position p of a row holds chunk p ^ (row & 7)
    // ---- burst 1: att_s | Wp_s K-tiles (R1) and the Wup rows of the stage-2 K-tiles (R2)
    for (int s = 0; s < S; ++s) {
        const T* att = static_cast<const T*>(P.att[s]);
        const T* Wp = static_cast<const T*>(P.Wp[s]);
        for (int kt = 0; kt < KT1; ++kt) {
            const int k0 = kt * BK + chunk * 8;
            for (int u = wv; u < (BM + D) / 8; u += 4) {
                const int row = u * 8 + rl;
                const T* p;
                if (u < BM / 8) {
                    int mr = m0 + row;
                    mr = mr < M ? mr : M - 1;
                    p = att + (int64_t)mr * P.ldatt + k0;
                } else {
                    p = Wp + (int64_t)(row - BM) * P.ldwp + k0;
                }
                glds16_gn(p, lds_base + (unsigned)((s * KT1 + kt) * ST1 + u * 8 * BKB));
            }
        }
    }
    {
        const T* Wup = static_cast<const T*>(P.Wup);
        for (int kt = 0; kt < KT2; ++kt)
            for (int u = wv; u < E / 8; u += 4)
                glds16_gn(Wup + (int64_t)(u * 8 + rl) * P.ldwup + kt * BK + chunk * 8, r2_base + (unsigned)(kt * ST2 + (BM + u * 8) * BKB));
    }
    // ---- epilogue operands of stages 2 and 3 (a few KB per workgroup, ordinary loads)
    float bv2[NI2][4], rv2[NI2][4];
#pragma unroll
    for (int j = 0; j < NI2; ++j) {
        const int n = wave * (E / 4) + j * 16 + g * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) bv2[j][q] = 0.f;
        if (P.bup != nullptr) {
            load4(P.bup + n, bv2[j]);
#pragma unroll
            for (int q = 0; q < 4; ++q) bv2[j][q] *= P.bias_scale;
        }
        load4(P.X + (int64_t)mc * P.ldx + n, rv2[j]);
    }
    NormEpilogue<T, NI3, true, true> epi3;
    if (P.has_down) epi3.prefetch(P.down, mc, wave * (D / 4), g);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA bursts are not tracked by the compiler
    __syncthreads();
    {
    // ---- stage 1: g = sum_s gelu(att_s . Wp_s^T), this wave's D/4 columns
    float gsum[NI1][4];
#pragma unroll
    for (int jb = 0; jb < NI1; ++jb)
#pragma unroll
        for (int q = 0; q < 4; ++q) gsum[jb][q] = 0.f;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
        if (s < S) {
            f32x4 acc1[NI1];
#pragma unroll
            for (int jb = 0; jb < NI1; ++jb) acc1[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KT1; ++kt) {
                const char* sA = R1 + (s * KT1 + kt) * ST1 + r * BKB;
                const char* sB = R1 + (s * KT1 + kt) * ST1 + (BM + wave * (D / 4) + r) * BKB;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
                    const uint4 af = *reinterpret_cast<const uint4*>(sA + off);
#pragma unroll
                    for (int jb = 0; jb < NI1; ++jb) mma16<T>(*reinterpret_cast<const uint4*>(sB + jb * 16 * BKB + off), af, acc1[jb]);
                }
            }
#pragma unroll
            for (int jb = 0; jb < NI1; ++jb)
#pragma unroll
                for (int q = 0; q < 4; ++q) gsum[jb][q] += gelu_erf(acc1[jb][q]);
        }
    }
#pragma unroll
    for (int jb = 0; jb < NI1; ++jb) {
        const int kk = wave * (D / 4) + jb * 16 + g * 4;      // this lane's 4 consecutive contraction indices of stage 2, row r
        const int kt = kk / BK, cc = kk % BK;
        store4(reinterpret_cast<T*>(R2 + kt * ST2 + r * BKB + (((cc >> 3) ^ (r & 7)) << 4) + (cc & 7) * 2), gsum[jb][0], gsum[jb][1], gsum[jb][2], gsum[jb][3]);
    }
    __syncthreads();   // g is in place; nobody reads R1 any more
    }
    auto burst_wdown = [&]() {   // Wdown K-tiles to the front of LDS (over R1)
        const T* Wd = static_cast<const T*>(P.down.W);
        for (int kt = 0; kt < KT3; ++kt)
            for (int u = wv; u < D / 8; u += 4)
                glds16_gn(Wd + (int64_t)(u * 8 + rl) * P.down.ldw + kt * BK + chunk * 8, lds_base + (unsigned)(kt * ST3 + u * 8 * BKB));
    };
    // ---- burst 2: landing under stage 2
    if (P.has_down) burst_wdown();
    // ---- stage 2
    f32x4 acc2[NI2];
#pragma unroll
    for (int j = 0; j < NI2; ++j) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT2; ++kt) {
        const char* sA = R2 + kt * ST2 + r * BKB;
        const char* sB = R2 + kt * ST2 + (BM + wave * (E / 4) + r) * BKB;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
            const uint4 af = *reinterpret_cast<const uint4*>(sA + off);
#pragma unroll
            for (int j = 0; j < NI2; ++j) mma16<T>(*reinterpret_cast<const uint4*>(sB + j * 16 * BKB + off), af, acc2[j]);
        }
    }
    float v2[NI2][4];
#pragma unroll
    for (int j = 0; j < NI2; ++j) {
        const int n = wave * (E / 4) + j * 16 + g * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) v2[j][q] = acc2[j][q] + bv2[j][q] + rv2[j][q];
        store4(reinterpret_cast<T*>(x3 + (n / BK) * (BM * BKB) + r * BKB + ((((n % BK) >> 3) ^ (r & 7)) << 4) + (n & 7) * 2), v2[j][0], v2[j][1], v2[j][2], v2[j][3]);
    }
    // Wdown has landed before anything else of this wave is put into the memory pipeline (stores count in vmcnt too: they are issued after the wait)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (mok) {
#pragma unroll
        for (int j = 0; j < NI2; ++j) {
            const int n = wave * (E / 4) + j * 16 + g * 4;
            store4(P.X + (int64_t)m * P.ldx + n, v2[j][0], v2[j][1], v2[j][2], v2[j][3]);
            if (P.Xact != nullptr) store4(static_cast<T*>(P.Xact) + (int64_t)m * P.ldxact + n, v2[j][0], v2[j][1], v2[j][2], v2[j][3]);
        }
    }
    if (!P.has_down) return;   // block-uniform
    __syncthreads();
    // ---- stage 3
    f32x4 acc3[NI3];
#pragma unroll
    for (int jb = 0; jb < NI3; ++jb) acc3[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT3; ++kt) {
        const char* sA = x3 + kt * (BM * BKB) + r * BKB;
        const char* sB = R1 + kt * ST3 + (wave * (D / 4) + r) * BKB;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
            const uint4 af = *reinterpret_cast<const uint4*>(sA + off);
#pragma unroll
            for (int jb = 0; jb < NI3; ++jb) mma16<T>(*reinterpret_cast<const uint4*>(sB + jb * 16 * BKB + off), af, acc3[jb]);
        }
    }
    epi3.finish(P.down, acc3, m, wave * (D / 4), r, g, wave, L.eps, red, nullptr);
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
        SEA_REQUIRE(G.M >= 1 && G.N >= 16 && G.N % 16 == 0 && G.N <= 256 && G.K >= 8 && G.K % 8 == 0 && G.n_seg >= 1,
                    "sea_gemm_rownorm[%d]: bad shape M=%d N=%d K=%d n_seg=%d (N a multiple of 16 up to 256)", i, G.M, G.N, G.K, G.n_seg);
        SEA_REQUIRE(G.lda % epc == 0 && G.ldw % epc == 0 && G.lda >= G.K && G.ldw >= G.K && G.a_seg_stride % epc == 0, "sea_gemm_rownorm[%d]: bad operand strides lda=%d ldw=%d", i, G.lda, G.ldw);
        SEA_REQUIRE((!G.R || (G.ldr % 4 == 0 && G.ldr >= G.N)) && (!G.C32 || (G.ldc32 % 4 == 0 && G.ldc32 >= G.N)) && (!G.Y32 || (G.ldy32 % 4 == 0 && G.ldy32 >= G.N)) &&
                        (!G.Yact || (G.ldyact % 4 == 0 && G.ldyact >= G.N)) && (!G.Cact || (G.ldcact % 4 == 0 && G.ldcact >= G.N)) && (!G.mod || (G.ldmod % 4 == 0 && G.ldmod >= 2 * G.N)),
                    "sea_gemm_rownorm[%d]: bad output / modulation strides", i);
        SEA_REQUIRE(sea_aligned16(G.A) && sea_aligned16(G.W) && sea_aligned16(G.bias) && sea_aligned16(G.R) && sea_aligned16(G.C32) && sea_aligned16(G.Cact) && sea_aligned16(G.mod) &&
                        sea_aligned16(G.gamma) && sea_aligned16(G.beta) && sea_aligned16(G.Y32) && sea_aligned16(G.Yact),
                    "sea_gemm_rownorm[%d]: pointers must be 16-byte aligned", i);
        if (G.ib_c != nullptr)
            SEA_REQUIRE(G.ib_w1 && G.ib_b1 && G.ib_lnw && G.ib_lnb && G.ib_w2 && G.ib_b2 && G.ib_h >= 4 && G.ib_h <= SEA_IB_FUSED_MAX_H && G.ib_h % 4 == 0 &&
                            sea_aligned16(G.ib_w2) && sea_aligned16(G.ib_b2),
                        "sea_gemm_rownorm[%d]: info-bottleneck addend needs all parameters, h in {4, 8}, w2 / b2 16-byte aligned (h=%d)", i, G.ib_h);
        nmax = G.N > nmax ? G.N : nmax;
        L.g[i] = G;
        L.tile_start[i] = total;
        total += (G.M + 63) / 64;
    }
    // 16-row tiles while the 64-row launch would leave most CUs without a workgroup (measured at cfg2: see DESIGN.md); SEA_TUNE=gemm_norm_rows=16|64 forces
    static const int forced = sea_tune("gemm_norm_rows", 0);
    const bool small = forced == 16 || (forced != 64 && total <= 512);
    if (small) {
        total = 0;
        for (int i = 0; i < n_groups; ++i) {
            L.tile_start[i] = total;
            total += (groups[i].M + 15) / 16;
        }
    }
    // whole-contraction LDS-DMA burst: every group's contraction is 1..4 whole K-tiles (128 bytes of K per row each); SEA_TUNE=gemm_norm_dma=0 disables
    static const int dma_on = sea_tune("gemm_norm_dma", 1);
    int dma_nk = 0;
    if (small && dma_on && total <= 256) {   // one workgroup per CU (its LDS is the whole contraction): only while the launch is a single round
        const int bk = dtype == SEA_BF16 ? 64 : 32;
        for (int i = 0; i < n_groups; ++i) {
            const long kt = (long)groups[i].K * groups[i].n_seg;
            const int nk = (groups[i].K % bk == 0 && kt / bk <= 4) ? (int)(kt / bk) : -1;
            if (nk < 0) { dma_nk = -1; break; }
            dma_nk = nk > dma_nk ? nk : dma_nk;
        }
        if (dma_nk < 0) dma_nk = 0;
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
        static int once = set_lds_gn(gemm_rownorm16_kernel<TT, BNN, false>, lds_);              \
        static int once2 = set_lds_gn(gemm_rownorm16_kernel<TT, BNN, true>, 4 * lds_);          \
        (void)once; (void)once2;                                                                \
        if (dma_nk > 0) gemm_rownorm16_kernel<TT, BNN, true><<<dim3(total), dim3(256), dma_nk * lds_, s>>>(L);  \
        else gemm_rownorm16_kernel<TT, BNN, false><<<dim3(total), dim3(256), lds_, s>>>(L);      \
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

extern "C" int sea_exchange_tail(const SeaExchangeTail* params, int n_groups, float eps, int dtype, void* stream) {
    SEA_REQUIRE(params != nullptr && n_groups >= 1 && n_groups <= SEA_XTAIL_MAX_GROUPS, "sea_exchange_tail: n_groups=%d out of range", n_groups);
    XTailLaunch L;
    memset(&L, 0, sizeof(L));
    L.eps = eps;
    const int D = params[0].D, E = params[0].E, S0 = params[0].n_seg;
    int m_max = 0;
    for (int gi = 0; gi < n_groups; ++gi) {
        const SeaExchangeTail& P = params[gi];
        const bool shape_ok = (P.D == 128 && P.E == 256) || (P.D == 64 && P.E == 128);
        const bool seg_ok = P.n_seg >= 1 && P.n_seg <= SEA_XTAIL_MAX_SEG && P.n_seg * P.D <= 256;   // stage-1 tiles of all segments share LDS with Wup
        if (dtype != SEA_BF16 || !shape_ok || !seg_ok) {
            sea_set_error("sea_exchange_tail: unsupported dtype / shape (dtype=%d D=%d E=%d n_seg=%d): bf16, (D,E) in {(128,256),(64,128)}, n_seg*D <= 256", dtype, P.D, P.E, P.n_seg);
            return SEA_EUNSUPPORTED;
        }
        SEA_REQUIRE(P.D == D && P.E == E && P.n_seg == S0, "sea_exchange_tail[%d]: the groups of a launch share their shape", gi);
        SEA_REQUIRE(P.M >= 1 && P.Wup && P.X && sea_aligned16(P.Wup) && sea_aligned16(P.bup) && sea_aligned16(P.X) && sea_aligned16(P.Xact), "sea_exchange_tail[%d]: null / misaligned pointer", gi);
        SEA_REQUIRE(P.ldatt % 8 == 0 && P.ldatt >= P.D && P.ldwp % 8 == 0 && P.ldwp >= P.D && P.ldwup % 8 == 0 && P.ldwup >= P.D && P.ldx % 4 == 0 && P.ldx >= P.E &&
                        (!P.Xact || (P.ldxact % 4 == 0 && P.ldxact >= P.E)), "sea_exchange_tail[%d]: bad strides", gi);
        for (int s = 0; s < P.n_seg; ++s)
            SEA_REQUIRE(P.att[s] && sea_aligned16(P.att[s]) && P.Wp[s] && sea_aligned16(P.Wp[s]), "sea_exchange_tail[%d]: segment %d: null / misaligned operand", gi, s);
        L.p[gi] = P;
        if (P.has_down) {
            const SeaGemmNormGroup& G = P.down;
            SEA_REQUIRE(G.W && G.gamma && (G.Y32 || G.Yact) && G.ldw % 8 == 0 && G.ldw >= P.E, "sea_exchange_tail[%d]: down: null pointer or bad ldw", gi);
            SEA_REQUIRE((!G.Y32 || (G.ldy32 % 4 == 0 && G.ldy32 >= P.D)) && (!G.Yact || (G.ldyact % 4 == 0 && G.ldyact >= P.D)) && (!G.mod || (G.ldmod % 4 == 0 && G.ldmod >= 2 * P.D)),
                        "sea_exchange_tail[%d]: down: bad output / modulation strides", gi);
            SEA_REQUIRE(sea_aligned16(G.W) && sea_aligned16(G.bias) && sea_aligned16(G.mod) && sea_aligned16(G.gamma) && sea_aligned16(G.beta) && sea_aligned16(G.Y32) && sea_aligned16(G.Yact),
                        "sea_exchange_tail[%d]: down: pointers must be 16-byte aligned", gi);
            // the fields NormEpilogue reads besides the pointers checked above
            SeaGemmNormGroup& Gd = L.p[gi].down;
            Gd.M = P.M; Gd.N = P.D; Gd.K = P.E; Gd.n_seg = 1; Gd.bias_scale = 1.0f;
            Gd.R = nullptr; Gd.C32 = nullptr; Gd.Cact = nullptr; Gd.ib_c = nullptr;
        }
        m_max = P.M > m_max ? P.M : m_max;
    }
    const dim3 grid((m_max + 15) / 16, n_groups);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define LAUNCH_XT(DD, EE)                                                                                            \
    do {                                                                                                            \
        static int once = set_lds_gn(exchange_tail_kernel<DD, EE>, XTailCfg<DD, EE>::lds_bytes(XTailCfg<DD, EE>::SMAX)); \
        (void)once;                                                                                                 \
        exchange_tail_kernel<DD, EE><<<grid, dim3(256), XTailCfg<DD, EE>::lds_bytes(S0), s>>>(L);                     \
    } while (0)
    if (D == 128) LAUNCH_XT(128, 256); else LAUNCH_XT(64, 128);
#undef LAUNCH_XT
    SEA_CHECK_LAUNCH("sea_exchange_tail");
    return SEA_OK;
}
