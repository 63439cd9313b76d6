// AdaLN condition MLPs of a scalar condition in one launch (gfx950, bf16): sea_cond_mlp.
//
//     Out[m, :] = W2 . silu(w1 * c[m] + b1) + b2          (models/base_blocks.py:337-345: Linear(1, K), SiLU, Linear(K, K); K = 2d)
//
// A workgroup (8 waves) owns 32 complete rows of one module.  The hidden rows are generated straight into LDS in the MFMA operand layout
// (32 x K bf16, never in HBM: the two-launch form writes and re-reads 24 MB at cfg2), every wave then holds the fragments of all 32 rows in
// registers for the whole contraction, and W2 goes L2 -> LDS by global_load_lds in stages of 64 rows x the full contraction (K = 512: 64 KiB, a
// ring of 2; K = 256: 32 KiB, a ring of 4), one barrier per stage.  Per stage 4 of the 8 waves compute (a 16-row block of W2 each, two row
// blocks of the tile), even stages the lower four waves, odd stages the upper four — the main loop of sea_mlp_fc1_ln_gelu with a generated
// operand.  The (scale | shift) rows leave through LDS as whole rows.  The information-bottleneck layers ride in the first workgroups of
// the grid, one wave per row, as in sea_silu_outer_ib.
#include "gemm_core.hpp"
#include "ib_rows.hpp"
#include <stdlib.h>

struct CondLaunch {
    SeaCondGroup g[SEA_MAX_COND_GROUPS];
    SeaIbParams ib[SEA_MAX_SILU_IB];
    const float* c;
    int M, n_groups, n_ib, ib_blocks, tiles_m;   // ib_blocks: workgroups per information-bottleneck layer (8 rows each)
};

__device__ __forceinline__ void glds16_cond(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// KT = K / 64 K-tiles (8, 4 or 2); the module is K x K
template <int KT>
__device__ __forceinline__ void cond_tile(const SeaCondGroup& G, const float* c, int M, int m0, char* smem) {
    using T = __bf16;
    constexpr int BM = 32, BKB = 128, BK = 64, NW = 8;
    constexpr int K = KT * 64, NSTAGE = K / 64, NSB = K / 128;
    constexpr int NS = KT == 8 ? 2 : 4;
    constexpr int A_BYTES = KT * BM * BKB;                 // generated hidden rows, K-tile major, chunk-swizzled
    constexpr int STAGE = KT * 64 * BKB;                   // 64 W2 rows x the whole contraction
    constexpr int LPS = KT;                                // DMA pieces (8 rows x 128 B) per wave per stage
    constexpr int SP = K * 2 + 16;                         // staging pitch of an output row
    static_assert(BM * SP <= NS * STAGE, "the output tile is staged over the ring");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int half = wave >> 2, wq = wave & 3;
    const T* W = static_cast<const T*>(G.W2);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const int rl = lane >> 3;
    const int chunk = (lane & 7) ^ (rl & 7);               // swizzle on the source side

    // ---- operands of the generated rows: this thread's 8 contraction indices (one 16-byte chunk) and the conditions of its rows
    constexpr int CPR = K / 8;                             // chunks per row
    constexpr int RPP = 512 / CPR;                         // rows per pass
    const int kc = tid % CPR, rr0 = tid / CPR;
    float w[8], b[8], cv[BM / RPP];
    load4(G.w1 + kc * 8, *reinterpret_cast<float(*)[4]>(w));
    load4(G.w1 + kc * 8 + 4, *reinterpret_cast<float(*)[4]>(w + 4));
    load4(G.b1 + kc * 8, *reinterpret_cast<float(*)[4]>(b));
    load4(G.b1 + kc * 8 + 4, *reinterpret_cast<float(*)[4]>(b + 4));
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
        const int m = m0 + rr0 + p * RPP;
        cv[p] = c[m < M ? m : M - 1];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the loads above are complete: from here on vmcnt counts ring pieces only

    auto dma_stage = [&](int s) {
        const unsigned base = lds_base + (unsigned)(A_BYTES + (s % NS) * STAGE);
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const int p = i * NW + wave;                      // piece: K-tile kt, 8 rows u
            const int kt = p >> 3, u = p & 7;
            glds16_cond(W + (int64_t)(s * 64 + u * 8 + rl) * G.ldw + kt * BK + chunk * 8, base + (unsigned)(kt * 64 * BKB + u * 8 * BKB));
        }
    };
#pragma unroll
    for (int s = 0; s < NS - 1 && s < NSTAGE; ++s) dma_stage(s);

    // ---- hidden rows under the first stages' flight: silu(w1 c + b1) as the two-launch form rounds it (fp32, then bf16)
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
        const int row = rr0 + p * RPP;
        float h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = silu_f(w[e] * cv[p] + b[e]);
        const int kt = kc >> 3, cw = kc & 7;
        T* dst = reinterpret_cast<T*>(smem + kt * BM * BKB + row * BKB + ((cw ^ (row & 7)) << 4));
        store4(dst, h[0], h[1], h[2], h[3]);
        store4(dst + 4, h[4], h[5], h[6], h[7]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint4 areg[KT][2][2];
    {
        const char* sA = smem + r * BKB;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int kcc = 0; kcc < 2; ++kcc) {
                const int off = ((kcc * 4 + g) ^ (r & 7)) << 4;
                areg[kt][kcc][0] = *reinterpret_cast<const uint4*>(sA + kt * BM * BKB + off);
                areg[kt][kcc][1] = *reinterpret_cast<const uint4*>(sA + kt * BM * BKB + 16 * BKB + off);
            }
    }

    f32x4 acc[NSB][2];
#pragma unroll
    for (int i = 0; i < NSB; ++i) {
        acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) {
        // stage s has landed once at most the pieces of the stages issued after it are outstanding
        if (s + NS - 2 < NSTAGE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + NS - 1 < NSTAGE) dma_stage(s + NS - 1);
        if ((s & 1) == half) {   // wave-uniform
            const char* sW = smem + A_BYTES + (s % NS) * STAGE + (wq * 16 + r) * BKB;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int kcc = 0; kcc < 2; ++kcc) {
                    const int off = ((kcc * 4 + g) ^ (r & 7)) << 4;
                    const uint4 wf = *reinterpret_cast<const uint4*>(sW + kt * 64 * BKB + off);
                    mma16<T>(wf, areg[kt][kcc][0], acc[s >> 1][0]);
                    mma16<T>(wf, areg[kt][kcc][1], acc[s >> 1][1]);
                }
        }
    }
    __syncthreads();   // the ring is free: the output tile goes over it

    // ---- epilogue: + b2, bf16, whole rows.  This lane: rows m0 + mb*16 + r, columns n(i) = (2 i + half) * 64 + wq * 16 + 4 g + q
    char* tile = smem + A_BYTES;
#pragma unroll
    for (int i = 0; i < NSB; ++i) {
        const int n = (2 * i + half) * 64 + wq * 16 + g * 4;
        float bv[4];
        load4(G.b2 + n, bv);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
            store4(reinterpret_cast<T*>(tile + (mb * 16 + r) * SP) + n, acc[i][mb][0] + bv[0], acc[i][mb][1] + bv[1], acc[i][mb][2] + bv[2], acc[i][mb][3] + bv[3]);
    }
    __syncthreads();
    T* Out = static_cast<T*>(G.Out);
    constexpr int CPO = K / 8;   // 16-byte chunks per output row
    for (int idx = tid; idx < BM * CPO; idx += 512) {
        const int row = idx / CPO, cc = idx - row * CPO;
        const int m = m0 + row;
        if (m < M) *reinterpret_cast<uint4*>(Out + (int64_t)m * G.ldo + cc * 8) = *reinterpret_cast<const uint4*>(tile + row * SP + cc * 16);
    }
}

__global__ __launch_bounds__(512) void cond_mlp_kernel(const CondLaunch L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bid = blockIdx.x;
    const int ib_total = L.n_ib * L.ib_blocks;
    if (bid < ib_total) {   // block-uniform: the information-bottleneck layers first (the longer dependent chains), one wave per row
        const int k = bid / L.ib_blocks;
        const int row = (bid - k * L.ib_blocks) * 8 + (int)(threadIdx.x >> 6);
        if (row < L.M) ib_store_row(L.ib[k], L.c[row], row, threadIdx.x & 63);
        return;
    }
    bid -= ib_total;
    const int gi = bid / L.tiles_m;
    const int m0 = (bid - gi * L.tiles_m) * 32;
    const SeaCondGroup& G = L.g[gi];
    if (G.K == 512) cond_tile<8>(G, L.c, L.M, m0, smem);
    else if (G.K == 256) cond_tile<4>(G, L.c, L.M, m0, smem);
    else cond_tile<2>(G, L.c, L.M, m0, smem);
}

extern "C" int sea_cond_mlp(const SeaCondGroup* groups, int n_groups, const float* c, int M, int dtype, const SeaIbParams* ibs, int n_ib, void* stream) {
    SEA_REQUIRE(groups != nullptr && c != nullptr && n_groups >= 1 && n_groups <= SEA_MAX_COND_GROUPS && M >= 1, "sea_cond_mlp: bad arguments (n_groups=%d, M=%d)", n_groups, M);
    SEA_REQUIRE(n_ib >= 0 && n_ib <= SEA_MAX_SILU_IB && (n_ib == 0 || ibs != nullptr), "sea_cond_mlp: n_ib=%d out of range", n_ib);
    bool shape_ok = dtype == SEA_BF16;
    for (int i = 0; i < n_groups && shape_ok; ++i) shape_ok = groups[i].K == 512 || groups[i].K == 256 || groups[i].K == 128;
    if (!shape_ok) {
        sea_set_error("sea_cond_mlp: unsupported dtype / width (dtype=%d): bf16, K in {128, 256, 512}", dtype);
        return SEA_EUNSUPPORTED;
    }
    CondLaunch L;
    memset(&L, 0, sizeof(L));
    for (int i = 0; i < n_groups; ++i) {
        const SeaCondGroup& G = groups[i];
        SEA_REQUIRE(G.w1 && G.b1 && G.W2 && G.b2 && G.Out, "sea_cond_mlp[%d]: null pointer", i);
        SEA_REQUIRE(G.ldw % 8 == 0 && G.ldw >= G.K && G.ldo % 8 == 0 && G.ldo >= G.K, "sea_cond_mlp[%d]: bad strides", i);
        SEA_REQUIRE(sea_aligned16(G.w1) && sea_aligned16(G.b1) && sea_aligned16(G.W2) && sea_aligned16(G.b2) && sea_aligned16(G.Out), "sea_cond_mlp[%d]: pointers must be 16-byte aligned", i);
        L.g[i] = G;
    }
    for (int k = 0; k < n_ib; ++k) {
        const SeaIbParams& P = ibs[k];
        if (P.mode == 0)
            SEA_REQUIRE(P.X[0] && sea_aligned16(P.X[0]) && P.E >= 4 && P.E % 4 == 0 && P.h >= 1 && P.h <= 64 && P.ldx >= P.E && P.ldx % 4 == 0 && P.w1 && P.b1 && P.lnw && P.lnb &&
                            P.w2 && P.b2 && sea_aligned16(P.b2), "sea_cond_mlp: ib[%d]: bad sizes / null / misaligned pointer", k);
        else
            SEA_REQUIRE((P.mode == 1 || P.mode == 2) && P.X[0] && sea_aligned16(P.X[0]) && P.E >= 8 && P.E % 8 == 0 && P.ldx >= P.E && P.ldx % 4 == 0 && P.w1 && sea_aligned16(P.w1) &&
                            (P.mode == 2 || (P.b1 && sea_aligned16(P.b1))), "sea_cond_mlp: ib[%d]: bad sizes / null / misaligned pointer (mode %d)", k, P.mode);
        L.ib[k] = P;
    }
    L.c = c; L.M = M; L.n_groups = n_groups; L.n_ib = n_ib;
    L.ib_blocks = (M + 7) / 8;
    L.tiles_m = (M + 31) / 32;
    const int total = n_ib * L.ib_blocks + n_groups * L.tiles_m;
    constexpr int lds = 8 * 32 * 128 + 2 * (8 * 64 * 128);   // K = 512: hidden rows 32 KiB + ring 2 x 64 KiB = 160 KiB (K = 256 uses 16 + 4 x 32 KiB of it)
    static int once = hipFuncSetAttribute(reinterpret_cast<const void*>(cond_mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess ? 0 : -1;
    (void)once;
    cond_mlp_kernel<<<dim3(total), dim3(512), lds, static_cast<hipStream_t>(stream)>>>(L);
    SEA_CHECK_LAUNCH("sea_cond_mlp");
    return SEA_OK;
}
