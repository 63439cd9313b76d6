// 256 x 256 tiles for the long GEMM launches (gfx950, bf16): the big-tile form of sea_gemm_grouped.
//
// The 128 x 128 tile of gemm_core.hpp gives a wave a 64 x 64 part: per 32 contraction indices it reads 4 + 4 operand fragments (8 KiB per wave) for 16 MFMAs —
// at the full MFMA rate the four waves of a CU would need 128 B/clk of LDS reads, all the LDS has, so that tile cannot pass half the matrix rate however its loads
// are pipelined (measured: 750 TFLOP/s with one workgroup per CU, 980 with two).  Here a workgroup of 8 waves owns 256 x 256 outputs, a wave
// 128 x 64 (8 x 4 MFMA blocks, 128 accumulator registers): 8 + 4 fragments for 32 MFMAs = 96 B/clk at the full rate.  K-tiles of 64 (A | W: 2 x 32 KiB per stage)
// go L2 -> LDS by global_load_lds through a two-stage ring (128 KiB: one workgroup, two waves per SIMD, per CU); tile order as gemm_grouped_kernel (XCD-contiguous,
// row tile fastest for skinny M).  Epilogue: bias, GELU / GELU' (act 1 / 2 with the pre-activation matrix Z), residual, fp32 and / or activation-dtype outputs —
// the activation-dtype output staged through LDS 128 rows at a time and stored as whole rows.  No dropout, no segments, no generated operand: sea_gemm_grouped
// keeps such launches on the 128 x 128 kernel.
#include "gemm_core.hpp"
#include <stdlib.h>

struct Gemm256Launch {
    SeaGemmGroup g[SEA_MAX_GROUPS];
    int tile_start[SEA_MAX_GROUPS + 1];
    int n_groups;
    unsigned n_major;
};

__device__ __forceinline__ void glds16_g256(const void* ubase, unsigned lane_off, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(ubase), "s"(lds_addr) : "memory");
}

template <bool PLAIN>
__global__ __launch_bounds__(512) void gemm256_kernel(const Gemm256Launch L) {
    using T = __bf16;
    constexpr int BM = 256, BN = 256, BKB = 128, BK = 64, NW = 8;
    constexpr int STAGE = (BM + BN) * BKB;   // 64 KiB
    constexpr int MI = 8, NI = 4;            // 16 x 16 blocks of a wave: 128 rows x 64 columns
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int gi = 0;
    while (gi + 1 < L.n_groups && bid >= L.tile_start[gi + 1]) ++gi;
    const SeaGemmGroup& G = L.g[gi];
    const int t = bid - L.tile_start[gi];
    const int tiles_n = (G.N + BN - 1) / BN;
    int tm = t / tiles_n, tn = t - tm * tiles_n;
    if ((L.n_major >> gi) & 1u) {   // block-uniform
        const int tiles_m = (G.M + BM - 1) / BM;
        tn = t / tiles_m;
        tm = t - tn * tiles_m;
    }
    const int m0 = tm * BM, n0 = tn * BN, M = G.M, N = G.N, K = G.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, g = lane >> 4;
    const int rl = lane >> 3, chunk = (lane & 7) ^ (rl & 7);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const T* A = static_cast<const T*>(G.A);
    const T* W = static_cast<const T*>(G.W);
    // a stage: 32 pieces (8 rows x 128 B) of A, 32 of W; wave w issues pieces w, w + 8, w + 16, w + 24 of each.  Rows past the matrix are clamped to its last row
    // (their products land in accumulator rows / columns the epilogue never stores).
    unsigned a_off[4], w_off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int ra = m0 + (u * NW + wave) * 8 + rl;
        ra = ra < M ? ra : M - 1;
        int rw = n0 + (u * NW + wave) * 8 + rl;
        rw = rw < N ? rw : N - 1;
        a_off[u] = (unsigned)(((int64_t)ra * G.lda + chunk * 8) * 2);
        w_off[u] = (unsigned)(((int64_t)rw * G.ldw + chunk * 8) * 2);
    }
    auto dma_stage = [&](int kt) {
        const unsigned st = lds_base + (unsigned)((kt & 1) * STAGE);
        const T* ak = A + kt * BK;
        const T* wk = W + kt * BK;
#pragma unroll
        for (int u = 0; u < 4; ++u) glds16_g256(ak, a_off[u], st + (unsigned)((u * NW + wave) * 1024));
#pragma unroll
        for (int u = 0; u < 4; ++u) glds16_g256(wk, w_off[u], st + (unsigned)(BM * BKB + (u * NW + wave) * 1024));
    };
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = K / BK;
    dma_stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // stage kt has landed (this wave's pieces); the fragment reads of stage kt - 1 are retired
        __builtin_amdgcn_s_barrier();                                 // ... every wave's; nobody still reads the other stage
        if (kt + 1 < nk) dma_stage(kt + 1);
        const char* sA = smem + (kt & 1) * STAGE + (wm * 128 + r) * BKB;
        const char* sW = smem + (kt & 1) * STAGE + BM * BKB + (wn * 64 + r) * BKB;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
            uint4 bf[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const uint4*>(sW + j * 16 * BKB + off);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const uint4 af = *reinterpret_cast<const uint4*>(sA + i * 16 * BKB + off);
#pragma unroll
                for (int j = 0; j < NI; ++j) mma16<T>(bf[j], af, acc[i][j]);
            }
        }
    }
    __syncthreads();
    // ---- epilogue: acc[i][j][q] = C[m0 + wm * 128 + 16 i + r][n0 + wn * 64 + 16 j + 4 g + q]
    const float* bias = G.bias;
    const float* R = G.R;
    float* C32 = G.C32;
    T* Cact = static_cast<T*>(G.Cact);
    T* Z = static_cast<T*>(G.Z);
    const int act = G.act;
    float bv[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * 64 + j * 16 + g * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[j][q] = 0.f;
        if (bias != nullptr && n < N) {
            load4(bias + n, bv[j]);
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[j][q] *= G.bias_scale;
        }
    }
    constexpr int SP = BN * 2 + 16;   // staging row pitch (bytes)
    const bool staged = Cact != nullptr && (N % 8 == 0) && (G.ldcact % 8 == 0);
    // An activation-dtype output with a fp32 RESIDUAL and nothing else (mlp.fc2 of the training forward: x + MLP(x) -> bf16) leaves through a path of its own:
    // all eight waves stage their 128 x 64 parts at once (256 rows x 528 B: the ring's 128 KiB + 4 KiB), one barrier, and the copy-out — whole rows, 16 bytes of
    // output per lane — reads the residual COALESCED (32 bytes per lane, 1 KB of a row per 32 lanes) and adds it there.  In the general path below a lane reads the
    // residual in the accumulator layout (16 rows x 64 B per instruction, 32 instructions): 50 MB as half cache lines, 33 of the launch's 98 us (the same GEMM
    // without a residual, bwd.fc1.dgrad, takes 65).  The sum is rounded twice (the staged product to bf16, then the sum): the product's rounding error is 2^-9 of the
    // MLP term, below the 2^-9 of the sum that the output's own rounding costs either way.
    if (PLAIN && staged && R != nullptr && C32 == nullptr && (G.ldr % 4 == 0)) {   // block-uniform
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            T* rowp = reinterpret_cast<T*>(smem + (wm * 128 + i * 16 + r) * SP) + (wn * 64 + g * 4);
#pragma unroll
            for (int j = 0; j < NI; ++j) store4(rowp + j * 16, acc[i][j][0] + bv[j][0], acc[i][j][1] + bv[j][1], acc[i][j][2] + bv[j][2], acc[i][j][3] + bv[j][3]);
        }
        __syncthreads();
        const int64_t ldc = G.ldcact, ldr = G.ldr;
#pragma unroll 2
        for (int idx = tid; idx < BM * (BN / 8); idx += 512) {
            const int row = idx >> 5, cc = idx & 31;
            const int m = m0 + row, n = n0 + cc * 8;
            if (m < M && n < N) {
                T pv[8];
                *reinterpret_cast<uint4*>(pv) = *reinterpret_cast<const uint4*>(smem + row * SP + cc * 16);
                float r0[4], r1[4];
                load4(R + m * ldr + n, r0);
                load4(R + m * ldr + n + 4, r1);
                store4(Cact + m * ldc + n, (float)pv[0] + r0[0], (float)pv[1] + r0[1], (float)pv[2] + r0[2], (float)pv[3] + r0[3]);
                store4(Cact + m * ldc + n + 4, (float)pv[4] + r1[0], (float)pv[5] + r1[1], (float)pv[6] + r1[2], (float)pv[7] + r1[3]);
            }
        }
        return;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {   // the activation-dtype rows of waves wm == half leave through LDS: 128 rows x 256 columns
        if (half) __syncthreads();
        if (wm == half || !staged) {         // (!staged: both halves in the first pass; the second pass then has nothing to do)
            if (staged || half == 0) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int m = m0 + wm * 128 + i * 16 + r;
                    if (m >= M) continue;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int n = n0 + wn * 64 + j * 16 + g * 4;
                        if (n >= N) continue;
                        float v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][q] + bv[j][q];
                        if (!PLAIN) {
                            if (act == 1) {
                                if (Z != nullptr) store4(Z + (int64_t)m * G.ldz + n, v[0], v[1], v[2], v[3]);
#pragma unroll
                                for (int q = 0; q < 4; ++q) v[q] = gelu_erf(v[q]);
                            } else if (act == 2) {
                                float z[4];
                                load4(Z + (int64_t)m * G.ldz + n, z);
#pragma unroll
                                for (int q = 0; q < 4; ++q) v[q] *= gelu_erf_grad(z[q]);
                            }
                        }
                        if (R != nullptr) {
                            float rv[4];
                            load4(R + (int64_t)m * G.ldr + n, rv);
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] += rv[q];
                        }
                        if (C32 != nullptr) store4(C32 + (int64_t)m * G.ldc32 + n, v[0], v[1], v[2], v[3]);
                        if (staged) store4(reinterpret_cast<T*>(smem + (i * 16 + r) * SP) + (wn * 64 + j * 16 + g * 4), v[0], v[1], v[2], v[3]);
                        else if (Cact != nullptr) store4(Cact + (int64_t)m * G.ldcact + n, v[0], v[1], v[2], v[3]);
                        if (!PLAIN) __builtin_amdgcn_sched_barrier(0);   // (one block's GELU temporaries at a time: interleaved, 32 blocks of them spill beside 128 accumulator registers)
                    }
                }
            }
        }
        if (staged) {   // block-uniform
            __syncthreads();
            for (int idx = tid; idx < 128 * (BN / 8); idx += 512) {
                const int row = idx >> 5, cc = idx & 31;
                const int m = m0 + half * 128 + row, n = n0 + cc * 8;
                if (m < M && n < N) *reinterpret_cast<uint4*>(Cact + (int64_t)m * G.ldcact + n) = *reinterpret_cast<const uint4*>(smem + row * SP + cc * 16);
            }
        }
    }
}

// Launches the big-tile kernel for groups sea_gemm_grouped has validated; returns false (nothing launched) when a group needs a feature it does not have.
bool sea_gemm256_try(const SeaGemmGroup* groups, int n_groups, unsigned n_major, hipStream_t s) {
    static const bool force_act = sea_tune("gemm256", -1) == 1;
    Gemm256Launch L;
    memset(&L, 0, sizeof(L));
    int total = 0;
    bool plain = true;
    for (int i = 0; i < n_groups; ++i) {
        const SeaGemmGroup& G = groups[i];
        if (G.silu_c != nullptr || G.n_seg != 1 || G.drop.thr != 0 || G.K % 64 != 0 || G.N % 4 != 0 || G.A == nullptr) return false;
        if ((int64_t)G.M * G.lda * 2 >= (1ll << 32) || (int64_t)G.N * G.ldw * 2 >= (1ll << 32)) return false;   // 32-bit operand offsets
        plain = plain && G.act == 0;
        if (G.M < 2048 && !force_act) return false;    // (M = 796 of the shipped configurations is 3.1 row tiles of 256: a fifth of the tile rows would be padding)
        if (G.act != 0 && !force_act) return false;   // (the GELU epilogues of this tile spill at 256 registers: such launches stay on the 128 x 128 kernel unless forced)
        L.g[i] = G;
        L.tile_start[i] = total;
        total += ((G.M + 255) / 256) * ((G.N + 255) / 256);
    }
    L.tile_start[n_groups] = total;
    L.n_groups = n_groups;
    L.n_major = n_major;
    constexpr int lds = 256 * (256 * 2 + 16);   // the ring (2 x 64 KiB) and, over it, the residual path's 256 staged rows of 528 B
    static const hipError_t o1 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    static const hipError_t o2 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)o1; (void)o2;
    if (plain) gemm256_kernel<true><<<dim3(total), dim3(512), lds, s>>>(L);
    else gemm256_kernel<false><<<dim3(total), dim3(512), lds, s>>>(L);
    return true;
}
