// Tile body of the grouped NT GEMM with fused epilogues (gfx950): shared by gemm.hip (sea_gemm_grouped) and chain.hip (rider tiles of sea_row_chain_riders).
#pragma once
#include "gemm_core.hpp"

// ---------------------------------------------------------------------------------------------- standard epilogue
// PLAIN: every group of the launch has act == 0 and no dropout (most launches): the epilogue is compiled without those options.
// SILUA: the A operand of every group is generated, A[m, k] = silu(silu_w1[k] * silu_c[m] + silu_b1[k]) (gemm_core.hpp, load_tile<true>).
// One BM x BN output tile (tm, tn) of group G — the body of gemm_grouped_kernel, also run by other kernels beside their own work (chain.hip: the AdaLN
// condition GEMMs of later launches ride on the CUs a row-chain launch leaves idle).  GT: SeaGemmGroup, or a struct with the same member names whose unused
// members are compile-time constants.
template <typename T, int BM, int BN, bool DMA, bool PLAIN, bool SILUA, int NSD = 4, typename GT>
__device__ __forceinline__ void gemm_tile_body(const GT& G, int tm, int tn, char* smem, int silu_lds_off) {
    using C = GemmCfg<T, BM, BN>;
    GemmMainloop<T, BM, BN> ml;
    ml.A = static_cast<const T*>(G.A);
    ml.W = static_cast<const T*>(G.W);
    ml.a_seg_stride = G.a_seg_stride;
    ml.lda = G.lda; ml.ldw = G.ldw; ml.M = G.M; ml.N = G.N; ml.K = G.K; ml.n_seg = G.n_seg;
    ml.m0 = tm * BM; ml.n0 = tn * BN;
    f32x4 acc[C::MI][C::NI];
    if constexpr (SILUA) {
        float* lw = reinterpret_cast<float*>(smem + silu_lds_off);
        for (int i = threadIdx.x; i < G.K; i += 256) {
            lw[i] = G.silu_w1[i];
            lw[G.K + i] = G.silu_b1[i];
        }
        ml.silu_c = G.silu_c;
        ml.s_w1 = lw;
        ml.s_b1 = lw + G.K;
        ml.init_silu(threadIdx.x);
        __syncthreads();
        ml.template run_single<true>(smem, acc);
    } else if constexpr (DMA) ml.template run_dma_n<NSD>(smem, acc);
    else ml.run_single(smem, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 15, g = lane >> 4;
    const float* bias = G.bias;
    const float* R = G.R;
    float* C32 = G.C32;
    T* Cact = static_cast<T*>(G.Cact);
    T* Z = static_cast<T*>(G.Z);
    const int act = G.act;
    // The activation-dtype output is the big one (fc1: [M, 8E]); per-lane 8-byte pieces at a row stride touch a quarter of a
    // 128-byte line per instruction, so it is staged through the (now free) LDS tile and written out as whole rows, 16 B per lane.
    constexpr int SP = BN * (int)sizeof(T) + 16;  // staging row pitch in bytes
    const bool staged = Cact != nullptr && (G.N % C::EPC == 0) && (G.ldcact % C::EPC == 0);
    if constexpr (PLAIN && sizeof(T) == 2) {
        // fp32-only output (the spatial decoder's [M, n_fields * n_inp] fields: 2.5 GB per rollout): staged through LDS too, in two halves
        // of the tile's row blocks (a half is BM/2 rows x BN floats, what the single-buffer LDS allocation holds), then written as whole
        // rows, 16 bytes per lane — the per-lane form (16 rows x 64 B per store instruction) ran this launch at 400 TFLOP/s against
        // 740 for the same shape with a bf16 output
        constexpr int SP32 = BN * 4 + 16;
        constexpr int HALF_ROWS = BM / 2;
        if (Cact == nullptr && C32 != nullptr && G.ldc32 % 4 == 0 && HALF_ROWS * SP32 <= (BM > BN ? BM : BN) * SP + 0) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half) __syncthreads();
#pragma unroll
                for (int ii = 0; ii < C::MI / 2; ++ii) {
                    const int i = half * (C::MI / 2) + ii;
                    const int m = ml.m0 + wm * C::WTM + i * 16 + r;
                    const int lrow = wm * (C::WTM / 2) + ii * 16 + r;   // row inside the half: wave wm contributes WTM/2 rows per half
#pragma unroll
                    for (int j = 0; j < C::NI; ++j) {
                        const int n = ml.n0 + wn * C::WTN + j * 16 + g * 4;
                        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        if (n < G.N) {
                            if (bias != nullptr) {
                                float bv[4];
                                load4(bias + n, bv);
#pragma unroll
                                for (int q = 0; q < 4; ++q) v[q] += bv[q] * G.bias_scale;
                            }
                            if (R != nullptr && m < G.M) {
                                float rv[4];
                                load4(R + (int64_t)m * G.ldr + n, rv);
#pragma unroll
                                for (int q = 0; q < 4; ++q) v[q] += rv[q];
                            }
                        }
                        *reinterpret_cast<float4*>(smem + lrow * SP32 + (wn * C::WTN + j * 16 + g * 4) * 4) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
                __syncthreads();
                constexpr int CPR32 = BN / 4;   // 16-byte chunks per staged row
                for (int idx = threadIdx.x; idx < HALF_ROWS * CPR32; idx += 256) {
                    const int lrow = idx / CPR32, cc = idx - lrow * CPR32;
                    // half-local row -> tile row: wave-row block wm' = lrow / (WTM/2), inside it ii*16 + r
                    const int wmr = lrow / (C::WTM / 2), rem = lrow - wmr * (C::WTM / 2);
                    const int m = ml.m0 + wmr * C::WTM + half * (C::WTM / 2) + rem, n = ml.n0 + cc * 4;
                    if (m < G.M && n < G.N) *reinterpret_cast<float4*>(C32 + (int64_t)m * G.ldc32 + n) = *reinterpret_cast<const float4*>(smem + lrow * SP32 + cc * 16);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
        const int n = ml.n0 + wn * C::WTN + j * 16 + g * 4;  // this lane's 4 consecutive output columns
        if (n >= G.N) continue;                               // N % 4 == 0: the 4 columns are valid together
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) {
            load4(bias + n, bv);
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] *= G.bias_scale;
        }
#pragma unroll
        for (int i = 0; i < C::MI; ++i) {
            const int m = ml.m0 + wm * C::WTM + i * 16 + r;
            if (m >= G.M) continue;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = acc[i][j][q] + bv[q];
            float df[4] = {1.f, 1.f, 1.f, 1.f};
            if (!PLAIN && G.drop.thr > 0) {
                const uint32_t w = drop_word(G.drop.seed, G.drop.stream, (uint32_t)m, (uint32_t)(n >> 2));
                const float sc = drop_scale(G.drop.thr);
#pragma unroll
                for (int q = 0; q < 4; ++q) df[q] = drop_factor(w, q, G.drop.thr, sc);
                if (G.drop.mode == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] *= df[q];
                }
            }
            if (PLAIN) {
            } else if (act == 1) {
                if (Z != nullptr) store4(Z + (int64_t)m * G.ldz + n, v[0], v[1], v[2], v[3]);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = gelu_erf(v[q]);
            } else if (act == 2) {
                float z[4];
                load4(Z + (int64_t)m * G.ldz + n, z);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] *= gelu_erf_grad(z[q]);
            }
            if (R != nullptr) {
                float rv[4];
                load4(R + (int64_t)m * G.ldr + n, rv);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += rv[q];
            }
            if (!PLAIN && G.drop.thr > 0 && G.drop.mode == 3) {  // on the COMPLETE value, residual included (PositionalEncoding: dropout(x + pe), models/base_blocks.py:370-372)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] *= df[q];
            }
            if (C32 != nullptr) store4(C32 + (int64_t)m * G.ldc32 + n, v[0], v[1], v[2], v[3]);
            if (!PLAIN && G.drop.thr > 0 && G.drop.mode == 2) {  // backward: only the copy that feeds the dropped branch is masked
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] *= df[q];
            }
            if (staged) {
                store4(reinterpret_cast<T*>(smem + (wm * C::WTM + i * 16 + r) * SP) + (wn * C::WTN + j * 16 + g * 4), v[0], v[1], v[2], v[3]);
            } else if (Cact != nullptr) {
                store4(Cact + (int64_t)m * G.ldcact + n, v[0], v[1], v[2], v[3]);
            }
        }
    }
    if (staged) {  // block-uniform
        __syncthreads();
        constexpr int CPRO = BN / C::EPC;  // 16-byte chunks per staged row
        for (int idx = threadIdx.x; idx < BM * CPRO; idx += 256) {
            const int row = idx / CPRO, cc = idx - row * CPRO;
            const int m = ml.m0 + row, n = ml.n0 + cc * C::EPC;
            if (m < G.M && n < G.N) *reinterpret_cast<uint4*>(Cact + (int64_t)m * G.ldcact + n) = *reinterpret_cast<const uint4*>(smem + row * SP + cc * 16);
        }
    }
}

// ---------------------------------------------------------------------------------------------- rider tiles
// A plain bf16 group (A, W, bias, activation-dtype output) as other kernels carry it beside their own work (chain.hip, adaln_qkv.hip): tiles of later launches'
// GEMMs that depend on nothing the carrying launch computes, run by extra workgroups on the CUs it leaves idle.
#define SEA_CHAIN_MAX_RIDERS 8
struct ChainRiderPod {
    const void* A;
    const void* W;
    const float* bias;
    void* Cact;
    int lda, ldw, ldcact, M, N, K, tile_start, pad_;
};
// what gemm_tile_body reads of a SeaGemmGroup, for a rider: A, W, bias, activation-dtype output — everything else a compile-time constant
struct ChainRiderGroup {
    static constexpr int64_t a_seg_stride = 0;
    static constexpr int n_seg = 1, act = 0, ldr = 0, ldc32 = 0, ldz = 0;
    static constexpr float bias_scale = 1.0f;
    static constexpr const float* R = nullptr;
    static constexpr float* C32 = nullptr;
    static constexpr void* Z = nullptr;
    static constexpr SeaDropout drop = {0u, 0u, 0, 0};
    static constexpr const float* silu_c = nullptr;
    static constexpr const float* silu_w1 = nullptr;
    static constexpr const float* silu_b1 = nullptr;
    const void* A;
    const void* W;
    const float* bias;
    void* Cact;
    int lda, ldw, ldcact, M, N, K;
};

