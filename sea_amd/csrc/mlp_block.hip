// The whole field MLP, proj and the final norm in ONE launch (gfx950, bf16): sea_mlp_block.
//
//     a   = AdaLN_2 / LayerNorm(x + ib)   (optional prologue, as sea_mlp_fc1_ln_gelu)        (models/temporal.py:139-145)
//     h   = gelu_erf(LayerNorm_S(a . W1^T + b1) * lnw + lnb)                                 (models/base_blocks.py:22-24)
//     x3  = h . W2^T + b2 + R                                                                (models/base_blocks.py:25; the residual of models/temporal.py:145)
//     y   = x3 . Wproj^T + bproj ; out = norm(y)                                             (models/temporal.py:146, 412-415)
//
// sea_mlp_fc1_ln_gelu + sea_mlp_fc2_proj_norm were two launches with the activated hidden rows (25 MB at cfg2) written by the first and streamed back by the
// second, 128 KiB per workgroup each way, and a launch boundary between two kernels that both run one workgroup per CU.  Here a workgroup (8 waves) owns its 32
// rows from the fp32 residual stream to the output rows and the hidden rows never leave its REGISTERS:
//
//   phase 1 = the fc1 kernel's main loop (W1 through a 4-stage LDS-DMA ring, the 32 operand rows as resident MFMA fragments), with ONE change: which hidden
//     column an MFMA row of a stage holds.  The accumulator layout of the transposed tile gives a lane 4 consecutive MFMA rows of a 16-row block; the W1 rows
//     DMA'd to a stage are chosen so that a lane's rows of block 2t are hidden columns c + 8g + {0..3} and those of block 2t + 1 are c + 8g + {4..7}
//     (c = col_base(t, wave)): after LayerNorm + GELU the two blocks, packed to bf16, ARE the lane's 16-byte operand fragment of the second layer for the 32
//     contraction indices c .. c + 31 — no shuffle, no LDS round trip.
//   phase 2 = fc2 with the contraction split over the waves: wave w multiplies its own 256 hidden columns (its fragments) with the matching 256 columns of W2,
//     whose fragments it loads from L2 straight into registers (no other wave needs them; 16 loads of 16 bytes per lane in flight), 128 output columns at a
//     time; the eight partial [32 x 128] tiles meet in LDS (a slab per wave, summed in wave order: deterministic), + b2 + residual -> the x3 tile (bf16, LDS).
//   phase 3 = proj from the x3 tile (Wproj fragments requested before the last reduction) and the row norm, as sea_mlp_fc2_proj_norm's epilogue.
#include "gemm_core.hpp"
#include <stdlib.h>

struct MlpBlockLaunch {
    SeaMlpGroup g1[SEA_MAX_MLP_GROUPS];
    SeaMlp2Group g2[SEA_MAX_MLP_GROUPS];
    int tile_start[SEA_MAX_MLP_GROUPS + 1];
    int xcd_start[SEA_MAX_MLP_GROUPS + 1];
    int n_groups;
    float eps;
    int per_xcd;
    int probe;   // development: SEA_TUNE=blk_probe=n ends every workgroup after stage n (1 main loop of the first layer, 2 its epilogue, 3 the second layer's loop, 4 its reduction); outputs are then not written
};

__device__ __forceinline__ void glds16_blk(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// ... with the address as a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: no 64-bit address registers per piece
__device__ __forceinline__ void glds16_blk_s(const void* ubase, unsigned lane_off, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(ubase), "s"(lds_addr) : "memory");
}

template <int KT, int NSB>
struct MlpBlockCfg {
    static constexpr int BM = 32, BKB = 128, NS = 4, NW = 8;
    static constexpr int E = KT * 64, S = NSB * 128;
    static constexpr int A_BYTES = KT * BM * BKB, STAGE = KT * 64 * BKB;
    static constexpr int PRM_OFF = A_BYTES + NS * STAGE;
    static constexpr int P1_BYTES = PRM_OFF + 2 * S * 4;                      // phase 1: A rows | ring | lnw | lnb
    static constexpr int SLAB_PITCH = 128 * 4 + 16, SLAB = BM * SLAB_PITCH;   // a wave's partial [32 x 128] f32 tile
    static constexpr int X3_OFF = NW * SLAB;                                  // the x3 tile: KT K-tiles of 32 rows x 128 B, swizzled like the A rows
    static constexpr int RED2_OFF = X3_OFF + KT * BM * BKB;                   // statistics of the final norm
    static constexpr int P2_BYTES = RED2_OFF + 2 * NW * BM * 4;
    // second layer: a private ring of W2 slots per wave (a slot = 16 output rows x 128 B of the wave's hidden columns = 2 KiB).  E = 256: eight slots per wave
    // exactly where the W1 ring was (dead once every wave has left the first layer's main loop; lnw | lnb behind it stay); E = 128: four slots behind lnw | lnb
    static constexpr int RS = KT == 4 ? 8 : 4, SLOT = 16 * 128;
    static constexpr int RING_OFF = KT == 4 ? A_BYTES : P1_BYTES;
    static constexpr int RING_END = RING_OFF + NW * RS * SLOT;
    static constexpr int BYTES_ = P1_BYTES > P2_BYTES ? P1_BYTES : P2_BYTES;
    static constexpr int BYTES = BYTES_ > RING_END ? BYTES_ : RING_END;
};

// first MFMA of an accumulator chain: C is the inline constant 0 (a zero-initialised register block is 64 VGPRs the scheduler materialises early)
__device__ __forceinline__ f32x4 mma16_first(const uint4& a, const uint4& b) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}

// hidden columns [c, c + 32) of MFMA step t of wave (half, wq): steps 2 tau and 2 tau + 1 are neighbours (a wave's two W2 loads of a row share a 128-byte line)
__device__ __forceinline__ constexpr int blk_col_base(int t, int half, int wq) { return (t >> 1) * 512 + half * 256 + wq * 64 + (t & 1) * 32; }

template <int KT, int NSB>
__global__ __launch_bounds__(512) void mlp_block_kernel(const MlpBlockLaunch L) {
    using T = __bf16;
    using Cf = MlpBlockCfg<KT, NSB>;
    constexpr int BM = 32, BKB = 128, BK = 64, NS = 4, NW = 8;
    constexpr int E = Cf::E, S = Cf::S, NSTAGE = S / 64, NT = NSB / 2;
    constexpr int A_BYTES = Cf::A_BYTES, STAGE = Cf::STAGE;
    constexpr int LPS = KT * 8 / NW;
    constexpr int PRM_OFF = Cf::PRM_OFF;
    constexpr int B1_OFF = 0, RED_OFF = S * 4;
    static_assert(RED_OFF + 2 * NW * BM * 4 <= A_BYTES && (S / 256) * 1024 <= A_BYTES && LPS * NW == KT * 8 && NT % 2 == 0 && Cf::BYTES <= 160 * 1024, "LDS layout of sea_mlp_block");
    static_assert((NT / 2) * 512 == S, "blk_col_base covers the hidden columns once");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // workgroups go to the XCDs round-robin.  per_xcd > 0 (several rounds): consecutive tiles (one field: one set of weights) on one XCD.  per_xcd < 0 (one
    // round, every group fits a whole number of XCDs): group gi owns XCDs xcd_start[gi] .. — an XCD's L2 then holds ONE field's W1 | W2 | Wproj (2.1 MiB of its 4)
    int tile, gi = 0;
    if (L.per_xcd < 0) {
        const int xcd = (int)(blockIdx.x & 7), idx = (int)(blockIdx.x >> 3);
        if (xcd >= L.xcd_start[L.n_groups]) return;
        while (gi + 1 < L.n_groups && xcd >= L.xcd_start[gi + 1]) ++gi;
        tile = L.tile_start[gi] + (xcd - L.xcd_start[gi]) * 32 + idx;
        if (tile >= L.tile_start[gi + 1]) return;
    } else {
        tile = L.per_xcd > 0 ? (int)(blockIdx.x & 7) * L.per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
        if (tile >= L.tile_start[L.n_groups]) return;
        while (gi + 1 < L.n_groups && tile >= L.tile_start[gi + 1]) ++gi;
    }
    const SeaMlpGroup& G = L.g1[gi];
    const SeaMlp2Group& H = L.g2[gi];
    const int m0 = (tile - L.tile_start[gi]) * BM, M = G.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int half = wave >> 2, wq = wave & 3;
    const T* A = static_cast<const T*>(G.A);
    const T* W = static_cast<const T*>(G.W1);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const int rl = lane >> 3;
    const int chunk = (lane & 7) ^ (rl & 7);               // swizzle on the source side
    // ================================================================================================ phase 1: fc1 + LayerNorm + GELU (mlp_fused.hip's loop)
    for (int p = wave; p < 2 * (S / 256); p += NW) {       // gains / shifts of the LayerNorm straight into LDS
        const float* src = p < S / 256 ? G.lnw : G.lnb;
        const int q = p < S / 256 ? p : p - S / 256;
        glds16_blk(src + q * 256 + lane * 4, lds_base + (unsigned)(PRM_OFF + p * 1024));
    }
    for (int p = wave; G.X32 == nullptr && p < KT * (BM / 8); p += NW) {
        const int kt = p / (BM / 8), u = p - kt * (BM / 8);
        int row = m0 + u * 8 + rl;
        row = row < M ? row : M - 1;
        glds16_blk(A + (int64_t)row * G.lda + kt * BK + chunk * 8, lds_base + (unsigned)(kt * BM * BKB + u * 8 * BKB));
    }
    // stage s = 4 t + 2 u + hs: for every wave quarter wq_, MFMA row 4 g_ + q_ of block 2 t + u of the waves (hs, wq_) = hidden column col_base(t, hs, wq_) + 8 g_ + 4 u + q_
    const unsigned w1_lane = (unsigned)(((8 * (rl >> 2) + (rl & 3)) * G.ldw + chunk * 8) * 2);
    auto dma_stage = [&](int s) {
        const unsigned base = lds_base + (unsigned)(A_BYTES + (s % NS) * STAGE);
        const int t = s >> 2, u = (s >> 1) & 1, hs = s & 1;
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const int p = i * NW + wave;                      // piece: K-tile kt, 8 rows u8
            const int kt = p >> 3, u8 = p & 7;
            // row inside the wave quarter's 16: within = (u8 & 1) * 8 + rl -> W1 row col_base + 8 (within >> 2) + 4 u + (within & 3); the lane's part of that is w1_lane
            const int urow = blk_col_base(t, hs, u8 >> 1) + 16 * (u8 & 1) + 4 * u;   // uniform
            glds16_blk_s(W + (int64_t)urow * G.ldw + kt * BK, w1_lane, base + (unsigned)(kt * 64 * BKB + u8 * 8 * BKB));
        }
    };
    // (Only the first weight stage goes in front of the prologue: its tracked loads end in a vmcnt(0).  Measured and not kept, round 4: the rows' operands by untracked
    // loads with all three stages behind them and one counted wait — 45.7 against 45.2 us: the ring fills under the prologue's arithmetic either way.)
    dma_stage(0);
    if (G.X32 != nullptr) {   // the operand rows normalised in place from the fp32 residual stream (x + addend), as sea_mlp_fc1_ln_gelu's prologue (no Xout here: the residual is re-formed below)
        constexpr int CPT = E / 16;
        const int prow = tid >> 4, pl = tid & 15;
        int row = m0 + prow;
        row = row < M ? row : M - 1;
        const int c0 = pl * CPT;
        float xv[CPT];
#pragma unroll
        for (int c = 0; c < CPT; c += 4) load4(G.X32 + (int64_t)row * G.ldx32 + c0 + c, *reinterpret_cast<float(*)[4]>(xv + c));
        if (G.addend != nullptr) {
#pragma unroll
            for (int c = 0; c < CPT; c += 4) {
                float av[4];
                load4(G.addend + (int64_t)row * G.ldadd + c0 + c, av);
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[c + e] += av[e];
            }
        }
        float gq[CPT], bq[CPT];
#pragma unroll
        for (int c = 0; c < CPT; c += 4) {
            load4(G.gamma + c0 + c, *reinterpret_cast<float(*)[4]>(gq + c));
#pragma unroll
            for (int e = 0; e < 4; ++e) bq[c + e] = 0.f;
            if (G.beta != nullptr) load4(G.beta + c0 + c, *reinterpret_cast<float(*)[4]>(bq + c));
            if (G.mod != nullptr) {
                float mw[4], mb[4];
                const T* mod = static_cast<const T*>(G.mod) + (int64_t)row * G.ldmod;
                load4(mod + c0 + c, mw);
                load4(mod + E + c0 + c, mb);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gq[c + e] += 1.0f + mw[e];
                    bq[c + e] += mb[e];
                }
            }
        }
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CPT; c += 4)
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e] = add1(s4[e], xv[c + e]);
        float sm_ = add1(add1(s4[0], s4[1]), add1(s4[2], s4[3]));
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) sm_ += __shfl_xor(sm_, o);
        const float mean = sm_ * (1.0f / (float)E);
        float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CPT; c += 4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d_ = xv[c + e] - mean;
                q4[e] = fma1(d_, d_, q4[e]);
            }
        float sq = add1(add1(q4[0], q4[1]), add1(q4[2], q4[3]));
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / (float)E) + G.norm_eps);
#pragma unroll
        for (int c = 0; c < CPT; c += 8) {
            const int col = c0 + c, kt = col >> 6, ck = (col & 63) >> 3;
            bf16x8 pv;
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = (__bf16)((xv[c + e] - mean) * rstd * gq[c + e] + bq[c + e]);
            *reinterpret_cast<bf16x8*>(smem + kt * BM * BKB + prow * BKB + ((ck ^ (prow & 7)) << 4)) = pv;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    for (int s = 1; s < NS - 1; ++s) dma_stage(s);

    // slot k = (tau, j) of this wave's W2 ring: output rows 16 j .. 16 j + 15, the wave's hidden columns of steps 2 tau and 2 tau + 1 (64 columns = 128 B per row),
    // two pieces of 8 rows, swizzled on the source side like every LDS-DMA tile
    const T* W2 = static_cast<const T*>(H.W2);
    const unsigned w2_lane = (unsigned)((rl * H.ldw2 + chunk * 8) * 2);
    // ... and behind the W2 slots, NX more of the same shape with THIS wave's rows of Wproj (E / 8 output rows x E: proj's weight fragments are read from the ring
    // before the reduction takes its memory — requested as fragments from global memory, 16 rows x 64 B per instruction, they cost 6 us)
    const T* Wp = static_cast<const T*>(H.Wproj);
    const unsigned wp_lane = (unsigned)((rl * H.ldwp + chunk * 8) * 2);
    constexpr int NSLOT = (E / 16) * (NT / 2);   // W2 slots of a wave: (tau, j), j fastest
    constexpr int NJ = E / NW / 16;              // 16-column blocks of a wave in proj
    constexpr int NX = NJ * KT;                  // Wproj slots of a wave: (j, kt), kt fastest
    auto w2_slot = [&](int k) {
        const unsigned dst = lds_base + (unsigned)(Cf::RING_OFF + (wave * Cf::RS + (k % Cf::RS)) * Cf::SLOT);
        if (k < NSLOT) {
            const int tau = k / (E / 16), j = k - tau * (E / 16);
            const T* ub = W2 + (int64_t)(j * 16) * H.ldw2 + blk_col_base(2 * tau, half, wq);   // uniform
#pragma unroll
            for (int u = 0; u < 2; ++u)
                glds16_blk_s(ub + (int64_t)(8 * u) * H.ldw2, w2_lane, dst + (unsigned)(u * 1024));
        } else {
            const int kk = k - NSLOT, j = kk / KT, kt = kk - j * KT;
            const T* ub = Wp + (int64_t)(wave * (E / NW) + j * 16) * H.ldwp + kt * BK;   // uniform
#pragma unroll
            for (int u = 0; u < 2; ++u)
                glds16_blk_s(ub + (int64_t)(8 * u) * H.ldwp, wp_lane, dst + (unsigned)(u * 1024));
        }
    };
    uint4 hid[NT][2];    // the activated hidden rows of this lane as second-layer operand fragments: (step t, row block) -> bf16 x 8 = blocks 2 t (.x .y) and 2 t + 1 (.z .w)
    float mean[2], rstd[2];
    {
        f32x4 acc[NSB][2];
        uint4 areg[KT][2][2];
#pragma unroll
        for (int i = 0; i < NSB; ++i) {
            acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s) {
            if (s + 2 < NSTAGE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LPS) : "memory");
            else if (s + 1 < NSTAGE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + NS - 1 < NSTAGE) dma_stage(s + NS - 1);
            if (s == 0) {
                const char* sA = smem + r * BKB;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
                        const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
                        areg[kt][kc][0] = *reinterpret_cast<const uint4*>(sA + kt * BM * BKB + off);
                        areg[kt][kc][1] = *reinterpret_cast<const uint4*>(sA + kt * BM * BKB + 16 * BKB + off);
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (s == 1)
                for (int p = wave; p < S / 256; p += NW) glds16_blk(G.b1 + p * 256 + lane * 4, lds_base + (unsigned)(B1_OFF + p * 1024));
            if ((s & 1) == half) {   // wave-uniform
                const char* sW = smem + A_BYTES + (s % NS) * STAGE + (wq * 16 + r) * BKB;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
                        const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
                        const uint4 wf = *reinterpret_cast<const uint4*>(sW + kt * 64 * BKB + off);
                        mma16<T>(wf, areg[kt][kc][0], acc[s >> 1][0]);
                        mma16<T>(wf, areg[kt][kc][1], acc[s >> 1][1]);
                    }
            }
        }
        if (L.probe == 1) return;
        // ---- + b1, LayerNorm over the S columns of a row (two-pass, fp32), * lnw + lnb, GELU -> bf16 fragments
        // this lane: rows mb * 16 + r, block i (= 2 t + u): hidden columns col_base(t) + 8 g + 4 u + q
        float* red = reinterpret_cast<float*>(smem + RED_OFF);
        const float* pb1 = reinterpret_cast<const float*>(smem + B1_OFF);
        const float* pgm = reinterpret_cast<const float*>(smem + PRM_OFF);
        const float* pbt = pgm + S;
        const float inv_s = 1.0f / (float)S;
        float sum[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NSB; ++i) {
            const int n = blk_col_base(i >> 1, half, wq) + 8 * g + 4 * (i & 1);
            float bv[4];
            load4(pb1 + n, bv);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][mb][q] += bv[q];
                sum[mb] += (acc[i][mb][0] + acc[i][mb][1]) + (acc[i][mb][2] + acc[i][mb][3]);
            }
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            sum[mb] += __shfl_xor(sum[mb], 16);
            sum[mb] += __shfl_xor(sum[mb], 32);
            if (g == 0) red[wave * 32 + mb * 16 + r] = sum[mb];
        }
        __syncthreads();
        // every wave has left the main loop: the W1 ring is dead, the second layer's weight stream starts HERE, under the rest of this epilogue.  From now
        // until the second layer's loop has drained its ring, barriers are raw (s_waitcnt lgkmcnt(0) + s_barrier): a __syncthreads would drain the DMA queue
#pragma unroll
        for (int k = 0; k < Cf::RS - 1; ++k) w2_slot(k);
        float sq[2] = {0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[w * 32 + mb * 16 + r];
            mean[mb] = t * inv_s;
        }
#pragma unroll
        for (int i = 0; i < NSB; ++i)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float c = acc[i][mb][q] - mean[mb];
                    sq[mb] += c * c;
                }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            sq[mb] += __shfl_xor(sq[mb], 16);
            sq[mb] += __shfl_xor(sq[mb], 32);
            if (g == 0) red[256 + wave * 32 + mb * 16 + r] = sq[mb];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[256 + w * 32 + mb * 16 + r];
            rstd[mb] = 1.0f / sqrtf(t * inv_s + L.eps);
        }
#pragma unroll
        for (int i = 0; i < NSB; ++i) {
            const int n = blk_col_base(i >> 1, half, wq) + 8 * g + 4 * (i & 1);
            float gm[4], bt[4];
            load4(pgm + n, gm);
            load4(pbt + n, bt);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                bf16x4 y;
#pragma unroll
                for (int q = 0; q < 4; ++q) y[q] = (__bf16)gelu_erf_bf16((acc[i][mb][q] - mean[mb]) * rstd[mb] * gm[q] + bt[q]);
                uint2 y2 = *reinterpret_cast<const uint2*>(&y);
                asm volatile("" : "+v"(y2.x), "+v"(y2.y));   // the packed values exist HERE: left alone, the optimiser sinks the GELU's last multiply-add and the packing to the fragments' first use in the second layer, and carries twice the registers across
                if (i & 1) {
                    hid[i >> 1][mb].z = y2.x;
                    hid[i >> 1][mb].w = y2.y;
                } else {
                    hid[i >> 1][mb].x = y2.x;
                    hid[i >> 1][mb].y = y2.y;
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // block by block: the accumulators of block i are dead before block i + 1's GELU temporaries are born (interleaved, the 128 + 64 registers of the two forms spill)
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // (nothing of the second layer — its address arithmetic — is scheduled into the first layer's epilogue, which is at the register limit)
    if (L.probe == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ================================================================================================ phase 2: fc2, contraction split over the waves
    // (the W2 ring was started under the first layer's epilogue: w2_slot / the prologue above)
    constexpr int NJB = E / 16;                  // 16-column output blocks
    constexpr int NTOT = NSLOT + NX;
    f32x4 acc2[2][NJB];
    uint4 wp[KT * 2][NJ];
#pragma unroll
    for (int k = 0; k < NTOT; ++k) {
        const int tau = k / NJB, j = k - tau * NJB;
        // slot k has landed once at most the pieces of the slots issued after it are outstanding (two pieces per slot); lgkmcnt(0): the fragment reads of slot
        // k - 1 are retired before its ring position is refilled
        {
            constexpr int AFTER_MAX = Cf::RS - 2;
            const int after = NTOT - 1 - k < AFTER_MAX ? NTOT - 1 - k : AFTER_MAX;
            switch (after) {   // (constant per unrolled iteration)
                case 6: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
            }
        }
        if (k + Cf::RS - 1 < NTOT) w2_slot(k + Cf::RS - 1);
        const char* sl = smem + Cf::RING_OFF + (wave * Cf::RS + (k % Cf::RS)) * Cf::SLOT + r * 128;
        if (k >= NSLOT) {   // a Wproj slot: its two fragments per lane go to registers
            const int kk = k - NSLOT, jp = kk / KT, kt = kk - jp * KT;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) wp[kt * 2 + kc][jp] = *reinterpret_cast<const uint4*>(sl + (((kc * 4 + g) ^ (r & 7)) << 4));
            continue;
        }
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {
            const uint4 w = *reinterpret_cast<const uint4*>(sl + (((u2 * 4 + g) ^ (r & 7)) << 4));
            const int t = 2 * tau + u2;
            if (t == 0) {
                acc2[0][j] = mma16_first(w, hid[0][0]);
                acc2[1][j] = mma16_first(w, hid[0][1]);
            } else {
                mma16<T>(w, hid[t][0], acc2[0][j]);
                mma16<T>(w, hid[t][1], acc2[1][j]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // every wave is done with its ring: the slabs (and the x3 tile) go over that memory
    if (L.probe == 3) return;
    {   // ---- everything below sees the lane's coordinates through an opaque copy of the thread index: computed from the original, the optimiser evaluates the
        // reduction's and the epilogue's address arithmetic (64-bit row offsets, a dozen registers) ahead of the loop above, whose 192 registers of fragments
        // and accumulators leave no room for it — and a spill inside that loop is a scratch access in the same queue as the counted LDS-DMA pieces
    int tid_l = threadIdx.x;
    asm volatile("" : "+v"(tid_l));
    const int tid = tid_l, lane = tid & 63, r = lane & 15, g = lane >> 4;
    (void)lane;
    // ---- the reduction's own operands (this thread: row srow, 8 columns from scb of every 128-column pass), proj's weight fragments and the epilogue operands
    constexpr int NH = E / 128, JH = 8;
    const int srow = tid >> 4, scb = (tid & 15) * 8;
    float resv[NH][8];
    {
        int row = m0 + srow;
        row = row < M ? row : M - 1;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int col = h * 128 + scb;
            float b2a[4], b2b[4];
            load4(H.b2 + col, b2a);
            load4(H.b2 + col + 4, b2b);
            float ra[4], rb[4];
            if (H.R != nullptr) {
                load4(H.R + (int64_t)row * H.ldr + col, ra);
                load4(H.R + (int64_t)row * H.ldr + col + 4, rb);
            } else {   // the residual is the prologue's x (+ addend), formed again by the same fp32 add
                load4(G.X32 + (int64_t)row * G.ldx32 + col, ra);
                load4(G.X32 + (int64_t)row * G.ldx32 + col + 4, rb);
                if (G.addend != nullptr) {
                    float aa[4], ab[4];
                    load4(G.addend + (int64_t)row * G.ldadd + col, aa);
                    load4(G.addend + (int64_t)row * G.ldadd + col + 4, ab);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        ra[q] += aa[q];
                        rb[q] += ab[q];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                resv[h][q] = b2a[q] + ra[q];
                resv[h][4 + q] = b2b[q] + rb[q];
            }
        }
    }
    constexpr int CPT = E / 16;                  // columns of a thread in the final pass (16 threads per row)
    const int ec0 = (tid & 15) * CPT;
    float ebp[CPT], egq[CPT], ebq[CPT];          // bproj; gamma (+ 1 + scale); beta (+ shift)
    const bool has_norm = H.gamma != nullptr;
    // ---- the eight partial tiles meet in LDS, 128 columns at a time: a slab per wave, summed in wave order
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        char* slab = smem + wave * Cf::SLAB;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int j = 0; j < JH; ++j)
                *reinterpret_cast<f32x4*>(slab + (mb * 16 + r) * Cf::SLAB_PITCH + (j * 16 + g * 4) * 4) = acc2[mb][h * JH + j];
        if (h == NH - 1) {   // the accumulators are gone: the epilogue operands are requested here, under the last reduction — in the ROW layout of the final pass
            __builtin_amdgcn_sched_barrier(0);   // (this thread: row srow, the CPT columns from ec0)
            int m = m0 + srow;
            m = m < M ? m : M - 1;
#pragma unroll
            for (int c = 0; c < CPT; c += 4) {
                load4(H.bproj + ec0 + c, *reinterpret_cast<float(*)[4]>(ebp + c));
#pragma unroll
                for (int q = 0; q < 4; ++q) egq[c + q] = ebq[c + q] = 0.f;
                if (has_norm) {
                    load4(H.gamma + ec0 + c, *reinterpret_cast<float(*)[4]>(egq + c));
                    if (H.beta != nullptr) load4(H.beta + ec0 + c, *reinterpret_cast<float(*)[4]>(ebq + c));
                    if (H.mod != nullptr) {
                        float mw[4], mb[4];
                        const T* mod = static_cast<const T*>(H.mod) + (int64_t)m * H.ldmod;
                        load4(mod + ec0 + c, mw);
                        load4(mod + E + ec0 + c, mb);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            egq[c + q] += 1.0f + mw[q];
                            ebq[c + q] += mb[q];
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (raw barriers: a __syncthreads would wait for the global requests above, which the sums below / proj wait for themselves)
        __builtin_amdgcn_s_barrier();
        {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = resv[h][q];
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float4 a = *reinterpret_cast<const float4*>(smem + w * Cf::SLAB + srow * Cf::SLAB_PITCH + scb * 4);
                const float4 b = *reinterpret_cast<const float4*>(smem + w * Cf::SLAB + srow * Cf::SLAB_PITCH + scb * 4 + 16);
                v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
                v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
            }
            const int col = h * 128 + scb, kt = col >> 6, ck = (col & 63) >> 3;
            bf16x8 pv;
#pragma unroll
            for (int q = 0; q < 8; ++q) pv[q] = (__bf16)v[q];
            *reinterpret_cast<bf16x8*>(smem + Cf::X3_OFF + kt * BM * BKB + srow * BKB + ((ck ^ (srow & 7)) << 4)) = pv;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // the slabs are free for the next pass; after the last pass the x3 tile is complete
        if (L.probe == 5 + h) return;
    }

    if (L.probe == 4) return;
    // ================================================================================================ phase 3: proj + the row norm (sea_mlp_fc2_proj_norm's tail)
    f32x4 acc3[2][NJ];
    {
        const char* sA = smem + Cf::X3_OFF + r * BKB;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
                const uint4 af0 = *reinterpret_cast<const uint4*>(sA + kt * BM * BKB + off);
                const uint4 af1 = *reinterpret_cast<const uint4*>(sA + kt * BM * BKB + 16 * BKB + off);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (kt == 0 && kc == 0) {
                        acc3[0][j] = mma16_first(wp[0][j], af0);
                        acc3[1][j] = mma16_first(wp[0][j], af1);
                    } else {
                        mma16<T>(wp[kt * 2 + kc][j], af0, acc3[0][j]);
                        mma16<T>(wp[kt * 2 + kc][j], af1, acc3[1][j]);
                    }
                }
            }
    }
    // ---- the output rows leave in ROW layout: y through an LDS tile (the slabs' memory), then 16 threads per row — statistics by lane exchanges inside the
    // 16, the norm operands and the stores as whole 64-byte pieces of a row (in the accumulator layout a store instruction is 16 rows x 64 B, and the
    // modulation rows 16 rows x 8 B per load: ~100 cycles of the memory pipeline each, 3 us per workgroup)
    constexpr int SEG = 80, YP = (E / 16) * SEG + 16;   // a row: 16-column segments of 64 B at a pitch of 80 (the readers' 64-byte strides would share banks four ways)
    static_assert(BM * YP <= Cf::X3_OFF, "the y tile lies in front of the x3 tile");
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = wave * (E / NW) + j * 16 + g * 4;
            *reinterpret_cast<f32x4*>(smem + (i * 16 + r) * YP + (n >> 4) * SEG + (n & 15) * 4) = acc3[i][j];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        float v[CPT];
#pragma unroll
        for (int c = 0; c < CPT; c += 4) {
            const float4 y = *reinterpret_cast<const float4*>(smem + srow * YP + ((ec0 + c) >> 4) * SEG + ((ec0 + c) & 15) * 4);
            v[c] = y.x + ebp[c]; v[c + 1] = y.y + ebp[c + 1]; v[c + 2] = y.z + ebp[c + 2]; v[c + 3] = y.w + ebp[c + 3];
        }
        if (has_norm) {   // block-uniform
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CPT; c += 4)
#pragma unroll
                for (int e = 0; e < 4; ++e) s4[e] = add1(s4[e], v[c + e]);
            float sm_ = add1(add1(s4[0], s4[1]), add1(s4[2], s4[3]));
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) sm_ += __shfl_xor(sm_, o);
            const float mean2 = sm_ * (1.0f / (float)E);
            float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CPT; c += 4)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d_ = v[c + e] - mean2;
                    q4[e] = fma1(d_, d_, q4[e]);
                }
            float sq = add1(add1(q4[0], q4[1]), add1(q4[2], q4[3]));
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
            const float rstd2 = 1.0f / sqrtf(sq * (1.0f / (float)E) + L.eps);
#pragma unroll
            for (int c = 0; c < CPT; ++c) v[c] = (v[c] - mean2) * rstd2 * egq[c] + ebq[c];
        }
        const int m = m0 + srow;
        if (m < M) {
#pragma unroll
            for (int c = 0; c < CPT; c += 4) {
                if (H.Y32 != nullptr) store4(H.Y32 + (int64_t)m * H.ldy32 + ec0 + c, v[c], v[c + 1], v[c + 2], v[c + 3]);
                if (H.Yact != nullptr) store4(static_cast<T*>(H.Yact) + (int64_t)m * H.ldyact + ec0 + c, v[c], v[c + 1], v[c + 2], v[c + 3]);
            }
        }
    }
    }
}

extern "C" int sea_mlp_block(const SeaMlpGroup* g1, const SeaMlp2Group* g2, int n_groups, float eps, int dtype, void* stream) {
    SEA_REQUIRE(g1 != nullptr && g2 != nullptr && n_groups >= 1 && n_groups <= SEA_MAX_MLP_GROUPS, "sea_mlp_block: n_groups=%d out of range", n_groups);
    const int E = g1[0].E, S = g1[0].S;
    const bool shape_ok = (E == 256 && S == 2048) || (E == 128 && S == 1024);
    if (dtype != SEA_BF16 || !shape_ok) {
        sea_set_error("sea_mlp_block: unsupported dtype / shape (dtype=%d E=%d S=%d): bf16, (E, S) in {(256, 2048), (128, 1024)}", dtype, E, S);
        return SEA_EUNSUPPORTED;
    }
    MlpBlockLaunch L;
    memset(&L, 0, sizeof(L));
    int total = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaMlpGroup& G = g1[i];
        const SeaMlp2Group& H = g2[i];
        SEA_REQUIRE(G.E == E && G.S == S && H.E == E && H.S == S && G.M >= 1 && H.M == G.M, "sea_mlp_block[%d]: the groups of a launch share E and S; both halves the same M", i);
        SEA_REQUIRE((G.A || G.X32) && G.W1 && G.b1 && G.lnw && G.lnb, "sea_mlp_block[%d]: null pointer (first layer)", i);
        SEA_REQUIRE((G.X32 || (G.lda % 8 == 0 && G.lda >= E)) && G.ldw % 8 == 0 && G.ldw >= E, "sea_mlp_block[%d]: bad strides (first layer)", i);
        if (G.X32 != nullptr)
            SEA_REQUIRE(G.gamma && G.ldx32 % 4 == 0 && G.ldx32 >= E && (!G.addend || (G.ldadd % 4 == 0 && G.ldadd >= E)) && G.Xout == nullptr &&
                            (!G.mod || (G.ldmod % 8 == 0 && G.ldmod >= 2 * E)) && sea_aligned16(G.X32) && sea_aligned16(G.addend) && sea_aligned16(G.mod) &&
                            sea_aligned16(G.gamma) && sea_aligned16(G.beta) && G.norm_eps > 0.f,
                        "sea_mlp_block[%d]: norm prologue: null / misaligned pointer, bad stride, or Xout (the block keeps x + addend to itself)", i);
        SEA_REQUIRE(sea_aligned16(G.A) && sea_aligned16(G.W1) && sea_aligned16(G.b1) && sea_aligned16(G.lnw) && sea_aligned16(G.lnb), "sea_mlp_block[%d]: pointers must be 16-byte aligned", i);
        SEA_REQUIRE(H.W2 && H.b2 && (H.R || G.X32) && H.Wproj && H.bproj && (H.Y32 || H.Yact), "sea_mlp_block[%d]: null pointer (second layer; R may be NULL with the norm prologue: the residual is then x + addend)", i);
        SEA_REQUIRE(H.ldw2 % 8 == 0 && H.ldw2 >= S && H.ldwp % 8 == 0 && H.ldwp >= E && (!H.R || (H.ldr % 4 == 0 && H.ldr >= E)) &&
                        (!H.Y32 || (H.ldy32 % 4 == 0 && H.ldy32 >= E)) && (!H.Yact || (H.ldyact % 4 == 0 && H.ldyact >= E)) && (!H.mod || (H.ldmod % 4 == 0 && H.ldmod >= 2 * E)),
                    "sea_mlp_block[%d]: bad strides (second layer)", i);
        SEA_REQUIRE(sea_aligned16(H.W2) && sea_aligned16(H.b2) && sea_aligned16(H.R) && sea_aligned16(H.Wproj) && sea_aligned16(H.bproj) && sea_aligned16(H.gamma) && sea_aligned16(H.beta) &&
                        sea_aligned16(H.mod) && sea_aligned16(H.Y32) && sea_aligned16(H.Yact),
                    "sea_mlp_block[%d]: pointers must be 16-byte aligned", i);
        SEA_REQUIRE(H.gamma || (!H.beta && !H.mod), "sea_mlp_block[%d]: beta / mod without gamma", i);
        L.g1[i] = G;
        L.g2[i] = H;
        L.tile_start[i] = total;
        total += (G.M + 31) / 32;
    }
    L.tile_start[n_groups] = total;
    L.n_groups = n_groups;
    L.eps = eps;
    static const int probe = sea_tune("blk_probe", 0);
    L.probe = probe;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int xcds = 0;
    for (int i = 0; i < n_groups; ++i) {
        L.xcd_start[i] = xcds;
        xcds += (L.tile_start[i + 1] - L.tile_start[i] + 31) / 32;
    }
    L.xcd_start[n_groups] = xcds;
    static const int xcd_excl = sea_tune("blk_xcd", 0);   // measured at cfg2: 55.5 us against 52.9 with the plain order (the second layer's loop 15.2 against 12.8: 32 CUs of an XCD asking one L2 for the same lines at once)
    static const int per_xcd_on = sea_tune("blk_perxcd", 1);   // tuning aid
    if (total > 256 && per_xcd_on) {   // several rounds of workgroups: consecutive tiles (one field: one set of weights) on one XCD
        L.per_xcd = (total + 7) / 8;
        total = 8 * L.per_xcd;
    } else if (xcds <= 8 && xcd_excl) {   // one round, whole XCDs per group (32 CUs each)
        L.per_xcd = -1;
        total = 256;
    }
    if (E == 256) {
        constexpr int lds = MlpBlockCfg<4, 16>::BYTES;
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_block_kernel<4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)once;
        mlp_block_kernel<4, 16><<<dim3(total), dim3(512), lds, s>>>(L);
    } else {
        constexpr int lds = MlpBlockCfg<2, 8>::BYTES;
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_block_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)once;
        mlp_block_kernel<2, 8><<<dim3(total), dim3(512), lds, s>>>(L);
    }
    SEA_CHECK_LAUNCH("sea_mlp_block");
    return SEA_OK;
}
