// First half of the field MLP in one launch (gfx950, bf16): sea_mlp_fc1_ln_gelu.
//
//     Hg = gelu_erf(LayerNorm_S(A[M, E] . W1[S, E]^T + b1) * lnw + lnb)          (models/base_blocks.py:22-24: Linear, nn.LayerNorm(S), GELU)
//
// A workgroup (8 waves) owns 32 COMPLETE rows of the [M, S] hidden matrix, so the LayerNorm statistics and the activation are an epilogue: the
// pre-activation matrix (25 MB at cfg2) is never written, the LayerNorm + GELU pass over it (a launch of its own, 17.9 us at cfg2) disappears.
// The price is that every workgroup streams the whole W1 (S x E bf16 = 1 MiB) from L2: W1 goes L2 -> LDS by global_load_lds in stages of 64 rows
// x the full contraction (32 KiB), a ring of 4 stages (3 in flight), one barrier per stage; the 32 A rows (16 KiB) stay resident.  Per stage 4 of
// the 8 waves compute (a 16-row block each, two row blocks: 16 MFMAs), even stages the lower four waves, odd stages the upper four, so that
// every wave ends with S / 128 column blocks x 2 row blocks of accumulators (128 VGPRs at S = 2048).  The activations leave through LDS as
// whole rows (the ring is free by then).
#include "gemm_core.hpp"
#include <stdlib.h>

struct MlpLaunch {
    SeaMlpGroup g[SEA_MAX_MLP_GROUPS];
    int tile_start[SEA_MAX_MLP_GROUPS + 1];
    int n_groups;
    float eps;
    int per_xcd;   // > 0: workgroup b takes tile (b % 8) * per_xcd + b / 8, so that the tiles of one field (one W1) sit on as few XCDs (L2s) as possible
};

__device__ __forceinline__ void glds16_mlp(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// LDS request of mlp_fc1_ln_gelu_kernel<KT, NSB>: resident A rows + the 4-stage W1 ring + lnw | lnb (E = 256, S = 2048: 16 + 128 + 16 KiB = the CU's 160 KiB)
template <int KT, int NSB>
struct Mlp1Lds {
    static constexpr int BYTES = KT * 32 * 128 + 4 * (KT * 64 * 128) + 2 * (NSB * 128) * 4;
};
// LDS request of mlp_fc2_proj_norm_kernel<KTE, KTS>: the 4-stage (A | W) ring (the x3 tile lives in the A pieces of its slots), the statistics scratch
template <int KTE, int KTS>
struct Mlp2Lds {
    static constexpr int BYTES = 4 * (32 * 128 + KTE * 64 * 128) + 2 * 8 * 32 * 4;
};

// KT = E / 64 K-tiles; NSB = S / 128 column blocks per wave
template <int KT, int NSB>
__global__ __launch_bounds__(512) void mlp_fc1_ln_gelu_kernel(const MlpLaunch L) {
    using T = __bf16;
    constexpr int BM = 32, BKB = 128, BK = 64, NS = 4, NW = 8;
    constexpr int S = NSB * 128, NSTAGE = S / 64;
    constexpr int A_BYTES = KT * BM * BKB;                 // resident A rows, K-tile major
    constexpr int STAGE = KT * 64 * BKB;                   // 64 W rows x the whole contraction
    constexpr int LPS = KT * 8 / NW;                       // DMA pieces per wave per stage (KT = 4: 4, KT = 2: 2)
    constexpr int SP = S * 2 + 16;                         // staging pitch of an output row
    constexpr int PRM_OFF = A_BYTES + NS * STAGE;          // lnw | lnb (2 S floats) behind the ring: DMA'd at the start, read by the epilogue
    constexpr int B1_OFF = 0, RED_OFF = S * 4;             // b1 and the statistics scratch go where the A rows were (they live in registers after stage 0)
    static_assert(BM * SP <= PRM_OFF && RED_OFF + 2 * NW * BM * 4 <= A_BYTES, "the output tile is staged over the operand memory");
    // Every LDS-DMA destination (M0 base + 1 KiB per wave-instruction) lies inside the launch's LDS request, Mlp1Lds<KT, NSB>::BYTES (the host passes
    // exactly that): the A pieces end at A_BYTES, stage s of the ring at A_BYTES + NS * STAGE = PRM_OFF, the 2 S / 256 gain / shift pieces at
    // PRM_OFF + 8 S, the b1 pieces (S / 256 of them from B1_OFF = 0) inside the A region.
    static_assert(PRM_OFF + 2 * S * 4 == Mlp1Lds<KT, NSB>::BYTES && Mlp1Lds<KT, NSB>::BYTES <= 160 * 1024, "LDS request of sea_mlp_fc1_ln_gelu");
    static_assert(KT * (BM / 8) * 1024 == A_BYTES && KT * 8 * 1024 == STAGE && (S / 256) * 1024 <= A_BYTES && LPS * NW == KT * 8, "LDS-DMA piece counts");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // workgroups go to the XCDs round-robin: consecutive tiles (the same field, the same W1) are given to the same XCD
    const int tile = L.per_xcd > 0 ? (int)(blockIdx.x & 7) * L.per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (tile >= L.tile_start[L.n_groups]) return;
    int gi = 0;
    while (gi + 1 < L.n_groups && tile >= L.tile_start[gi + 1]) ++gi;
    const SeaMlpGroup& G = L.g[gi];
    const int m0 = (tile - L.tile_start[gi]) * BM, M = G.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int half = wave >> 2, wq = wave & 3;             // which stages this wave computes, and its 16-row block inside a stage
    const T* A = static_cast<const T*>(G.A);
    const T* W = static_cast<const T*>(G.W1);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const int rl = lane >> 3;
    const int chunk = (lane & 7) ^ (rl & 7);               // swizzle on the source side
    // ---- gains / shifts of the LayerNorm: 1 KiB pieces straight into LDS, oldest in the queue (the epilogue finds them there)
    for (int p = wave; p < 2 * (S / 256); p += NW) {
        const float* src = p < S / 256 ? G.lnw : G.lnb;
        const int q = p < S / 256 ? p : p - S / 256;
        glds16_mlp(src + q * 256 + lane * 4, lds_base + (unsigned)(PRM_OFF + p * 1024));
    }
    // ---- A rows: KT x 4 pieces of 8 rows
    for (int p = wave; G.X32 == nullptr && p < KT * (BM / 8); p += NW) {
        const int kt = p / (BM / 8), u = p - kt * (BM / 8);
        int row = m0 + u * 8 + rl;
        row = row < M ? row : M - 1;
        glds16_mlp(A + (int64_t)row * G.lda + kt * BK + chunk * 8, lds_base + (unsigned)(kt * BM * BKB + u * 8 * BKB));
    }
    constexpr int A_PIECES = (KT * (BM / 8) + NW - 1) / NW;   // per wave, upper bound (waves past the count issue none: wave-uniform)
    auto dma_stage = [&](int s) {
        const unsigned base = lds_base + (unsigned)(A_BYTES + (s % NS) * STAGE);
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const int p = i * NW + wave;                      // piece: K-tile kt, 8 rows u
            const int kt = p >> 3, u = p & 7;
            glds16_mlp(W + (int64_t)(s * 64 + u * 8 + rl) * G.ldw + kt * BK + chunk * 8, base + (unsigned)(kt * 64 * BKB + u * 8 * BKB));
        }
    };
    dma_stage(0);
    (void)A_PIECES;
    // ---- optional prologue: the operand rows are the normalised rows of the fp32 residual stream (x + addend, AdaLN / LayerNorm), written straight into
    // the A region in the K-tile-major, swizzled layout the LDS-DMA of ready-made rows would have produced.  16 lanes own a row (16 columns each), the
    // statistics are two-pass like sea_rownorm's.  Only the first weight stage is requested in front of it: the compiler's wait for these loads is a
    // vmcnt(0), which behind the whole ring fill would wait for all of that (measured: 32.1 us per launch against 31.2 with nothing in front).
    if (G.X32 != nullptr) {
        constexpr int E = KT * 64, CPT = E / 16;              // columns per lane
        const int prow = tid >> 4, pl = tid & 15;              // 512 threads = 32 rows x 16 lanes
        int row = m0 + prow;
        const bool rok = row < M;
        row = rok ? row : M - 1;
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
        if (G.Xout != nullptr && rok) {
#pragma unroll
            for (int c = 0; c < CPT; c += 4) store4(G.Xout + (int64_t)row * G.ldxout + c0 + c, xv[c], xv[c + 1], xv[c + 2], xv[c + 3]);
        }
        float s4[4] = {0.f, 0.f, 0.f, 0.f};   // (add1 / fma1: the vectoriser's packed horizontal adds are the form the build's lint refuses beside MFMAs)
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
        // column c0 + c of row prow -> K-tile (c0 + c) / 64, 16-byte chunk ((c0 + c) % 64) / 8 at position chunk ^ (row & 7)
#pragma unroll
        for (int c = 0; c < CPT; c += 8) {
            const int col = c0 + c, kt = col >> 6, ck = (col & 63) >> 3;
            bf16x8 pv;
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = (__bf16)((xv[c + e] - mean) * rstd * gq[c + e] + bq[c + e]);
            *reinterpret_cast<bf16x8*>(smem + kt * BM * BKB + prow * BKB + ((ck ^ (prow & 7)) << 4)) = pv;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the Xout stores too: nothing of this wave is in the memory pipeline when the DMAs start)
    }
    for (int s = 1; s < NS - 1; ++s) dma_stage(s);

    f32x4 acc[NSB][2];
    uint4 areg[KT][2][2];
#pragma unroll
    for (int i = 0; i < NSB; ++i) {
        acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) {
        // stage s (and the A rows, older) have landed once at most the pieces of the stages issued after it are outstanding
        // (lgkmcnt(0): the fragment reads of the previous stage are retired before the barrier that releases its slot to the next DMA, see gemm_core.hpp)
        if (s + 2 < NSTAGE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LPS) : "memory");
        else if (s + 1 < NSTAGE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + NS - 1 < NSTAGE) dma_stage(s + NS - 1);
        if (s == 0) {            // the A rows have landed: this lane's fragments of all 32 rows stay in registers for the whole contraction
            const char* sA = smem + r * BKB;   // (read from LDS per stage they were two of every three fragment loads of the main loop)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
                    areg[kt][kc][0] = *reinterpret_cast<const uint4*>(sA + kt * BM * BKB + off);
                    areg[kt][kc][1] = *reinterpret_cast<const uint4*>(sA + kt * BM * BKB + 16 * BKB + off);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are complete before the next barrier releases the b1 DMA over these rows
        }
        if (s == 1)              // every wave has its A fragments (barrier above): b1 goes where the A rows were
            for (int p = wave; p < S / 256; p += NW) glds16_mlp(G.b1 + p * 256 + lane * 4, lds_base + (unsigned)(B1_OFF + p * 1024));
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
    // ---- epilogue: + b1, LayerNorm over the S columns of a row (two-pass, fp32), * lnw + lnb, GELU, bf16
    // this lane: rows m0 + mb*16 + r, columns n(i) = (2 i + half) * 64 + wq * 16 + 4 g + q
    // b1 | lnw | lnb are in LDS already (DMA'd under the main loop; the last stage's vmcnt(0) + barrier made them visible): the per-lane form
    // is 3 x 16 dependent 16-byte global loads, and a staging pass here put a cold L2 round trip in front of the statistics (probe build:
    // 4.9 us between the last MFMA and the GELU, for 0.3 us of arithmetic).  Nothing below touches the ring until the output tile is written.
    float* red = reinterpret_cast<float*>(smem + RED_OFF);   // [2 passes][8 waves][32 rows]
    const float* pb1 = reinterpret_cast<const float*>(smem + B1_OFF);
    const float* pgm = reinterpret_cast<const float*>(smem + PRM_OFF);
    const float* pbt = pgm + S;
    const float inv_s = 1.0f / (float)S;
    float sum[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NSB; ++i) {
        const int n = (2 * i + half) * 64 + wq * 16 + g * 4;
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
    float mean[2], sq[2] = {0.f, 0.f};
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
    __syncthreads();
    float rstd[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[256 + w * 32 + mb * 16 + r];
        rstd[mb] = 1.0f / sqrtf(t * inv_s + L.eps);
    }
    __syncthreads();   // every wave is past the main loop and has read the statistics: the output tile may go over the A region and the ring
    // The tile leaves in NCK column chunks: a chunk is activated (VALU), staged in LDS, and its whole-row stores are issued while the next chunk
    // is being activated — the stores of a workgroup are 128 KiB, and every workgroup of the launch reaches this point at the same time
    // (probe build: GELU 7 us, stores 4.9 us when they followed each other).
    constexpr int NCK = 4, IPC = NSB / NCK, CW = S / NCK;   // chunks, column blocks per chunk, columns per chunk
    T* Hg = static_cast<T*>(G.Hg);
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
#pragma unroll
        for (int ii = 0; ii < IPC; ++ii) {
            const int i = c * IPC + ii;
            const int n = (2 * i + half) * 64 + wq * 16 + g * 4;
            float gm[4], bt[4];
            load4(pgm + n, gm);
            load4(pbt + n, bt);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                float y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) y[q] = gelu_erf_bf16((acc[i][mb][q] - mean[mb]) * rstd[mb] * gm[q] + bt[q]);
                store4(reinterpret_cast<T*>(smem + (mb * 16 + r) * SP) + n, y[0], y[1], y[2], y[3]);
            }
        }
        __syncthreads();
        constexpr int CPR = CW / 8;   // 16-byte pieces per row of the chunk
#pragma unroll
        for (int k = 0; k < BM * CPR / 512; ++k) {
            const int idx = tid + k * 512;
            const int row = idx / CPR, cc = idx - row * CPR;
            const int m = m0 + row;
            if (m < M) *reinterpret_cast<uint4*>(Hg + (int64_t)m * G.ldh + c * CW + cc * 8) = *reinterpret_cast<const uint4*>(smem + row * SP + c * CW * 2 + cc * 16);
        }
    }
}

// ================================================================================================ second half of the field MLP, proj and the final norm
//     x3  = Hg[M, S] . W2[E, S]^T + b2 + R                       (models/base_blocks.py:25, the residual of models/temporal.py:145)
//     y   = x3 . Wproj[E, E]^T + bproj                            (models/temporal.py:146)
//     out = norm(y) (gamma / beta / mod as SeaNormGroup) or y     (the model's final per-field norm, models/temporal.py:412-415)
// A workgroup (8 waves) owns 32 COMPLETE rows through both Linear layers, so x3 never leaves the CU and the norm is an epilogue: three launches
// (fc2 19.8 us, proj 7.1, final norm 5.6 at cfg2) become one.  The price is the one the fc1 kernel pays: every workgroup streams the whole W2 (E x S
// bf16 = 1 MiB) — but fc2's own 64 x 64-tile launch is bound by the same megabyte per CU, and there is no heavy epilogue here.
// K-tiles of 64: a stage = the A tile (32 rows of Hg) + the W tile (E rows of W2), both by global_load_lds, a ring of 3 stages (2 in flight beside the
// one being multiplied); the E / 64 K-tiles of Wproj follow as further stages of the same ring.  Wave w owns output columns [E/8 w, E/8 (w+1)) of all
// 32 rows in both layers (transposed MFMA tiles: a lane holds 4 consecutive columns of a row).
struct Mlp2Launch {
    SeaMlp2Group g[SEA_MAX_MLP_GROUPS];
    int tile_start[SEA_MAX_MLP_GROUPS + 1];
    int n_groups;
    float eps;
    int per_xcd;
};

template <int KTE, int KTS>
__global__ __launch_bounds__(512) void mlp_fc2_proj_norm_kernel(const Mlp2Launch L) {
    using T = __bf16;
    // Round 4: FOUR ring stages (three in flight beside the one being multiplied) instead of three.  The launch is a chain of 36 tiny stages (4 MFMAs per
    // wave and K-tile) fed by a 36 KiB DMA piece each: its time is the weight stream's latency x bytes / bytes in flight (23 us at 72 KiB in flight).  The 16 KiB
    // of the x3 tile that used to cap the ring at three slots now live in the A pieces of the slots themselves: the second layer's K-tile kt is stage KTS + kt
    // = slot kt (KTS % NS == 0), whose 4 KiB A piece is not used by that stage's DMA.
    constexpr int BM = 32, BKB = 128, BK = 64, NS = 4, NW = 8;
    constexpr int E = KTE * 64, S = KTS * 64;
    constexpr int STAGE_A = BM * BKB, STAGE_W = E * BKB, STAGE = STAGE_A + STAGE_W;
    constexpr int RED_OFF = NS * STAGE;                                       // statistics scratch behind the ring
    static_assert(KTS % NS == 0 && KTE <= NS, "the x3 K-tiles sit in the A pieces of slots 0 .. KTE - 1");
    constexpr int NSTAGE = KTS + KTE;
    constexpr int WPW = E / 8 / NW;                                          // 8-row W pieces per wave per stage (E = 256: 4)
    constexpr int LPS1 = WPW + 1, LPS2 = WPW;                                // DMA pieces per wave and stage: first layer + one A piece (waves 4 .. 7 repeat pieces 0 .. 3: same bytes, same place), second layer W only
    constexpr int NJ = E / NW / 16;                                          // 16-column blocks per wave (E = 256: 2)
    // every LDS-DMA destination lies inside the ring (stage slot + the A piece's 4 KiB + WPW * NW = E / 8 W pieces of 1 KiB), the ring, the x3 tile and
    // the statistics inside the launch's LDS request
    static_assert(STAGE_A == 4 * 1024 && WPW * NW * 1024 == STAGE_W && RED_OFF + 2 * NW * BM * 4 == Mlp2Lds<KTE, KTS>::BYTES && Mlp2Lds<KTE, KTS>::BYTES <= 160 * 1024 && 2 * LPS1 <= 63,
                  "LDS request of sea_mlp_fc2_proj_norm");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = L.per_xcd > 0 ? (int)(blockIdx.x & 7) * L.per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (tile >= L.tile_start[L.n_groups]) return;
    int gi = 0;
    while (gi + 1 < L.n_groups && tile >= L.tile_start[gi + 1]) ++gi;
    const SeaMlp2Group& G = L.g[gi];
    const int m0 = (tile - L.tile_start[gi]) * BM, M = G.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const T* A = static_cast<const T*>(G.Hg);
    const T* W2 = static_cast<const T*>(G.W2);
    const T* Wp = static_cast<const T*>(G.Wproj);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const int rl = lane >> 3;
    const int chunk = (lane & 7) ^ (rl & 7);               // swizzle on the source side
    // ---- epilogue operands: ordinary loads, requested before the weight stream (older than every DMA piece: they never hold a counted wait up)
    // this lane: rows m0 + 16 i + r (i < 2), columns n(j) = wave * (E / 8) + 16 j + 4 g + q
    float b2v[NJ][4], bpv[NJ][4], rv[2][NJ][4], gq[NJ][4], bq[NJ][4], mwv[2][NJ][4], mbv[2][NJ][4];
    const bool has_norm = G.gamma != nullptr;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = wave * (E / NW) + j * 16 + g * 4;
        load4(G.b2 + n, b2v[j]);
        load4(G.bproj + n, bpv[j]);
#pragma unroll
        for (int q = 0; q < 4; ++q) gq[j][q] = bq[j][q] = 0.f;
        if (has_norm) {
            load4(G.gamma + n, gq[j]);
            if (G.beta != nullptr) load4(G.beta + n, bq[j]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = m0 + i * 16 + r;
            m = m < M ? m : M - 1;
            load4(G.R + (int64_t)m * G.ldr + n, rv[i][j]);
#pragma unroll
            for (int q = 0; q < 4; ++q) mwv[i][j][q] = mbv[i][j][q] = 0.f;
            if (has_norm && G.mod != nullptr) {
                const T* mod = static_cast<const T*>(G.mod) + (int64_t)m * G.ldmod;
                load4(mod + n, mwv[i][j]);
                load4(mod + E + n, mbv[i][j]);
            }
        }
    }
    auto dma_stage = [&](int s) {
        const unsigned base = lds_base + (unsigned)((s % NS) * STAGE);
        const bool second = s >= KTS;                         // Wproj K-tiles behind the W2 ones
        const T* W = second ? Wp : W2;
        const int ldw = second ? G.ldwp : G.ldw2;
        const int kt = second ? s - KTS : s;
        if (!second) {   // the A piece of this wave (4 pieces of 8 rows); the second layer's operand is the x3 tile, which lives where this piece would go
            const int u = wave & 3;
            int row = m0 + u * 8 + rl;
            row = row < M ? row : M - 1;
            glds16_mlp(A + (int64_t)row * G.ldh + kt * BK + chunk * 8, base + (unsigned)(u * 8 * BKB));
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int u = i * NW + wave;                      // 8-row piece of the W tile
            glds16_mlp(W + (int64_t)(u * 8 + rl) * ldw + kt * BK + chunk * 8, base + (unsigned)(STAGE_A + u * 8 * BKB));
        }
    };
    for (int s = 0; s < NS - 1; ++s) dma_stage(s);

    f32x4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float x3v[2][NJ][4];
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) {
        // stage s has landed once at most the pieces of the stage issued after it are outstanding.  lgkmcnt(0): this wave's fragment reads of the
        // PREVIOUS stage are retired before it arrives at the barrier — the slot they read is re-staged right behind the barrier, one phase after its
        // last read, and nothing else orders an LDS-DMA write against an earlier ds_read (seen as rare wrong 8-row pieces with two workgroups per CU)
        {
            int later = 0;                                    // pieces of the stages issued after stage s (a constant per unrolled iteration)
            for (int t = s + 1; t <= s + NS - 2 && t < NSTAGE; ++t) later += t < KTS ? LPS1 : LPS2;
            switch (later) {
                case 2 * LPS1: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LPS1) : "memory"); break;
                case LPS1 + LPS2: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS1 + LPS2) : "memory"); break;
                case 2 * LPS2: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LPS2) : "memory"); break;
                case LPS2: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS2) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
            }
        }
        __builtin_amdgcn_s_barrier();                         // ... for every wave; and nobody still reads the slot the next DMA goes to
        if (s + NS - 1 < NSTAGE) dma_stage(s + NS - 1);
        if (s == KTS) {
            // ---- between the layers: x3 = acc + b2 + R, as bf16 into the x3 tile (the second layer's A operand); acc restarts
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = wave * (E / NW) + j * 16 + g * 4, m = i * 16 + r;
#pragma unroll
                    for (int q = 0; q < 4; ++q) x3v[i][j][q] = acc[i][j][q] + b2v[j][q] + rv[i][j][q];
                    store4(reinterpret_cast<T*>(smem + (n >> 6) * STAGE + m * BKB + ((((n & 63) >> 3) ^ (m & 7)) << 4) + (n & 7) * 2), x3v[i][j][0], x3v[i][j][1],
                           x3v[i][j][2], x3v[i][j][3]);
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const char* sA = smem + (s % NS) * STAGE + r * BKB;   // (second layer: the x3 K-tile s - KTS sits in slot (s - KTS) = s % NS's A piece)
        const char* sW = smem + (s % NS) * STAGE + STAGE_A + (wave * (E / NW) + r) * BKB;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
            uint4 af[2], wf[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const uint4*>(sA + i * 16 * BKB + off);
#pragma unroll
            for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const uint4*>(sW + j * 16 * BKB + off);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) mma16<T>(wf[j], af[i], acc[i][j]);
        }
    }
    // ---- epilogue: + bproj, optional row norm over the E columns (two-pass, fp32; the row's columns are spread over the 8 waves), outputs
    float* red = reinterpret_cast<float*>(smem + RED_OFF);   // [2 passes][8 waves][32 rows]
    float v[2][NJ][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[i][j][q] = acc[i][j][q] + bpv[j][q];
    float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
    if (has_norm) {   // block-uniform
        const float inv_e = 1.0f / (float)E;
        float sum[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) t4[q] = add1(t4[q], v[i][j][q]);
            sum[i] = add1(add1(t4[0], t4[1]), add1(t4[2], t4[3]));
            sum[i] += __shfl_xor(sum[i], 16);
            sum[i] += __shfl_xor(sum[i], 32);
            if (g == 0) red[wave * 32 + i * 16 + r] = sum[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[w * 32 + i * 16 + r];
            mean[i] = t * inv_e;
            float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float c = v[i][j][q] - mean[i];
                    q4[q] = fma1(c, c, q4[q]);
                }
            float sq = add1(add1(q4[0], q4[1]), add1(q4[2], q4[3]));
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
            if (g == 0) red[256 + wave * 32 + i * 16 + r] = sq;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[256 + w * 32 + i * 16 + r];
            rstd[i] = 1.0f / sqrtf(t * inv_e + L.eps);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + r;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = wave * (E / NW) + j * 16 + g * 4;
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (has_norm) {
                    const float gg = G.mod != nullptr ? gq[j][q] + 1.0f + mwv[i][j][q] : gq[j][q];
                    o[q] = (v[i][j][q] - mean[i]) * rstd[i] * gg + (bq[j][q] + mbv[i][j][q]);
                } else {
                    o[q] = v[i][j][q];
                }
            }
            if (G.Y32 != nullptr) store4(G.Y32 + (int64_t)m * G.ldy32 + n, o[0], o[1], o[2], o[3]);
            if (G.Yact != nullptr) store4(static_cast<T*>(G.Yact) + (int64_t)m * G.ldyact + n, o[0], o[1], o[2], o[3]);
        }
    }
}

template <typename K>
static int set_lds_mlp(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : -1;
}

extern "C" int sea_mlp_fc1_ln_gelu(const SeaMlpGroup* groups, int n_groups, float eps, int dtype, void* stream) {
    SEA_REQUIRE(groups != nullptr && n_groups >= 1 && n_groups <= SEA_MAX_MLP_GROUPS, "sea_mlp_fc1_ln_gelu: n_groups=%d out of range", n_groups);
    const int E = groups[0].E, S = groups[0].S;
    const bool shape_ok = (E == 256 && S == 2048) || (E == 128 && S == 1024);
    if (dtype != SEA_BF16 || !shape_ok) {
        sea_set_error("sea_mlp_fc1_ln_gelu: unsupported dtype / shape (dtype=%d E=%d S=%d): bf16, (E, S) in {(256, 2048), (128, 1024)}", dtype, E, S);
        return SEA_EUNSUPPORTED;
    }
    MlpLaunch L;
    memset(&L, 0, sizeof(L));
    int total = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaMlpGroup& G = groups[i];
        SEA_REQUIRE(G.E == E && G.S == S && G.M >= 1, "sea_mlp_fc1_ln_gelu[%d]: the groups of a launch share E and S", i);
        SEA_REQUIRE((G.A || G.X32) && G.W1 && G.b1 && G.lnw && G.lnb && G.Hg, "sea_mlp_fc1_ln_gelu[%d]: null pointer", i);
        SEA_REQUIRE((G.X32 || (G.lda % 8 == 0 && G.lda >= E)) && G.ldw % 8 == 0 && G.ldw >= E && G.ldh % 8 == 0 && G.ldh >= S, "sea_mlp_fc1_ln_gelu[%d]: bad strides", i);
        if (G.X32 != nullptr)
            SEA_REQUIRE(G.gamma && G.ldx32 % 4 == 0 && G.ldx32 >= E && (!G.addend || (G.ldadd % 4 == 0 && G.ldadd >= E)) && (!G.Xout || (G.ldxout % 4 == 0 && G.ldxout >= E)) &&
                            (!G.mod || (G.ldmod % 8 == 0 && G.ldmod >= 2 * E)) && sea_aligned16(G.X32) && sea_aligned16(G.addend) && sea_aligned16(G.Xout) && sea_aligned16(G.mod) &&
                            sea_aligned16(G.gamma) && sea_aligned16(G.beta) && G.norm_eps > 0.f,
                        "sea_mlp_fc1_ln_gelu[%d]: norm prologue: null / misaligned pointer or bad stride", i);
        SEA_REQUIRE(sea_aligned16(G.A) && sea_aligned16(G.W1) && sea_aligned16(G.b1) && sea_aligned16(G.lnw) && sea_aligned16(G.lnb) && sea_aligned16(G.Hg),
                    "sea_mlp_fc1_ln_gelu[%d]: pointers must be 16-byte aligned", i);
        L.g[i] = G;
        L.tile_start[i] = total;
        total += (G.M + 31) / 32;
    }
    L.tile_start[n_groups] = total;
    L.n_groups = n_groups;
    L.eps = eps;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const bool xcd_map = sea_tune("mlp_xcd", 1) != 0;
    if (xcd_map && total > 256) {   // one round of workgroups: the plain order measured 0.6 us faster (cfg2, 190 tiles); several rounds: XCD-local wins (B=8)
        L.per_xcd = (total + 7) / 8;
        total = 8 * L.per_xcd;
    }
    if (E == 256) {
        constexpr int lds = Mlp1Lds<4, 16>::BYTES;   // A 16 KiB + ring 128 KiB + lnw | lnb 16 KiB = all 160 KiB of the CU
        static int once = set_lds_mlp(mlp_fc1_ln_gelu_kernel<4, 16>, lds);
        (void)once;
        mlp_fc1_ln_gelu_kernel<4, 16><<<dim3(total), dim3(512), lds, s>>>(L);
    } else {
        constexpr int lds0 = Mlp1Lds<2, 8>::BYTES;  // 8 KiB + 64 KiB (>= the output tile, 32 * (1024 * 2 + 16) = 66048) + lnw | lnb 8 KiB
        static int once = set_lds_mlp(mlp_fc1_ln_gelu_kernel<2, 8>, lds0);
        (void)once;
        mlp_fc1_ln_gelu_kernel<2, 8><<<dim3(total), dim3(512), lds0, s>>>(L);
    }
    SEA_CHECK_LAUNCH("sea_mlp_fc1_ln_gelu");
    return SEA_OK;
}

extern "C" int sea_mlp_fc2_proj_norm(const SeaMlp2Group* groups, int n_groups, float eps, int dtype, void* stream) {
    SEA_REQUIRE(groups != nullptr && n_groups >= 1 && n_groups <= SEA_MAX_MLP_GROUPS, "sea_mlp_fc2_proj_norm: n_groups=%d out of range", n_groups);
    const int E = groups[0].E, S = groups[0].S;
    const bool shape_ok = (E == 256 && S == 2048) || (E == 128 && S == 1024);
    if (dtype != SEA_BF16 || !shape_ok) {
        sea_set_error("sea_mlp_fc2_proj_norm: unsupported dtype / shape (dtype=%d E=%d S=%d): bf16, (E, S) in {(256, 2048), (128, 1024)}", dtype, E, S);
        return SEA_EUNSUPPORTED;
    }
    Mlp2Launch L;
    memset(&L, 0, sizeof(L));
    int total = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaMlp2Group& G = groups[i];
        SEA_REQUIRE(G.E == E && G.S == S && G.M >= 1, "sea_mlp_fc2_proj_norm[%d]: the groups of a launch share E and S", i);
        SEA_REQUIRE(G.Hg && G.W2 && G.b2 && G.R && G.Wproj && G.bproj && (G.Y32 || G.Yact), "sea_mlp_fc2_proj_norm[%d]: null pointer", i);
        SEA_REQUIRE(G.ldh % 8 == 0 && G.ldh >= S && G.ldw2 % 8 == 0 && G.ldw2 >= S && G.ldwp % 8 == 0 && G.ldwp >= E && G.ldr % 4 == 0 && G.ldr >= E &&
                        (!G.Y32 || (G.ldy32 % 4 == 0 && G.ldy32 >= E)) && (!G.Yact || (G.ldyact % 4 == 0 && G.ldyact >= E)) && (!G.mod || (G.ldmod % 4 == 0 && G.ldmod >= 2 * E)),
                    "sea_mlp_fc2_proj_norm[%d]: bad strides", i);
        SEA_REQUIRE(sea_aligned16(G.Hg) && sea_aligned16(G.W2) && sea_aligned16(G.b2) && sea_aligned16(G.R) && sea_aligned16(G.Wproj) && sea_aligned16(G.bproj) &&
                        sea_aligned16(G.gamma) && sea_aligned16(G.beta) && sea_aligned16(G.mod) && sea_aligned16(G.Y32) && sea_aligned16(G.Yact),
                    "sea_mlp_fc2_proj_norm[%d]: pointers must be 16-byte aligned", i);
        SEA_REQUIRE(G.gamma || (!G.beta && !G.mod), "sea_mlp_fc2_proj_norm[%d]: beta / mod without gamma", i);
        L.g[i] = G;
        L.tile_start[i] = total;
        total += (G.M + 31) / 32;
    }
    L.tile_start[n_groups] = total;
    L.n_groups = n_groups;
    L.eps = eps;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (total > 256) {   // several rounds of workgroups: consecutive tiles (one field: one W2, one Wproj) on one XCD
        L.per_xcd = (total + 7) / 8;
        total = 8 * L.per_xcd;
    }
    if (E == 256) {
        constexpr int lds = Mlp2Lds<4, 32>::BYTES;   // ring 144 KiB (x3 tile inside) + statistics 2 KiB
        static int once = set_lds_mlp(mlp_fc2_proj_norm_kernel<4, 32>, lds);
        (void)once;
        mlp_fc2_proj_norm_kernel<4, 32><<<dim3(total), dim3(512), lds, s>>>(L);
    } else {
        constexpr int lds = Mlp2Lds<2, 16>::BYTES;    // ring 80 KiB (x3 tile inside) + statistics 2 KiB
        static int once = set_lds_mlp(mlp_fc2_proj_norm_kernel<2, 16>, lds);
        (void)once;
        mlp_fc2_proj_norm_kernel<2, 16><<<dim3(total), dim3(512), lds, s>>>(L);
    }
    SEA_CHECK_LAUNCH("sea_mlp_fc2_proj_norm");
    return SEA_OK;
}
