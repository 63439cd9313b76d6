// Grouped NT GEMM kernels with fused epilogues (gfx950): sea_gemm_grouped, sea_qkv_rope_grouped.
#include "gemm_tile.hpp"
#include <stdlib.h>

struct GemmLaunch {
    SeaGemmGroup g[SEA_MAX_GROUPS];
    int tile_start[SEA_MAX_GROUPS + 1];
    int n_groups;
    int single_buffer;   // register-staged main loop on one LDS buffer (more workgroups per CU)
    int silu_lds_off;    // generated-A launches: byte offset of the w1 | b1 staging area in LDS
    int dma_ns;          // LDS-DMA launches: ring stages (4, or 2: half the LDS, two workgroups per CU)
    unsigned n_major;    // bit i: group i numbers its tiles with the ROW tile fastest (skinny M: the tiles that share a W panel are neighbours, see sea_gemm_grouped)
};

struct QkvLaunch {
    SeaQkvGroup g[SEA_MAX_GROUPS];
    int tile_start[SEA_MAX_GROUPS + 1];
    int n_groups;
    int single_buffer;
    SeaQkvCommon c;
};

template <typename L>
__device__ __forceinline__ int find_group(const L& launch, int bid) {
    int gi = 0;
    while (gi + 1 < launch.n_groups && bid >= launch.tile_start[gi + 1]) ++gi;
    return gi;
}

bool sea_gemm256_try(const SeaGemmGroup* groups, int n_groups, unsigned n_major, hipStream_t s);   // gemm256.hip
bool sea_gemm_ws_try(const SeaGemmGroup* groups, int n_groups, hipStream_t s);                       // gemm_ws.hip

// ---------------------------------------------------------------------------------------------- standard epilogue (tile body: gemm_tile.hpp)
template <typename T, int BM, int BN, bool DMA, bool PLAIN, bool SILUA = false>
__global__ __launch_bounds__(256) void gemm_grouped_kernel(const GemmLaunch L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int gi = find_group(L, bid);
    const SeaGemmGroup& G = L.g[gi];
    const int t = bid - L.tile_start[gi];
    const int tiles_n = (G.N + BN - 1) / BN;
    int tm = t / tiles_n, tn = t - tm * tiles_n;
    if ((L.n_major >> gi) & 1u) {   // block-uniform
        const int tiles_m = (G.M + BM - 1) / BM;
        tn = t / tiles_m;
        tm = t - tn * tiles_m;
    }
    if constexpr (DMA) {
        if (L.dma_ns == 2) {   // block-uniform
            gemm_tile_body<T, BM, BN, DMA, PLAIN, SILUA, 2>(G, tm, tn, smem, L.silu_lds_off);
            return;
        }
    }
    gemm_tile_body<T, BM, BN, DMA, PLAIN, SILUA>(G, tm, tn, smem, L.silu_lds_off);
}

// ---------------------------------------------------------------------------------------------- a few rows (KV-cache rollout steps)
// M <= 16 rows: the tiled kernels above give such a launch to N / 64 workgroups that each walk the whole contraction K-tile by K-tile (fc2 of the
// cfg2 step: 12 workgroups x 32 dependent K-tiles = 13 us for 3 rows).  Here a workgroup of 4 waves owns 16 output columns of one group, wave w
// the quarter w of the contraction: EVERY fragment of its quarter — 16 W rows and the (up to 16) A rows, 16 bytes per lane — is requested before
// the first MFMA (one memory round trip), the four partial 16 x 16 tiles meet in LDS, wave 0 runs the epilogue.  bf16, K % 128 == 0, K <= 2048.
struct SkinnyLaunch {
    SeaGemmGroup g[SEA_MAX_GROUPS];
    int blk_start[SEA_MAX_GROUPS + 1];
    int n_groups;
};

__device__ __forceinline__ int find_group_blk(const SkinnyLaunch& launch, int bid) {
    int gi = 0;
    while (gi + 1 < launch.n_groups && bid >= launch.blk_start[gi + 1]) ++gi;
    return gi;
}

// epilogue operands of the few-row kernels (this lane: output row r, columns n0 + 4 g .. + 3), requested by wave 0 BEFORE the contraction: behind it they
// were a second memory round trip on a launch that is little more than one (12 such launches per KV-cache step at the shipped widths)
struct SkinnyEpiPre {
    float bv[4], rv[4];
};
__device__ __forceinline__ void skinny_epilogue_request(const SeaGemmGroup& G, int r, int g, int n0, SkinnyEpiPre& E) {
    const int nc = n0 + g * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) E.bv[q] = E.rv[q] = 0.f;
    if (r >= G.M || nc >= G.N) return;
    if (G.bias != nullptr) load4(G.bias + nc, E.bv);
    if (G.R != nullptr) load4(G.R + (int64_t)r * G.ldr + nc, E.rv);
}

// epilogue of the few-row kernels: bias, GELU (act 1, pre-activation kept in Z), residual, fp32 / bf16 outputs
__device__ __forceinline__ void skinny_epilogue(const SeaGemmGroup& G, const f32x4& acc, int r, int g, int n0, const SkinnyEpiPre& E) {
    const int nc = n0 + g * 4;
    if (r >= G.M || nc >= G.N) return;
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (G.bias != nullptr) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += E.bv[q] * G.bias_scale;
    }
    if (G.act == 1) {
        if (G.Z != nullptr) store4(static_cast<__bf16*>(G.Z) + (int64_t)r * G.ldz + nc, v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = gelu_erf(v[q]);
    }
    if (G.R != nullptr) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += E.rv[q];
    }
    if (G.C32 != nullptr) store4(G.C32 + (int64_t)r * G.ldc32 + nc, v[0], v[1], v[2], v[3]);
    if (G.Cact != nullptr) store4(static_cast<__bf16*>(G.Cact) + (int64_t)r * G.ldcact + nc, v[0], v[1], v[2], v[3]);
}

template <int KSTEPS>   // 32-wide contraction steps per wave = K / 128
__device__ __forceinline__ void skinny_quarter(const __bf16* Arow, const __bf16* Wrow, int k0, f32x4& acc) {
    uint4 wf[KSTEPS], af[KSTEPS];
#pragma unroll
    for (int i = 0; i < KSTEPS; ++i) {
        wf[i] = *reinterpret_cast<const uint4*>(Wrow + k0 + i * 32);
        af[i] = *reinterpret_cast<const uint4*>(Arow + k0 + i * 32);
    }
#pragma unroll
    for (int i = 0; i < KSTEPS; ++i) mma16<__bf16>(wf[i], af[i], acc);
}

__global__ __launch_bounds__(256) void gemm_skinny_kernel(const SkinnyLaunch L) {
    __shared__ __attribute__((aligned(16))) float red[3][64][4];
    const int gi = find_group_blk(L, (int)blockIdx.x);
    const SeaGemmGroup& G = L.g[gi];
    const int n0 = ((int)blockIdx.x - L.blk_start[gi]) * 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, g = lane >> 4;
    const int m = r < G.M ? r : G.M - 1;                  // rows beyond M repeat the last row; their results are never stored
    const int n = n0 + r < G.N ? n0 + r : G.N - 1;
    const int kq = G.K >> 2;                              // this wave's quarter of the contraction
    const __bf16* Arow = static_cast<const __bf16*>(G.A) + (int64_t)m * G.lda + g * 8;
    const __bf16* Wrow = static_cast<const __bf16*>(G.W) + (int64_t)n * G.ldw + g * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    SkinnyEpiPre epi;
    if (wave == 0) skinny_epilogue_request(G, r, g, n0, epi);
    const int k0 = wave * kq;
    switch (kq >> 5) {   // block-uniform
        case 2: skinny_quarter<2>(Arow, Wrow, k0, acc); break;
        case 4: skinny_quarter<4>(Arow, Wrow, k0, acc); break;
        case 8: skinny_quarter<8>(Arow, Wrow, k0, acc); break;
        case 16: skinny_quarter<16>(Arow, Wrow, k0, acc); break;
        default:
            for (int i = 0; i < (kq >> 5); ++i)
                mma16<__bf16>(*reinterpret_cast<const uint4*>(Wrow + k0 + i * 32), *reinterpret_cast<const uint4*>(Arow + k0 + i * 32), acc);
    }
    if (wave > 0) *reinterpret_cast<float4*>(red[wave - 1][lane]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const float4 p = *reinterpret_cast<const float4*>(red[w][lane]);
        acc[0] += p.x; acc[1] += p.y; acc[2] += p.z; acc[3] += p.w;
    }
    skinny_epilogue(G, acc, r, g, n0, epi);
}

// The same for LONG contractions (fc2 of the shipped widths: K = 8192 / 16384 for one row per field; the tiled kernels give such a launch to N / 128
// workgroups that each walk 4 MB of weights K-tile by K-tile: 81 us at the multiphase width).  Eight waves per 16 output columns, wave w the eighth w of the
// contraction in rounds of 8 steps, the next round's fragments requested before the current round's MFMAs (two register sets): 128 KiB of weights in flight per
// workgroup, one workgroup per 16 columns -> every CU streams its own 16 weight rows end to end.  K % 2048 == 0 (whole rounds).
__global__ __launch_bounds__(512) void gemm_skinny_long_kernel(const SkinnyLaunch L) {
    constexpr int NWV = 8, RS = 8;
    __shared__ __attribute__((aligned(16))) float red[NWV - 1][64][4];
    const int gi = find_group_blk(L, (int)blockIdx.x);
    const SeaGemmGroup& G = L.g[gi];
    const int n0 = ((int)blockIdx.x - L.blk_start[gi]) * 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, g = lane >> 4;
    const int m = r < G.M ? r : G.M - 1;
    const int n = n0 + r < G.N ? n0 + r : G.N - 1;
    const int kw = G.K / NWV;                             // this wave's share of the contraction
    const __bf16* Arow = static_cast<const __bf16*>(G.A) + (int64_t)m * G.lda + g * 8 + wave * kw;
    const __bf16* Wrow = static_cast<const __bf16*>(G.W) + (int64_t)n * G.ldw + g * 8 + wave * kw;
    const int rounds = kw / (32 * RS);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    SkinnyEpiPre epi;
    if (wave == 0) skinny_epilogue_request(G, r, g, n0, epi);
    uint4 w0[RS], a0[RS], w1[RS], a1[RS];
#pragma unroll
    for (int i = 0; i < RS; ++i) {
        w0[i] = *reinterpret_cast<const uint4*>(Wrow + i * 32);
        a0[i] = *reinterpret_cast<const uint4*>(Arow + i * 32);
    }
    for (int rd = 0; rd < rounds; rd += 2) {
        const bool more1 = rd + 1 < rounds, more2 = rd + 2 < rounds;
        if (more1) {
#pragma unroll
            for (int i = 0; i < RS; ++i) {
                w1[i] = *reinterpret_cast<const uint4*>(Wrow + (rd + 1) * 32 * RS + i * 32);
                a1[i] = *reinterpret_cast<const uint4*>(Arow + (rd + 1) * 32 * RS + i * 32);
            }
        }
#pragma unroll
        for (int i = 0; i < RS; ++i) mma16<__bf16>(w0[i], a0[i], acc);
        if (more2) {
#pragma unroll
            for (int i = 0; i < RS; ++i) {
                w0[i] = *reinterpret_cast<const uint4*>(Wrow + (rd + 2) * 32 * RS + i * 32);
                a0[i] = *reinterpret_cast<const uint4*>(Arow + (rd + 2) * 32 * RS + i * 32);
            }
        }
        if (more1) {
#pragma unroll
            for (int i = 0; i < RS; ++i) mma16<__bf16>(w1[i], a1[i], acc);
        }
    }
    if (wave > 0) *reinterpret_cast<float4*>(red[wave - 1][lane]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < NWV - 1; ++w) {
        const float4 p = *reinterpret_cast<const float4*>(red[w][lane]);
        acc[0] += p.x; acc[1] += p.y; acc[2] += p.z; acc[3] += p.w;
    }
    skinny_epilogue(G, acc, r, g, n0, epi);
}

// ---------------------------------------------------------------------------------------------- QKV + RoPE epilogue
template <typename T, int BM, int BN, bool DMA>
__global__ __launch_bounds__(256) void qkv_rope_kernel(const QkvLaunch L) {
    using C = GemmCfg<T, BM, BN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int gi = find_group(L, bid);
    const SeaQkvGroup& G = L.g[gi];
    const int t = bid - L.tile_start[gi];
    const int tiles_n = (G.N + BN - 1) / BN;
    const int tm = t / tiles_n, tn = t - tm * tiles_n;

    GemmMainloop<T, BM, BN> ml;
    ml.A = static_cast<const T*>(G.A);
    ml.W = static_cast<const T*>(G.W);
    ml.a_seg_stride = 0;
    ml.lda = G.lda; ml.ldw = G.ldw; ml.M = G.M; ml.N = G.N; ml.K = G.K; ml.n_seg = 1;
    ml.m0 = tm * BM; ml.n0 = tn * BN;
    f32x4 acc[C::MI][C::NI];
    if constexpr (DMA) ml.run_dma(smem, acc);
    else ml.run_single(smem, acc);

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave >> 1, wn = wave & 1, r = lane & 15, g = lane >> 4;
    const int H = L.c.H, hd = L.c.hd, Tlen = L.c.T, cap = L.c.cap;
    const int Ea = H * hd, hd2 = hd >> 1;
    // a 16-column MFMA block lies inside one head of one of q / k / v when hd and the group's column offset are multiples of 16: part, head and
    // the q / k / v branch are then WAVE-uniform (scalar branches, one integer division per block instead of one per lane)
    const bool blk_uniform = (hd & 15) == 0 && (G.col0 & 15) == 0;
    const float2* rope = reinterpret_cast<const float2*>(L.c.rope);
    T* Qo = static_cast<T*>(G.Qout);
    T* Ko = static_cast<T*>(G.Kout);
    T* Vto = static_cast<T*>(G.Vtout);
    T* Vo = static_cast<T*>(G.Vout);
    // per-row bookkeeping once (the integer division by T is ~20 instructions; it used to run per 4-column piece: PMC showed 19 VALU
    // instructions per MFMA in this kernel), 32-bit element offsets (the host checks the tensors stay below 2^31 elements)
    int rb[C::MI], rt[C::MI];
    bool rok[C::MI];
#pragma unroll
    for (int i = 0; i < C::MI; ++i) {
        const int m = ml.m0 + wm * C::WTM + i * 16 + r;
        rok[i] = m < G.M;
        const int mc = rok[i] ? m : G.M - 1;
        rb[i] = mc / Tlen;
        rt[i] = mc - rb[i] * Tlen;
    }
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
        const int n = ml.n0 + wn * C::WTN + j * 16 + g * 4;  // 4 consecutive columns = two (even, odd) rotation pairs
        if (n >= G.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (G.bias != nullptr) load4(G.bias + n, bv);
        const int nn = G.col0 + n;       // column in the virtual [q | k | v] row; hd % 4 == 0 keeps the 4 columns in one head
        int part, h, dd;
        if (blk_uniform) {   // block-uniform
            const int nb = __builtin_amdgcn_readfirstlane(nn - g * 4);   // first column of the 16-column block (the same in every lane)
            part = nb >= 2 * Ea ? 2 : (nb >= Ea ? 1 : 0);                // 0 q, 1 k, 2 v
            const int hb = nb - part * Ea;
            h = hb / hd;
            dd = hb - h * hd + g * 4;
        } else {
            part = nn >= 2 * Ea ? 2 : (nn >= Ea ? 1 : 0);
            const int hcol = nn - part * Ea;
            h = hcol / hd;
            dd = hcol - h * hd;
        }
        // the (cos, sin) pairs of every row block of this column block, requested together ahead of their use
        float4 csr[C::MI];
#pragma unroll
        for (int i = 0; i < C::MI; ++i) {
            csr[i] = make_float4(1.f, 0.f, 1.f, 0.f);
            if (part < 2) csr[i] = *reinterpret_cast<const float4*>(rope + (uint32_t)(L.c.pos0 + rt[i]) * (uint32_t)hd2 + (dd >> 1));   // rt is clamped: always in range
        }
#pragma unroll
        for (int i = 0; i < C::MI; ++i) {
            if (!rok[i]) continue;
            const int tt = rt[i], pos = L.c.pos0 + tt;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = acc[i][j][q] + bv[q];
            const uint32_t bh = (uint32_t)(rb[i] * H + h);
            if (part < 2) {
                // (xe + i xo)(c + i s): even' = xe c - xo s ; odd' = xe s + xo c   (fp32, before rounding)
                const float4 cs = csr[i];   // two (cos, sin) pairs
                float o[4];
                rope_pair(v[0], v[1], cs.x, cs.y, o[0], o[1]);   // scalar-lane ops: the packed form of this rotation hits the gfx950 erratum (sea_common.hpp)
                rope_pair(v[2], v[3], cs.z, cs.w, o[2], o[3]);
                if (part == 0) {
                    const float sc = L.c.q_scale;
                    store4(Qo + ((bh * (uint32_t)Tlen + tt) * (uint32_t)hd + dd), o[0] * sc, o[1] * sc, o[2] * sc, o[3] * sc);
                } else {
                    store4(Ko + ((bh * (uint32_t)cap + pos) * (uint32_t)hd + dd), o[0], o[1], o[2], o[3]);
                }
            } else {
                T* dst = Vto + ((bh * (uint32_t)hd + dd) * (uint32_t)cap + pos);
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[(uint32_t)q * (uint32_t)cap] = from_f32<T>(v[q]);
                if (Vo != nullptr) store4(Vo + ((bh * (uint32_t)cap + pos) * (uint32_t)hd + dd), v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- QKV + RoPE for a few rows
// The same split of the contraction over four waves as gemm_skinny_kernel, with the epilogue of qkv_rope_kernel (bias, rotary embedding on q / k,
// q scale, attention layouts) on the one 16-column block of the workgroup.  M <= 16 rows, bf16, K % 128 == 0, K <= 2048, head dim and the group's
// column offset multiples of 16 (a block lies inside one head of one of q / k / v).
struct QkvSkinnyLaunch {
    SeaQkvGroup g[SEA_MAX_GROUPS];
    int blk_start[SEA_MAX_GROUPS + 1];
    int n_groups;
    SeaQkvCommon c;
};

__global__ __launch_bounds__(256) void qkv_skinny_kernel(const QkvSkinnyLaunch L) {
    using T = __bf16;
    __shared__ __attribute__((aligned(16))) float red[3][64][4];
    int gi = 0;
    while (gi + 1 < L.n_groups && (int)blockIdx.x >= L.blk_start[gi + 1]) ++gi;
    const SeaQkvGroup& G = L.g[gi];
    const int n0 = ((int)blockIdx.x - L.blk_start[gi]) * 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, g = lane >> 4;
    const int m = r < G.M ? r : G.M - 1;
    const int nr = n0 + r < G.N ? n0 + r : G.N - 1;
    const int kq = G.K >> 2;
    const T* Arow = static_cast<const T*>(G.A) + (int64_t)m * G.lda + g * 8;
    const T* Wrow = static_cast<const T*>(G.W) + (int64_t)nr * G.ldw + g * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // epilogue operands (bias, the rotary pair of this lane's columns): requested by wave 0 before the contraction, like skinny_epilogue_request
    const int H = L.c.H, hd = L.c.hd, Tlen = L.c.T, cap = L.c.cap;
    const int Ea = H * hd, hd2 = hd >> 1;
    const int n = n0 + g * 4;                             // this lane: row r, columns n .. n + 3 of the group
    const int nb = G.col0 + n0;                           // first column of the block in the virtual [q | k | v] row (block-uniform)
    const int part = nb >= 2 * Ea ? 2 : (nb >= Ea ? 1 : 0);
    const int hb = nb - part * Ea;
    const int h = hb / hd;
    const int dd = hb - h * hd + g * 4;
    const int bidx = r / Tlen, tt = r - bidx * Tlen, pos = L.c.pos0 + tt;
    const bool live = r < G.M && n < G.N;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    float4 cs = make_float4(1.f, 0.f, 1.f, 0.f);
    if (wave == 0 && live) {
        if (G.bias != nullptr) load4(G.bias + n, bv);
        if (part < 2) cs = *reinterpret_cast<const float4*>(reinterpret_cast<const float2*>(L.c.rope) + (uint32_t)pos * (uint32_t)hd2 + (dd >> 1));
    }
    const int k0 = wave * kq;
    switch (kq >> 5) {   // block-uniform
        case 2: skinny_quarter<2>(Arow, Wrow, k0, acc); break;
        case 4: skinny_quarter<4>(Arow, Wrow, k0, acc); break;
        case 8: skinny_quarter<8>(Arow, Wrow, k0, acc); break;
        case 16: skinny_quarter<16>(Arow, Wrow, k0, acc); break;
        default:
            for (int i = 0; i < (kq >> 5); ++i)
                mma16<T>(*reinterpret_cast<const uint4*>(Wrow + k0 + i * 32), *reinterpret_cast<const uint4*>(Arow + k0 + i * 32), acc);
    }
    if (wave > 0) *reinterpret_cast<float4*>(red[wave - 1][lane]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const float4 p = *reinterpret_cast<const float4*>(red[w][lane]);
        acc[0] += p.x; acc[1] += p.y; acc[2] += p.z; acc[3] += p.w;
    }
    if (!live) return;
    float v[4] = {acc[0] + bv[0], acc[1] + bv[1], acc[2] + bv[2], acc[3] + bv[3]};   // bv is zero without a bias
    const uint32_t bh = (uint32_t)(bidx * H + h);
    if (part < 2) {
        float o[4];
        rope_pair(v[0], v[1], cs.x, cs.y, o[0], o[1]);
        rope_pair(v[2], v[3], cs.z, cs.w, o[2], o[3]);
        if (part == 0) {
            const float sc = L.c.q_scale;
            store4(static_cast<T*>(G.Qout) + ((bh * (uint32_t)Tlen + tt) * (uint32_t)hd + dd), o[0] * sc, o[1] * sc, o[2] * sc, o[3] * sc);
        } else {
            store4(static_cast<T*>(G.Kout) + ((bh * (uint32_t)cap + pos) * (uint32_t)hd + dd), o[0], o[1], o[2], o[3]);
        }
    } else {
        T* dst = static_cast<T*>(G.Vtout) + ((bh * (uint32_t)hd + dd) * (uint32_t)cap + pos);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[(uint32_t)q * (uint32_t)cap] = from_f32<T>(v[q]);
        if (G.Vout != nullptr) store4(static_cast<T*>(G.Vout) + ((bh * (uint32_t)cap + pos) * (uint32_t)hd + dd), v[0], v[1], v[2], v[3]);
    }
}

// ---------------------------------------------------------------------------------------------- host side
template <typename K>
static int set_lds(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess
               ? 0 : -1;
}

static int pick_tile(long tiles128, bool dma, long kmax = 0) {
    // 64x64 tiles until the launch is a couple of rounds of 128x128 tiles deep: below that its time is one tile's latency, which the small
    // tile halves.  Measured crossovers on the cfg2/cfg3 shapes at B = 1, 2, 4, 8 (single-buffered register-staged main loop):
    // condition GEMM 672 tiles 31.8 us (64) vs 28.0 (128); fc1 768: 22.0 vs 21.2, 6144: 135 vs 118; qkv 288: 16.7 vs 19.5, 2304: 61 vs 59;
    // LDS-DMA main loop (fc2) 384: 70 vs 79, 768: 158 vs 127.
    static const int forced = sea_tune("gemm_tile", 0);  // tuning aid
    if (forced == 64 || forced == 128) return forced;
    // very long contractions with the chip nearly filled by 128 x 128 tiles (the reference's multiphase MLP: M = 796, N = 2048, K = 16384 -> 224 tiles): the launch
    // runs at the rate of its L2 -> LDS operand traffic, (BM + BN) K per tile — 3.5 GB with 64 x 64 tiles, 1.9 GB with 128 x 128: mlp.fc2 291 -> 173 us,
    // bwd.fc1.dgrad 289 -> 166 us (round 4; at K = 8192 with 112 tiles the 64 tile stays faster: 60 against 94 us)
    if (dma && kmax >= 16384 && tiles128 >= 192) return 128;
    return tiles128 >= (dma ? 512 : 600) ? 128 : 64;
}

extern "C" int sea_gemm_grouped(const SeaGemmGroup* groups, int n_groups, int dtype, void* stream) {
    SEA_REQUIRE(groups != nullptr && n_groups >= 1 && n_groups <= SEA_MAX_GROUPS, "sea_gemm_grouped: n_groups=%d out of range", n_groups);
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_gemm_grouped: bad dtype %d", dtype);
    const int epc = dtype == SEA_BF16 ? 8 : 4;
    long t128 = 0, t64 = 0;
    int n_silu = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaGemmGroup& G = groups[i];
        SEA_REQUIRE((G.A || G.silu_c) && G.W, "sea_gemm_grouped[%d]: null operand", i);
        n_silu += G.silu_c != nullptr;
        if (G.silu_c != nullptr)
            SEA_REQUIRE(G.silu_w1 && G.silu_b1 && sea_aligned16(G.silu_w1) && sea_aligned16(G.silu_b1) && G.n_seg == 1 && G.K <= 1024 && G.K % 8 == 0,
                        "sea_gemm_grouped[%d]: generated A needs w1 / b1 (16-byte aligned), n_seg = 1, K <= 1024", i);
        SEA_REQUIRE(G.M >= 1 && G.N >= 1 && G.K >= 8 && G.n_seg >= 1, "sea_gemm_grouped[%d]: bad shape M=%d N=%d K=%d n_seg=%d", i, G.M, G.N, G.K, G.n_seg);
        SEA_REQUIRE(G.K % 8 == 0 && (G.silu_c || G.lda % epc == 0) && G.ldw % epc == 0 && G.a_seg_stride % epc == 0,
                    "sea_gemm_grouped[%d]: K=%d lda=%d ldw=%d must be multiples of 8/%d", i, G.K, G.lda, G.ldw, epc);
        SEA_REQUIRE(sea_aligned16(G.A) && sea_aligned16(G.W), "sea_gemm_grouped[%d]: A/W not 16-byte aligned", i);
        SEA_REQUIRE((G.silu_c || G.lda >= G.K) && G.ldw >= G.K, "sea_gemm_grouped[%d]: leading dimension smaller than K", i);
        SEA_REQUIRE(G.C32 || G.Cact, "sea_gemm_grouped[%d]: no output", i);
        SEA_REQUIRE(G.drop.thr >= 0 && G.drop.thr <= 255 && (G.drop.thr == 0 || (G.drop.mode >= 1 && G.drop.mode <= 3)), "sea_gemm_grouped[%d]: bad dropout", i);
        SEA_REQUIRE(G.N % 4 == 0, "sea_gemm_grouped[%d]: N=%d must be a multiple of 4", i, G.N);
        SEA_REQUIRE(G.act >= 0 && G.act <= 2 && (G.act != 2 || G.Z) && (!G.Z || (G.ldz >= G.N && G.ldz % 4 == 0)), "sea_gemm_grouped[%d]: bad act/Z", i);
        SEA_REQUIRE(sea_aligned16(G.bias) && sea_aligned16(G.R) && sea_aligned16(G.C32) && sea_aligned16(G.Cact) && sea_aligned16(G.Z),
                    "sea_gemm_grouped[%d]: bias/R/C32/Cact/Z must be 16-byte aligned", i);
        SEA_REQUIRE((!G.R || G.ldr % 4 == 0) && (!G.C32 || G.ldc32 % 4 == 0) && (!G.Cact || G.ldcact % 4 == 0), "sea_gemm_grouped[%d]: output strides must be multiples of 4", i);
        SEA_REQUIRE((!G.R || G.ldr >= G.N) && (!G.C32 || G.ldc32 >= G.N) && (!G.Cact || G.ldcact >= G.N), "sea_gemm_grouped[%d]: output stride < N", i);
        t128 += (long)((G.M + 127) / 128) * ((G.N + 127) / 128);
        t64 += (long)((G.M + 63) / 64) * ((G.N + 63) / 64);
    }
    // a few rows (a KV-cache rollout step): one workgroup per 16 output columns, the contraction split over its four waves
    static const int skinny_env = sea_tune("gemm_skinny", 1);  // tuning aid: 0 keeps the tiled kernels
    bool skinny = skinny_env != 0 && dtype == SEA_BF16 && n_silu == 0;
    for (int i = 0; i < n_groups && skinny; ++i) {
        const SeaGemmGroup& G = groups[i];
        skinny = G.M <= 16 && G.n_seg == 1 && G.K % 128 == 0 && (G.K <= 2048 || (G.K % 2048 == 0 && G.K <= 32768)) && G.act <= 1 && G.drop.thr == 0 && G.N % 4 == 0;
    }
    bool long_k = false;   // one kernel per launch: the long-contraction form as soon as one group needs it (every group must then have whole rounds)
    for (int i = 0; i < n_groups && skinny; ++i) long_k = long_k || groups[i].K > 2048;
    for (int i = 0; i < n_groups && skinny && long_k; ++i) skinny = groups[i].K % 2048 == 0;
    if (skinny) {
        SkinnyLaunch S;
        memset(&S, 0, sizeof(S));
        int blocks = 0;
        for (int i = 0; i < n_groups; ++i) {
            S.g[i] = groups[i];
            S.blk_start[i] = blocks;
            blocks += (groups[i].N + 15) / 16;
        }
        S.blk_start[n_groups] = blocks;
        S.n_groups = n_groups;
        if (long_k) gemm_skinny_long_kernel<<<dim3(blocks), dim3(512), 0, static_cast<hipStream_t>(stream)>>>(S);
        else gemm_skinny_kernel<<<dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(S);
        SEA_CHECK_LAUNCH("sea_gemm_grouped");
        return SEA_OK;
    }
    // LDS-DMA ring: needs whole K-tiles (128 bytes of K per row per stage) and pays off only on long contractions (its 4 stages
    // cost a workgroup per CU at 128x128; measured: K = 2048 +5 %, K = 256 -20 % against the register-staged double buffer)
    static const int dma_min_k = sea_tune("gemm_dma_min_k", 1024);  // tuning aid
    SEA_REQUIRE(n_silu == 0 || n_silu == n_groups, "sea_gemm_grouped: generated-A groups cannot share a launch with ordinary ones");
    const bool silu = n_silu > 0;
    bool dma = !silu;
    for (int i = 0; i < n_groups; ++i)
        dma = dma && (groups[i].K % (dtype == SEA_BF16 ? 64 : 32) == 0) && (long)groups[i].K * groups[i].n_seg >= dma_min_k;
    // ... and only while the launch is a few tiles deep per CU (its time then is the serial chain of K-tiles of one tile, which the ring
    // shortens: fc2 at B = 1 23 us vs 28); with many tiles per CU the single-buffered loop's occupancy wins (B = 8: 109 us vs 128)
    static const int dma_force = sea_tune("gemm_dma_force", 0);  // tuning aid
    // ... or, round 4, when it is at least two 128 x 128 tiles per CU deep: then with TWO ring stages (64 KiB: two workgroups per CU whose waves fill each other's
    // fragment-read and barrier waits) — 4096^3 183 -> 141 us (977 TFLOP/s), the multiphase fc1 shape (796 x 16384 x 2048, two fields) 139 -> 120 us, cfg3's fc2
    // 67.9 -> 62.0 us (tools/skinny_probe.py, tools/bench_ops.py)
    const bool dma_long = dma && t128 >= 512;
    dma = dma && (t64 <= 1536 || dma_long || dma_force);
    (void)t64;
    long kmax_all = 0;
    for (int i = 0; i < n_groups; ++i) kmax_all = (long)groups[i].K * groups[i].n_seg > kmax_all ? (long)groups[i].K * groups[i].n_seg : kmax_all;
    const int tile = pick_tile(t128, dma, kmax_all);
    GemmLaunch L;
    memset(&L, 0, sizeof(L));
    L.n_groups = n_groups;
    int total = 0;
    for (int i = 0; i < n_groups; ++i) {
        L.g[i] = groups[i];
        L.tile_start[i] = total;
        total += ((groups[i].M + tile - 1) / tile) * ((groups[i].N + tile - 1) / tile);
    }
    L.tile_start[n_groups] = total;
    // Skinny M (the reference's own widths: M = B T ~ 800 rows against N or K = 16384, configs/multiphase_flow.py:112-141): W is the big operand (67 MB per
    // field MLP matrix) and every row tile streams all of it.  With the column tile fastest (the default: neighbours share an A panel) the few row tiles of a
    // W panel are tiles_n apart, i.e. on different XCDs — every XCD's L2 pulls the WHOLE of W from the Infinity Cache / HBM (8 x 134 MB for mlp.fc2: the
    // launch ran at the rate of that traffic, 292 us).  With the row tile fastest the tiles of a W panel are neighbours inside one XCD's contiguous range
    // (xcd_remap) and walk the panel together: W crosses the fabric once, the (small) A matrix once per XCD.  SEA_TUNE=gemm_n_major=0|1 forces.
    {
        static const int nm_forced = sea_tune("gemm_n_major", -1);
        for (int i = 0; i < n_groups; ++i) {
            const bool skinny_m = (long)groups[i].N >= 2L * groups[i].M && (long)groups[i].N * groups[i].K * (dtype == SEA_BF16 ? 2 : 4) >= (4L << 20);
            if (nm_forced == 1 || (nm_forced < 0 && skinny_m)) L.n_major |= 1u << i;
        }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // short contractions (K = 256 / 512) of long plain launches: the weight-stationary streaming kernel (gemm_ws.hip: weights in registers, activation rows once
    // per 256 output columns, the tile's store under the next tile's MFMAs)
    if (dtype == SEA_BF16 && !silu && sea_gemm_ws_try(groups, n_groups, s)) {
        SEA_CHECK_LAUNCH("sea_gemm_grouped");
        return SEA_OK;
    }
    // 256 x 256 tiles (gemm256.hip: a wave owns 128 x 64, 96 B/clk of fragment reads at the full MFMA rate instead of the 128 x 128 tile's 128) for launches that
    // give every CU at least one such tile and a half.  SEA_TUNE=gemm256=0|1 forces.
    {
        static const int g256 = sea_tune("gemm256", -1);
        long t256 = 0;
        for (int i = 0; i < n_groups; ++i) t256 += (long)((groups[i].M + 255) / 256) * ((groups[i].N + 255) / 256);
        // measured (tools/bench_ops.py, SEA_TUNE=gemm256=0|1): 4096^3 139 -> 118 us (1164 TFLOP/s); 9 x (16192, 512, 512) 128.5 -> 122 us; 3 x (16192, 256, 2048) 61.9 -> 58.7;
        // the multiphase fc1 shape 121.6 -> 115; but 3 x (16192, 2048, 256) 91 -> 96 (four K-tiles: the launch is its epilogue) — hence the contraction floor
        const bool deep = (t256 >= 384 && kmax_all >= 512) || (t256 >= 192 && kmax_all >= 2048);
        if (dtype == SEA_BF16 && !silu && (g256 == 1 || (g256 < 0 && deep)) && sea_gemm256_try(groups, n_groups, L.n_major, s)) {
            SEA_CHECK_LAUNCH("sea_gemm_grouped");
            return SEA_OK;
        }
    }
    bool plain = true;
    for (int i = 0; i < n_groups; ++i) plain = plain && groups[i].act == 0 && groups[i].drop.thr == 0;
    SEA_REQUIRE(!silu || plain, "sea_gemm_grouped: generated-A launches take no activation / dropout epilogue");
    int kmax = 0;
    for (int i = 0; i < n_groups; ++i) kmax = groups[i].K > kmax ? groups[i].K : kmax;
    L.single_buffer = !dma;
    // two ring stages (64 KiB at 128 x 128: two workgroups per CU) when the launch has at least two 128 x 128 tiles per CU: their waves fill each other's
    // fragment-read and barrier waits (one workgroup per CU ran the shipped MLP GEMMs at a third of a CU's MFMA rate).  SEA_TUNE=gemm_dma_ns=2|4 forces.
    static const int ns_forced = sea_tune("gemm_dma_ns", 0);
    L.dma_ns = ns_forced == 2 || ns_forced == 4 ? ns_forced : ((dma && tile == 128 && total >= 2 * 256 && t64 > 1536) ? 2 : 4);
    const int sb = L.single_buffer;
#define LAUNCH_GEMM(TT, BMN, DM)                                                                                          \
    do {                                                                                                                  \
        const int main_ = DM ? (L.dma_ns == 2 ? 2 : 4) * GemmCfg<TT, BMN, BMN>::BUF_BYTES : (sb ? GemmCfg<TT, BMN, BMN>::BUF_BYTES : GemmCfg<TT, BMN, BMN>::LDS_BYTES); \
        constexpr int stage_ = BMN * (BMN * (int)sizeof(TT) + 16);                                                         \
        const int lds_ = main_ > stage_ ? main_ : stage_;                                                                  \
        static int once = set_lds(gemm_grouped_kernel<TT, BMN, BMN, DM, false>, DM ? GemmMainloop<TT, BMN, BMN>::DMA_LDS_BYTES : (GemmCfg<TT, BMN, BMN>::LDS_BYTES > stage_ ? GemmCfg<TT, BMN, BMN>::LDS_BYTES : stage_)); \
        (void)once;                                                                                                       \
        static int once2 = set_lds(gemm_grouped_kernel<TT, BMN, BMN, DM, true>, DM ? GemmMainloop<TT, BMN, BMN>::DMA_LDS_BYTES : (GemmCfg<TT, BMN, BMN>::LDS_BYTES > stage_ ? GemmCfg<TT, BMN, BMN>::LDS_BYTES : stage_)); \
        (void)once2;                                                                                                      \
        if (plain) gemm_grouped_kernel<TT, BMN, BMN, DM, true><<<dim3(total), dim3(256), lds_, s>>>(L);                    \
        else gemm_grouped_kernel<TT, BMN, BMN, DM, false><<<dim3(total), dim3(256), lds_, s>>>(L);                         \
    } while (0)
    // generated-A launches on the ordinary square tiles.  (64 x 256 tiles — half the recomputation of the operand, which every column tile of a
    // row panel repeats — measured slower: 36.7 against 33.7 us at cfg2, 194 against 174 us at B = 8: two waves per SIMD instead of three.)
#define LAUNCH_GEMM_SILU(TT, BMN)                                                                                          \
    do {                                                                                                                  \
        constexpr int stage_ = BMN * (BMN * (int)sizeof(TT) + 16);                                                         \
        constexpr int main_ = GemmCfg<TT, BMN, BMN>::BUF_BYTES > stage_ ? GemmCfg<TT, BMN, BMN>::BUF_BYTES : stage_;       \
        static int once = set_lds(gemm_grouped_kernel<TT, BMN, BMN, false, true, true>, main_ + 8 * 1024);                 \
        (void)once;                                                                                                       \
        L.silu_lds_off = main_;                                                                                           \
        gemm_grouped_kernel<TT, BMN, BMN, false, true, true><<<dim3(total), dim3(256), main_ + 8 * kmax, s>>>(L);          \
    } while (0)
#define LAUNCH_GEMM_T(TT)                                                 \
    do {                                                                  \
        if (silu) { if (tile == 128) LAUNCH_GEMM_SILU(TT, 128); else LAUNCH_GEMM_SILU(TT, 64); } \
        else if (tile == 128) { if (dma) LAUNCH_GEMM(TT, 128, true); else LAUNCH_GEMM(TT, 128, false); } \
        else { if (dma) LAUNCH_GEMM(TT, 64, true); else LAUNCH_GEMM(TT, 64, false); }              \
    } while (0)
    if (dtype == SEA_BF16) LAUNCH_GEMM_T(__bf16); else LAUNCH_GEMM_T(float);
#undef LAUNCH_GEMM_T
#undef LAUNCH_GEMM_SILU
#undef LAUNCH_GEMM
    SEA_CHECK_LAUNCH("sea_gemm_grouped");
    return SEA_OK;
}

extern "C" int sea_qkv_rope_grouped(const SeaQkvGroup* groups, int n_groups, const SeaQkvCommon* common, int dtype, void* stream) {
    SEA_REQUIRE(groups != nullptr && common != nullptr && n_groups >= 1 && n_groups <= SEA_MAX_GROUPS, "sea_qkv_rope_grouped: bad arguments");
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_qkv_rope_grouped: bad dtype %d", dtype);
    const SeaQkvCommon& c = *common;
    SEA_REQUIRE(c.rope != nullptr && c.H >= 1 && c.hd >= 4 && c.hd % 4 == 0 && c.T >= 1 && c.pos0 >= 0 && c.cap >= c.pos0 + c.T,
                "sea_qkv_rope_grouped: bad common H=%d hd=%d T=%d pos0=%d cap=%d", c.H, c.hd, c.T, c.pos0, c.cap);
    const int epc = dtype == SEA_BF16 ? 8 : 4;
    const int Ea = c.H * c.hd;
    long t128 = 0, t64 = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaQkvGroup& G = groups[i];
        SEA_REQUIRE(G.A && G.W, "sea_qkv_rope_grouped[%d]: null operand", i);
        SEA_REQUIRE(G.M >= 1 && G.M % c.T == 0 && G.N >= 4 && G.N % 4 == 0 && G.K >= 8 && G.K % 8 == 0, "sea_qkv_rope_grouped[%d]: bad shape M=%d N=%d K=%d T=%d", i, G.M, G.N, G.K, c.T);
        SEA_REQUIRE(G.lda % epc == 0 && G.ldw % epc == 0 && G.lda >= G.K && G.ldw >= G.K, "sea_qkv_rope_grouped[%d]: bad strides", i);
        SEA_REQUIRE(sea_aligned16(G.A) && sea_aligned16(G.W), "sea_qkv_rope_grouped[%d]: A/W not 16-byte aligned", i);
        SEA_REQUIRE(G.col0 >= 0 && G.col0 % 4 == 0 && G.col0 + G.N <= 3 * Ea, "sea_qkv_rope_grouped[%d]: columns [%d,%d) outside [0,%d)", i, G.col0, G.col0 + G.N, 3 * Ea);
        const bool hasq = G.col0 < Ea, hask = G.col0 < 2 * Ea && G.col0 + G.N > Ea, hasv = G.col0 + G.N > 2 * Ea;
        SEA_REQUIRE((!hasq || G.Qout) && (!hask || G.Kout) && (!hasv || G.Vtout), "sea_qkv_rope_grouped[%d]: missing output pointer", i);
        SEA_REQUIRE(sea_aligned16(G.bias) && sea_aligned16(G.Qout) && sea_aligned16(G.Kout) && sea_aligned16(G.Vtout) && sea_aligned16(G.Vout),
                    "sea_qkv_rope_grouped[%d]: pointers must be 16-byte aligned", i);
        SEA_REQUIRE((int64_t)(G.M / c.T + 1) * c.H * c.cap * c.hd < ((int64_t)1 << 31) && (int64_t)(c.pos0 + c.T) * (c.hd / 2) < ((int64_t)1 << 30),
                    "sea_qkv_rope_grouped[%d]: attention tensors too large for 32-bit element offsets", i);
        t128 += (long)((G.M + 127) / 128) * ((G.N + 127) / 128);
        t64 += (long)((G.M + 63) / 64) * ((G.N + 63) / 64);
    }
    // a few rows (a KV-cache rollout step): one workgroup per 16-column block, the contraction split over its four waves
    static const int skinny_env = sea_tune("gemm_skinny", 1);
    bool skinny = skinny_env != 0 && dtype == SEA_BF16 && c.hd % 16 == 0;
    for (int i = 0; i < n_groups && skinny; ++i) skinny = groups[i].M <= 16 && groups[i].K % 128 == 0 && groups[i].K <= 2048 && groups[i].col0 % 16 == 0 && groups[i].N % 16 == 0;
    if (skinny) {
        QkvSkinnyLaunch S;
        memset(&S, 0, sizeof(S));
        int blocks = 0;
        for (int i = 0; i < n_groups; ++i) {
            S.g[i] = groups[i];
            S.blk_start[i] = blocks;
            blocks += groups[i].N / 16;
        }
        S.blk_start[n_groups] = blocks;
        S.n_groups = n_groups;
        S.c = c;
        qkv_skinny_kernel<<<dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(S);
        SEA_CHECK_LAUNCH("sea_qkv_rope_grouped");
        return SEA_OK;
    }
    bool dma = true;
    for (int i = 0; i < n_groups; ++i) dma = dma && (groups[i].K % (dtype == SEA_BF16 ? 64 : 32) == 0) && groups[i].K >= 1024;
    (void)t64;
    const int tile = pick_tile(t128, dma);
    QkvLaunch L;
    memset(&L, 0, sizeof(L));
    L.n_groups = n_groups;
    L.c = c;
    int total = 0;
    for (int i = 0; i < n_groups; ++i) {
        L.g[i] = groups[i];
        L.tile_start[i] = total;
        total += ((groups[i].M + tile - 1) / tile) * ((groups[i].N + tile - 1) / tile);
    }
    L.tile_start[n_groups] = total;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const int sbq_env = sea_tune("gemm_single", 1);
    L.single_buffer = sbq_env && !dma;
    const int sbq = L.single_buffer;
#define LAUNCH_QKV(TT, BMN, DM)                                                                                           \
    do {                                                                                                                  \
        const int lds_ = DM ? GemmMainloop<TT, BMN, BMN>::DMA_LDS_BYTES : (sbq ? GemmCfg<TT, BMN, BMN>::BUF_BYTES : GemmCfg<TT, BMN, BMN>::LDS_BYTES); \
        static int once = set_lds(qkv_rope_kernel<TT, BMN, BMN, DM>, DM ? GemmMainloop<TT, BMN, BMN>::DMA_LDS_BYTES : GemmCfg<TT, BMN, BMN>::LDS_BYTES); \
        (void)once;                                                                                                       \
        qkv_rope_kernel<TT, BMN, BMN, DM><<<dim3(total), dim3(256), lds_, s>>>(L);                                         \
    } while (0)
#define LAUNCH_QKV_T(TT)                                                  \
    do {                                                                  \
        if (tile == 128) { if (dma) LAUNCH_QKV(TT, 128, true); else LAUNCH_QKV(TT, 128, false); }   \
        else { if (dma) LAUNCH_QKV(TT, 64, true); else LAUNCH_QKV(TT, 64, false); }                \
    } while (0)
    if (dtype == SEA_BF16) LAUNCH_QKV_T(__bf16); else LAUNCH_QKV_T(float);
#undef LAUNCH_QKV_T
#undef LAUNCH_QKV
    SEA_CHECK_LAUNCH("sea_qkv_rope_grouped");
    return SEA_OK;
}
