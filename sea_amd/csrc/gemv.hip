// gemv.hip — Linear layers on a FEW rows (M <= 4 per group: the KV-cache rollout step at the shipped widths, embed_dim 1024 / 2048, where a step is one row
// per trajectory and field and its time is the launch count — 22 launches of ~4.7 us at embed_dim 1024 against a 20 us weight stream).
// sea_gemm_fewrows / sea_qkv_rope_fewrows are sea_gemm_grouped / sea_qkv_rope_grouped for such launches, with the row passes on either side folded in:
//   * the weights are read as what they are — a stream: a wave owns CW consecutive output columns, its 64 lanes split the contraction 16 bytes each
//     (one fully coalesced 1 KiB access per instruction), EVERY weight fragment of the wave (CW x K / 512 <= 32 of them) is requested before anything else;
//     the products are v_dot2c_f32_bf16 on the VALU (M <= 4: an MFMA tile would be 75-94 % padding), the 64 partial sums of a column meet by a
//     halving exchange that leaves (row, column) number l >> s in lane l, which runs the epilogue of that one element;
//   * `pre` (optional, per group): the A operand is sea_rownorm of fp32 rows (LayerNorm / AdaLN, optionally x + addend first, SeaNormGroup's arithmetic) evaluated
//     by EVERY workgroup for itself under the weight loads (K <= 2048: 8 KiB of L2 reads) — the norm launch in front of the layer disappears;
// (Not done — measured: the norm BEHIND a layer by the group's last workgroup.  With device-scope fences, 88 / 223 us for fc1 of the two shipped widths — every
// buffer_wbl2 walks an L2; with write-through stores, an arrival ticket and agent-scope loads instead, 30 us against 11.3 + 8.8 for the two launches: each hand-over
// through memory is a 1.5-2 us round trip and there are four of them, more than the launch boundary they replace.)
// Reference call sites: the nn.Linear / LayerNorm / AdaLN lines of models/temporal.py:126-146, 170-192 and models/base_blocks.py:22-25, 179-188, 271-280, 345-350 on one
// row per (trajectory, field) — the KV-cache form of the rollout loop utils/train_utils.py:202-209.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/sea_hip.h"
#include "sea_common.hpp"

namespace {

constexpr int FR_MAX_GROUPS = 8;

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot8(const uint4& w, const uint4& a, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.x), __builtin_bit_cast(bf16x2_t, a.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.y), __builtin_bit_cast(bf16x2_t, a.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.z), __builtin_bit_cast(bf16x2_t, a.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.w), __builtin_bit_cast(bf16x2_t, a.w), acc, false);
    return acc;
}

struct GemmFewLaunch {
    SeaGemmGroup g[FR_MAX_GROUPS];
    SeaNormGroup pre[FR_MAX_GROUPS];    // X == NULL: the group's A operand is read as it is
    int blk_start[FR_MAX_GROUPS + 1];
    int n_groups;
    int pre_mode;                       // 0: pre[g].X are fp32 rows; 1: rows in the activation dtype (plain LayerNorm); 3: ... followed by GELU
    float eps;
};
struct QkvFewLaunch {
    SeaQkvGroup g[FR_MAX_GROUPS];
    SeaNormGroup pre[FR_MAX_GROUPS];
    int blk_start[FR_MAX_GROUPS + 1];
    int n_groups;
    float eps;
    SeaQkvCommon c;
};

template <typename LaunchT>
__device__ __forceinline__ int few_group(const LaunchT& L, int bid) {
    int gi = 0;
    while (gi + 1 < L.n_groups && bid >= L.blk_start[gi + 1]) ++gi;
    return gi;
}

// ---------------------------------------------------------------------------------------------- the A rows of a workgroup, in LDS as bf16
// a_lds[m][q]: 16-byte piece q (columns 8 q .. 8 q + 7) of row m; K = KC * 512 columns.  Either a copy of A, or sea_rownorm of the fp32 rows of `P` (two-pass
// statistics in fp32, the gains / shifts / modulations requested together with the rows: one memory round trip).  In two halves: `request` issues every load
// BEFORE the wave's weight fragments (loads return in order: the rows are then complete while the weights are still on their way and the statistics run under
// the weight stream), `finish` turns them into the LDS rows.  `writer`: this workgroup stores x + addend.
template <int MR, int KC>
struct RowStage {
    static constexpr int NPC = KC * 64;            // pieces per row
    static constexpr int CP = (NPC + 255) / 256;   // pieces per thread of a plain copy
    float gm[8], bt[8], xv[MR][8], av[MR][8];
    uint4 mw[MR], mb[MR], cp[MR][CP];

    __device__ __forceinline__ void request(const __bf16* A, int lda, int M, const SeaNormGroup& P, int mode, int tid) {
        if (P.X == nullptr || mode != 0) {   // rows in the activation dtype: the operand itself, or the rows a LayerNorm (+ GELU) will be applied to
            const __bf16* src = P.X == nullptr ? A : static_cast<const __bf16*>(P.X);
            const int ld = P.X == nullptr ? lda : P.ldx;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
#pragma unroll
                for (int k = 0; k < CP; ++k) {
                    const int q = k * 256 + tid;
                    cp[m][k] = make_uint4(0u, 0u, 0u, 0u);
                    if (m < M && q < NPC) cp[m][k] = *reinterpret_cast<const uint4*>(src + (int64_t)m * ld + q * 8);
                }
            }
            return;
        }
        const bool valid = tid < NPC;    // KC <= 4 (checked by the host): one piece per thread
        const int i0 = tid * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) gm[e] = bt[e] = 0.f;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[m][e] = av[m][e] = 0.f;
            mw[m] = mb[m] = make_uint4(0u, 0u, 0u, 0u);
            if (m < M && valid) {
                load8(static_cast<const float*>(P.X) + (int64_t)m * P.ldx + i0, xv[m]);
                if (P.addend != nullptr) load8(P.addend + (int64_t)m * P.ldadd + i0, av[m]);
                if (P.mod != nullptr) {
                    const __bf16* mod = static_cast<const __bf16*>(P.mod) + (int64_t)m * P.ldmod;
                    mw[m] = *reinterpret_cast<const uint4*>(mod + i0);
                    mb[m] = *reinterpret_cast<const uint4*>(mod + KC * 512 + i0);
                }
            }
        }
        if (valid) {
            load8(P.gamma + i0, gm);
            if (P.beta != nullptr) load8(P.beta + i0, bt);
        }
    }

    // nn.LayerNorm (+ GELU) of rows held in the activation dtype (the hidden rows of the MLP in front of its second Linear, models/base_blocks.py:23-25), row by
    // row from the copy registers; the gains / shifts are requested once, before the statistics
    __device__ __forceinline__ void finish_act(int M, const SeaNormGroup& P, bool gelu, float eps, uint4* a_lds, float* red, int tid) {
        const int lane = tid & 63, wave = tid >> 6;
        const float inv_d = 1.0f / (float)(KC * 512);
        float g8[CP][8], b8[CP][8];
#pragma unroll
        for (int k = 0; k < CP; ++k) {
            const int q = k * 256 + tid, qc = q < NPC ? q : 0;
            load8(P.gamma + qc * 8, g8[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) b8[k][e] = 0.f;
            if (P.beta != nullptr) load8(P.beta + qc * 8, b8[k]);
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m >= M) break;   // block-uniform
            float xf[CP][8];
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < CP; ++k) {
                const bf16x8 v = __builtin_bit_cast(bf16x8, cp[m][k]);   // zeros beyond the row
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[k][e] = (float)v[e];
                s += ((xf[k][0] + xf[k][1]) + (xf[k][2] + xf[k][3])) + ((xf[k][4] + xf[k][5]) + (xf[k][6] + xf[k][7]));
            }
            s = wave_sum_xor(s, lane);
            if (lane == 0) red[m * 4 + wave] = s;
            __syncthreads();
            const float mean = (red[m * 4] + red[m * 4 + 1] + red[m * 4 + 2] + red[m * 4 + 3]) * inv_d;
            float sq = 0.f;
#pragma unroll
            for (int k = 0; k < CP; ++k) {
                if (k * 256 + tid < NPC) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float c = xf[k][e] - mean;
                        sq += c * c;
                    }
                }
            }
            sq = wave_sum_xor(sq, lane);
            if (lane == 0) red[MR * 4 + m * 4 + wave] = sq;
            __syncthreads();
            const float rstd = 1.0f / sqrtf((red[MR * 4 + m * 4] + red[MR * 4 + m * 4 + 1] + red[MR * 4 + m * 4 + 2] + red[MR * 4 + m * 4 + 3]) * inv_d + eps);
#pragma unroll
            for (int k = 0; k < CP; ++k) {
                const int q = k * 256 + tid;
                if (q < NPC) {
                    bf16x8 pk;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float o = (xf[k][e] - mean) * rstd * g8[k][e] + b8[k][e];
                        if (gelu) o = gelu_for<__bf16>(o);   // as sea_rownorm for an output that only exists in bf16
                        pk[e] = (__bf16)o;
                    }
                    a_lds[m * NPC + q] = __builtin_bit_cast(uint4, pk);
                }
            }
        }
    }

    __device__ __forceinline__ void finish(int M, const SeaNormGroup& P, int mode, bool writer, float eps, uint4* a_lds, float* red, int tid) {
        if (P.X != nullptr && mode != 0) {
            finish_act(M, P, (mode & 2) != 0, eps, a_lds, red, tid);
            return;
        }
        if (P.X == nullptr) {
#pragma unroll
            for (int m = 0; m < MR; ++m) {
#pragma unroll
                for (int k = 0; k < CP; ++k) {
                    const int q = k * 256 + tid;
                    if (m < M && q < NPC) a_lds[m * NPC + q] = cp[m][k];
                }
            }
            return;
        }
        const int lane = tid & 63, wave = tid >> 6;
        const bool valid = tid < NPC;
        const int i0 = tid * 8;
        const float inv_d = 1.0f / (float)(KC * 512);
        float mean[MR], rstd[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (P.addend != nullptr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[m][e] += av[m][e];
                if (writer && P.Xout != nullptr && m < M && valid) {
                    float* xo = P.Xout + (int64_t)m * P.ldxout + i0;
                    store4(xo, xv[m][0], xv[m][1], xv[m][2], xv[m][3]);
                    store4(xo + 4, xv[m][4], xv[m][5], xv[m][6], xv[m][7]);
                }
            }
            float s = ((xv[m][0] + xv[m][1]) + (xv[m][2] + xv[m][3])) + ((xv[m][4] + xv[m][5]) + (xv[m][6] + xv[m][7]));
            s = wave_sum_xor(s, lane);
            if (lane == 0) red[m * 4 + wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            mean[m] = (red[m * 4] + red[m * 4 + 1] + red[m * 4 + 2] + red[m * 4 + 3]) * inv_d;
            float sq = 0.f;
            if (valid) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float c = xv[m][e] - mean[m];
                    sq += c * c;
                }
            }
            sq = wave_sum_xor(sq, lane);
            if (lane == 0) red[MR * 4 + m * 4 + wave] = sq;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            rstd[m] = 1.0f / sqrtf((red[MR * 4 + m * 4] + red[MR * 4 + m * 4 + 1] + red[MR * 4 + m * 4 + 2] + red[MR * 4 + m * 4 + 3]) * inv_d + eps);
            if (m < M && valid) {
                float w8[8], b8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) w8[e] = b8[e] = 0.f;
                if (P.mod != nullptr) {
                    const bf16x8 wv = __builtin_bit_cast(bf16x8, mw[m]), bv = __builtin_bit_cast(bf16x8, mb[m]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) w8[e] = (float)wv[e], b8[e] = (float)bv[e];
                }
                bf16x8 pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gq = P.mod != nullptr ? gm[e] + 1.0f + w8[e] : gm[e];
                    const float bq = P.mod != nullptr ? bt[e] + b8[e] : bt[e];
                    pk[e] = (__bf16)((xv[m][e] - mean[m]) * rstd[m] * gq + bq);
                }
                a_lds[m * NPC + tid] = __builtin_bit_cast(uint4, pk);
            }
        }
    }
};

template <int MR, int CW, int KC>
struct FewShape {
    static constexpr int NV = MR * CW;                       // (row, column) results per wave
    static constexpr int LSTRIDE = 64 / NV;                  // lanes per result after the exchange
    static constexpr int LDS_BYTES = MR * KC * 1024 + 2 * MR * 4 * 4 + 16;
    static_assert(NV >= 1 && NV <= 64 && (NV & (NV - 1)) == 0 && CW * KC <= 32, "shape of a few-row launch");
};

// (Measured and not kept: for short contractions the A operand built in registers by every wave for itself — a lane's operand pieces are its own 16 bytes of the
// rows, so a wave can load the rows, take the statistics by wave reductions and normalise in place: no LDS, no barrier.  Slower: 0.0776 -> 0.0840 ms per step at the
// cylinder width, 0.152 -> 0.162 at the multiphase width — four times the row / gain / modulation requests per workgroup and 114-224 registers, which take the
// 768-workgroup launches from one resident round to two.)
// weights of the wave's CW columns (every fragment requested at once), the staged rows, the dot products, the exchange; returns this lane's (row, column) total
template <int MR, int CW, int KC>
__device__ __forceinline__ float few_core(const __bf16* A, const __bf16* W, int lda, int ldw, int M, int N, int nw0, const SeaNormGroup& pre, int pre_mode, bool writer, float eps,
                                          char* smem, int tid) {
    using S = FewShape<MR, CW, KC>;
    uint4* a_lds = reinterpret_cast<uint4*>(smem);
    float* red = reinterpret_cast<float*>(smem + MR * KC * 1024);
    const int lane = tid & 63;
    RowStage<MR, KC> rows;
    rows.request(A, lda, M, pre, pre_mode, tid);
    uint4 w[CW][KC];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const int n = nw0 + c < N ? nw0 + c : N - 1;         // columns beyond N repeat the last one; their results are never stored
        const uint4* wp = reinterpret_cast<const uint4*>(W + (int64_t)n * ldw) + lane;
#pragma unroll
        for (int j = 0; j < KC; ++j) w[c][j] = wp[j * 64];
    }
    rows.finish(M, pre, pre_mode, writer, eps, a_lds, red, tid);
    __syncthreads();
    float acc[S::NV];
#pragma unroll
    for (int i = 0; i < S::NV; ++i) acc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const uint4 a = a_lds[m * KC * 64 + j * 64 + lane];   // rows >= M: whatever the LDS holds; their results are never stored
#pragma unroll
            for (int c = 0; c < CW; ++c) acc[m * CW + c] = dot8(w[c][j], a, acc[m * CW + c]);
        }
    }
    return lane_scatter_sum<S::NV>(acc, lane);
}

// ---------------------------------------------------------------------------------------------- Linear (+ bias, GELU, residual), optional norm prologue
template <int MR, int CW, int KC>
__global__ __launch_bounds__(256) void gemm_fewrows_kernel(const GemmFewLaunch L) {
    using S = FewShape<MR, CW, KC>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = few_group(L, (int)blockIdx.x);
    const SeaGemmGroup& G = L.g[gi];
    const int blk = (int)blockIdx.x - L.blk_start[gi];
    const int nw0 = blk * 4 * CW + wave * CW;
    // this lane's result after the exchange: row me, column ne; its epilogue operands are requested now
    const int idx = lane / S::LSTRIDE, me = idx / CW, ne = nw0 + idx % CW;
    const bool live = lane % S::LSTRIDE == 0 && me < G.M && ne < G.N;
    float bv = 0.f, rv = 0.f;
    if (live) {
        if (G.bias != nullptr) bv = G.bias[ne];
        if (G.R != nullptr) rv = G.R[(int64_t)me * G.ldr + ne];
    }
    float v = few_core<MR, CW, KC>(static_cast<const __bf16*>(G.A), static_cast<const __bf16*>(G.W), G.lda, G.ldw, G.M, G.N, nw0, L.pre[gi], L.pre_mode, blk == 0, L.eps, smem, tid);
    if (live) {
        v += bv * G.bias_scale;
        if (G.act == 1) {
            if (G.Z != nullptr) static_cast<__bf16*>(G.Z)[(int64_t)me * G.ldz + ne] = (__bf16)v;
            v = gelu_erf(v);
        }
        v += rv;
        if (G.C32 != nullptr) G.C32[(int64_t)me * G.ldc32 + ne] = v;
        if (G.Cact != nullptr) static_cast<__bf16*>(G.Cact)[(int64_t)me * G.ldcact + ne] = (__bf16)v;
    }
}

// ---------------------------------------------------------------------------------------------- q / k / v projection + rotary embedding + cache append
template <int MR, int CW, int KC>
__global__ __launch_bounds__(256) void qkv_fewrows_kernel(const QkvFewLaunch L) {
    using S = FewShape<MR, CW, KC>;
    using T = __bf16;
    static_assert(CW >= 2, "a rotation pair lives in one wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = few_group(L, (int)blockIdx.x);
    const SeaQkvGroup& G = L.g[gi];
    const int blk = (int)blockIdx.x - L.blk_start[gi];
    const int nw0 = blk * 4 * CW + wave * CW;
    const int idx = lane / S::LSTRIDE, me = idx / CW, ne = nw0 + idx % CW;
    const bool live = lane % S::LSTRIDE == 0 && me < G.M && ne < G.N;
    const int H = L.c.H, hd = L.c.hd, Tlen = L.c.T, cap = L.c.cap;
    const int Ea = H * hd, hd2 = hd >> 1;
    const int nn = G.col0 + (ne < G.N ? ne : G.N - 1);       // column in the virtual [q | k | v] row
    const int part = nn >= 2 * Ea ? 2 : (nn >= Ea ? 1 : 0);
    const int hcol = nn - part * Ea;
    const int h = hcol / hd, dd = hcol - h * hd;
    const int mr = me < G.M ? me : G.M - 1;
    const int bidx = mr / Tlen, tt = mr - bidx * Tlen, pos = L.c.pos0 + tt;
    float bv = 0.f;
    float2 cs = make_float2(1.f, 0.f);
    if (live) {
        if (G.bias != nullptr) bv = G.bias[ne];
        if (part < 2) cs = reinterpret_cast<const float2*>(L.c.rope)[(uint32_t)pos * (uint32_t)hd2 + (dd >> 1)];
    }
    float v = few_core<MR, CW, KC>(static_cast<const T*>(G.A), static_cast<const T*>(G.W), G.lda, G.ldw, G.M, G.N, nw0, L.pre[gi], 0, false, L.eps, smem, tid);
    v += bv;
    // the other half of the rotation pair (columns 2 p, 2 p + 1 sit LSTRIDE lanes apart; both lanes of a pair are live or neither: N and col0 are even)
    const float pv = lane_xor<S::LSTRIDE>(live ? v : 0.f, lane);
    if (!live) return;
    const uint32_t bh = (uint32_t)(bidx * H + h);
    if (part < 2) {
        float oe, oo;
        if ((dd & 1) == 0) rope_pair(v, pv, cs.x, cs.y, oe, oo);
        else rope_pair(pv, v, cs.x, cs.y, oe, oo);
        const float o = (dd & 1) == 0 ? oe : oo;
        if (part == 0) static_cast<T*>(G.Qout)[(bh * (uint32_t)Tlen + tt) * (uint32_t)hd + dd] = (T)(o * L.c.q_scale);
        else static_cast<T*>(G.Kout)[(bh * (uint32_t)cap + pos) * (uint32_t)hd + dd] = (T)o;
    } else {
        static_cast<T*>(G.Vtout)[(bh * (uint32_t)hd + dd) * (uint32_t)cap + pos] = (T)v;
        if (G.Vout != nullptr) static_cast<T*>(G.Vout)[(bh * (uint32_t)cap + pos) * (uint32_t)hd + dd] = (T)v;
    }
}

// ---------------------------------------------------------------------------------------------- host side
template <typename K>
static bool few_set_lds(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}

static int few_cw(int kc) { return kc <= 8 ? 4 : (kc == 16 ? 2 : 1); }

static bool few_k_ok(int K) { return K == 512 || K == 1024 || K == 2048 || K == 4096 || K == 8192 || K == 16384; }

static int check_pre(const SeaNormGroup& P, int K, const char* who, int i) {
    SEA_REQUIRE(K <= 2048, "%s[%d]: a norm prologue needs K <= 2048 (K=%d)", who, i, K);
    SEA_REQUIRE(P.gamma != nullptr && sea_aligned16(P.X) && sea_aligned16(P.gamma) && sea_aligned16(P.beta) && sea_aligned16(P.mod) && sea_aligned16(P.addend) && sea_aligned16(P.Xout),
                "%s[%d]: prologue pointers must be 16-byte aligned, gamma non-NULL", who, i);
    SEA_REQUIRE(P.ldx >= K && P.ldx % 4 == 0 && (!P.mod || (P.ldmod >= 2 * K && P.ldmod % 8 == 0)) && (!P.addend || (P.ldadd >= K && P.ldadd % 4 == 0)) &&
                    (!P.Xout || (P.ldxout >= K && P.ldxout % 4 == 0)), "%s[%d]: bad prologue strides", who, i);
    SEA_REQUIRE(P.Xout == nullptr || P.Xout != P.X, "%s[%d]: the prologue runs in every workgroup: Xout must not be X", who, i);
    SEA_REQUIRE(P.Y32 == nullptr && P.Yact == nullptr && P.mean == nullptr && P.rstd == nullptr, "%s[%d]: a prologue has no outputs of its own besides Xout", who, i);
    return SEA_OK;
}

#define FEW_DISPATCH(KERNEL, MRV, KCV, ...)                                                              \
    do {                                                                                                  \
        constexpr int cw_ = (KCV) <= 8 ? 4 : ((KCV) == 16 ? 2 : 1);                                       \
        constexpr int lds_ = FewShape<MRV, cw_, KCV>::LDS_BYTES;                                          \
        static const bool ok_ = few_set_lds(KERNEL<MRV, cw_, KCV>, lds_);                                 \
        SEA_REQUIRE(ok_, "few-row launch: cannot reserve %d bytes of LDS", lds_);                         \
        KERNEL<MRV, cw_, KCV><<<dim3(blocks), dim3(256), lds_, s>>>(__VA_ARGS__);                         \
    } while (0)

#define FEW_DISPATCH_KC(KERNEL, MRV, kc, ...)                                  \
    switch (kc) {                                                              \
        case 1: FEW_DISPATCH(KERNEL, MRV, 1, __VA_ARGS__); break;              \
        case 2: FEW_DISPATCH(KERNEL, MRV, 2, __VA_ARGS__); break;              \
        case 4: FEW_DISPATCH(KERNEL, MRV, 4, __VA_ARGS__); break;              \
        case 8: FEW_DISPATCH(KERNEL, MRV, 8, __VA_ARGS__); break;              \
        case 16: FEW_DISPATCH(KERNEL, MRV, 16, __VA_ARGS__); break;            \
        default: FEW_DISPATCH(KERNEL, MRV, 32, __VA_ARGS__); break;            \
    }

#define FEW_DISPATCH_KC4(KERNEL, MRV, kc, ...)                                 \
    switch (kc) {                                                              \
        case 1: FEW_DISPATCH(KERNEL, MRV, 1, __VA_ARGS__); break;              \
        case 2: FEW_DISPATCH(KERNEL, MRV, 2, __VA_ARGS__); break;              \
        default: FEW_DISPATCH(KERNEL, MRV, 4, __VA_ARGS__); break;             \
    }

}  // namespace

extern "C" int sea_gemm_fewrows(const SeaGemmGroup* groups, const SeaNormGroup* pre, int n_groups, int pre_x_is_act, int pre_gelu, float eps, int dtype, void* stream) {
    SEA_REQUIRE(groups != nullptr && n_groups >= 1 && n_groups <= FR_MAX_GROUPS, "sea_gemm_fewrows: n_groups=%d out of range (1..%d)", n_groups, FR_MAX_GROUPS);
    SEA_REQUIRE(pre_x_is_act || !pre_gelu, "sea_gemm_fewrows: pre_gelu goes with pre_x_is_act (LayerNorm + GELU of activation rows)");
    if (dtype != SEA_BF16) {
        sea_set_error("sea_gemm_fewrows: bf16 only (dtype=%d)", dtype);
        return SEA_EUNSUPPORTED;
    }
    GemmFewLaunch L;
    memset(&L, 0, sizeof(L));
    const int K = groups[0].K;
    if (!few_k_ok(K)) {
        sea_set_error("sea_gemm_fewrows: K=%d is not one of 512, 1024, 2048, 4096, 8192, 16384", K);
        return SEA_EUNSUPPORTED;
    }
    const int kc = K / 512, cw = few_cw(kc);
    int blocks = 0, mmax = 1;
    for (int i = 0; i < n_groups; ++i) {
        const SeaGemmGroup& G = groups[i];
        SEA_REQUIRE(G.W && (G.A || (pre && pre[i].X)), "sea_gemm_fewrows[%d]: null operand", i);
        SEA_REQUIRE(G.K == K, "sea_gemm_fewrows[%d]: the groups of a launch share K (%d vs %d)", i, G.K, K);
        SEA_REQUIRE(G.M >= 1 && G.M <= 4 && G.N >= 1, "sea_gemm_fewrows[%d]: M=%d N=%d (1 <= M <= 4)", i, G.M, G.N);
        SEA_REQUIRE(G.n_seg == 1 && G.act <= 1 && G.act >= 0 && G.drop.thr == 0 && G.silu_c == nullptr, "sea_gemm_fewrows[%d]: one segment, act 0 / 1, no dropout, no generated operand", i);
        SEA_REQUIRE(sea_aligned16(G.A) && sea_aligned16(G.W) && G.ldw >= K && G.ldw % 8 == 0 && (!G.A || (G.lda >= K && G.lda % 8 == 0)), "sea_gemm_fewrows[%d]: A / W alignment or strides", i);
        SEA_REQUIRE(G.C32 || G.Cact, "sea_gemm_fewrows[%d]: no output", i);
        SEA_REQUIRE((!G.R || G.ldr >= G.N) && (!G.C32 || G.ldc32 >= G.N) && (!G.Cact || G.ldcact >= G.N) && (!G.Z || G.ldz >= G.N), "sea_gemm_fewrows[%d]: output stride < N", i);
        L.g[i] = G;
        if (pre != nullptr && pre[i].X != nullptr) {
            if (pre_x_is_act) {
                const SeaNormGroup& P = pre[i];
                // every workgroup re-reads the row and its fp32 gains / shifts (10 K bytes): affordable while that stays below the weight stream (K <= 8192: 20 MB against 34)
                SEA_REQUIRE(K <= 8192, "sea_gemm_fewrows[%d]: a LayerNorm prologue on activation rows needs K <= 8192 (K=%d)", i, K);
                SEA_REQUIRE(P.gamma && sea_aligned16(P.X) && sea_aligned16(P.gamma) && sea_aligned16(P.beta) && P.ldx >= K && P.ldx % 8 == 0, "sea_gemm_fewrows[%d]: prologue pointers / strides", i);
                SEA_REQUIRE(!P.mod && !P.addend && !P.Xout && !P.Y32 && !P.Yact && !P.mean && !P.rstd, "sea_gemm_fewrows[%d]: a prologue on activation rows is a plain LayerNorm (gamma, beta)", i);
            } else {
                const int rc = check_pre(pre[i], K, "sea_gemm_fewrows", i);
                if (rc != SEA_OK) return rc;
            }
            L.pre[i] = pre[i];
        }
        L.blk_start[i] = blocks;
        blocks += (G.N + 4 * cw - 1) / (4 * cw);
        mmax = G.M > mmax ? G.M : mmax;
    }
    L.blk_start[n_groups] = blocks;
    L.n_groups = n_groups;
    L.pre_mode = pre_x_is_act ? (pre_gelu ? 3 : 1) : 0;
    L.eps = eps;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mmax == 1) {
        FEW_DISPATCH_KC(gemm_fewrows_kernel, 1, kc, L);
    } else if (mmax == 2) {
        FEW_DISPATCH_KC(gemm_fewrows_kernel, 2, kc, L);
    } else {
        FEW_DISPATCH_KC(gemm_fewrows_kernel, 4, kc, L);
    }
    SEA_CHECK_LAUNCH("sea_gemm_fewrows");
    return SEA_OK;
}

extern "C" int sea_qkv_rope_fewrows(const SeaQkvGroup* groups, const SeaNormGroup* pre, int n_groups, const SeaQkvCommon* common, float eps, int dtype, void* stream) {
    SEA_REQUIRE(groups != nullptr && common != nullptr && n_groups >= 1 && n_groups <= FR_MAX_GROUPS, "sea_qkv_rope_fewrows: n_groups=%d out of range (1..%d)", n_groups, FR_MAX_GROUPS);
    if (dtype != SEA_BF16) {
        sea_set_error("sea_qkv_rope_fewrows: bf16 only (dtype=%d)", dtype);
        return SEA_EUNSUPPORTED;
    }
    const SeaQkvCommon& c = *common;
    SEA_REQUIRE(c.rope && c.H >= 1 && c.hd >= 2 && c.hd % 2 == 0 && c.T >= 1 && c.pos0 >= 0 && c.cap >= c.pos0 + c.T, "sea_qkv_rope_fewrows: bad common (H=%d hd=%d T=%d pos0=%d cap=%d)", c.H, c.hd, c.T, c.pos0, c.cap);
    const int K = groups[0].K;
    if (!(K == 512 || K == 1024 || K == 2048)) {
        sea_set_error("sea_qkv_rope_fewrows: K=%d is not one of 512, 1024, 2048", K);
        return SEA_EUNSUPPORTED;
    }
    QkvFewLaunch L;
    memset(&L, 0, sizeof(L));
    const int kc = K / 512, cw = 4;
    const int Ea = c.H * c.hd;
    int blocks = 0, mmax = 1;
    for (int i = 0; i < n_groups; ++i) {
        const SeaQkvGroup& G = groups[i];
        SEA_REQUIRE(G.W && G.bias && (G.A || (pre && pre[i].X)), "sea_qkv_rope_fewrows[%d]: null operand", i);
        SEA_REQUIRE(G.K == K, "sea_qkv_rope_fewrows[%d]: the groups of a launch share K (%d vs %d)", i, G.K, K);
        SEA_REQUIRE(G.M >= 1 && G.M <= 4 && G.N >= 2 && G.N % 2 == 0 && G.col0 % 2 == 0 && G.col0 >= 0 && G.col0 + G.N <= 3 * Ea, "sea_qkv_rope_fewrows[%d]: M=%d N=%d col0=%d", i, G.M, G.N, G.col0);
        SEA_REQUIRE(sea_aligned16(G.A) && sea_aligned16(G.W) && G.ldw >= K && G.ldw % 8 == 0 && (!G.A || (G.lda >= K && G.lda % 8 == 0)), "sea_qkv_rope_fewrows[%d]: A / W alignment or strides", i);
        SEA_REQUIRE((G.col0 >= Ea || G.Qout) && ((G.col0 + G.N <= Ea || G.col0 >= 2 * Ea) || G.Kout) && (G.col0 + G.N <= 2 * Ea || G.Vtout), "sea_qkv_rope_fewrows[%d]: missing output for the column range", i);
        L.g[i] = G;
        if (pre != nullptr && pre[i].X != nullptr) {
            const int rc = check_pre(pre[i], K, "sea_qkv_rope_fewrows", i);
            if (rc != SEA_OK) return rc;
            SEA_REQUIRE(pre[i].Xout == nullptr, "sea_qkv_rope_fewrows[%d]: no Xout here", i);
            L.pre[i] = pre[i];
        }
        L.blk_start[i] = blocks;
        blocks += (G.N + 4 * cw - 1) / (4 * cw);
        mmax = G.M > mmax ? G.M : mmax;
    }
    L.blk_start[n_groups] = blocks;
    L.n_groups = n_groups;
    L.eps = eps;
    L.c = c;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mmax == 1) {
        FEW_DISPATCH_KC4(qkv_fewrows_kernel, 1, kc, L);
    } else if (mmax == 2) {
        FEW_DISPATCH_KC4(qkv_fewrows_kernel, 2, kc, L);
    } else {
        FEW_DISPATCH_KC4(qkv_fewrows_kernel, 4, kc, L);
    }
    SEA_CHECK_LAUNCH("sea_qkv_rope_fewrows");
    return SEA_OK;
}
