// Row-local operator chains (gfx950): sea_rowchain.  See include/sea_hip.h for the stage semantics.
//
// Why: at one trajectory (M = 2024 rows) every Linear / norm of the temporal block between two attention launches is a
// [2024 x <=256] x [<=256 x <=256] problem — 5-7 us as its own launch (launch + one tile's latency), a few hundred MFMAs
// of real work.  Everything that only mixes the columns of a row is therefore chained inside one workgroup that owns 32 rows:
//   * the row tile lives in LDS between stages ([32][<=256] act dtype per slot, 16-byte row padding => conflict-free
//     16-byte fragment reads), it is the MFMA B operand (transposed product: D^T[n][m] = W[n][:] . A[m][:]);
//   * the weights are the MFMA A operand and are read STRAIGHT from global memory / L2 into registers: wave w owns output
//     columns [w N/4, (w+1) N/4), so no two waves of a workgroup want the same weight rows and LDS staging would buy nothing —
//     there is no barrier inside a GEMM stage;
//   * in the transposed product a lane holds 4 CONSECUTIVE output columns of one row, so bias / GELU / residual / RoPE / stores are
//     vectorised, and the LayerNorm statistics of a row need one reduction over the 4 lane groups (permlane swaps) and one over the 4
//     waves (16 floats of LDS).
#include "sea_common.hpp"

namespace {

constexpr int CH_ROWS = SEA_CHAIN_ROWS;   // rows per workgroup
constexpr int CH_RT = CH_ROWS / 16;       // 16-row MFMA tiles per workgroup

template <typename T>
struct ChainCfg {
    static constexpr int EPC = ActTraits<T>::EPC;
    static constexpr int CK = ActTraits<T>::CK;
    static constexpr int STRIDE = SEA_CHAIN_MAX_WIDTH * (int)sizeof(T) + 16;   // bytes per LDS row
    static constexpr int SLOT = CH_ROWS * STRIDE;
    static constexpr int RED_OFF = SEA_CHAIN_SLOTS * SLOT;                     // 2 x [32 rows][4 waves] floats
    static constexpr int TAB_OFF = RED_OFF + 2 * CH_ROWS * 4 * 4;               // the group's stage table
    static constexpr int SINK_OFF = TAB_OFF + 24 * (int)sizeof(SeaChainStage);   // 4 x 256 bytes: target of the L2 touches
    static constexpr int LDS_BYTES = SINK_OFF + 1024;
};

__device__ __forceinline__ float group_sum4(float x) {  // sum over the 4 lane groups {l, l^16, l^32, l^48}
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <typename T>
__device__ __forceinline__ void lds_store4(char* p, const float (&v)[4]) {
    if constexpr (sizeof(T) == 2) {
        bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        *reinterpret_cast<bf16x4*>(p) = o;
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// Per-lane row bookkeeping, computed once per workgroup (integer division by T is ~20 instructions).
struct RowInfo {
    int m[CH_RT];      // global row of (rt, r), clamped to M - 1: rows past the end re-read the last row (finite, never stored)
    bool ok[CH_RT];    // row < M
    int b[CH_RT], tt[CH_RT];   // trajectory and time step of the row (qkv epilogue)
};

// Touch every 128-byte line of [rows x row_bytes] (row stride ld_bytes): brings it into this XCD's L2 before it is needed.
// A chain reads each operand exactly once, always cold (the rest of the forward has flushed the 4 MB L2 in between), and every stage
// would otherwise pay a ~1.4 us miss; issued together at workgroup start they overlap into one.
// The touch is an LDS-DMA load (global_load_lds_dword) into a 256-byte sink nobody reads: no destination register, so nothing ever
// waits for it and all the touches of a workgroup are in flight together.
__device__ __forceinline__ void l2_touch(const void* base, int rows, int row_bytes, int64_t ld_bytes, unsigned sink) {
    if (base == nullptr) return;
    const int lpr = (row_bytes + 127) >> 7;
    const int total = rows * lpr;
    const char* p = static_cast<const char*>(base);
    for (int i = threadIdx.x; i < total; i += 256) {
        const int row = i / lpr, ln = i - row * lpr;
        const void* src = p + (int64_t)row * ld_bytes + ln * 128;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(sink) : "memory");
    }
}

// One stage for a compile-time number NT of 16-column tiles per wave (N = 64 NT).
template <typename T, int NT>
__device__ __forceinline__ void chain_stage(const SeaChainStage& S, const SeaChainLaunch& L, char* lds, f32x4 (&sum)[CH_RT][2], const RowInfo& ri, uint64_t* dbgp) {
    using C = ChainCfg<T>;
    constexpr int KS_MAX = SEA_CHAIN_MAX_WIDTH / C::CK;   // contraction steps of the widest tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int N = NT * 64;
    const int nb = wave * (NT * 16);            // first column of this wave

    f32x4 acc[CH_RT][NT];
#pragma unroll
    for (int rt = 0; rt < CH_RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (S.kind == 0) {
        // D^T[n][m] = sum_k W[n][k] A[m][k]: MFMA A operand = W rows (global), B operand = activation rows (LDS).
        // ALL weight fragments of the stage are requested before the first MFMA: one memory round trip per stage instead of one per
        // contraction step (a single wave per SIMD has nothing else to hide them behind).
        const T* W = static_cast<const T*>(S.W) + (uint32_t)(nb + r) * (uint32_t)S.ldw + g * C::EPC;
        const char* A = lds + S.a_slot * C::SLOT + r * C::STRIDE + g * 16;
        const int KS = S.K / C::CK;
        if constexpr (sizeof(T) == 2) {
            uint4 wf[NT][KS_MAX];
#pragma unroll
            for (int ks = 0; ks < KS_MAX; ++ks)
                if (ks < KS) {
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) wf[ct][ks] = *reinterpret_cast<const uint4*>(W + (uint32_t)(ct * 16) * (uint32_t)S.ldw + ks * C::CK);
                }
#pragma unroll
            for (int ks = 0; ks < KS_MAX; ++ks)
                if (ks < KS) {
                    uint4 b[CH_RT];
#pragma unroll
                    for (int rt = 0; rt < CH_RT; ++rt) b[rt] = *reinterpret_cast<const uint4*>(A + rt * 16 * C::STRIDE + ks * C::CK * (int)sizeof(T));
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                        for (int rt = 0; rt < CH_RT; ++rt) mma16<T>(wf[ct][ks], b[rt], acc[rt][ct]);
                }
        } else {
            // f32: 16 contraction steps of 16 — half of them in flight at a time
            constexpr int HALF = KS_MAX / 2;
            for (int k0 = 0; k0 < KS; k0 += HALF) {
                uint4 wf[NT][HALF];
#pragma unroll
                for (int ks = 0; ks < HALF; ++ks)
                    if (k0 + ks < KS) {
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct) wf[ct][ks] = *reinterpret_cast<const uint4*>(W + (uint32_t)(ct * 16) * (uint32_t)S.ldw + (k0 + ks) * C::CK);
                    }
#pragma unroll
                for (int ks = 0; ks < HALF; ++ks)
                    if (k0 + ks < KS) {
                        uint4 b[CH_RT];
#pragma unroll
                        for (int rt = 0; rt < CH_RT; ++rt) b[rt] = *reinterpret_cast<const uint4*>(A + rt * 16 * C::STRIDE + (k0 + ks) * C::CK * (int)sizeof(T));
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                            for (int rt = 0; rt < CH_RT; ++rt) mma16<T>(wf[ct][ks], b[rt], acc[rt][ct]);
                    }
            }
        }
    } else {
        const char* X = static_cast<const char*>(S.X);
        if (S.ext & 1) X = reinterpret_cast<const char*>(L.x) + reinterpret_cast<intptr_t>(S.X);
        float xv[CH_RT][NT][4];
        if (S.x_is_act) {
#pragma unroll
            for (int rt = 0; rt < CH_RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) load4(reinterpret_cast<const T*>(X) + (uint32_t)ri.m[rt] * (uint32_t)S.ldx + nb + g * 4 + ct * 16, xv[rt][ct]);
        } else {
#pragma unroll
            for (int rt = 0; rt < CH_RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) load4(reinterpret_cast<const float*>(X) + (uint32_t)ri.m[rt] * (uint32_t)S.ldx + nb + g * 4 + ct * 16, xv[rt][ct]);
        }
#pragma unroll
        for (int rt = 0; rt < CH_RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[rt][ct] = f32x4{xv[rt][ct][0], xv[rt][ct][1], xv[rt][ct][2], xv[rt][ct][3]};
    }

    if (dbgp) dbgp[0] = wall_clock64();
    // ---- bias, activation, running sum
    if (S.bias != nullptr) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            float bv[4];
            load4(S.bias + nb + ct * 16 + g * 4, bv);
#pragma unroll
            for (int rt = 0; rt < CH_RT; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][ct][q] += S.bias_scale * bv[q];
        }
    }
    if (S.act == 1) {
#pragma unroll
        for (int rt = 0; rt < CH_RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][ct][q] = gelu_erf(acc[rt][ct][q]);
    }
    if constexpr (NT <= 2) {
        if (S.sum_op == 1 || S.sum_op == 3) {
#pragma unroll
            for (int rt = 0; rt < CH_RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) sum[rt][ct] = S.sum_op == 1 ? acc[rt][ct] : sum[rt][ct] + acc[rt][ct];
            return;
        }
        if (S.sum_op == 2) {
#pragma unroll
            for (int rt = 0; rt < CH_RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) acc[rt][ct] += sum[rt][ct];
        }
    }

    // ---- RoPE + attention layouts (sea_qkv_rope_grouped's epilogue)
    if (S.qkv) {
        const int H = L.H, hd = S.hd, Tlen = L.T, cap = L.cap;
        const int Ea = H * hd, hd2 = hd >> 1;
        const float2* rope = reinterpret_cast<const float2*>(S.rope);
        T* Qo = static_cast<T*>(S.Qout);
        T* Ko = static_cast<T*>(S.Kout);
        T* Vto = static_cast<T*>(S.Vtout);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const int nn = S.col0 + nb + ct * 16 + g * 4;
            const int part = nn >= 2 * Ea ? 2 : (nn >= Ea ? 1 : 0);
            const int hcol = nn - part * Ea;
            const int h = hcol / hd;
            const int dd = hcol - h * hd;
#pragma unroll
            for (int rt = 0; rt < CH_RT; ++rt) {
                if (!ri.ok[rt]) continue;
                const int tt = ri.tt[rt], pos = L.pos0 + tt;
                const f32x4 v = acc[rt][ct];
                const uint32_t bh = (uint32_t)(ri.b[rt] * H + h);
                if (part < 2) {
                    const float4 cs = *reinterpret_cast<const float4*>(rope + (uint32_t)pos * (uint32_t)hd2 + (dd >> 1));  // two (cos, sin) pairs
                    const float o[4] = {v[0] * cs.x - v[1] * cs.y, v[0] * cs.y + v[1] * cs.x, v[2] * cs.z - v[3] * cs.w, v[2] * cs.w + v[3] * cs.z};
                    if (part == 0) {
                        const float sc = S.q_scale;
                        store4(Qo + ((int64_t)(bh * (uint32_t)Tlen + tt) * hd + dd), o[0] * sc, o[1] * sc, o[2] * sc, o[3] * sc);
                    } else {
                        store4(Ko + ((int64_t)(bh * (uint32_t)cap + pos) * hd + dd), o[0], o[1], o[2], o[3]);
                    }
                } else {
                    T* dst = Vto + ((int64_t)(bh * (uint32_t)hd + dd) * cap + pos);
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst[(int64_t)q * cap] = from_f32<T>(v[q]);
                }
            }
        }
        return;
    }

    // ---- residual, raw LDS copy, info-bottleneck term, plain stores
    if (S.R != nullptr || (S.ext & 2)) {
        const float* R = (S.ext & 2) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(L.x) + reinterpret_cast<intptr_t>(S.R)) : S.R;
        float rv[CH_RT][NT][4];
#pragma unroll
        for (int rt = 0; rt < CH_RT; ++rt) {
            const uint32_t off = (uint32_t)ri.m[rt] * (uint32_t)S.ldr + nb + g * 4;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) load4(R + off + ct * 16, rv[rt][ct]);
        }
#pragma unroll
        for (int rt = 0; rt < CH_RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][ct][q] += rv[rt][ct][q];
    }
    if (S.raw_slot >= 0) {
        char* dst = lds + S.raw_slot * C::SLOT + r * C::STRIDE + (nb + g * 4) * (int)sizeof(T);
#pragma unroll
        for (int rt = 0; rt < CH_RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const float v[4] = {acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]};
                lds_store4<T>(dst + rt * 16 * C::STRIDE + ct * 16 * (int)sizeof(T), v);
            }
    }
    if (S.ib_w1 != nullptr) {
        // info-bottleneck term W2 . gelu(LN_8(w1 c + b1)) + b2, hidden width exactly 8: the [rows, 8] x [8, N] product is one more
        // contraction step of the same transposed MFMA (A operand = rows of W2, B operand = the row's hidden vector, zero-padded), added
        // straight into the accumulators
        float w1[8], b1[8], lw[8], lb[8];
        load4(S.ib_w1, *reinterpret_cast<float(*)[4]>(&w1[0]));
        load4(S.ib_w1 + 4, *reinterpret_cast<float(*)[4]>(&w1[4]));
        load4(S.ib_b1, *reinterpret_cast<float(*)[4]>(&b1[0]));
        load4(S.ib_b1 + 4, *reinterpret_cast<float(*)[4]>(&b1[4]));
        load4(S.ib_lnw, *reinterpret_cast<float(*)[4]>(&lw[0]));
        load4(S.ib_lnw + 4, *reinterpret_cast<float(*)[4]>(&lw[4]));
        load4(S.ib_lnb, *reinterpret_cast<float(*)[4]>(&lb[0]));
        load4(S.ib_lnb + 4, *reinterpret_cast<float(*)[4]>(&lb[4]));
        uint4 wfrag[NT];
        float o[NT][4];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            load4(S.ib_b2 + nb + ct * 16 + g * 4, o[ct]);
            const float* wrow = S.ib_w2 + (uint32_t)(nb + ct * 16 + r) * 8u;   // MFMA A-operand row = lane & 15
            wfrag[ct] = make_uint4(0, 0, 0, 0);
            if constexpr (sizeof(T) == 2) {
                if (g == 0) {
                    float w[8];
                    load4(wrow, *reinterpret_cast<float(*)[4]>(&w[0]));
                    load4(wrow + 4, *reinterpret_cast<float(*)[4]>(&w[4]));
                    bf16x8 pv = {(__bf16)w[0], (__bf16)w[1], (__bf16)w[2], (__bf16)w[3], (__bf16)w[4], (__bf16)w[5], (__bf16)w[6], (__bf16)w[7]};
                    wfrag[ct] = __builtin_bit_cast(uint4, pv);
                }
            } else {
                if (g < 2) wfrag[ct] = *reinterpret_cast<const uint4*>(wrow + 4 * g);
            }
        }
#pragma unroll
        for (int rt = 0; rt < CH_RT; ++rt) {
            const float cv = L.cond[ri.m[rt]];
            float hid[8], mean = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                hid[k] = w1[k] * cv + b1[k];
                mean += hid[k];
            }
            mean *= 0.125f;
            float var = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                hid[k] -= mean;
                var += hid[k] * hid[k];
            }
            const float rstd = 1.0f / sqrtf(var * 0.125f + 1e-5f);
#pragma unroll
            for (int k = 0; k < 8; ++k) hid[k] = gelu_erf(hid[k] * rstd * lw[k] + lb[k]);
            uint4 hfrag = make_uint4(0, 0, 0, 0);
            if constexpr (sizeof(T) == 2) {
                if (g == 0) {
                    bf16x8 pv = {(__bf16)hid[0], (__bf16)hid[1], (__bf16)hid[2], (__bf16)hid[3], (__bf16)hid[4], (__bf16)hid[5], (__bf16)hid[6], (__bf16)hid[7]};
                    hfrag = __builtin_bit_cast(uint4, pv);
                }
            } else {
                if (g == 0) hfrag = __builtin_bit_cast(uint4, f32x4{hid[0], hid[1], hid[2], hid[3]});
                if (g == 1) hfrag = __builtin_bit_cast(uint4, f32x4{hid[4], hid[5], hid[6], hid[7]});
            }
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][ct][q] += o[ct][q];
                mma16<T>(wfrag[ct], hfrag, acc[rt][ct]);
            }
        }
    }
    if (S.C32 != nullptr || S.Cact != nullptr) {
#pragma unroll
        for (int rt = 0; rt < CH_RT; ++rt) {
            if (!ri.ok[rt]) continue;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const int n = nb + ct * 16 + g * 4;
                const f32x4 v = acc[rt][ct];
                if (S.C32 != nullptr) store4(S.C32 + (uint32_t)ri.m[rt] * (uint32_t)S.ldc32 + n, v[0], v[1], v[2], v[3]);
                if (S.Cact != nullptr) store4(static_cast<T*>(S.Cact) + (uint32_t)ri.m[rt] * (uint32_t)S.ldcact + n, v[0], v[1], v[2], v[3]);
            }
        }
    }

    if (dbgp) dbgp[1] = wall_clock64();
    // ---- LayerNorm / AdaLN over the N columns of a row (two-pass statistics on the register-resident row)
    if (S.norm) {
        float* red = reinterpret_cast<float*>(lds + C::RED_OFF);
        float* red2 = red + CH_ROWS * 4;
        const float inv_n = 1.0f / (float)N;
        float mean[CH_RT], rstd[CH_RT];
        // modulation / gain / shift: requested before the statistics so that their latency hides behind the two reductions
        float gm[NT][4], bt[NT][4], mw[CH_RT][NT][4], mb[CH_RT][NT][4];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const int n = nb + ct * 16 + g * 4;
            load4(S.gamma + n, gm[ct]);
#pragma unroll
            for (int q = 0; q < 4; ++q) bt[ct][q] = 0.f;
            if (S.beta != nullptr) load4(S.beta + n, bt[ct]);
        }
        if (S.mod != nullptr) {
#pragma unroll
            for (int rt = 0; rt < CH_RT; ++rt) {
                const T* mod = static_cast<const T*>(S.mod) + (uint32_t)ri.m[rt] * (uint32_t)S.ldmod + nb + g * 4;
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
                    load4(mod + ct * 16, mw[rt][ct]);
                    load4(mod + N + ct * 16, mb[rt][ct]);
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < CH_RT; ++rt) {
            float s = 0.f;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) s += (acc[rt][ct][0] + acc[rt][ct][1]) + (acc[rt][ct][2] + acc[rt][ct][3]);
            s = group_sum4(s);
            if (g == 0) red[(rt * 16 + r) * 4 + wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < CH_RT; ++rt) {
            const float4 p = *reinterpret_cast<const float4*>(red + (rt * 16 + r) * 4);
            mean[rt] = ((p.x + p.y) + (p.z + p.w)) * inv_n;
            float s = 0.f;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float c = acc[rt][ct][q] - mean[rt];
                    s += c * c;
                }
            s = group_sum4(s);
            if (g == 0) red2[(rt * 16 + r) * 4 + wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < CH_RT; ++rt) {
            const float4 p = *reinterpret_cast<const float4*>(red2 + (rt * 16 + r) * 4);
            rstd[rt] = 1.0f / sqrtf(((p.x + p.y) + (p.z + p.w)) * inv_n + L.eps);
        }
        float* N32 = (S.ext & 4) ? reinterpret_cast<float*>(reinterpret_cast<char*>(L.out) + reinterpret_cast<intptr_t>(S.N32)) : S.N32;
        const bool has_n32 = S.N32 != nullptr || (S.ext & 4);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const int n = nb + ct * 16 + g * 4;
#pragma unroll
            for (int rt = 0; rt < CH_RT; ++rt) {
                float y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gq = S.mod != nullptr ? gm[ct][q] + 1.0f + mw[rt][ct][q] : gm[ct][q];
                    const float bq = S.mod != nullptr ? bt[ct][q] + mb[rt][ct][q] : bt[ct][q];
                    y[q] = (acc[rt][ct][q] - mean[rt]) * rstd[rt] * gq + bq;
                    if (S.norm == 2) y[q] = gelu_erf(y[q]);
                }
                if (S.norm_slot >= 0) lds_store4<T>(lds + S.norm_slot * C::SLOT + (rt * 16 + r) * C::STRIDE + n * (int)sizeof(T), y);
                if (ri.ok[rt]) {
                    if (S.Nact != nullptr) store4(static_cast<T*>(S.Nact) + (uint32_t)ri.m[rt] * (uint32_t)S.ldnact + n, y[0], y[1], y[2], y[3]);
                    if (has_n32) store4(N32 + (uint32_t)ri.m[rt] * (uint32_t)S.ldn32 + n, y[0], y[1], y[2], y[3]);
                }
            }
        }
    }
}

constexpr int STAGE_WORDS = (int)(sizeof(SeaChainStage) / 4);
// A stage descriptor from the LDS copy of the table, every word made wave-uniform (SGPR) so that the stage's branches stay scalar.
__device__ __forceinline__ void load_stage(SeaChainStage& S, const uint32_t* src) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(&S);
#pragma unroll
    for (int w = 0; w < STAGE_WORDS; ++w) dst[w] = __builtin_amdgcn_readfirstlane(src[w]);
}

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void rowchain_kernel(const SeaChainLaunch L) {
    using C = ChainCfg<T>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int grp = blockIdx.y;
    const int row0 = blockIdx.x * CH_ROWS;
    const int lane = threadIdx.x & 63, r = lane & 15;
    const int s0 = L.first[grp], s1 = L.first[grp + 1];
    const int rows_here = L.M - row0 < CH_ROWS ? L.M - row0 : CH_ROWS;

    if (L.dbg != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) L.dbg[0] = wall_clock64();
    // ---- the group's stage table -> LDS, all threads at once (one memory round trip instead of one per stage)
    uint32_t* tab = reinterpret_cast<uint32_t*>(lds + C::TAB_OFF);
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(L.stages + s0);
        const int nw = (s1 - s0) * STAGE_WORDS;
        for (int i = threadIdx.x; i < nw; i += 256) tab[i] = src[i];
    }
    __syncthreads();
    // ---- warm this XCD's L2 with everything the chain will read: weights of every stage, this workgroup's residual / modulation rows
    const unsigned sink = (unsigned)__builtin_amdgcn_readfirstlane(C::SINK_OFF + (int)(threadIdx.x >> 6) * 256);
    for (int si = s0; si < s1; ++si) {
        SeaChainStage S;
        load_stage(S, tab + (si - s0) * STAGE_WORDS);
        const int esz = (int)sizeof(T);
        if (S.kind == 0) l2_touch(S.W, S.N, S.K * esz, S.ldw * esz, sink);
        if (S.R != nullptr && !(S.ext & 2)) l2_touch(S.R + (int64_t)row0 * S.ldr, rows_here, S.N * 4, S.ldr * 4, sink);
        if (S.mod != nullptr) l2_touch(static_cast<const T*>(S.mod) + (int64_t)row0 * S.ldmod, rows_here, 2 * S.N * esz, S.ldmod * esz, sink);
    }

    const bool dbg = L.dbg != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    if (dbg) L.dbg[1] = wall_clock64();
    RowInfo ri;
#pragma unroll
    for (int rt = 0; rt < CH_RT; ++rt) {
        const int m = row0 + rt * 16 + r;
        ri.ok[rt] = m < L.M;
        ri.m[rt] = m < L.M ? m : L.M - 1;
        ri.b[rt] = ri.m[rt] / L.T;
        ri.tt[rt] = ri.m[rt] - ri.b[rt] * L.T;
    }
    f32x4 sum[CH_RT][2];
#pragma unroll
    for (int rt = 0; rt < CH_RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) sum[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int si = s0; si < s1; ++si) {
        SeaChainStage S;
        load_stage(S, tab + (si - s0) * STAGE_WORDS);
        if (S.kind == 2) {
            // COPY: [rows, N] act-dtype rows straight into an LDS slot, 16 bytes per thread per step; consecutive COPY stages are issued
            // back to back (no barrier in between) so that their loads overlap
            const int cpr = S.N * (int)sizeof(T) / 16;  // chunks per row
            const char* X = static_cast<const char*>(S.X);
            for (int i = threadIdx.x; i < CH_ROWS * cpr; i += 256) {
                const int rr = i / cpr, cc = i - rr * cpr;
                int m = row0 + rr;
                m = m < L.M ? m : L.M - 1;
                const uint4 v = *reinterpret_cast<const uint4*>(X + ((int64_t)m * S.ldx) * (int64_t)sizeof(T) + cc * 16);
                *reinterpret_cast<uint4*>(lds + S.raw_slot * C::SLOT + rr * C::STRIDE + cc * 16) = v;
            }
            bool next_is_copy = false;
            if (si + 1 < s1) {
                next_is_copy = tab[(si + 1 - s0) * STAGE_WORDS] == 2u;  // .kind is the first field
            }
            if (!next_is_copy) __syncthreads();
            if (dbg) L.dbg[2 + si - s0] = wall_clock64();
            continue;
        }
        switch (S.N >> 6) {
            case 1: chain_stage<T, 1>(S, L, lds, sum, ri, dbg ? L.dbg + 32 + 2 * (si - s0) : nullptr); break;
            case 2: chain_stage<T, 2>(S, L, lds, sum, ri, dbg ? L.dbg + 32 + 2 * (si - s0) : nullptr); break;
            case 3: chain_stage<T, 3>(S, L, lds, sum, ri, dbg ? L.dbg + 32 + 2 * (si - s0) : nullptr); break;
            default: chain_stage<T, 4>(S, L, lds, sum, ri, dbg ? L.dbg + 32 + 2 * (si - s0) : nullptr); break;
        }
        // LDS written by this stage is read by the next one; its readers are done before anything overwrites it because every wave
        // passes this barrier only after its own reads of the stage
        if (S.raw_slot >= 0 || (S.norm && S.norm_slot >= 0)) __syncthreads();
        if (dbg) L.dbg[2 + si - s0] = wall_clock64();
    }
}

}  // namespace

extern "C" int sea_rowchain(const SeaChainLaunch* launch, const SeaChainStage* hs, int dtype, void* stream) {
    SEA_REQUIRE(launch != nullptr && hs != nullptr, "sea_rowchain: null argument");
    const SeaChainLaunch& L = *launch;
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_rowchain: bad dtype %d", dtype);
    SEA_REQUIRE(L.stages != nullptr && L.n_groups >= 1 && L.n_groups <= SEA_CHAIN_MAX_GROUPS, "sea_rowchain: n_groups=%d out of range", L.n_groups);
    SEA_REQUIRE(L.M >= 1 && L.T >= 1 && L.H >= 1 && L.pos0 >= 0 && L.cap >= 0, "sea_rowchain: bad sizes M=%d T=%d H=%d pos0=%d cap=%d", L.M, L.T, L.H, L.pos0, L.cap);
    SEA_REQUIRE(L.first[0] == 0, "sea_rowchain: first[0] must be 0");
    const int ck = dtype == SEA_BF16 ? 32 : 16;
    for (int gi = 0; gi < L.n_groups; ++gi) {
        SEA_REQUIRE(L.first[gi + 1] > L.first[gi] && L.first[gi + 1] - L.first[gi] <= 24, "sea_rowchain: group %d has a bad stage range (1..24 stages)", gi);
        uint32_t written = 0;  // LDS slots holding a tile, with its width
        int width[SEA_CHAIN_SLOTS] = {0};
        int sum_n = 0;
        for (int si = L.first[gi]; si < L.first[gi + 1]; ++si) {
            const SeaChainStage& S = hs[si];
            SEA_REQUIRE(S.N >= 64 && S.N <= SEA_CHAIN_MAX_WIDTH && S.N % 64 == 0, "sea_rowchain[%d]: N=%d must be a multiple of 64 up to %d", si, S.N, SEA_CHAIN_MAX_WIDTH);
            if (S.kind == 0) {
                SEA_REQUIRE(S.W != nullptr && sea_aligned16(S.W) && S.K >= ck && S.K <= SEA_CHAIN_MAX_WIDTH && S.K % (2 * ck) == 0 && S.ldw >= S.K && S.ldw % 8 == 0,
                            "sea_rowchain[%d]: bad GEMM operand (K=%d ldw=%lld)", si, S.K, (long long)S.ldw);
                SEA_REQUIRE(S.a_slot >= 0 && S.a_slot < SEA_CHAIN_SLOTS && (written >> S.a_slot & 1) && width[S.a_slot] == S.K,
                            "sea_rowchain[%d]: A slot %d does not hold a [rows, %d] tile", si, S.a_slot, S.K);
                SEA_REQUIRE(S.raw_slot != S.a_slot && (S.norm == 0 || S.norm_slot != S.a_slot), "sea_rowchain[%d]: stage writes the slot it reads", si);
            } else if (S.kind == 2) {
                SEA_REQUIRE(S.X != nullptr && sea_aligned16(S.X) && S.ldx >= S.N && S.ldx % 8 == 0 && S.raw_slot >= 0 && S.raw_slot < SEA_CHAIN_SLOTS && !S.norm && !S.qkv &&
                                !S.sum_op && !(S.ext & 1),
                            "sea_rowchain[%d]: bad COPY stage", si);
            } else {
                SEA_REQUIRE(S.kind == 1 && (S.X != nullptr || (S.ext & 1)) && S.ldx >= S.N && S.ldx % 4 == 0, "sea_rowchain[%d]: bad LOAD stage", si);
                SEA_REQUIRE((S.ext & 1) ? (L.x != nullptr && reinterpret_cast<intptr_t>(S.X) % 16 == 0) : sea_aligned16(S.X), "sea_rowchain[%d]: X misaligned", si);
            }
            SEA_REQUIRE(S.act == 0 || S.act == 1, "sea_rowchain[%d]: bad act", si);
            {
                const int64_t lim = (int64_t)1 << 31, m = L.M;
                SEA_REQUIRE(m * S.ldx < lim && m * S.ldr < lim && m * S.ldc32 < lim && m * S.ldcact < lim && m * S.ldmod < lim && m * S.ldnact < lim && m * S.ldn32 < lim &&
                                (int64_t)S.N * S.ldw < lim,
                            "sea_rowchain[%d]: operand too large for 32-bit element offsets", si);
            }
            SEA_REQUIRE(S.sum_op >= 0 && S.sum_op <= 3, "sea_rowchain[%d]: bad sum_op", si);
            SEA_REQUIRE(S.sum_op == 0 || S.N <= 128, "sea_rowchain[%d]: the running sum holds at most 128 columns", si);
            if (S.sum_op == 1) sum_n = S.N;
            if (S.sum_op == 2 || S.sum_op == 3) SEA_REQUIRE(sum_n == S.N, "sea_rowchain[%d]: running sum has width %d, stage has %d", si, sum_n, S.N);
            SEA_REQUIRE(sea_aligned16(S.bias) && sea_aligned16(S.gamma) && sea_aligned16(S.beta) && sea_aligned16(S.mod) && sea_aligned16(S.C32) && sea_aligned16(S.Cact) &&
                            sea_aligned16(S.Nact) && sea_aligned16(S.ib_b2),
                        "sea_rowchain[%d]: misaligned pointer", si);
            SEA_REQUIRE(!(S.ext & 2) || (L.x != nullptr && reinterpret_cast<intptr_t>(S.R) % 16 == 0), "sea_rowchain[%d]: bad ext residual", si);
            SEA_REQUIRE((S.ext & 2) || sea_aligned16(S.R), "sea_rowchain[%d]: R misaligned", si);
            SEA_REQUIRE((!(S.R || (S.ext & 2)) || (S.ldr >= S.N && S.ldr % 4 == 0)) && (!S.C32 || (S.ldc32 >= S.N && S.ldc32 % 4 == 0)) && (!S.Cact || (S.ldcact >= S.N && S.ldcact % 4 == 0)),
                        "sea_rowchain[%d]: bad output stride", si);
            SEA_REQUIRE(S.raw_slot >= -1 && S.raw_slot < SEA_CHAIN_SLOTS, "sea_rowchain[%d]: bad raw_slot", si);
            if (S.ib_w1 != nullptr)
                SEA_REQUIRE(S.ib_b1 && S.ib_lnw && S.ib_lnb && S.ib_w2 && S.ib_b2 && S.ib_h == 8 && L.cond != nullptr && sea_aligned16(S.ib_w1) && sea_aligned16(S.ib_b1) &&
                                sea_aligned16(S.ib_lnw) && sea_aligned16(S.ib_lnb) && sea_aligned16(S.ib_w2),
                            "sea_rowchain[%d]: bad info-bottleneck parameters (hidden width must be 8, got %d; 16-byte aligned)", si, S.ib_h);
            if (S.qkv) {
                SEA_REQUIRE(S.hd >= 4 && S.hd % 4 == 0 && S.rope != nullptr && S.col0 >= 0 && S.col0 % 4 == 0 && S.col0 + S.N <= 3 * L.H * S.hd && L.cap >= L.pos0 + L.T && L.M % L.T == 0,
                            "sea_rowchain[%d]: bad qkv epilogue (hd=%d col0=%d N=%d H=%d cap=%d)", si, S.hd, S.col0, S.N, L.H, L.cap);
                const int Ea = L.H * S.hd;
                SEA_REQUIRE((S.col0 >= Ea || S.Qout) && ((S.col0 + S.N <= Ea || S.col0 >= 2 * Ea) || S.Kout) && (S.col0 + S.N <= 2 * Ea || S.Vtout), "sea_rowchain[%d]: missing q/k/v output", si);
                SEA_REQUIRE(sea_aligned16(S.Qout) && sea_aligned16(S.Kout) && sea_aligned16(S.Vtout), "sea_rowchain[%d]: q/k/v outputs misaligned", si);
            }
            if (S.norm) {
                SEA_REQUIRE((S.norm == 1 || S.norm == 2) && S.gamma != nullptr && !S.qkv && S.sum_op != 1 && S.sum_op != 3, "sea_rowchain[%d]: bad norm stage", si);
                SEA_REQUIRE(S.norm_slot >= -1 && S.norm_slot < SEA_CHAIN_SLOTS && (S.norm_slot < 0 || S.norm_slot != S.raw_slot), "sea_rowchain[%d]: bad norm_slot", si);
                SEA_REQUIRE((!S.mod || (S.ldmod >= 2 * S.N && S.ldmod % 4 == 0)) && (!S.Nact || (S.ldnact >= S.N && S.ldnact % 4 == 0)) && (!(S.N32 || (S.ext & 4)) || (S.ldn32 >= S.N && S.ldn32 % 4 == 0)),
                            "sea_rowchain[%d]: bad norm strides", si);
                SEA_REQUIRE(!(S.ext & 4) || (L.out != nullptr && reinterpret_cast<intptr_t>(S.N32) % 16 == 0), "sea_rowchain[%d]: bad ext output", si);
            }
            const bool ends = S.sum_op == 1 || S.sum_op == 3 || S.qkv;
            if (!ends && S.raw_slot >= 0) { written |= 1u << S.raw_slot; width[S.raw_slot] = S.N; }
            if (!ends && S.norm && S.norm_slot >= 0) { written |= 1u << S.norm_slot; width[S.norm_slot] = S.N; }
        }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((L.M + CH_ROWS - 1) / CH_ROWS, L.n_groups), block(256);
    if (dtype == SEA_BF16) {
        constexpr int lds_b = ChainCfg<__bf16>::LDS_BYTES;
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(rowchain_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
        (void)once;
        rowchain_kernel<__bf16><<<grid, block, lds_b, s>>>(L);
    } else {
        constexpr int lds_f = ChainCfg<float>::LDS_BYTES;
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(rowchain_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_f);
        (void)once;
        rowchain_kernel<float><<<grid, block, lds_f, s>>>(L);
    }
    SEA_CHECK_LAUNCH("sea_rowchain");
    return SEA_OK;
}
