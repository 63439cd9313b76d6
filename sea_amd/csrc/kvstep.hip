// kvstep.hip — the exact KV-cache rollout step as SEVEN launches per layer (sea_kv_rollout), for models small enough that a workgroup can carry a whole
// row through a Linear layer (E <= 512): the step of the generic plan is 22 launches of ~5 us each on B x F rows, every one a dispatch plus two or
// three dependent memory round trips.  What this file changes (all exact, same arithmetic order per output element up to fp32 reassociation):
//   * the AdaLN modulations and the info-bottleneck term depend on the condition only: computed for ALL steps in one batched pass before the loop
//     (the caller hands in mod[n_steps * B, 2d] / ib[n_steps * B, E]); the step reads its rows;
//   * activations between launches stay fp32 vectors (B x F x E floats), every Linear is a GEMV on the VALU: a wave team per output row, 16-byte
//     weight loads, the input vector in LDS;
//   * q/k/v projections ride in the attention launch (a workgroup per (field or pair, trajectory, head) projects ITS head's rows: 3 hd x E weights),
//     the AdaLN in front of them is recomputed by every workgroup (E floats);
//   * cross-attention over the CACHED keys of all F (F - 1) pairs is one launch; for the pairs whose source field is updated earlier in the same
//     Gauss-Seidel sweep (j < i) the current position's key/value is merged in the tail (flash-decoding merge of (m, l, o) with one more key);
//   * the Gauss-Seidel tails of all fields are ONE launch: workgroup (i, b) waits for the normalised down-projection of fields j < i through
//     data-tagged 8-byte granules (one agent-scope store each, polled by the consumer; every spin is bounded and reports through an error word).
// Launches per layer and step: self (LN + QKV + RoPE + append + attention), out-proj + residual + down + ln_cross, cross attention, tails, fc1,
// fc2, proj (+ final norm).
// Reference: the loop of utils/train_utils.py:202-209 around models/temporal.py:120-200, 398-417 (one row per call instead of the whole prefix).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sea_hip.h"
#include "sea_common.hpp"

namespace {

constexpr float KV_EPS = 1e-5f;
constexpr int KV_SPIN_LIMIT = 1 << 18;   // polls of one granule before a consumer gives up (each poll is a memory round trip: >= 0.1 s)

template <typename T>
__device__ __forceinline__ void unpack_w(const uint4& r, float (&o)[ActTraits<T>::EPC]);
template <>
__device__ __forceinline__ void unpack_w<float>(const uint4& r, float (&o)[4]) {
    o[0] = __builtin_bit_cast(float, r.x); o[1] = __builtin_bit_cast(float, r.y); o[2] = __builtin_bit_cast(float, r.z); o[3] = __builtin_bit_cast(float, r.w);
}
template <>
__device__ __forceinline__ void unpack_w<__bf16>(const uint4& r, float (&o)[8]) {
    o[0] = __builtin_bit_cast(float, r.x << 16); o[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
    o[2] = __builtin_bit_cast(float, r.y << 16); o[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
    o[4] = __builtin_bit_cast(float, r.z << 16); o[5] = __builtin_bit_cast(float, r.z & 0xffff0000u);
    o[6] = __builtin_bit_cast(float, r.w << 16); o[7] = __builtin_bit_cast(float, r.w & 0xffff0000u);
}

__device__ __forceinline__ float round_to(float v, float) { return v; }
__device__ __forceinline__ float round_to(float v, __bf16) { return (float)(__bf16)v; }

// ------------------------------------------------------------------------------------------------ reductions without the LDS crossbar
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// a[u] <- sum of a[u] over the TL consecutive lanes of this lane's team (TL a power of two <= 64); every lane of the team gets the sum.
// quad_perm swaps, then half-row / row mirrors (the groups below are already uniform), then gfx950's row / half swaps.
template <int N>
__device__ __forceinline__ void team_reduce(float (&a)[N], int TL) {
    if (TL >= 2) {
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] += dpp_f<0xB1>(a[u]);
    }
    if (TL >= 4) {
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] += dpp_f<0x4E>(a[u]);
    }
    if (TL >= 8) {
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] += dpp_f<0x141>(a[u]);
    }
    if (TL >= 16) {
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] += dpp_f<0x140>(a[u]);
    }
    if (TL >= 32) {
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const unsigned x = __float_as_uint(a[u]);
            const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
            a[u] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
    }
    if (TL >= 64) {
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const unsigned x = __float_as_uint(a[u]);
            const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
            a[u] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
    }
}
__device__ __forceinline__ float wave_sum_fast(float v) {
    float a[1] = {v};
    team_reduce<1>(a, 64);
    return a[0];
}
__device__ __forceinline__ float wave_max_fast(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return group_max4(v);
}

// ------------------------------------------------------------------------------------------------ workgroup reductions
// red: LDS, >= 32 floats.  Every thread of the workgroup calls; contains two barriers.
__device__ __forceinline__ float wg_sum(float v, float* red, int tid, int nthreads) {
    v = wave_sum_fast(v);
    __syncthreads();                     // red may still be read from a previous call
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < (nthreads >> 6); ++w) s += red[w];
    return s;
}
__device__ __forceinline__ float wg_max(float v, float* red, int tid, int nthreads) {
    v = wave_max_fast(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = red[0];
    for (int w = 1; w < (nthreads >> 6); ++w) s = fmaxf(s, red[w]);
    return s;
}

// ------------------------------------------------------------------------------------------------ row norm (models/base_blocks.py:320-352)
// ys = (xs - mean) * rstd * gq + bq over d elements (both in LDS, may alias), two-pass statistics like rownorm_kernel; gq = gamma + 1 + mod[0:d],
// bq = beta + mod[d:2d] with a modulation row, else gamma / beta (beta may be null).  Optional GELU (the MLP's LayerNorm + GELU).
// NormRegs: this thread's gain / shift of up to NPT elements (i = tid + k nthreads), requested at kernel entry so that they are not a round trip
// of their own behind the statistics.
template <int NPT>
struct NormRegs {
    float gq[NPT], bq[NPT];
};
template <typename T, int NPT>
__device__ __forceinline__ void norm_issue(NormRegs<NPT>& R, int d, const SeaKvNorm& nm, int64_t modrow, int tid, int nthreads) {
    const T* mod = nm.mod != nullptr ? static_cast<const T*>(nm.mod) + modrow * nm.ldmod : nullptr;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int i = tid + k * nthreads;
        R.gq[k] = R.bq[k] = 0.f;
        if (i < d) {
            R.gq[k] = nm.gamma[i];
            if (nm.beta != nullptr) R.bq[k] = nm.beta[i];
            if (mod != nullptr) {
                R.gq[k] += 1.0f + to_f32(mod[i]);
                R.bq[k] += to_f32(mod[d + i]);
            }
        }
    }
}
template <int NPT, typename T>
__device__ __forceinline__ void wg_norm_r(const float* xs, float* ys, T* yT, int d, const NormRegs<NPT>& R, bool gelu, float* red, int tid, int nthreads) {
    float s = 0.f;
    for (int i = tid; i < d; i += nthreads) s += xs[i];
    const float mean = wg_sum(s, red, tid, nthreads) / (float)d;
    float q = 0.f;
    for (int i = tid; i < d; i += nthreads) {
        const float c = xs[i] - mean;
        q = fma1(c, c, q);
    }
    const float rstd = 1.0f / sqrtf(wg_sum(q, red, tid, nthreads) / (float)d + KV_EPS);
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int i = tid + k * nthreads;
        if (i < d) {
            float o = (xs[i] - mean) * rstd * R.gq[k] + R.bq[k];
            if (gelu) o = gelu_erf(o);
            ys[i] = o;
            if (yT != nullptr) yT[i] = from_f32<T>(o);   // the matrix-core operand copy
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ GEMV
// ys[r] = sum_k W[rowmap(r), k] * xs[k], r < nrows; xs, ys in LDS (fp32).  A team of TL lanes owns a row: lane tl of the team holds the 16-byte
// chunks tl, tl + TL, ... of it (CPL per lane; CPL == 1: K / EPC <= 64 chunks, several teams per wave).  The workgroup covers rpp rows per
// "block"; gemv_issue requests the weights of NB blocks into registers (no dependence on the input vector: issued at kernel entry, or one
// stage ahead), gemv_apply consumes them.  The caller puts a barrier between gemv_apply and the first read of ys.
struct Geo {
    int TL, tpw, team, tl, rpp, wave;
};
template <int CPL>
__device__ __forceinline__ Geo team_geo(int kc, int tid, int nthreads) {
    Geo g;
    const int lane = tid & 63;
    g.wave = tid >> 6;
    g.TL = CPL == 1 ? kc : 64;                   // power of two <= 64
    const int l2 = 31 - __builtin_clz((unsigned)g.TL);
    g.tpw = 64 >> l2;
    g.team = lane >> l2;
    g.tl = lane & (g.TL - 1);
    g.rpp = (nthreads >> 6) * g.tpw;
    return g;
}
template <int CPL, int NB>
struct WRegs {
    uint4 w[NB][CPL];
};
template <typename T, int CPL, int NB, typename RowMap>
__device__ __forceinline__ void gemv_issue(WRegs<CPL, NB>& R, const T* __restrict__ W, int ldw, int K, int nrows, int blk0, RowMap rowmap, int tid, int nthreads) {
    constexpr int EPC = ActTraits<T>::EPC;
    const Geo g = team_geo<CPL>(K / EPC, tid, nthreads);
#pragma unroll
    for (int u = 0; u < NB; ++u) {   // no branch per block (blocks beyond the rows re-read the last row: a cache hit; a branch here costs the allocator every register)
        int row = (blk0 + u) * g.rpp + g.wave * g.tpw + g.team;
        row = row < nrows ? row : nrows - 1;
        const T* wr = W + (int64_t)rowmap(row) * ldw;
#pragma unroll
        for (int t = 0; t < CPL; ++t) R.w[u][t] = *reinterpret_cast<const uint4*>(wr + (g.tl + t * g.TL) * EPC);
    }
}
template <typename T, int CPL, int NB>
__device__ __forceinline__ void gemv_apply(const WRegs<CPL, NB>& R, int K, int nrows, int blk0, const float* xs, float* ys, int tid, int nthreads) {
    constexpr int EPC = ActTraits<T>::EPC;
    const Geo g = team_geo<CPL>(K / EPC, tid, nthreads);
    float xv[CPL][EPC];
#pragma unroll
    for (int t = 0; t < CPL; ++t)
#pragma unroll
        for (int e = 0; e < EPC; ++e) xv[t][e] = xs[(g.tl + t * g.TL) * EPC + e];
    float acc[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            float wv[EPC];
            unpack_w<T>(R.w[u][t], wv);
#pragma unroll
            for (int e = 0; e < EPC; e += 2) {   // (fma1: plain fmaf pairs become v_pk_fma_f32 op_sel:[0,1,0], the form the build's lint refuses)
                a0 = fma1(wv[e], xv[t][e], a0);
                a1 = fma1(wv[e + 1], xv[t][e + 1], a1);
            }
        }
        acc[u] = a0 + a1;
    }
    team_reduce<NB>(acc, g.TL);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int row = (blk0 + u) * g.rpp + g.wave * g.tpw + g.team;
        if (g.tl == 0 && row < nrows) ys[row] = acc[u];
    }
}
// blocks blk0, blk0 + 1, ... until the rows are covered: 8 loads per lane in flight
template <typename T, int CPL, typename RowMap>
__device__ __forceinline__ void gemv_stream(const T* __restrict__ W, int ldw, int K, int nrows, int blk0, const float* xs, float* ys, RowMap rowmap, int tid, int nthreads) {
    constexpr int NB = CPL >= 8 ? 1 : 8 / CPL;
    const Geo g = team_geo<CPL>(K / ActTraits<T>::EPC, tid, nthreads);
    for (int b = blk0; b * g.rpp < nrows; b += NB) {
        WRegs<CPL, NB> R;
        gemv_issue<T, CPL, NB>(R, W, ldw, K, nrows, b, rowmap, tid, nthreads);
        gemv_apply<T, CPL, NB>(R, K, nrows, b, xs, ys, tid, nthreads);
    }
}
template <typename T, typename RowMap>
__device__ __forceinline__ void wg_gemv(const T* __restrict__ W, int ldw, int K, int nrows, const float* xs, float* ys, RowMap rowmap, int tid, int nthreads) {
    const int kc = K / ActTraits<T>::EPC;
    if (kc <= 64) gemv_stream<T, 1>(W, ldw, K, nrows, 0, xs, ys, rowmap, tid, nthreads);
    else if (kc == 128) gemv_stream<T, 2>(W, ldw, K, nrows, 0, xs, ys, rowmap, tid, nthreads);
    else if (kc == 256) gemv_stream<T, 4>(W, ldw, K, nrows, 0, xs, ys, rowmap, tid, nthreads);
    else gemv_stream<T, 8>(W, ldw, K, nrows, 0, xs, ys, rowmap, tid, nthreads);   // kc == 512 (the host checks)
}
// A Linear layer on the matrix cores, for the kernels whose widths are template constants (KDIM = the contraction length; 0 = run-time widths,
// the streaming VALU form above).  A wave owns 16-row tiles t = wave, wave + nwaves, ... (NT of them); the weight fragments of a tile are the A
// operand (lane (r, g): row 16 t + r, 16-byte chunk 4 kk + g), the input vector — kept in LDS in the weights' dtype — is the B operand with every
// column the same, so C[row 4 g + q][any column] = y[16 t + 4 g + q]: lanes with r = 0 store.  `pre_issue` requests all fragments (kernel entry),
// `pre_finish` multiplies once the vector is in LDS.
template <int KS, int NT>
struct MRegs {
    uint4 w[NT][KS];
};
template <typename T, int KDIM>
struct MCfg {
    static constexpr int KS = KDIM > 0 ? KDIM / ActTraits<T>::CK : 1;   // mma16 steps per tile
};
template <int KDIM, typename T, int NT, typename RowMap>
__device__ __forceinline__ void pre_issue(MRegs<MCfg<T, KDIM>::KS, NT>& R, const T* __restrict__ W, int ldw, int nrows, RowMap rowmap, int tid, int nthreads) {
    if constexpr (KDIM > 0) {
        constexpr int EPC = ActTraits<T>::EPC, KS = MCfg<T, KDIM>::KS;
        const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6, r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            int row = (u * nw + wave) * 16 + r;
            row = row < nrows ? row : nrows - 1;
            const T* wr = W + (int64_t)rowmap(row) * ldw + g * EPC;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) R.w[u][kk] = *reinterpret_cast<const uint4*>(wr + kk * 4 * EPC);
        }
    }
}
template <int KDIM, typename T, int NT, typename RowMap>
__device__ __forceinline__ void pre_finish(const MRegs<MCfg<T, KDIM>::KS, NT>& R, const T* __restrict__ W, int ldw, int K, int nrows, const float* xs, const T* xT, float* ys,
                                           RowMap rowmap, int tid, int nthreads) {
    if constexpr (KDIM > 0) {
        constexpr int EPC = ActTraits<T>::EPC, KS = MCfg<T, KDIM>::KS;
        const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6, r = lane & 15, g = lane >> 4;
        uint4 xf[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) xf[kk] = *reinterpret_cast<const uint4*>(xT + (kk * 4 + g) * EPC);
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) mma16<T>(R.w[u][kk], xf[kk], acc);
            const int row0 = (u * nw + wave) * 16 + g * 4;
            if (r == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (row0 + q < nrows) ys[row0 + q] = acc[q];
            }
        }
    } else {
        wg_gemv<T>(W, ldw, K, nrows, xs, ys, rowmap, tid, nthreads);
    }
}
// fp32 vector in LDS -> the same vector in the weights' dtype (the B operand above); the caller's next barrier publishes it
template <typename T>
__device__ __forceinline__ void to_operand(const float* xs, T* xT, int n, int tid, int nthreads) {
    for (int i = tid; i < n; i += nthreads) xT[i] = from_f32<T>(xs[i]);
}

// 16-byte load of cache rows.  COH (the persistent kernel, where a row written by another workgroup in an earlier step must be seen): a buffer load
// with sc1 — past this CU's L1, served coherently — matching sc1 (write-through) stores on the producer side; otherwise a plain load.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <bool COH>
__device__ __forceinline__ uint4 ld16(const void* base, int64_t byte_off) {
    if constexpr (COH) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const uint4*>(static_cast<const char*>(base) + byte_off);
    }
}
// two adjacent cache elements as ONE store (COH: sc1, 4 or 8 bytes: the hand-off forms validated for this part are >= 4-byte stores)
template <bool COH>
__device__ __forceinline__ void store_pair(__bf16* p, float a, float b) {
    const uint32_t lo = (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)a), hi = (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)b);
    const uint32_t pk = lo | (hi << 16);
    if constexpr (COH) __hip_atomic_store(reinterpret_cast<uint32_t*>(p), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *reinterpret_cast<uint32_t*>(p) = pk;
}
template <bool COH>
__device__ __forceinline__ void store_pair(float* p, float a, float b) {
    const unsigned long long pk = (unsigned long long)__builtin_bit_cast(uint32_t, a) | ((unsigned long long)__builtin_bit_cast(uint32_t, b) << 32);
    if constexpr (COH) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *reinterpret_cast<unsigned long long*>(p) = pk;
}

struct IdentityRow {
    __device__ __forceinline__ int operator()(int r) const { return r; }
};

// ------------------------------------------------------------------------------------------------ one query against a row-major cache
// Workgroup-wide softmax(q . K^T) V over keys 0 .. nk_cached-1 of the cache (K, V: [cap, HD] rows of T) plus — has_cur — one more key held in
// LDS (kcur / vcur, fp32).  Returns through LDS: oacc[HD] = sum_k exp(s_k - m) v_k, and (m, l) to every thread.  prob: LDS [>= nk_cached + 1].
// q (LDS) is pre-scaled.  nk_cached + has_cur may be 0: m = -inf, l = 0, oacc = 0.
// PRE (cap <= KPT * 512 positions): the key rows do not depend on this step's query — k_prefetch requests every key row of a thread (scores: thread t
// owns keys t, t + nthreads, ...) at kernel entry; the value chunks (thread (slot, chunk) owns keys slot, slot + nthreads / CPK, ...) are requested
// as soon as the scores have freed those registers and land under the two softmax reductions.  Otherwise both stream.
template <typename T, int HD>
struct KRegs {
    static constexpr int EPC = ActTraits<T>::EPC;
    static constexpr int CPK = HD / EPC;
    static constexpr int KPT = 4, VPT = 4 * CPK;
    uint4 k[KPT][CPK];
};
template <bool PRE, bool COH, typename T, int HD>
__device__ __forceinline__ void k_prefetch(KRegs<T, HD>& R, const T* __restrict__ Kg, int nk_cached, int tid, int nthreads) {
    if constexpr (PRE) {
        constexpr int CPK = KRegs<T, HD>::CPK;
#pragma unroll
        for (int j = 0; j < KRegs<T, HD>::KPT; ++j) {
            const int key = j * nthreads + tid;
            if (key < nk_cached) {
#pragma unroll
                for (int c = 0; c < CPK; ++c) R.k[j][c] = ld16<COH>(Kg, ((int64_t)key * HD) * (int64_t)sizeof(T) + c * 16);
            }
        }
    }
}
// the row of ONE key (the last cached one) into its owner's register slot: for caches whose newest row is written by another workgroup late in the
// previous step, everything but that row is requested early and this one once the step's hand-off says it is there
template <bool COH, typename T, int HD>
__device__ __forceinline__ void k_prefetch_one(KRegs<T, HD>& R, const T* __restrict__ Kg, int key, int tid, int nthreads) {
    constexpr int CPK = KRegs<T, HD>::CPK;
#pragma unroll
    for (int j = 0; j < KRegs<T, HD>::KPT; ++j) {
        if (key >= 0 && j * nthreads + tid == key) {
#pragma unroll
            for (int c = 0; c < CPK; ++c) R.k[j][c] = ld16<COH>(Kg, ((int64_t)key * HD) * (int64_t)sizeof(T) + c * 16);
        }
    }
}
template <bool PRE, bool COH, typename T, int HD>
__device__ __forceinline__ void wg_attend(const KRegs<T, HD>& R, const T* __restrict__ Kg, const T* __restrict__ Vg, int nk_cached, bool has_cur, const float* q_l, const float* kcur,
                                          const float* vcur, float* prob, float* part /* [nw][HD] */, float* oacc, float* red, float& m_out, float& l_out, int tid, int nthreads) {
    constexpr int EPC = ActTraits<T>::EPC;
    constexpr int CPK = HD / EPC;                 // 16-byte chunks per key row
    constexpr int KPT = KRegs<T, HD>::KPT, VPT = KRegs<T, HD>::VPT;
    constexpr float LOG2E = 1.4426950408889634f;
    const int nk = nk_cached + (has_cur ? 1 : 0);
    float mx = -INFINITY;
    {
        float q[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) q[c] = q_l[c];
        auto score = [&](const uint4 (&raw)[CPK], int key) __attribute__((always_inline)) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int c = 0; c < CPK; ++c) {
                float kv[EPC];
                unpack_w<T>(raw[c], kv);
#pragma unroll
                for (int e = 0; e < EPC; e += 2) {
                    a0 = fma1(q[c * EPC + e], kv[e], a0);
                    a1 = fma1(q[c * EPC + e + 1], kv[e + 1], a1);
                }
            }
            const float acc = a0 + a1;
            prob[key] = acc;
            mx = fmaxf(mx, acc);
        };
        if constexpr (PRE) {
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const int key = j * nthreads + tid;
                if (key < nk_cached) score(R.k[j], key);
            }
        } else {
            constexpr int KB = CPK >= 8 ? 1 : (CPK >= 4 ? 2 : 4);   // 8 loads in flight per lane
            for (int k0 = 0; k0 < nk_cached; k0 += nthreads * KB) {
                uint4 raw[KB][CPK];
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const int key = k0 + j * nthreads + tid;
                    const uint4* kr = reinterpret_cast<const uint4*>(Kg + (int64_t)(key < nk_cached ? key : nk_cached - 1) * HD);
#pragma unroll
                    for (int c = 0; c < CPK; ++c) raw[j][c] = kr[c];
                }
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const int key = k0 + j * nthreads + tid;
                    if (key < nk_cached) score(raw[j], key);
                }
            }
        }
        if (has_cur && tid == 0) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < HD; ++c) acc = fma1(q[c], kcur[c], acc);
            prob[nk_cached] = acc;
            mx = fmaxf(mx, acc);
        }
    }
    const int ch = tid % CPK, slot = tid / CPK, nslot = nthreads / CPK;
    uint4 vr[VPT];
    if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int key = j * nslot + slot;
            if (key < nk_cached) vr[j] = ld16<COH>(Vg, ((int64_t)key * HD + ch * EPC) * (int64_t)sizeof(T));
        }
    }
    const float m = wg_max(mx, red, tid, nthreads);   // barriers inside: prob is complete
    float ls = 0.f;
    for (int key = tid; key < nk; key += nthreads) {
        const float pv = __builtin_amdgcn_exp2f((prob[key] - m) * LOG2E);
        prob[key] = pv;
        ls += pv;
    }
    const float l = wg_sum(ls, red, tid, nthreads);   // barriers inside: prob holds the probabilities
    // ---- o = sum_k p_k v_k
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    auto pv_add = [&](const uint4& raw, float pv) __attribute__((always_inline)) {
        float vv[EPC];
        unpack_w<T>(raw, vv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = fma1(pv, vv[e], acc[e]);
    };
    if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int key = j * nslot + slot;
            if (key < nk_cached) pv_add(vr[j], prob[key]);
        }
    } else {
        for (int k0 = 0; k0 < nk_cached; k0 += nslot * 4) {
            uint4 raw[4];
            float pv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = k0 + j * nslot + slot;
                raw[j] = *reinterpret_cast<const uint4*>(Vg + (int64_t)(key < nk_cached ? key : nk_cached - 1) * HD + ch * EPC);
                pv[j] = key < nk_cached ? prob[key] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) pv_add(raw[j], pv[j]);
        }
    }
    if (has_cur && slot == 0) {
        const float pv = prob[nk_cached];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = fma1(pv, vcur[ch * EPC + e], acc[e]);
    }
    // lanes of a wave with the same chunk: lane % CPK (CPK divides 64)
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
#pragma unroll
        for (int o = 32; o >= CPK; o >>= 1) acc[e] += __shfl_xor(acc[e], o);
    }
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    if (lane < CPK) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) part[wave * HD + lane * EPC + e] = acc[e];
    }
    __syncthreads();
    if (tid < HD) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += part[w * HD + tid];
        oacc[tid] = s;
    }
    __syncthreads();
    m_out = m;
    l_out = l;
}

// bias + rotary embedding (interleaved pairs, models/base_blocks.py:81-88) + q scale on a projected head held in LDS: qkv = [q | k | v] (HD each).
// k and v are rounded to the cache dtype (later steps read them back from the cache) and appended at `pos`.  Thread roles: tid < HD/2 rotates a
// q pair, HD/2 <= tid < HD a k pair, 64 <= tid < 64 + HD adds a v bias; head_issue requests each role's bias / (cos, sin) at kernel entry.
struct HeadRegs {
    float b0, b1;
    float2 cs;
};
template <int HD>
__device__ __forceinline__ void head_issue(HeadRegs& R, const float* bq, const float* bk, const float* bv, const float* rope, int pos, bool with_kv, int tid) {
    constexpr int HD2 = HD / 2;
    const float2* cs = reinterpret_cast<const float2*>(rope) + (int64_t)pos * HD2;
    R.b0 = R.b1 = 0.f;
    R.cs = make_float2(1.f, 0.f);
    if (tid < HD2) {
        R.cs = cs[tid];
        R.b0 = bq[2 * tid];
        R.b1 = bq[2 * tid + 1];
    } else if (with_kv && tid < 2 * HD2) {
        const int t = tid - HD2;
        R.cs = cs[t];
        R.b0 = bk[2 * t];
        R.b1 = bk[2 * t + 1];
    } else if (with_kv && tid >= 64 && tid < 64 + HD) {
        R.b0 = bv[tid - 64];
    }
}
template <bool COH, typename T, int HD>
__device__ __forceinline__ void head_finish(const HeadRegs& R, float* qkv, bool with_kv, T* Krow, T* Vrow, int tid) {
    constexpr int HD2 = HD / 2;
    const float scale = 1.0f / sqrtf((float)HD);
    if (tid < HD2) {
        float oe, oo;
        rope_pair(qkv[2 * tid] + R.b0, qkv[2 * tid + 1] + R.b1, R.cs.x, R.cs.y, oe, oo);
        qkv[2 * tid] = oe * scale;
        qkv[2 * tid + 1] = oo * scale;
    } else if (with_kv && tid < 2 * HD2) {
        const int t = tid - HD2;
        float oe, oo;
        rope_pair(qkv[HD + 2 * t] + R.b0, qkv[HD + 2 * t + 1] + R.b1, R.cs.x, R.cs.y, oe, oo);
        oe = round_to(oe, T());
        oo = round_to(oo, T());
        qkv[HD + 2 * t] = oe;
        qkv[HD + 2 * t + 1] = oo;
        store_pair<COH>(Krow + 2 * t, oe, oo);
    } else if (with_kv && tid >= 64 && tid < 64 + HD) {
        const int t = tid - 64;
        const float v = round_to(qkv[2 * HD + t] + R.b0, T());
        qkv[2 * HD + t] = v;
        if constexpr (!COH) Vrow[t] = from_f32<T>(v);
    }
    __syncthreads();
    if constexpr (COH) {   // the value row as pairs (>= 4-byte stores); every element of it was rounded above
        if (with_kv && tid < HD2) store_pair<true>(Vrow + 2 * tid, qkv[2 * HD + 2 * tid], qkv[2 * HD + 2 * tid + 1]);
    }
}

struct KvArgs {
    SeaKvLayer L;
    SeaKvGlobal G;
    int32_t pos;         // position of this step's row in the caches / trajectory
    int32_t layer;
    uint32_t tag;        // unique per (step, layer) launch of the tails kernel
    int32_t last_layer;
    const float* xin;    // [B, F, E] input rows of this layer
    float* xout;         // [B, F, E] output rows of this layer
};

// Every kernel below opens by REQUESTING whatever does not depend on the activations — weights of its Linear layers (registers), gains, shifts,
// modulation rows, biases, cache rows — so that the launch is one memory round trip deep instead of one per stage.  E, D <= 512: a 512-thread
// workgroup holds one element of a row per thread.
// KE / KD > 0: E and D are these template constants and the Linear layers run on the matrix cores (pre_issue / pre_finish); KE = 0: run-time
// widths, streaming VALU GEMVs.
constexpr int tiles_per_wave(int rows, int nwaves) { return rows <= 0 ? 1 : (rows + 16 * nwaves - 1) / (16 * nwaves); }

// ------------------------------------------------------------------------------------------------ A: self attention (models/temporal.py:127-136 up to the projection)
// grid F * B * H, block 512.  LDS: xs[E] ns[E] nsT[E] qkv[3 HD] oacc[HD] red[32] part[8 HD] prob[cap + 8]
template <int KE, typename T, int HD>
__global__ __launch_bounds__(512) void kv_self_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr bool PRE = KE > 0;
    const int E = PRE ? KE : A.G.E;
    const int H = A.G.H, B = A.G.B, F = A.G.F, cap = A.G.cap, pos = A.pos;
    const int tid = threadIdx.x, nth = 512;
    const int h = blockIdx.x % H, ib_ = blockIdx.x / H, b = ib_ % B, i = ib_ / B;
    float* xs = sm;
    float* ns = xs + E;
    T* nsT = reinterpret_cast<T*>(ns + E);
    float* qkv = ns + 2 * E;
    float* oacc = qkv + 3 * HD;
    float* red = oacc + HD;
    float* part = red + 32;
    float* prob = part + 8 * HD;
    const SeaKvField& Fd = A.L.f[i];
    const int64_t crow = (int64_t)pos * B + b;
    const int64_t bh = (int64_t)b * H + h;
    T* Kc = static_cast<T*>(Fd.Ks) + bh * cap * HD;
    T* Vc = static_cast<T*>(Fd.Vs) + bh * cap * HD;
    const int hh = h;
    auto rowmap = [=](int r) { return (r / HD) * E + hh * HD + (r % HD); };
    MRegs<MCfg<T, KE>::KS, 1> rw;
    pre_issue<KE, T, 1>(rw, static_cast<const T*>(Fd.Wqkv), E, 3 * HD, rowmap, tid, nth);
    const float* x = A.xin + ((int64_t)b * F + i) * E;
    const float* ibp = (A.L.ib != nullptr && !A.G.ib_after_cross) ? A.L.ib + crow * E : nullptr;
    if (tid < E) xs[tid] = x[tid] + (ibp != nullptr ? ibp[tid] : 0.f);
    NormRegs<1> nr;
    norm_issue<T, 1>(nr, E, Fd.ln0, crow, tid, nth);
    HeadRegs hr;
    head_issue<HD>(hr, Fd.bqkv + h * HD, Fd.bqkv + E + h * HD, Fd.bqkv + 2 * E + h * HD, A.G.rope_self, pos, true, tid);
    KRegs<T, HD> kr;
    k_prefetch<PRE, false, T, HD>(kr, Kc, pos, tid, nth);
    __syncthreads();
    wg_norm_r<1, T>(xs, ns, PRE ? nsT : nullptr, E, nr, false, red, tid, nth);
    pre_finish<KE, T, 1>(rw, static_cast<const T*>(Fd.Wqkv), E, E, 3 * HD, ns, nsT, qkv, rowmap, tid, nth);
    __syncthreads();
    head_finish<false, T, HD>(hr, qkv, true, Kc + (int64_t)pos * HD, Vc + (int64_t)pos * HD, tid);
    float m, l;
    wg_attend<PRE, false, T, HD>(kr, Kc, Vc, pos, true, qkv, qkv + HD, qkv + 2 * HD, prob, part, oacc, red, m, l, tid, nth);
    if (tid < HD) A.G.att_e[((int64_t)b * F + i) * E + h * HD + tid] = oacc[tid] / l;
}

// ------------------------------------------------------------------------------------------------ B: out-projection + residual, down-projection + ln_cross (models/temporal.py:136, 177-178)
// grid F * B, block 512.  LDS: att[E] xs[E] y[E] attT[E] xsT[E] red[32]
template <int KE, int KD, typename T>
__global__ __launch_bounds__(512) void kv_oproj_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr bool PRE = KE > 0;
    const int E = PRE ? KE : A.G.E, D = PRE ? KD : A.G.D;
    const int B = A.G.B, F = A.G.F;
    const int tid = threadIdx.x, nth = 512;
    const int b = blockIdx.x % B, i = blockIdx.x / B;
    float* att = sm;
    float* xs = att + E;
    float* y = xs + E;
    T* attT = reinterpret_cast<T*>(y + E);
    T* xsT = reinterpret_cast<T*>(y + 2 * E);
    float* red = y + 3 * E;
    const SeaKvField& Fd = A.L.f[i];
    const bool ex = A.G.exchange != 0;
    const int64_t crow = (int64_t)A.pos * B + b, ro = ((int64_t)b * F + i);
    MRegs<MCfg<T, KE>::KS, tiles_per_wave(KE, 8)> rwo;
    MRegs<MCfg<T, KE>::KS, tiles_per_wave(KD, 8)> rwd;
    pre_issue<KE, T, tiles_per_wave(KE, 8)>(rwo, static_cast<const T*>(Fd.Wo), E, E, IdentityRow(), tid, nth);
    if (ex) pre_issue<KE, T, tiles_per_wave(KD, 8)>(rwd, static_cast<const T*>(Fd.Wdown), E, D, IdentityRow(), tid, nth);
    const float* ibp = (A.L.ib != nullptr && !A.G.ib_after_cross) ? A.L.ib + crow * E : nullptr;
    if (tid < E) {
        const float av = A.G.att_e[ro * E + tid];
        att[tid] = av;
        if constexpr (PRE) attT[tid] = from_f32<T>(av);
        xs[tid] = A.xin[ro * E + tid] + (ibp != nullptr ? ibp[tid] : 0.f);
    }
    NormRegs<1> nr;
    float bd = 0.f;
    if (ex) {
        norm_issue<T, 1>(nr, D, Fd.ln_cross, crow, tid, nth);
        if (tid < D) bd = Fd.bdown[tid];
    }
    __syncthreads();
    pre_finish<KE, T, tiles_per_wave(KE, 8)>(rwo, static_cast<const T*>(Fd.Wo), E, E, E, att, attT, y, IdentityRow(), tid, nth);
    __syncthreads();
    if (tid < E) {
        const float v = xs[tid] + y[tid];
        xs[tid] = v;
        if constexpr (PRE) xsT[tid] = from_f32<T>(v);
        A.G.xr[ro * E + tid] = v;
    }
    __syncthreads();
    if (!ex) return;
    pre_finish<KE, T, tiles_per_wave(KD, 8)>(rwd, static_cast<const T*>(Fd.Wdown), E, E, D, xs, xsT, y, IdentityRow(), tid, nth);
    __syncthreads();
    if (tid < D) y[tid] += bd;
    __syncthreads();
    wg_norm_r<1, T>(y, y, nullptr, D, nr, false, red, tid, nth);
    if (tid < D) A.G.nd_old[ro * D + tid] = y[tid];
}

// ------------------------------------------------------------------------------------------------ C: cross attention of every pair (models/temporal.py:181-186; base_blocks.py:232-293)
// grid F (F-1) * B * H, block 512.  Pair (i, j): query from nd_old_i; source j > i: key / value of this position from nd_old_j, appended, the head's
// output is final; source j < i (updated earlier in this sweep): only the cached keys here — (o, m, l) and q go to the tail, which merges this
// position's key.  LDS: ni[D] nj[D] niT[D] njT[D] qkv[3 HD] oacc[HD] red[32] part[8 HD] prob[cap + 8]
template <int KD, typename T, int HD>
__global__ __launch_bounds__(512) void kv_cross_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr bool PRE = KD > 0;
    const int D = PRE ? KD : A.G.D;
    const int H = A.G.H, B = A.G.B, F = A.G.F, cap = A.G.cap, pos = A.pos;
    const int tid = threadIdx.x, nth = 512;
    const int h = blockIdx.x % H, pb = blockIdx.x / H, b = pb % B, p = pb / B;
    const int i = p / (F - 1), s = p % (F - 1), j = s < i ? s : s + 1;
    const bool old_src = j > i;
    float* ni = sm;
    float* nj = ni + D;
    T* niT = reinterpret_cast<T*>(nj + D);
    T* njT = reinterpret_cast<T*>(nj + 2 * D);
    float* qkv = nj + 3 * D;
    float* oacc = qkv + 3 * HD;
    float* red = oacc + HD;
    float* part = red + 32;
    float* prob = part + 8 * HD;
    const SeaKvPair& P = A.L.p[i][j];
    const int64_t bh = (int64_t)b * H + h;
    T* Kc = static_cast<T*>(P.Kc) + bh * cap * HD;
    T* Vc = static_cast<T*>(P.Vc) + bh * cap * HD;
    const int hh = h;
    auto qmap = [=](int r) { return hh * HD + r; };
    auto kvmap = [=](int r) { return (r / HD) * D + hh * HD + (r % HD); };
    MRegs<MCfg<T, KD>::KS, 1> rq, rkv;
    pre_issue<KD, T, 1>(rq, static_cast<const T*>(P.Wq), D, HD, qmap, tid, nth);
    if (old_src) pre_issue<KD, T, 1>(rkv, static_cast<const T*>(P.Wkv), D, 2 * HD, kvmap, tid, nth);
    if (tid < D) {
        const float a = A.G.nd_old[((int64_t)b * F + i) * D + tid], c = A.G.nd_old[((int64_t)b * F + j) * D + tid];
        ni[tid] = a;
        nj[tid] = c;
        if constexpr (PRE) {
            niT[tid] = from_f32<T>(a);
            njT[tid] = from_f32<T>(c);
        }
    }
    HeadRegs hr;
    head_issue<HD>(hr, P.bq + h * HD, P.bkv + h * HD, P.bkv + D + h * HD, A.G.rope_cross, pos, old_src, tid);
    KRegs<T, HD> kr;
    k_prefetch<PRE, false, T, HD>(kr, Kc, pos, tid, nth);
    __syncthreads();
    pre_finish<KD, T, 1>(rq, static_cast<const T*>(P.Wq), D, D, HD, ni, niT, qkv, qmap, tid, nth);
    if (old_src) pre_finish<KD, T, 1>(rkv, static_cast<const T*>(P.Wkv), D, D, 2 * HD, nj, njT, qkv + HD, kvmap, tid, nth);
    __syncthreads();
    head_finish<false, T, HD>(hr, qkv, old_src, Kc + (int64_t)pos * HD, Vc + (int64_t)pos * HD, tid);
    float m, l;
    wg_attend<PRE, false, T, HD>(kr, Kc, Vc, pos, old_src, qkv, qkv + HD, qkv + 2 * HD, prob, part, oacc, red, m, l, tid, nth);
    const int64_t po = ((int64_t)p * B + b) * D + h * HD;
    if (tid < HD) {
        A.G.oc[po + tid] = old_src ? oacc[tid] / l : oacc[tid];
        if (!old_src) A.G.qc[po + tid] = qkv[tid];
    }
    if (!old_src && tid == 0) {
        float* ml = A.G.ml + (((int64_t)p * B + b) * H + h) * 2;
        ml[0] = m;
        ml[1] = l;
    }
}

// ------------------------------------------------------------------------------------------------ T: the Gauss-Seidel tails (models/temporal.py:187-192)
// grid F * B (field-major: workgroup (i, b) only ever waits for workgroups with a smaller index), block 512.
// LDS: nj[D] kv[2 D] o[D] g[D] gs[D] njT[D] oT[D] gsT[D] xs[E] y[E] xsT[E] red[32]
// The body is specialised on (number of fields, this field): which pairs wait, which weights are requested when and how many register blocks are
// alive is then static — with run-time control flow around the register blocks the allocator spills.
template <int KE, int KD, typename T, int NF, int I>
__device__ __forceinline__ void tail_body(const KvArgs& A, float* sm, int b) {
    constexpr bool PRE = KE > 0;
    constexpr bool HAS_DOWN = I < NF - 1;
    constexpr int NNEW = I;                          // sources updated before this field in the sweep: j = 0 .. I - 1
    constexpr int NT_2D = tiles_per_wave(2 * KD, 8), NT_D = tiles_per_wave(KD, 8), NT_E = tiles_per_wave(KE, 8);
    const int E = PRE ? KE : A.G.E, D = PRE ? KD : A.G.D;
    const int H = A.G.H, B = A.G.B, cap = A.G.cap, pos = A.pos;
    const int hd = D / H, hd2 = hd >> 1;
    const int tid = threadIdx.x, nth = 512;
    float* nj = sm;
    float* kv = nj + D;
    float* o = kv + 2 * D;
    float* g = o + D;
    float* gs = g + D;
    T* njT = reinterpret_cast<T*>(gs + D);
    T* oT = reinterpret_cast<T*>(gs + 2 * D);
    T* gsT = reinterpret_cast<T*>(gs + 3 * D);
    float* xs = gs + 4 * D;
    float* y = xs + E;
    T* xsT = reinterpret_cast<T*>(y + E);
    float* red = y + 2 * E;
    const SeaKvField& Fd = A.L.f[I];
    const int64_t crow = (int64_t)pos * B + b, ro = (int64_t)b * NF + I;
    // ---- requested before the first wait: the projections of every pair, the k / v weights of the first two updated sources
    MRegs<MCfg<T, KD>::KS, NT_D> rp[NF - 1];
    MRegs<MCfg<T, KD>::KS, NT_2D> rkv[NNEW > 2 ? 2 : (NNEW > 0 ? NNEW : 1)];
    MRegs<MCfg<T, KD>::KS, NT_E> rup;
    MRegs<MCfg<T, KE>::KS, NT_D> rdn;
    if constexpr (NNEW >= 1) pre_issue<KD, T, NT_2D>(rkv[0], static_cast<const T*>(A.L.p[I][0].Wkv), D, 2 * D, IdentityRow(), tid, nth);
    if constexpr (NNEW >= 2) pre_issue<KD, T, NT_2D>(rkv[1], static_cast<const T*>(A.L.p[I][1].Wkv), D, 2 * D, IdentityRow(), tid, nth);
#pragma unroll
    for (int s = 0; s < NF - 1; ++s) pre_issue<KD, T, NT_D>(rp[s], static_cast<const T*>(A.L.p[I][s < I ? s : s + 1].Wp), D, D, IdentityRow(), tid, nth);
    if constexpr (NNEW == 0) {   // the first field waits for nobody: everything at entry
        pre_issue<KD, T, NT_E>(rup, static_cast<const T*>(Fd.Wup), D, E, IdentityRow(), tid, nth);
        if constexpr (HAS_DOWN) pre_issue<KE, T, NT_D>(rdn, static_cast<const T*>(Fd.Wdown), E, D, IdentityRow(), tid, nth);
    }
    NormRegs<1> nr;
    float bd = 0.f, bu = 0.f;
    if constexpr (HAS_DOWN) {
        norm_issue<T, 1>(nr, D, Fd.ln_cross, crow, tid, nth);
        if (tid < D) bd = Fd.bdown[tid];
    }
    if (tid < E) {
        bu = Fd.bup[tid];
        xs[tid] = A.G.xr[ro * E + tid];
    }
    float gsum = 0.f;                                // element tid of sum_j gelu(proj_j(o_j))
    // ---- sources not yet updated in this sweep (j > I): their heads' outputs are final
#pragma unroll
    for (int s = I; s < NF - 1; ++s) {
        const int p = I * (NF - 1) + s;
        const int64_t po = ((int64_t)p * B + b) * D;
        __syncthreads();
        if (tid < D) {
            const float v = A.G.oc[po + tid];
            o[tid] = v;
            if constexpr (PRE) oT[tid] = from_f32<T>(v);
        }
        __syncthreads();
        pre_finish<KD, T, NT_D>(rp[s], static_cast<const T*>(A.L.p[I][s + 1].Wp), D, D, D, o, oT, g, IdentityRow(), tid, nth);
        __syncthreads();
        if (tid < D) gsum += gelu_erf(g[tid]);
    }
    if constexpr (NNEW > 0) pre_issue<KD, T, NT_E>(rup, static_cast<const T*>(Fd.Wup), D, E, IdentityRow(), tid, nth);
    // ---- sources updated earlier in this sweep (j < I): granules {value, tag} published by workgroup (j, b)
#pragma unroll
    for (int j = 0; j < NNEW; ++j) {
        const int p = I * (NF - 1) + j;
        const SeaKvPair& P = A.L.p[I][j];
        const int64_t po = ((int64_t)p * B + b) * D;
        const unsigned long long* hg = A.G.handoff + ((int64_t)b * NF + j) * D;
        __syncthreads();
        if (tid < D) {
            float v = 0.f;
            int it = 0;
            for (; it < KV_SPIN_LIMIT; ++it) {
                const unsigned long long pk = __hip_atomic_load(hg + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(pk >> 32) == A.tag) {
                    v = __builtin_bit_cast(float, (uint32_t)pk);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (it == KV_SPIN_LIMIT) __hip_atomic_store(A.G.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nj[tid] = v;
            if constexpr (PRE) njT[tid] = from_f32<T>(v);
        }
        // what the merge needs besides k / v: requested while the k / v projection runs
        float qv = 0.f, ov = 0.f, m0 = 0.f, l0 = 0.f, bk0 = 0.f, bk1 = 0.f, bvv = 0.f;
        float2 cs = make_float2(1.f, 0.f);
        if (tid < D) {
            const int hh = tid / hd;
            qv = A.G.qc[po + tid];
            ov = A.G.oc[po + tid];
            const float* ml = A.G.ml + (((int64_t)p * B + b) * H + hh) * 2;
            m0 = ml[0];
            l0 = ml[1];
            bvv = P.bkv[D + tid];
        }
        if (tid < D / 2) {
            const int hh = tid / hd2, t = tid - hh * hd2;
            cs = reinterpret_cast<const float2*>(A.G.rope_cross)[(int64_t)pos * hd2 + t];
            bk0 = P.bkv[hh * hd + 2 * t];
            bk1 = P.bkv[hh * hd + 2 * t + 1];
        }
        __syncthreads();
        if constexpr (PRE) {
            if (j < 2) pre_finish<KD, T, NT_2D>(rkv[j < 2 ? j : 0], static_cast<const T*>(P.Wkv), D, D, 2 * D, nj, njT, kv, IdentityRow(), tid, nth);
            else wg_gemv<T>(static_cast<const T*>(P.Wkv), D, D, 2 * D, nj, kv, IdentityRow(), tid, nth);   // (a fourth field's third source)
        } else {
            wg_gemv<T>(static_cast<const T*>(P.Wkv), D, D, 2 * D, nj, kv, IdentityRow(), tid, nth);
        }
        __syncthreads();
        // bias, rotary embedding of k, append, merge this key into the head's (o, m, l)
        if (tid < D / 2) {
            const int hh = tid / hd2, t = tid - hh * hd2;
            const int c0 = hh * hd + 2 * t;
            float oe, oo;
            rope_pair(kv[c0] + bk0, kv[c0 + 1] + bk1, cs.x, cs.y, oe, oo);
            oe = round_to(oe, T());
            oo = round_to(oo, T());
            kv[c0] = oe;
            kv[c0 + 1] = oo;
            T* Kr = static_cast<T*>(P.Kc) + (((int64_t)b * H + hh) * cap + pos) * hd;
            Kr[2 * t] = from_f32<T>(oe);
            Kr[2 * t + 1] = from_f32<T>(oo);
        }
        float vv = 0.f;
        if (tid < D) {
            vv = round_to(kv[D + tid] + bvv, T());
            const int hh = tid / hd;
            static_cast<T*>(P.Vc)[(((int64_t)b * H + hh) * cap + pos) * hd + (tid - hh * hd)] = from_f32<T>(vv);
        }
        __syncthreads();
        // score of this position's key: sum over the head's hd consecutive lanes (hd a power of two <= 64: heads do not straddle waves)
        float sc[1] = {tid < D ? qv * kv[tid] : 0.f};
        team_reduce<1>(sc, hd);
        if (tid < D) {
            const float m1 = fmaxf(m0, sc[0]);
            const float w0 = l0 > 0.f ? __expf(m0 - m1) : 0.f, w1 = __expf(sc[0] - m1);
            const float ov2 = (ov * w0 + vv * w1) / (l0 * w0 + w1);
            o[tid] = ov2;
            if constexpr (PRE) oT[tid] = from_f32<T>(ov2);
        }
        __syncthreads();
        pre_finish<KD, T, NT_D>(rp[j], static_cast<const T*>(P.Wp), D, D, D, o, oT, g, IdentityRow(), tid, nth);
        __syncthreads();
        if (tid < D) gsum += gelu_erf(g[tid]);
    }
    if (tid < D) {
        gs[tid] = gsum;
        if constexpr (PRE) gsT[tid] = from_f32<T>(gsum);
    }
    if constexpr (NNEW > 0 && HAS_DOWN) pre_issue<KE, T, NT_D>(rdn, static_cast<const T*>(Fd.Wdown), E, D, IdentityRow(), tid, nth);
    __syncthreads();
    pre_finish<KD, T, NT_E>(rup, static_cast<const T*>(Fd.Wup), D, D, E, gs, gsT, y, IdentityRow(), tid, nth);
    __syncthreads();
    if (tid < E) {
        const float v = xs[tid] + y[tid] + (float)(NF - 1) * bu;
        xs[tid] = v;
        if constexpr (PRE) xsT[tid] = from_f32<T>(v);
        A.G.xr[ro * E + tid] = v;
    }
    __syncthreads();
    if constexpr (HAS_DOWN) {
        pre_finish<KE, T, NT_D>(rdn, static_cast<const T*>(Fd.Wdown), E, E, D, xs, xsT, y, IdentityRow(), tid, nth);
        __syncthreads();
        if (tid < D) y[tid] += bd;
        __syncthreads();
        wg_norm_r<1, T>(y, y, nullptr, D, nr, false, red, tid, nth);
        unsigned long long* hgo = A.G.handoff + ((int64_t)b * NF + I) * D;
        if (tid < D)
            __hip_atomic_store(hgo + tid, ((unsigned long long)A.tag << 32) | (unsigned long long)__builtin_bit_cast(uint32_t, y[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int KE, int KD, typename T>
__global__ __launch_bounds__(512) void kv_tail_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = A.G.B;
    const int b = blockIdx.x % B, i = blockIdx.x / B;
    switch (A.G.F * 4 + i) {   // block-uniform
        case 2 * 4 + 0: tail_body<KE, KD, T, 2, 0>(A, sm, b); break;
        case 2 * 4 + 1: tail_body<KE, KD, T, 2, 1>(A, sm, b); break;
        case 3 * 4 + 0: tail_body<KE, KD, T, 3, 0>(A, sm, b); break;
        case 3 * 4 + 1: tail_body<KE, KD, T, 3, 1>(A, sm, b); break;
        case 3 * 4 + 2: tail_body<KE, KD, T, 3, 2>(A, sm, b); break;
        case 4 * 4 + 0: tail_body<KE, KD, T, 4, 0>(A, sm, b); break;
        case 4 * 4 + 1: tail_body<KE, KD, T, 4, 1>(A, sm, b); break;
        case 4 * 4 + 2: tail_body<KE, KD, T, 4, 2>(A, sm, b); break;
        default: tail_body<KE, KD, T, 4, 3>(A, sm, b); break;
    }
}

// ------------------------------------------------------------------------------------------------ D: info-bottleneck add + AdaLN_2 + fc1 (models/temporal.py:139-145; base_blocks.py:22)
// grid (F * B, S / 32), block 256: 32 rows of W1 per workgroup.  LDS: xs[E] ns[E] nsT[E] hs[32] red[32]
template <int KE, typename T>
__global__ __launch_bounds__(256) void kv_fc1_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int ROWS = 32;
    constexpr bool PRE = KE > 0;
    const int E = PRE ? KE : A.G.E;
    const int S = A.G.S, B = A.G.B, F = A.G.F;
    const int tid = threadIdx.x, nth = 256;
    const int b = blockIdx.x % B, i = blockIdx.x / B, r0 = blockIdx.y * ROWS;
    float* xs = sm;
    float* ns = xs + E;
    T* nsT = reinterpret_cast<T*>(ns + E);
    float* hs = ns + 2 * E;
    float* red = hs + ROWS;
    const SeaKvField& Fd = A.L.f[i];
    const int64_t crow = (int64_t)A.pos * B + b, ro = (int64_t)b * F + i;
    const int nr_ = r0 + ROWS <= S ? ROWS : S - r0;
    const T* W = static_cast<const T*>(Fd.W1) + (int64_t)r0 * E;
    MRegs<MCfg<T, KE>::KS, 1> rw;
    pre_issue<KE, T, 1>(rw, W, E, nr_, IdentityRow(), tid, nth);
    const float* ibp = (A.L.ib != nullptr && A.G.ib_after_cross) ? A.L.ib + crow * E : nullptr;
    for (int e = tid; e < E; e += nth) {
        const float v = A.G.xr[ro * E + e] + (ibp != nullptr ? ibp[e] : 0.f);
        xs[e] = v;
        if (blockIdx.y == 0) A.G.xq[ro * E + e] = v;
    }
    NormRegs<2> nrg;
    norm_issue<T, 2>(nrg, E, Fd.ln2, crow, tid, nth);
    const float b1v = tid < nr_ ? Fd.b1[r0 + tid] : 0.f;
    __syncthreads();
    wg_norm_r<2, T>(xs, ns, PRE ? nsT : nullptr, E, nrg, false, red, tid, nth);
    pre_finish<KE, T, 1>(rw, W, E, E, nr_, ns, nsT, hs, IdentityRow(), tid, nth);
    __syncthreads();
    if (tid < nr_) A.G.hbuf[ro * S + r0 + tid] = hs[tid] + b1v;
}

// ------------------------------------------------------------------------------------------------ E: nn.LayerNorm(S) + GELU + fc2 + residual (base_blocks.py:23-25; temporal.py:145)
// grid (F * B, E / 4), block 256: 4 rows of W2 per workgroup (a wave per row).  LDS: hs[S] ys[4] red[32]
template <typename T>
__global__ __launch_bounds__(256) void kv_fc2_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int ROWS = 4, EPC = ActTraits<T>::EPC;
    const int E = A.G.E, S = A.G.S, B = A.G.B, F = A.G.F;
    const int tid = threadIdx.x, nth = 256;
    const int b = blockIdx.x % B, i = blockIdx.x / B, r0 = blockIdx.y * ROWS;
    float* hs = sm;
    float* ys = hs + S;
    float* red = ys + ROWS;
    const SeaKvField& Fd = A.L.f[i];
    const int64_t ro = (int64_t)b * F + i;
    const int nr_ = r0 + ROWS <= E ? ROWS : E - r0;
    const T* W = static_cast<const T*>(Fd.W2) + (int64_t)r0 * S;
    const int kc = S / EPC;
    WRegs<4, 1> r4;
    WRegs<2, 1> r2;
    if (kc == 256) gemv_issue<T, 4, 1>(r4, W, S, S, nr_, 0, IdentityRow(), tid, nth);
    else if (kc == 128) gemv_issue<T, 2, 1>(r2, W, S, S, nr_, 0, IdentityRow(), tid, nth);
    for (int e = tid * 4; e < S; e += nth * 4) *reinterpret_cast<float4*>(hs + e) = *reinterpret_cast<const float4*>(A.G.hbuf + ro * S + e);
    SeaKvNorm nm;
    nm.gamma = Fd.lnw; nm.beta = Fd.lnb; nm.mod = nullptr; nm.ldmod = 0; nm.pad_ = 0;
    NormRegs<16> nrg;
    norm_issue<T, 16>(nrg, S, nm, 0, tid, nth);
    const float b2v = tid < nr_ ? Fd.b2[r0 + tid] + A.G.xq[ro * E + r0 + tid] : 0.f;
    __syncthreads();
    wg_norm_r<16, T>(hs, hs, nullptr, S, nrg, true, red, tid, nth);
    if (kc == 256) gemv_apply<T, 4, 1>(r4, S, nr_, 0, hs, ys, tid, nth);
    else if (kc == 128) gemv_apply<T, 2, 1>(r2, S, nr_, 0, hs, ys, tid, nth);
    else wg_gemv<T>(W, S, S, nr_, hs, ys, IdentityRow(), tid, nth);
    __syncthreads();
    if (tid < nr_) A.G.x3[ro * E + r0 + tid] = ys[tid] + b2v;
}

// ------------------------------------------------------------------------------------------------ Fin: proj (+ the model's final norm after the last layer) (temporal.py:146, 412-415)
// grid F * B, block 512.  LDS: xs[E] y[E] xsT[E] red[32]
template <int KE, typename T>
__global__ __launch_bounds__(512) void kv_proj_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr bool PRE = KE > 0;
    const int E = PRE ? KE : A.G.E;
    const int B = A.G.B, F = A.G.F;
    const int tid = threadIdx.x, nth = 512;
    const int b = blockIdx.x % B, i = blockIdx.x / B;
    float* xs = sm;
    float* y = xs + E;
    T* xsT = reinterpret_cast<T*>(y + E);
    float* red = y + 2 * E;
    const SeaKvField& Fd = A.L.f[i];
    const int64_t crow = (int64_t)A.pos * B + b, ro = (int64_t)b * F + i;
    MRegs<MCfg<T, KE>::KS, tiles_per_wave(KE, 8)> rw;
    pre_issue<KE, T, tiles_per_wave(KE, 8)>(rw, static_cast<const T*>(Fd.Wproj), E, E, IdentityRow(), tid, nth);
    if (tid < E) {
        const float v = A.G.x3[ro * E + tid];
        xs[tid] = v;
        if constexpr (PRE) xsT[tid] = from_f32<T>(v);
    }
    NormRegs<1> nr;
    if (A.last_layer) norm_issue<T, 1>(nr, E, A.G.final_ln[i], crow, tid, nth);
    const float bp = tid < E ? Fd.bproj[tid] : 0.f;
    __syncthreads();
    pre_finish<KE, T, tiles_per_wave(KE, 8)>(rw, static_cast<const T*>(Fd.Wproj), E, E, E, xs, xsT, y, IdentityRow(), tid, nth);
    __syncthreads();
    if (tid < E) y[tid] += bp;
    __syncthreads();
    if (A.last_layer) wg_norm_r<1, T>(y, y, nullptr, E, nr, false, red, tid, nth);
    if (tid < E) A.xout[ro * E + tid] = y[tid];
}

// ================================================================================================ the whole rollout as ONE launch
// Persistent form of the seven phases for B = 1, one layer, fixed widths (KE / KD): every workgroup has ONE role for the whole rollout and keeps the
// weights of that role in registers across steps (requested once, before the step loop) — no weight streaming and no launch boundary per step.
// What a phase hands to the next travels as tagged 8-byte granules {value, step tag} (one agent-scope store each; the consumer polls the words it
// needs — first one sentinel word from one lane with s_sleep, then all of them); every buffer is single: the consumer of step s has read it before
// the data dependencies let its producer reach step s + 1.  Cache rows are written with sc1 (write-through) stores and read with sc1 loads: the key /
// value row the tail of field i appends for a source j < i is read by another workgroup from the next step on.  Every spin is bounded and reports
// through the error word; a role never waits for a role with a larger block index within a step chain that it feeds, and all roles fit the chip at
// once (<= 256 workgroups of 512 threads, checked by the host), so the grid drains whatever happens.
// Roles by block index: self (F H) | oproj (F) | cross (F (F-1) H) | tail (F) | proj (F) | fc (F n_fcf: fc1 rows, then fc2 rows of the same field).
struct KvPersist {
    SeaKvLayer L;
    SeaKvGlobal G;
    int32_t pos0, n_steps;
    uint32_t tag0;
    int32_t n_fcf;                 // fc workgroups per field
    int32_t r1, r2;                // fc1 / fc2 rows per fc workgroup
    unsigned long long* gx;        // [F, E]   next step's input rows (proj -> self, oproj)
    unsigned long long* gatt;      // [F, E]   self-attention outputs (self -> oproj)
    unsigned long long* gxr;       // [F, E]   x after the out-projection (oproj -> tail)
    unsigned long long* gnd;       // [F, D]   ln_cross(cross_down(x)) of the pre-exchange rows (oproj -> cross)
    unsigned long long* goc;       // [P, D]   cross-attention outputs / unnormalised partials (cross -> tail)
    unsigned long long* gqc;       // [P, D]   queries of the pairs whose source is updated in this sweep
    unsigned long long* gml;       // [P, H, 2]
    unsigned long long* gnew;      // [F, D]   the updated field's normalised down-projection (tail -> later tails)
    unsigned long long* gxr2;      // [F, E]   x after the exchange (tail -> fc)
    unsigned long long* gh;        // [F, S]   fc1 rows (fc -> fc)
    unsigned long long* gx3;       // [F, E]   x after fc2 (fc -> proj)
    unsigned long long* stamps;    // tuning aid (sea_kv_debug_stamps): [n_steps, 16 roles, 4] values of the 100 MHz device clock, or NULL
};

// event k (< 4) of role slot `slot` (< 16) at step s: the device-wide 100 MHz clock (tuning aid, off unless a buffer was registered)
__device__ __forceinline__ void stamp(const KvPersist& A, int s, int slot, int k) {
    if (A.stamps != nullptr && threadIdx.x == 0) A.stamps[((int64_t)s * 16 + slot) * 4 + k] = wall_clock64();
}
__device__ __forceinline__ void gr_put(unsigned long long* g, float v, uint32_t tag) {
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A wait that gives up sets the error word; every other wait looks at that word every 256 polls and, once it is set, stops waiting for good
// (`dead`): a rollout whose workgroups are not all on the chip drains in seconds instead of n_steps x the spin limit.
struct ErrCtx {
    int32_t* err;
    int dead;
};
__device__ __forceinline__ bool gr_give_up(ErrCtx& ec, int it) {
    if ((it & 255) == 255 && __hip_atomic_load(ec.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ec.dead = 1;
    if (it == KV_SPIN_LIMIT - 1) {
        __hip_atomic_store(ec.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ec.dead = 1;
    }
    return ec.dead != 0;
}
__device__ __forceinline__ float gr_get(const unsigned long long* g, uint32_t tag, ErrCtx& ec) {
    if (ec.dead) return 0.f;
    for (int it = 0; it < KV_SPIN_LIMIT; ++it) {
        const unsigned long long pk = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(pk >> 32) == tag) return __builtin_bit_cast(float, (uint32_t)pk);
        if (gr_give_up(ec, it)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    return 0.f;
}
// two words with both loads in flight at once (one round trip when both are there)
__device__ __forceinline__ void gr_get2(const unsigned long long* g0, const unsigned long long* g1, uint32_t tag, ErrCtx& ec, float& v0, float& v1) {
    v0 = v1 = 0.f;
    if (ec.dead) return;
    for (int it = 0; it < KV_SPIN_LIMIT; ++it) {
        const unsigned long long p0 = __hip_atomic_load(g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long p1 = __hip_atomic_load(g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(p0 >> 32) == tag && (uint32_t)(p1 >> 32) == tag) {
            v0 = __builtin_bit_cast(float, (uint32_t)p0);
            v1 = __builtin_bit_cast(float, (uint32_t)p1);
            return;
        }
        if (gr_give_up(ec, it)) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
// One lane waits (politely) until the first word of a vector carries the tag — the producers are then in this phase — before the whole workgroup
// starts polling its own words: hundreds of spinning lanes would otherwise sit on the fabric for most of every step.
__device__ __forceinline__ void gr_wait_first(const unsigned long long* g, uint32_t tag, ErrCtx& ec, int tid) {
    if (tid == 0 && !ec.dead) {
        for (int it = 0; it < KV_SPIN_LIMIT; ++it) {
            const unsigned long long pk = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(pk >> 32) == tag) break;
            if (gr_give_up(ec, it)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

template <int KE, typename T, int HD>
__device__ __forceinline__ void role_self(const KvPersist& A, float* sm, int i, int h) {
    constexpr int E = KE;
    const int H = A.G.H, F = A.G.F, cap = A.G.cap;
    const int tid = threadIdx.x, nth = 512;
    ErrCtx ec{A.G.err, 0};
    float* xs = sm;
    float* ns = xs + E;
    T* nsT = reinterpret_cast<T*>(ns + E);
    float* qkv = ns + 2 * E;
    float* oacc = qkv + 3 * HD;
    float* red = oacc + HD;
    float* part = red + 32;
    float* prob = part + 8 * HD;
    const SeaKvField& Fd = A.L.f[i];
    T* Kc = static_cast<T*>(Fd.Ks) + (int64_t)h * cap * HD;
    T* Vc = static_cast<T*>(Fd.Vs) + (int64_t)h * cap * HD;
    const int hh = h;
    auto rowmap = [=](int r) { return (r / HD) * E + hh * HD + (r % HD); };
    MRegs<MCfg<T, KE>::KS, 1> rw;
    pre_issue<KE, T, 1>(rw, static_cast<const T*>(Fd.Wqkv), E, 3 * HD, rowmap, tid, nth);
    for (int s = 0; s < A.n_steps; ++s) {
        const int pos = A.pos0 + s;
        const uint32_t tag = A.tag0 + (uint32_t)s;
        const int64_t crow = pos;
        NormRegs<1> nr;
        norm_issue<T, 1>(nr, E, Fd.ln0, crow, tid, nth);
        HeadRegs hr;
        head_issue<HD>(hr, Fd.bqkv + h * HD, Fd.bqkv + E + h * HD, Fd.bqkv + 2 * E + h * HD, A.G.rope_self, pos, true, tid);
        KRegs<T, HD> kr;
        k_prefetch<true, true, T, HD>(kr, Kc, pos, tid, nth);
        const float* ibp = (A.L.ib != nullptr && !A.G.ib_after_cross) ? A.L.ib + crow * E : nullptr;
        if (tid < E) {
            const float xv = s == 0 ? A.G.traj[((int64_t)pos * F + i) * E + tid] : gr_get(A.gx + i * E + tid, tag - 1, ec);
            xs[tid] = xv + (ibp != nullptr ? ibp[tid] : 0.f);
        }
        __syncthreads();
        if (h == 0 && i == 0) stamp(A, s, 0, 0);
        wg_norm_r<1, T>(xs, ns, nsT, E, nr, false, red, tid, nth);
        pre_finish<KE, T, 1>(rw, static_cast<const T*>(Fd.Wqkv), E, E, 3 * HD, ns, nsT, qkv, rowmap, tid, nth);
        __syncthreads();
        head_finish<true, T, HD>(hr, qkv, true, Kc + (int64_t)pos * HD, Vc + (int64_t)pos * HD, tid);
        if (h == 0 && i == 0) stamp(A, s, 0, 1);
        float m, l;
        wg_attend<true, true, T, HD>(kr, Kc, Vc, pos, true, qkv, qkv + HD, qkv + 2 * HD, prob, part, oacc, red, m, l, tid, nth);
        if (tid < HD) gr_put(A.gatt + i * E + h * HD + tid, oacc[tid] / l, tag);
        if (h == 0 && i == 0) stamp(A, s, 0, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this step's cache row has left the CU before the next step's loads are issued
        __syncthreads();
    }
}

template <int KE, int KD, typename T>
__device__ __forceinline__ void role_oproj(const KvPersist& A, float* sm, int i) {
    constexpr int E = KE, D = KD;
    const int F = A.G.F;
    const int tid = threadIdx.x, nth = 512;
    ErrCtx ec{A.G.err, 0};
    float* att = sm;
    float* xs = att + E;
    float* y = xs + E;
    T* attT = reinterpret_cast<T*>(y + E);
    T* xsT = reinterpret_cast<T*>(y + 2 * E);
    float* red = y + 3 * E;
    const SeaKvField& Fd = A.L.f[i];
    const bool ex = A.G.exchange != 0;
    MRegs<MCfg<T, KE>::KS, tiles_per_wave(KE, 8)> rwo;
    MRegs<MCfg<T, KE>::KS, tiles_per_wave(KD, 8)> rwd;
    pre_issue<KE, T, tiles_per_wave(KE, 8)>(rwo, static_cast<const T*>(Fd.Wo), E, E, IdentityRow(), tid, nth);
    if (ex) pre_issue<KE, T, tiles_per_wave(KD, 8)>(rwd, static_cast<const T*>(Fd.Wdown), E, D, IdentityRow(), tid, nth);
    const float bd = (ex && tid < D) ? Fd.bdown[tid] : 0.f;
    for (int s = 0; s < A.n_steps; ++s) {
        const int pos = A.pos0 + s;
        const uint32_t tag = A.tag0 + (uint32_t)s;
        const int64_t crow = pos;
        NormRegs<1> nr;
        if (ex) norm_issue<T, 1>(nr, D, Fd.ln_cross, crow, tid, nth);
        const float* ibp = (A.L.ib != nullptr && !A.G.ib_after_cross) ? A.L.ib + crow * E : nullptr;
        float xv = 0.f;
        if (tid < E) xv = (s == 0 ? A.G.traj[((int64_t)pos * F + i) * E + tid] : gr_get(A.gx + i * E + tid, tag - 1, ec)) + (ibp != nullptr ? ibp[tid] : 0.f);
        if (tid < E) {
            const float av = gr_get(A.gatt + i * E + tid, tag, ec);
            att[tid] = av;
            attT[tid] = from_f32<T>(av);
            xs[tid] = xv;
        }
        __syncthreads();
        if (i == 0) stamp(A, s, 1, 0);
        pre_finish<KE, T, tiles_per_wave(KE, 8)>(rwo, static_cast<const T*>(Fd.Wo), E, E, E, att, attT, y, IdentityRow(), tid, nth);
        __syncthreads();
        if (tid < E) {
            const float v = xs[tid] + y[tid];
            xs[tid] = v;
            xsT[tid] = from_f32<T>(v);
            gr_put(A.gxr + i * E + tid, v, tag);
        }
        __syncthreads();
        if (ex) {
            pre_finish<KE, T, tiles_per_wave(KD, 8)>(rwd, static_cast<const T*>(Fd.Wdown), E, E, D, xs, xsT, y, IdentityRow(), tid, nth);
            __syncthreads();
            if (tid < D) y[tid] += bd;
            __syncthreads();
            wg_norm_r<1, T>(y, y, nullptr, D, nr, false, red, tid, nth);
            if (tid < D) gr_put(A.gnd + i * D + tid, y[tid], tag);
        }
        if (i == 0) stamp(A, s, 1, 1);
        __syncthreads();
    }
}

template <int KD, typename T, int HD>
__device__ __forceinline__ void role_cross(const KvPersist& A, float* sm, int p, int h) {
    constexpr int D = KD;
    const int H = A.G.H, F = A.G.F, cap = A.G.cap;
    const int tid = threadIdx.x, nth = 512;
    ErrCtx ec{A.G.err, 0};
    const int i = p / (F - 1), sx = p % (F - 1), j = sx < i ? sx : sx + 1;
    const bool old_src = j > i;
    float* ni = sm;
    float* nj = ni + D;
    T* niT = reinterpret_cast<T*>(nj + D);
    T* njT = reinterpret_cast<T*>(nj + 2 * D);
    float* qkv = nj + 3 * D;
    float* oacc = qkv + 3 * HD;
    float* red = oacc + HD;
    float* part = red + 32;
    float* prob = part + 8 * HD;
    const SeaKvPair& P = A.L.p[i][j];
    T* Kc = static_cast<T*>(P.Kc) + (int64_t)h * cap * HD;
    T* Vc = static_cast<T*>(P.Vc) + (int64_t)h * cap * HD;
    const int hh = h;
    auto qmap = [=](int r) { return hh * HD + r; };
    auto kvmap = [=](int r) { return (r / HD) * D + hh * HD + (r % HD); };
    MRegs<MCfg<T, KD>::KS, 1> rq, rkv;
    pre_issue<KD, T, 1>(rq, static_cast<const T*>(P.Wq), D, HD, qmap, tid, nth);
    if (old_src) pre_issue<KD, T, 1>(rkv, static_cast<const T*>(P.Wkv), D, 2 * HD, kvmap, tid, nth);
    for (int s = 0; s < A.n_steps; ++s) {
        const int pos = A.pos0 + s;
        const uint32_t tag = A.tag0 + (uint32_t)s;
        HeadRegs hr;
        head_issue<HD>(hr, P.bq + h * HD, P.bkv + h * HD, P.bkv + D + h * HD, A.G.rope_cross, pos, old_src, tid);
        // cached rows: this workgroup's own (source not yet updated in the sweep) are all there; a row appended by the TAIL of field i at the previous step
        // is there once this step's hand-off chain has reached this workgroup — everything but that row is requested now, that row after the wait
        KRegs<T, HD> kr;
        k_prefetch<true, true, T, HD>(kr, Kc, old_src ? pos : pos - 1, tid, nth);
        gr_wait_first(A.gnd + (i > j ? i : j) * D, tag, ec, tid);
        if (tid < D) {
            float a, c;
            gr_get2(A.gnd + i * D + tid, A.gnd + j * D + tid, tag, ec, a, c);
            ni[tid] = a;
            nj[tid] = c;
            niT[tid] = from_f32<T>(a);
            njT[tid] = from_f32<T>(c);
        }
        if (!old_src) k_prefetch_one<true, T, HD>(kr, Kc, pos - 1, tid, nth);
        __syncthreads();
        if (h == 0 && (p == 0 || p == F * (F - 1) - 1)) stamp(A, s, p == 0 ? 2 : 3, 0);
        pre_finish<KD, T, 1>(rq, static_cast<const T*>(P.Wq), D, D, HD, ni, niT, qkv, qmap, tid, nth);
        if (old_src) pre_finish<KD, T, 1>(rkv, static_cast<const T*>(P.Wkv), D, D, 2 * HD, nj, njT, qkv + HD, kvmap, tid, nth);
        __syncthreads();
        head_finish<true, T, HD>(hr, qkv, old_src, Kc + (int64_t)pos * HD, Vc + (int64_t)pos * HD, tid);
        float m, l;
        wg_attend<true, true, T, HD>(kr, Kc, Vc, pos, old_src, qkv, qkv + HD, qkv + 2 * HD, prob, part, oacc, red, m, l, tid, nth);
        const int po = p * D + h * HD;
        if (tid < HD) {
            if (!old_src) gr_put(A.gqc + po + tid, qkv[tid], tag);
            gr_put(A.goc + po + tid, old_src ? oacc[tid] / l : oacc[tid], tag);
        }
        if (!old_src && tid < 2) gr_put(A.gml + (p * H + h) * 2 + tid, tid == 0 ? m : l, tag);
        if (h == 0 && (p == 0 || p == F * (F - 1) - 1)) stamp(A, s, p == 0 ? 2 : 3, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

template <int KE, int KD, typename T, int NF, int I>
__device__ __forceinline__ void role_tail(const KvPersist& A, float* sm) {
    constexpr bool HAS_DOWN = I < NF - 1;
    constexpr int NNEW = I;
    constexpr int NT_2D = tiles_per_wave(2 * KD, 8), NT_D = tiles_per_wave(KD, 8), NT_E = tiles_per_wave(KE, 8);
    constexpr int E = KE, D = KD;
    const int H = A.G.H, cap = A.G.cap;
    const int hd = D / H, hd2 = hd >> 1;
    const int tid = threadIdx.x, nth = 512;
    ErrCtx ec{A.G.err, 0};
    float* nj = sm;
    float* kv = nj + D;
    float* o = kv + 2 * D;
    float* g = o + D;
    float* gs = g + D;
    T* njT = reinterpret_cast<T*>(gs + D);
    T* oT = reinterpret_cast<T*>(gs + 2 * D);
    T* gsT = reinterpret_cast<T*>(gs + 3 * D);
    float* xs = gs + 4 * D;
    float* y = xs + E;
    T* xsT = reinterpret_cast<T*>(y + E);
    float* red = y + 2 * E;
    const SeaKvField& Fd = A.L.f[I];
    // resident for the whole rollout: projections of every pair, k / v weights of up to two updated sources, up- and down-projection
    MRegs<MCfg<T, KD>::KS, NT_D> rp[NF - 1];
    MRegs<MCfg<T, KD>::KS, NT_2D> rkv[NNEW > 2 ? 2 : (NNEW > 0 ? NNEW : 1)];
    MRegs<MCfg<T, KD>::KS, NT_E> rup;
    MRegs<MCfg<T, KE>::KS, NT_D> rdn;
    if constexpr (NNEW >= 1) pre_issue<KD, T, NT_2D>(rkv[0], static_cast<const T*>(A.L.p[I][0].Wkv), D, 2 * D, IdentityRow(), tid, nth);
    if constexpr (NNEW >= 2) pre_issue<KD, T, NT_2D>(rkv[1], static_cast<const T*>(A.L.p[I][1].Wkv), D, 2 * D, IdentityRow(), tid, nth);
#pragma unroll
    for (int s = 0; s < NF - 1; ++s) pre_issue<KD, T, NT_D>(rp[s], static_cast<const T*>(A.L.p[I][s < I ? s : s + 1].Wp), D, D, IdentityRow(), tid, nth);
    pre_issue<KD, T, NT_E>(rup, static_cast<const T*>(Fd.Wup), D, E, IdentityRow(), tid, nth);
    if constexpr (HAS_DOWN) pre_issue<KE, T, NT_D>(rdn, static_cast<const T*>(Fd.Wdown), E, D, IdentityRow(), tid, nth);
    const float bd = (HAS_DOWN && tid < D) ? Fd.bdown[tid] : 0.f;
    const float bu = tid < E ? Fd.bup[tid] : 0.f;
    for (int st = 0; st < A.n_steps; ++st) {
        const int pos = A.pos0 + st;
        const uint32_t tag = A.tag0 + (uint32_t)st;
        const int64_t crow = pos;
        NormRegs<1> nr;
        if constexpr (HAS_DOWN) norm_issue<T, 1>(nr, D, Fd.ln_cross, crow, tid, nth);
        float gsum = 0.f;
        if (tid < E) xs[tid] = gr_get(A.gxr + I * E + tid, tag, ec);   // (published before the cross phase)
        // ---- sources not yet updated in this sweep (j > I): their heads' outputs are final
#pragma unroll
        for (int s = I; s < NF - 1; ++s) {
            const int p = I * (NF - 1) + s;
            if (s == I) stamp(A, st, 4 + I, 0);
            if (tid < D) {
                const float v = gr_get(A.goc + p * D + tid, tag, ec);
                o[tid] = v;
                oT[tid] = from_f32<T>(v);
            }
            __syncthreads();
            pre_finish<KD, T, NT_D>(rp[s], static_cast<const T*>(A.L.p[I][s + 1].Wp), D, D, D, o, oT, g, IdentityRow(), tid, nth);
            __syncthreads();
            if (tid < D) gsum += gelu_erf(g[tid]);
        }
        // ---- sources updated earlier in this sweep (j < I): what the cross-attention launch left for them (partials, queries) is there long before the
        // updated rows are — gathered now, off the chain
        float qvs[NNEW > 0 ? NNEW : 1], ovs[NNEW > 0 ? NNEW : 1], m0s[NNEW > 0 ? NNEW : 1], l0s[NNEW > 0 ? NNEW : 1];
#pragma unroll
        for (int j = 0; j < NNEW; ++j) {
            qvs[j] = ovs[j] = m0s[j] = l0s[j] = 0.f;
            if (tid < D) {
                const int p = I * (NF - 1) + j, hh = tid / hd;
                gr_get2(A.gqc + p * D + tid, A.goc + p * D + tid, tag, ec, qvs[j], ovs[j]);
                gr_get2(A.gml + (p * H + hh) * 2, A.gml + (p * H + hh) * 2 + 1, tag, ec, m0s[j], l0s[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < NNEW; ++j) {
            const int p = I * (NF - 1) + j;
            const SeaKvPair& P = A.L.p[I][j];
            const float qv = qvs[j], ov = ovs[j], m0 = m0s[j], l0 = l0s[j];
            float bk0 = 0.f, bk1 = 0.f, bv0 = 0.f, bv1 = 0.f;
            float2 cs = make_float2(1.f, 0.f);
            if (tid < D / 2) {
                const int hh = tid / hd2, t = tid - hh * hd2;
                cs = reinterpret_cast<const float2*>(A.G.rope_cross)[(int64_t)pos * hd2 + t];
                bk0 = P.bkv[hh * hd + 2 * t];
                bk1 = P.bkv[hh * hd + 2 * t + 1];
                bv0 = P.bkv[D + 2 * tid];
                bv1 = P.bkv[D + 2 * tid + 1];
            }
            if (tid < D) {
                const float v = gr_get(A.gnew + j * D + tid, tag, ec);
                nj[tid] = v;
                njT[tid] = from_f32<T>(v);
            }
            __syncthreads();
            if (j == NNEW - 1) stamp(A, st, 4 + I, 1);
            if (j < 2) pre_finish<KD, T, NT_2D>(rkv[j < 2 ? j : 0], static_cast<const T*>(P.Wkv), D, D, 2 * D, nj, njT, kv, IdentityRow(), tid, nth);
            else wg_gemv<T>(static_cast<const T*>(P.Wkv), D, D, 2 * D, nj, kv, IdentityRow(), tid, nth);
            __syncthreads();
            // bias, rotary embedding of k, append (pairs: >= 4-byte write-through stores), merge this key into the head's (o, m, l)
            if (tid < D / 2) {
                const int hh = tid / hd2, t = tid - hh * hd2;
                const int c0 = hh * hd + 2 * t;
                float oe, oo;
                rope_pair(kv[c0] + bk0, kv[c0 + 1] + bk1, cs.x, cs.y, oe, oo);
                oe = round_to(oe, T());
                oo = round_to(oo, T());
                kv[c0] = oe;
                kv[c0 + 1] = oo;
                store_pair<true>(static_cast<T*>(P.Kc) + ((int64_t)hh * cap + pos) * hd + 2 * t, oe, oo);
                const float v0 = round_to(kv[D + 2 * tid] + bv0, T()), v1 = round_to(kv[D + 2 * tid + 1] + bv1, T());
                kv[D + 2 * tid] = v0;
                kv[D + 2 * tid + 1] = v1;
                const int hv = (2 * tid) / hd;
                store_pair<true>(static_cast<T*>(P.Vc) + ((int64_t)hv * cap + pos) * hd + (2 * tid - hv * hd), v0, v1);
            }
            __syncthreads();
            float sc[1] = {tid < D ? qv * kv[tid] : 0.f};
            team_reduce<1>(sc, hd);
            if (tid < D) {
                const float m1 = fmaxf(m0, sc[0]);
                const float w0 = l0 > 0.f ? __expf(m0 - m1) : 0.f, w1 = __expf(sc[0] - m1);
                const float ov2 = (ov * w0 + kv[D + tid] * w1) / (l0 * w0 + w1);
                o[tid] = ov2;
                oT[tid] = from_f32<T>(ov2);
            }
            __syncthreads();
            pre_finish<KD, T, NT_D>(rp[j], static_cast<const T*>(P.Wp), D, D, D, o, oT, g, IdentityRow(), tid, nth);
            __syncthreads();
            if (tid < D) gsum += gelu_erf(g[tid]);
        }
        if (tid < D) {
            gs[tid] = gsum;
            gsT[tid] = from_f32<T>(gsum);
        }
        __syncthreads();
        pre_finish<KD, T, NT_E>(rup, static_cast<const T*>(Fd.Wup), D, D, E, gs, gsT, y, IdentityRow(), tid, nth);
        __syncthreads();
        float xnew = 0.f;
        if (tid < E) {
            xnew = xs[tid] + y[tid] + (float)(NF - 1) * bu;
            xs[tid] = xnew;
            xsT[tid] = from_f32<T>(xnew);
        }
        __syncthreads();
        if constexpr (HAS_DOWN) {
            pre_finish<KE, T, NT_D>(rdn, static_cast<const T*>(Fd.Wdown), E, E, D, xs, xsT, y, IdentityRow(), tid, nth);
            __syncthreads();
            if (tid < D) y[tid] += bd;
            __syncthreads();
            wg_norm_r<1, T>(y, y, nullptr, D, nr, false, red, tid, nth);
            if (tid < D) gr_put(A.gnew + I * D + tid, y[tid], tag);
            stamp(A, st, 4 + I, 2);
        }
        // the rows appended above must have left this CU before anything downstream of x can lead to their being read (next step's cross attention)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < E) gr_put(A.gxr2 + I * E + tid, xnew, tag);
        stamp(A, st, 4 + I, 3);
        __syncthreads();
    }
}

template <int KE, int KD, typename T>
__device__ __forceinline__ void role_tail_dispatch(const KvPersist& A, float* sm, int i) {
    switch (A.G.F * 4 + i) {   // block-uniform
        case 2 * 4 + 0: role_tail<KE, KD, T, 2, 0>(A, sm); break;
        case 2 * 4 + 1: role_tail<KE, KD, T, 2, 1>(A, sm); break;
        case 3 * 4 + 0: role_tail<KE, KD, T, 3, 0>(A, sm); break;
        case 3 * 4 + 1: role_tail<KE, KD, T, 3, 1>(A, sm); break;
        case 3 * 4 + 2: role_tail<KE, KD, T, 3, 2>(A, sm); break;
        case 4 * 4 + 0: role_tail<KE, KD, T, 4, 0>(A, sm); break;
        case 4 * 4 + 1: role_tail<KE, KD, T, 4, 1>(A, sm); break;
        case 4 * 4 + 2: role_tail<KE, KD, T, 4, 2>(A, sm); break;
        default: role_tail<KE, KD, T, 4, 3>(A, sm); break;
    }
}

// fc1 rows [k r1, (k+1) r1) and fc2 rows [k r2, (k+1) r2) of field i.  LDS: xs[E] ns[E] nsT[E] hs[S] ys[128] red[32]
template <int KE, typename T>
__device__ __forceinline__ void role_fc(const KvPersist& A, float* sm, int i, int k) {
    constexpr int E = KE, EPC = ActTraits<T>::EPC;
    const int S = A.G.S, exch = A.G.exchange;
    const int tid = threadIdx.x, nth = 512;
    ErrCtx ec{A.G.err, 0};
    float* xs = sm;
    float* ns = xs + E;
    T* nsT = reinterpret_cast<T*>(ns + E);
    float* hs = ns + 2 * E;
    float* ys = hs + S;
    float* red = ys + 128;
    const SeaKvField& Fd = A.L.f[i];
    const int r1 = A.r1, r2 = A.r2;
    const int a0 = k * r1, n1 = a0 >= S ? 0 : (a0 + r1 <= S ? r1 : S - a0);          // fc1 rows of this workgroup (a multiple of 16 except at the end)
    const int c0 = k * r2, n2 = c0 >= E ? 0 : (c0 + r2 <= E ? r2 : E - c0);          // fc2 rows (<= 8)
    const T* W1 = static_cast<const T*>(Fd.W1) + (int64_t)(n1 > 0 ? a0 : 0) * E;
    const T* W2 = static_cast<const T*>(Fd.W2) + (int64_t)(n2 > 0 ? c0 : 0) * S;
    MRegs<MCfg<T, KE>::KS, 1> rw1;                      // up to 8 tiles of 16 rows: r1 <= 128
    pre_issue<KE, T, 1>(rw1, W1, E, n1 > 0 ? n1 : 1, IdentityRow(), tid, nth);
    const int kc = S / EPC;
    WRegs<4, 1> w4;                                      // fc2: a wave per row, 8 rows per pass
    WRegs<2, 1> w2;
    if (kc == 256) gemv_issue<T, 4, 1>(w4, W2, S, S, n2 > 0 ? n2 : 1, 0, IdentityRow(), tid, nth);
    else if (kc == 128) gemv_issue<T, 2, 1>(w2, W2, S, S, n2 > 0 ? n2 : 1, 0, IdentityRow(), tid, nth);
    SeaKvNorm nm;
    nm.gamma = Fd.lnw; nm.beta = Fd.lnb; nm.mod = nullptr; nm.ldmod = 0; nm.pad_ = 0;
    NormRegs<8> nrs;                                     // S <= 4096
    norm_issue<T, 8>(nrs, S, nm, 0, tid, nth);
    const float b1v = tid < n1 ? Fd.b1[a0 + tid] : 0.f;
    const float b2v = tid < n2 ? Fd.b2[c0 + tid] : 0.f;
    const unsigned long long* gin = exch ? A.gxr2 : A.gxr;   // no exchange: the out-projection's rows go straight to the MLP
    for (int s = 0; s < A.n_steps; ++s) {
        const int pos = A.pos0 + s;
        const uint32_t tag = A.tag0 + (uint32_t)s;
        const int64_t crow = pos;
        NormRegs<1> nr;
        norm_issue<T, 1>(nr, E, Fd.ln2, crow, tid, nth);
        const float* ibp = (A.L.ib != nullptr && A.G.ib_after_cross) ? A.L.ib + crow * E : nullptr;
        const float ibv = (ibp != nullptr && tid < E) ? ibp[tid] : 0.f;
        gr_wait_first(gin + i * E, tag, ec, tid);
        if (tid < E) xs[tid] = gr_get(gin + i * E + tid, tag, ec) + ibv;
        __syncthreads();
        if (k == 0 && (i == 0 || i == A.G.F - 1)) stamp(A, s, i == 0 ? 8 : 9, 0);
        const float xq = (tid < n2) ? xs[c0 + tid] : 0.f;                 // the residual of this workgroup's fc2 rows
        wg_norm_r<1, T>(xs, ns, nsT, E, nr, false, red, tid, nth);
        if (n1 > 0) pre_finish<KE, T, 1>(rw1, W1, E, E, n1, ns, nsT, ys, IdentityRow(), tid, nth);
        __syncthreads();
        if (tid < n1) gr_put(A.gh + (int64_t)i * S + a0 + tid, ys[tid] + b1v, tag);
        if (k == 0 && (i == 0 || i == A.G.F - 1)) stamp(A, s, i == 0 ? 8 : 9, 1);
        // ---- all rows of the field's hidden vector (every fc workgroup of the field contributes r1 of them)
        gr_wait_first(A.gh + (int64_t)i * S + S - 1, tag, ec, tid);
        for (int e = tid; e < S; e += nth) hs[e] = gr_get(A.gh + (int64_t)i * S + e, tag, ec);
        __syncthreads();
        if (k == 0 && (i == 0 || i == A.G.F - 1)) stamp(A, s, i == 0 ? 8 : 9, 2);
        wg_norm_r<8, T>(hs, hs, nullptr, S, nrs, true, red, tid, nth);
        if (n2 > 0) {
            if (kc == 256) gemv_apply<T, 4, 1>(w4, S, n2, 0, hs, ys, tid, nth);
            else if (kc == 128) gemv_apply<T, 2, 1>(w2, S, n2, 0, hs, ys, tid, nth);
            else wg_gemv<T>(W2, S, S, n2, hs, ys, IdentityRow(), tid, nth);
        }
        __syncthreads();
        if (tid < n2) gr_put(A.gx3 + i * E + c0 + tid, ys[tid] + b2v + xq, tag);
        if (k == 0 && (i == 0 || i == A.G.F - 1)) stamp(A, s, i == 0 ? 8 : 9, 3);
        __syncthreads();
    }
}

template <int KE, typename T>
__device__ __forceinline__ void role_proj(const KvPersist& A, float* sm, int i) {
    constexpr int E = KE;
    const int F = A.G.F;
    const int tid = threadIdx.x, nth = 512;
    ErrCtx ec{A.G.err, 0};
    float* xs = sm;
    float* y = xs + E;
    T* xsT = reinterpret_cast<T*>(y + E);
    float* red = y + 2 * E;
    const SeaKvField& Fd = A.L.f[i];
    MRegs<MCfg<T, KE>::KS, tiles_per_wave(KE, 8)> rw;
    pre_issue<KE, T, tiles_per_wave(KE, 8)>(rw, static_cast<const T*>(Fd.Wproj), E, E, IdentityRow(), tid, nth);
    const float bp = tid < E ? Fd.bproj[tid] : 0.f;
    for (int s = 0; s < A.n_steps; ++s) {
        const int pos = A.pos0 + s;
        const uint32_t tag = A.tag0 + (uint32_t)s;
        NormRegs<1> nr;
        norm_issue<T, 1>(nr, E, A.G.final_ln[i], (int64_t)pos, tid, nth);
        if (tid < E) {
            const float v = gr_get(A.gx3 + i * E + tid, tag, ec);
            xs[tid] = v;
            xsT[tid] = from_f32<T>(v);
        }
        __syncthreads();
        if (i == 0 || i == F - 1) stamp(A, s, i == 0 ? 10 : 11, 0);
        pre_finish<KE, T, tiles_per_wave(KE, 8)>(rw, static_cast<const T*>(Fd.Wproj), E, E, E, xs, xsT, y, IdentityRow(), tid, nth);
        __syncthreads();
        if (tid < E) y[tid] += bp;
        __syncthreads();
        wg_norm_r<1, T>(y, y, nullptr, E, nr, false, red, tid, nth);
        if (tid < E) {
            A.G.traj[((int64_t)(pos + 1) * F + i) * E + tid] = y[tid];
            gr_put(A.gx + i * E + tid, y[tid], tag);
        }
        if (i == 0 || i == F - 1) stamp(A, s, i == 0 ? 10 : 11, 1);
        __syncthreads();
    }
}

template <int KE, typename T, int HDS, int HDC>
__global__ __launch_bounds__(512) void kv_persistent_kernel(const KvPersist A_) {
    // The roles index the per-field / per-pair tables of the argument struct with their (workgroup-uniform) field and pair numbers.  Through the by-value
    // parameter the compiler first copies the whole 2.3-3.1 KB struct to SCRATCH ("memcpy-split": one private copy per lane, flat addressing) and indexes
    // that; through the kernarg segment itself the same reads are scalar loads of constant memory and the kernel needs no scratch at all.
    const KvPersist& A = *(const KvPersist*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int KD = KE / 2;
    const int F = A.G.F, H = A.G.H, ex = A.G.exchange;
    int w = blockIdx.x;
    const int n_self = F * H, n_cross = ex ? F * (F - 1) * H : 0, n_tail = ex ? F : 0;
    if (w < n_self) { role_self<KE, T, HDS>(A, sm, w / H, w % H); return; }
    w -= n_self;
    if (w < F) { role_oproj<KE, KD, T>(A, sm, w); return; }
    w -= F;
    if (w < n_cross) { role_cross<KD, T, HDC>(A, sm, w / H, w % H); return; }
    w -= n_cross;
    if (w < n_tail) { role_tail_dispatch<KE, KD, T>(A, sm, w); return; }
    w -= n_tail;
    if (w < F) { role_proj<KE, T>(A, sm, w); return; }
    w -= F;
    if (w < F * A.n_fcf) role_fc<KE, T>(A, sm, w / A.n_fcf, w % A.n_fcf);
}

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static bool kdim_ok(int K, int epc) {
    if (K % epc) return false;
    const int kc = K / epc;
    return (kc <= 64 && pow2(kc) && kc >= 2) || kc == 128 || kc == 256 || kc == 512;
}

// Widths with matrix-core kernels: E in {64, 128, 256} with D = E / 2 (or no exchange), heads of at most 32 columns, a cache of at most 2048
// positions (the key rows a thread prefetches).  SEA_TUNE=kv_pre=0 keeps the run-time-width kernels (tuning aid).
static int pre_width(const SeaKvGlobal& G) {
    const int env = sea_tune("kv_pre", 1);
    if (!env || G.cap > 2048 || (G.E != 64 && G.E != 128 && G.E != 256) || G.E / G.H > 32) return 0;
    if (G.exchange && (G.D * 2 != G.E || G.D / G.H > 32)) return 0;
    return G.E;
}

template <int KE, typename T>
static int run_steps(const SeaKvGlobal& G, const SeaKvLayer* layers, int pos0, int n_steps, uint32_t tag0, hipStream_t s) {
    constexpr int KD = KE / 2;
    const int F = G.F, E = G.E, D = G.D, S = G.S, H = G.H, B = G.B, Ln = G.L, cap = G.cap;
    const int hd_s = E / H, hd_c = G.exchange ? D / H : 0;
    const int lds_attn_s = (3 * E + 3 * hd_s + hd_s + 32 + 8 * hd_s + cap + 8) * 4;
    const int lds_attn_c = G.exchange ? (4 * D + 3 * hd_c + hd_c + 32 + 8 * hd_c + cap + 8) * 4 : 0;
    const int lds_b = (5 * E + 32) * 4, lds_t = (9 * D + 3 * E + 32) * 4, lds_p = (3 * E + 32) * 4;
    const int r1 = 32, r2 = 4;
    const int lds_d = (3 * E + r1 + 32) * 4, lds_e = (S + r2 + 32) * 4;
    const int64_t slab = (int64_t)B * F * E;
#define KV_SELF(HDV) kv_self_kernel<KE, T, HDV><<<dim3(F * B * H), dim3(512), lds_attn_s, s>>>(A)
#define KV_CROSS(HDV) kv_cross_kernel<KD, T, HDV><<<dim3(F * (F - 1) * B * H), dim3(512), lds_attn_c, s>>>(A)
    for (int k = 0; k < n_steps; ++k) {
        const int pos = pos0 + k;
        for (int l = 0; l < Ln; ++l) {
            KvArgs A;
            A.L = layers[l];
            A.G = G;
            A.pos = pos;
            A.layer = l;
            A.tag = tag0 + (uint32_t)(k * Ln + l);
            A.last_layer = l == Ln - 1;
            A.xin = l == 0 ? G.traj + (int64_t)pos * slab : G.xl[(l - 1) & 1];
            A.xout = l == Ln - 1 ? G.traj + (int64_t)(pos + 1) * slab : G.xl[l & 1];
            switch (hd_s) {
                case 8: KV_SELF(8); break;
                case 16: KV_SELF(16); break;
                case 32: KV_SELF(32); break;
                default:
                    if constexpr (KE == 0) KV_SELF(64);   // (the matrix-core kernels stop at head dim 32)
                    break;
            }
            kv_oproj_kernel<KE, KD, T><<<dim3(F * B), dim3(512), lds_b, s>>>(A);
            if (G.exchange) {
                switch (hd_c) {
                    case 8: KV_CROSS(8); break;
                    case 16: KV_CROSS(16); break;
                    case 32: KV_CROSS(32); break;
                    default:
                        if constexpr (KE == 0) KV_CROSS(64);
                        break;
                }
                kv_tail_kernel<KE, KD, T><<<dim3(F * B), dim3(512), lds_t, s>>>(A);
            }
            kv_fc1_kernel<KE, T><<<dim3(F * B, (S + r1 - 1) / r1), dim3(256), lds_d, s>>>(A);
            kv_fc2_kernel<T><<<dim3(F * B, (E + r2 - 1) / r2), dim3(256), lds_e, s>>>(A);
            kv_proj_kernel<KE, T><<<dim3(F * B), dim3(512), lds_p, s>>>(A);
        }
    }
#undef KV_SELF
#undef KV_CROSS
    return 0;
}

static unsigned long long* g_kv_stamps = nullptr;   // sea_kv_debug_stamps

// Words of the granule arena the persistent form needs (SeaKvGlobal.handoff; the first F * D words are the tails' hand-off of the seven-launch form).
static int64_t persist_words(const SeaKvGlobal& G) {
    const int64_t F = G.F, E = G.E, D = G.D, S = G.S, H = G.H, P = F * (F - 1);
    return F * D * 2 + F * E * 5 + P * D * 2 + P * H * 2 + F * S;
}

// The persistent form applies to: one trajectory, one layer, the fixed-width kernels (pre_width) with cross heads of 8 or 16 columns, F >= 1, and a
// role count that fits the chip (one 512-thread workgroup per CU is always resident).  SEA_TUNE=kv_persist=0 keeps the seven launches per step.
template <int KE, typename T>
static bool run_persistent(const SeaKvGlobal& G, const SeaKvLayer* layers, int pos0, int n_steps, uint32_t tag0, hipStream_t s) {
    const int env = sea_tune("kv_persist", 1);   // read per call: tests compare the two forms in one process
    if constexpr (KE == 0) return false;
    else {
        if (!env || G.B != 1 || G.L != 1 || n_steps < 1 || G.handoff_words < persist_words(G) || G.S > 4096) return false;
        const int F = G.F, E = G.E, D = G.D, S = G.S, H = G.H, ex = G.exchange;
        const int hd_s = E / H;
        if (hd_s != 16 && hd_s != 32) return false;
        // the CU count of the device this call runs on (one process per GPU: rank r's current device is not device 0)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
        static int cu_of[64] = {0};
        if (cu_of[dev] == 0) {
            int n = 0;
            cu_of[dev] = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : -1;
        }
        const int cus = cu_of[dev];
        const int fixed = F * H + F + (ex ? F * (F - 1) * H + F : 0) + F;
        const int room = (cus - fixed) / F;
        if (room < 1) return false;
        int r1 = ((S + room - 1) / room + 15) / 16 * 16;
        const int n_fcf = (S + r1 - 1) / r1;
        const int r2 = (E + n_fcf - 1) / n_fcf;
        if (r1 > 128 || r2 > 8 || fixed + F * n_fcf > cus) return false;
        KvPersist A;
        A.L = layers[0];
        A.G = G;
        A.pos0 = pos0; A.n_steps = n_steps; A.tag0 = tag0; A.n_fcf = n_fcf; A.r1 = r1; A.r2 = r2;
        const int64_t P = (int64_t)F * (F - 1);
        unsigned long long* g = G.handoff;
        A.gnew = g; g += F * D;
        A.gnd = g; g += F * D;
        A.gx = g; g += F * E;
        A.gatt = g; g += F * E;
        A.gxr = g; g += F * E;
        A.gxr2 = g; g += F * E;
        A.gx3 = g; g += F * E;
        A.goc = g; g += P * D;
        A.gqc = g; g += P * D;
        A.gml = g; g += P * H * 2;
        A.gh = g;
        A.stamps = g_kv_stamps;
        const int hd_c = hd_s / 2, cap = G.cap;
        int lds = (3 * E + 4 * hd_s + 32 + 8 * hd_s + cap + 8);
        const int lds_c = (4 * D + 4 * hd_c + 32 + 8 * hd_c + cap + 8), lds_t = 9 * D + 3 * E + 32, lds_f = 3 * E + S + 128 + 32, lds_b = 5 * E + 32;
        lds = lds > lds_c ? lds : lds_c;
        lds = lds > lds_t ? lds : lds_t;
        lds = lds > lds_f ? lds : lds_f;
        lds = lds > lds_b ? lds : lds_b;
        // Every workgroup of this launch waits for others: all of them must be resident at once.  The grid is at most one 512-thread workgroup per CU;
        // the launch is COOPERATIVE, so that the runtime checks the grid against the kernel's occupancy (registers, this LDS request) on this device and
        // refuses it (hipErrorCooperativeLaunchTooLarge -> the seven-launch form) instead of starting a grid that can never finish its first hand-off.  What
        // no launch-time check can see — another process or stream holding CUs — is covered by the bounded spins and the error word (kv_engine.py).
        const dim3 grid(fixed + F * n_fcf), block(512);
        const void* fn = hd_s == 32 ? reinterpret_cast<const void*>(&kv_persistent_kernel<KE, T, 32, 16>) : reinterpret_cast<const void*>(&kv_persistent_kernel<KE, T, 16, 8>);
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, (size_t)lds * 4) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            return false;
        }
        void* args[] = {&A};
        // Under a rocprofiler-sdk tool (rocprofv3) the launch is a PLAIN one behind the same residency check done by hand (grid <= CUs x resident workgroups
        // per CU): a process that has used the cooperative queue dies with SIGSEGV inside exit() when the tool is loaded — resolved in round 4 against the
        // process map (profiles/failures/r04_rocprof_kv_exit_stack.txt): libc exit -> libamdhip64's exit handler (+0x3b8b8a) -> hsa_shut_down
        // (libhsa-runtime64.so.1.18 +0x60080) -> the agent's queue teardown (+0x6359e), i.e. ROCr destroying the cooperative (GWS) queue under the tool's
        // queue interception, after the tool has finalised.  Nothing of this library is on that stack (it keeps no HIP object alive); the same run with
        // SEA_TUNE=kv_persist=0 exits 0.  SEA_TUNE=kv_coop=0 / 1 forces the plain / cooperative launch.
        static const int coop = []() {
            const int forced = sea_tune("kv_coop", -1);
            if (forced >= 0) return forced;
            const char* tool = getenv("ROCP_TOOL_LIBRARIES");
            const char* pre = getenv("LD_PRELOAD");
            const bool profiled = (tool != nullptr && tool[0] != 0) || (pre != nullptr && strstr(pre, "rocprofiler") != nullptr);
            return profiled ? 0 : 1;
        }();
        if (!coop) {
            if ((int)grid.x > cus * per_cu) return false;
            if (hipLaunchKernel(fn, grid, block, args, (size_t)(lds * 4), s) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            return true;
        }
        if (hipLaunchCooperativeKernel(fn, grid, block, args, (unsigned)(lds * 4), s) != hipSuccess) {
            (void)hipGetLastError();   // refused (too large for this device / a device without cooperative launch): the seven launches per step
            return false;
        }
        return true;
    }
}

template <typename T>
static int run_steps_t(const SeaKvGlobal& G, const SeaKvLayer* layers, int pos0, int n_steps, uint32_t tag0, hipStream_t s) {
    switch (pre_width(G)) {
        case 64: return run_persistent<64, T>(G, layers, pos0, n_steps, tag0, s) ? 0 : run_steps<64, T>(G, layers, pos0, n_steps, tag0, s);
        case 128: return run_persistent<128, T>(G, layers, pos0, n_steps, tag0, s) ? 0 : run_steps<128, T>(G, layers, pos0, n_steps, tag0, s);
        case 256: return run_persistent<256, T>(G, layers, pos0, n_steps, tag0, s) ? 0 : run_steps<256, T>(G, layers, pos0, n_steps, tag0, s);
        default: return run_steps<0, T>(G, layers, pos0, n_steps, tag0, s);
    }
}

}  // namespace

// Tuning aid: a device buffer of n_steps * 64 8-byte words receives time stamps of the persistent form's hand-offs (NULL: off).  Not part of the ABI proper.
extern "C" void sea_kv_debug_stamps(unsigned long long* buf) { g_kv_stamps = buf; }

extern "C" int64_t sea_kv_arena_words(const SeaKvGlobal* G) { return G != nullptr ? persist_words(*G) : 0; }

extern "C" int sea_kv_rollout(const SeaKvGlobal* G, const SeaKvLayer* layers, int pos0, int n_steps, uint32_t tag0, int dtype, void* stream) {
    SEA_REQUIRE(G != nullptr && layers != nullptr && n_steps >= 0 && pos0 >= 0, "sea_kv_rollout: bad arguments");
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_kv_rollout: bad dtype %d", dtype);
    const int epc = dtype == SEA_BF16 ? 8 : 4;
    const int F = G->F, E = G->E, D = G->D, S = G->S, H = G->H, B = G->B;
    SEA_REQUIRE(F >= 1 && F <= SEA_KV_MAX_FIELDS && B >= 1 && B <= 64 && H >= 1 && G->L >= 1 && G->cap >= pos0 + n_steps && G->cap <= 8192,
                "sea_kv_rollout: bad sizes (F=%d B=%d H=%d L=%d cap=%d pos0=%d n_steps=%d)", F, B, H, G->L, G->cap, pos0, n_steps);
    SEA_REQUIRE(E % H == 0 && E <= 512 && kdim_ok(E, epc) && S <= 4096 && kdim_ok(S, epc) && S % 4 == 0, "sea_kv_rollout: unsupported widths E=%d S=%d", E, S);
    const int hd_s = E / H;
    SEA_REQUIRE(hd_s == 8 || hd_s == 16 || hd_s == 32 || hd_s == 64, "sea_kv_rollout: self head dim %d (8 / 16 / 32 / 64)", hd_s);
    if (G->exchange) {
        SEA_REQUIRE(F >= 2 && D % H == 0 && D <= 512 && kdim_ok(D, epc), "sea_kv_rollout: unsupported exchange width D=%d (F=%d)", D, F);
        const int hd_c = D / H;
        SEA_REQUIRE(hd_c == 8 || hd_c == 16 || hd_c == 32 || hd_c == 64, "sea_kv_rollout: cross head dim %d (8 / 16 / 32 / 64)", hd_c);
        SEA_REQUIRE(G->nd_old && G->oc && G->qc && G->ml && G->handoff && G->handoff_words >= (int64_t)B * F * D && G->rope_cross, "sea_kv_rollout: null / short exchange workspace");
    }
    SEA_REQUIRE(G->traj && G->att_e && G->xr && G->xq && G->x3 && G->hbuf && G->err && G->rope_self && (G->L == 1 || (G->xl[0] && G->xl[1])), "sea_kv_rollout: null workspace");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == SEA_BF16) run_steps_t<__bf16>(*G, layers, pos0, n_steps, tag0, s);
    else run_steps_t<float>(*G, layers, pos0, n_steps, tag0, s);
    SEA_CHECK_LAUNCH("sea_kv_rollout");
    return SEA_OK;
}
