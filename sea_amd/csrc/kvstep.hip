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

// ------------------------------------------------------------------------------------------------ workgroup reductions
// red: LDS, >= 32 floats.  Every thread of the workgroup calls; contains two barriers.
__device__ __forceinline__ float wg_sum(float v, float* red, int tid, int nthreads) {
    v = wave_sum(v);
    __syncthreads();                     // red may still be read from a previous call
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < (nthreads >> 6); ++w) s += red[w];
    return s;
}
__device__ __forceinline__ float wg_max(float v, float* red, int tid, int nthreads) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
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
template <typename T>
__device__ __forceinline__ void wg_norm(const float* xs, float* ys, int d, const SeaKvNorm& nm, int64_t modrow, bool gelu, float* red, int tid, int nthreads) {
    float s = 0.f;
    for (int i = tid; i < d; i += nthreads) s += xs[i];
    const float mean = wg_sum(s, red, tid, nthreads) / (float)d;
    float q = 0.f;
    for (int i = tid; i < d; i += nthreads) {
        const float c = xs[i] - mean;
        q = fma1(c, c, q);
    }
    const float rstd = 1.0f / sqrtf(wg_sum(q, red, tid, nthreads) / (float)d + KV_EPS);
    const T* mod = nm.mod != nullptr ? static_cast<const T*>(nm.mod) + modrow * nm.ldmod : nullptr;
    for (int i = tid; i < d; i += nthreads) {
        float gq = nm.gamma[i], bq = nm.beta != nullptr ? nm.beta[i] : 0.f;
        if (mod != nullptr) {
            gq += 1.0f + to_f32(mod[i]);
            bq += to_f32(mod[d + i]);
        }
        float o = (xs[i] - mean) * rstd * gq + bq;
        if (gelu) o = gelu_erf(o);
        ys[i] = o;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ GEMV
// ys[r] = sum_k W[rowmap(r), k] * xs[k], r < nrows; xs, ys in LDS (fp32).  A team of TL lanes owns a row: lane tl of the team holds the 16-byte
// chunks tl, tl + TL, ... of it (CPL per lane; CPL == 1: K / EPC <= 64 chunks, several teams per wave); 8 (or CPL) loads per lane are in flight.
// The caller puts a barrier between this and the first read of ys.
template <typename T, int CPL, typename RowMap>
__device__ __forceinline__ void wg_gemv_c(const T* __restrict__ W, int ldw, int K, int nrows, const float* xs, float* ys, RowMap rowmap, int tid, int nthreads) {
    constexpr int EPC = ActTraits<T>::EPC;
    constexpr int RB = CPL >= 8 ? 1 : 8 / CPL;
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    const int kc = K / EPC;
    const int TL = CPL == 1 ? kc : 64;            // power of two <= 64
    const int tpw = 64 / TL;
    const int team = lane / TL, tl = lane - team * TL;
    float xv[CPL][EPC];
#pragma unroll
    for (int t = 0; t < CPL; ++t)
#pragma unroll
        for (int e = 0; e < EPC; ++e) xv[t][e] = xs[(tl + t * TL) * EPC + e];
    const int rpp = nw * tpw;                     // rows per pass of the workgroup
    for (int r0 = 0; r0 < nrows; r0 += rpp * RB) {
        uint4 w[RB][CPL];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            int row = r0 + u * rpp + wave * tpw + team;
            row = row < nrows ? row : nrows - 1;
            const T* wr = W + (int64_t)rowmap(row) * ldw;
#pragma unroll
            for (int t = 0; t < CPL; ++t) w[u][t] = *reinterpret_cast<const uint4*>(wr + (tl + t * TL) * EPC);
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int row = r0 + u * rpp + wave * tpw + team;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                float wv[EPC];
                unpack_w<T>(w[u][t], wv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc = fma1(wv[e], xv[t][e], acc);
            }
            for (int o = TL >> 1; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
            if (tl == 0 && row < nrows) ys[row] = acc;
        }
    }
}

template <typename T, typename RowMap>
__device__ __forceinline__ void wg_gemv(const T* __restrict__ W, int ldw, int K, int nrows, const float* xs, float* ys, RowMap rowmap, int tid, int nthreads) {
    const int kc = K / ActTraits<T>::EPC;
    if (kc <= 64) wg_gemv_c<T, 1>(W, ldw, K, nrows, xs, ys, rowmap, tid, nthreads);
    else if (kc == 128) wg_gemv_c<T, 2>(W, ldw, K, nrows, xs, ys, rowmap, tid, nthreads);
    else if (kc == 256) wg_gemv_c<T, 4>(W, ldw, K, nrows, xs, ys, rowmap, tid, nthreads);
    else wg_gemv_c<T, 8>(W, ldw, K, nrows, xs, ys, rowmap, tid, nthreads);   // kc == 512 (the host checks)
}

struct IdentityRow {
    __device__ __forceinline__ int operator()(int r) const { return r; }
};

// ------------------------------------------------------------------------------------------------ one query against a row-major cache
// Workgroup-wide softmax(q . K^T) V over keys 0 .. nk_cached-1 of the cache (K, V: [cap, HD] rows of T) plus — has_cur — one more key held in
// LDS (kcur / vcur, fp32).  Returns through LDS: oacc[HD] = sum_k exp(s_k - m) v_k, and (m, l) to every thread.  prob: LDS [>= nk_cached + 1].
// q (LDS) is pre-scaled.  nk_cached + has_cur may be 0: m = -inf, l = 0, oacc = 0.
template <typename T, int HD>
__device__ __forceinline__ void wg_attend(const T* __restrict__ Kg, const T* __restrict__ Vg, int nk_cached, bool has_cur, const float* q_l, const float* kcur, const float* vcur,
                                          float* prob, float* part /* [nw][HD] */, float* oacc, float* red, float& m_out, float& l_out, int tid, int nthreads) {
    constexpr int EPC = ActTraits<T>::EPC;
    constexpr int CPK = HD / EPC;                 // 16-byte chunks per key row
    constexpr float LOG2E = 1.4426950408889634f;
    float q[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) q[c] = q_l[c];
    const int nk = nk_cached + (has_cur ? 1 : 0);
    // ---- scores: thread t owns keys t, t + nthreads, ...; the rows of KB keys requested together (8 loads in flight per lane)
    constexpr int KB = CPK >= 8 ? 1 : (CPK >= 4 ? 2 : 4);
    float mx = -INFINITY;
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
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CPK; ++c) {
                float kv[EPC];
                unpack_w<T>(raw[j][c], kv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc = fma1(q[c * EPC + e], kv[e], acc);
            }
            if (key < nk_cached) {
                prob[key] = acc;
                mx = fmaxf(mx, acc);
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
    const float m = wg_max(mx, red, tid, nthreads);   // barriers inside: prob is complete
    float ls = 0.f;
    for (int key = tid; key < nk; key += nthreads) {
        const float pv = __builtin_amdgcn_exp2f((prob[key] - m) * LOG2E);
        prob[key] = pv;
        ls += pv;
    }
    const float l = wg_sum(ls, red, tid, nthreads);   // barriers inside: prob holds the probabilities
    // ---- o = sum_k p_k v_k: thread (slot, chunk) walks keys slot, slot + nthreads / CPK, ... of its 16-byte column chunk
    const int ch = tid % CPK, slot = tid / CPK, nslot = nthreads / CPK;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
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
        for (int j = 0; j < 4; ++j) {
            float vv[EPC];
            unpack_w<T>(raw[j], vv);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = fma1(pv[j], vv[e], acc[e]);
        }
    }
    if (has_cur && slot == 0) {
        const float pv = prob[nk_cached];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = fma1(pv, vcur[ch * EPC + e], acc[e]);
    }
    // lanes of a wave with the same chunk: lane % CPK (CPK divides 64)
#pragma unroll
    for (int e = 0; e < EPC; ++e)
        for (int o = 32; o >= CPK; o >>= 1) acc[e] += __shfl_xor(acc[e], o);
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
// k and v are rounded to the cache dtype (later steps read them back from the cache) and appended at `pos`.
template <typename T, int HD>
__device__ __forceinline__ void head_finish(float* qkv, const float* bq, const float* bk, const float* bv, const float* rope, int pos, bool with_kv, T* Krow, T* Vrow, int tid) {
    constexpr int HD2 = HD / 2;
    const float2* cs = reinterpret_cast<const float2*>(rope) + (int64_t)pos * HD2;
    const float scale = 1.0f / sqrtf((float)HD);
    if (tid < HD2) {
        const float2 c = cs[tid];
        float oe, oo;
        rope_pair(qkv[2 * tid] + bq[2 * tid], qkv[2 * tid + 1] + bq[2 * tid + 1], c.x, c.y, oe, oo);
        qkv[2 * tid] = oe * scale;
        qkv[2 * tid + 1] = oo * scale;
    } else if (with_kv && tid < 2 * HD2) {
        const int t = tid - HD2;
        const float2 c = cs[t];
        float oe, oo;
        rope_pair(qkv[HD + 2 * t] + bk[2 * t], qkv[HD + 2 * t + 1] + bk[2 * t + 1], c.x, c.y, oe, oo);
        oe = round_to(oe, T());
        oo = round_to(oo, T());
        qkv[HD + 2 * t] = oe;
        qkv[HD + 2 * t + 1] = oo;
        Krow[2 * t] = from_f32<T>(oe);
        Krow[2 * t + 1] = from_f32<T>(oo);
    } else if (with_kv && tid >= 64 && tid < 64 + HD) {
        const int t = tid - 64;
        const float v = round_to(qkv[2 * HD + t] + bv[t], T());
        qkv[2 * HD + t] = v;
        Vrow[t] = from_f32<T>(v);
    }
    __syncthreads();
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

__device__ __forceinline__ int pair_index(int i, int j, int F) { return i * (F - 1) + (j < i ? j : j - 1); }

// ------------------------------------------------------------------------------------------------ A: self attention (models/temporal.py:127-136 up to the projection)
// grid F * B * H, block 512.  LDS: xs[E] ns[E] qkv[3 HD] oacc[HD] red[32] part[8 HD] prob[cap + 8]
template <typename T, int HD>
__global__ __launch_bounds__(512) void kv_self_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = A.G.E, H = A.G.H, B = A.G.B, F = A.G.F, cap = A.G.cap, pos = A.pos;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int h = blockIdx.x % H, ib_ = blockIdx.x / H, b = ib_ % B, i = ib_ / B;
    float* xs = sm;
    float* ns = xs + E;
    float* qkv = ns + E;
    float* oacc = qkv + 3 * HD;
    float* red = oacc + HD;
    float* part = red + 32;
    float* prob = part + 8 * HD;
    const SeaKvField& Fd = A.L.f[i];
    const int64_t crow = (int64_t)pos * B + b;
    const float* x = A.xin + ((int64_t)b * F + i) * E;
    const float* ibp = (A.L.ib != nullptr && !A.G.ib_after_cross) ? A.L.ib + crow * E : nullptr;
    for (int e = tid; e < E; e += nth) xs[e] = x[e] + (ibp != nullptr ? ibp[e] : 0.f);
    __syncthreads();
    wg_norm<T>(xs, ns, E, Fd.ln0, crow, false, red, tid, nth);
    const int hh = h;
    auto rowmap = [=](int r) { return (r / HD) * E + hh * HD + (r % HD); };
    wg_gemv<T>(static_cast<const T*>(Fd.Wqkv), E, E, 3 * HD, ns, qkv, rowmap, tid, nth);
    __syncthreads();
    const int64_t bh = (int64_t)b * H + h;
    T* Kc = static_cast<T*>(Fd.Ks) + bh * cap * HD;
    T* Vc = static_cast<T*>(Fd.Vs) + bh * cap * HD;
    head_finish<T, HD>(qkv, Fd.bqkv + h * HD, Fd.bqkv + E + h * HD, Fd.bqkv + 2 * E + h * HD, A.G.rope_self, pos, true, Kc + (int64_t)pos * HD, Vc + (int64_t)pos * HD, tid);
    float m, l;
    wg_attend<T, HD>(Kc, Vc, pos, true, qkv, qkv + HD, qkv + 2 * HD, prob, part, oacc, red, m, l, tid, nth);
    if (tid < HD) A.G.att_e[((int64_t)b * F + i) * E + h * HD + tid] = oacc[tid] / l;
}

// ------------------------------------------------------------------------------------------------ B: out-projection + residual, down-projection + ln_cross (models/temporal.py:136, 177-178)
// grid F * B, block 512.  LDS: att[E] xs[E] y[E] red[32]
template <typename T>
__global__ __launch_bounds__(512) void kv_oproj_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = A.G.E, D = A.G.D, B = A.G.B, F = A.G.F;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int b = blockIdx.x % B, i = blockIdx.x / B;
    float* att = sm;
    float* xs = att + E;
    float* y = xs + E;
    float* red = y + E;
    const SeaKvField& Fd = A.L.f[i];
    const int64_t crow = (int64_t)A.pos * B + b, ro = ((int64_t)b * F + i);
    const float* ibp = (A.L.ib != nullptr && !A.G.ib_after_cross) ? A.L.ib + crow * E : nullptr;
    for (int e = tid; e < E; e += nth) {
        att[e] = A.G.att_e[ro * E + e];
        xs[e] = A.xin[ro * E + e] + (ibp != nullptr ? ibp[e] : 0.f);
    }
    __syncthreads();
    wg_gemv<T>(static_cast<const T*>(Fd.Wo), E, E, E, att, y, IdentityRow(), tid, nth);
    __syncthreads();
    for (int e = tid; e < E; e += nth) {
        const float v = xs[e] + y[e];
        xs[e] = v;
        A.G.xr[ro * E + e] = v;
    }
    __syncthreads();
    if (!A.G.exchange) return;
    wg_gemv<T>(static_cast<const T*>(Fd.Wdown), E, E, D, xs, y, IdentityRow(), tid, nth);
    __syncthreads();
    for (int e = tid; e < D; e += nth) y[e] += Fd.bdown[e];
    __syncthreads();
    wg_norm<T>(y, y, D, Fd.ln_cross, crow, false, red, tid, nth);
    for (int e = tid; e < D; e += nth) A.G.nd_old[ro * D + e] = y[e];
}

// ------------------------------------------------------------------------------------------------ C: cross attention of every pair (models/temporal.py:181-186; base_blocks.py:232-293)
// grid F (F-1) * B * H, block 512.  Pair (i, j): query from nd_old_i; source j > i: key / value of this position from nd_old_j, appended, the head's
// output is final; source j < i (updated earlier in this sweep): only the cached keys here — (o, m, l) and q go to the tail, which merges this
// position's key.  LDS: ni[D] nj[D] qkv[3 HD] oacc[HD] red[32] part[8 HD] prob[cap + 8]
template <typename T, int HD>
__global__ __launch_bounds__(512) void kv_cross_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = A.G.D, H = A.G.H, B = A.G.B, F = A.G.F, cap = A.G.cap, pos = A.pos;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int h = blockIdx.x % H, pb = blockIdx.x / H, b = pb % B, p = pb / B;
    const int i = p / (F - 1), s = p % (F - 1), j = s < i ? s : s + 1;
    const bool old_src = j > i;
    float* ni = sm;
    float* nj = ni + D;
    float* qkv = nj + D;
    float* oacc = qkv + 3 * HD;
    float* red = oacc + HD;
    float* part = red + 32;
    float* prob = part + 8 * HD;
    const SeaKvPair& P = A.L.p[i][j];
    for (int e = tid; e < D; e += nth) {
        ni[e] = A.G.nd_old[((int64_t)b * F + i) * D + e];
        nj[e] = A.G.nd_old[((int64_t)b * F + j) * D + e];
    }
    __syncthreads();
    const int hh = h;
    auto qmap = [=](int r) { return hh * HD + r; };
    wg_gemv<T>(static_cast<const T*>(P.Wq), D, D, HD, ni, qkv, qmap, tid, nth);
    if (old_src) {
        auto kvmap = [=](int r) { return (r / HD) * D + hh * HD + (r % HD); };
        wg_gemv<T>(static_cast<const T*>(P.Wkv), D, D, 2 * HD, nj, qkv + HD, kvmap, tid, nth);
    }
    __syncthreads();
    const int64_t bh = (int64_t)b * H + h;
    T* Kc = static_cast<T*>(P.Kc) + bh * cap * HD;
    T* Vc = static_cast<T*>(P.Vc) + bh * cap * HD;
    head_finish<T, HD>(qkv, P.bq + h * HD, P.bkv + h * HD, P.bkv + D + h * HD, A.G.rope_cross, pos, old_src, Kc + (int64_t)pos * HD, Vc + (int64_t)pos * HD, tid);
    float m, l;
    wg_attend<T, HD>(Kc, Vc, pos, old_src, qkv, qkv + HD, qkv + 2 * HD, prob, part, oacc, red, m, l, tid, nth);
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
// LDS: nj[D] kv[2 D] o[D] g[D] gs[D] xs[E] y[E] red[32]
template <typename T>
__global__ __launch_bounds__(512) void kv_tail_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = A.G.E, D = A.G.D, H = A.G.H, B = A.G.B, F = A.G.F, cap = A.G.cap, pos = A.pos;
    const int hd = D / H, hd2 = hd >> 1;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int b = blockIdx.x % B, i = blockIdx.x / B;
    float* nj = sm;
    float* kv = nj + D;
    float* o = kv + 2 * D;
    float* g = o + D;
    float* gs = g + D;
    float* xs = gs + D;
    float* y = xs + E;
    float* red = y + E;
    const SeaKvField& Fd = A.L.f[i];
    const int64_t crow = (int64_t)pos * B + b, ro = (int64_t)b * F + i;
    for (int e = tid; e < D; e += nth) gs[e] = 0.f;
    for (int e = tid; e < E; e += nth) xs[e] = A.G.xr[ro * E + e];
    // old-source pairs first: nothing to wait for
    for (int pass = 0; pass < 2; ++pass) {
        for (int j = 0; j < F; ++j) {
            if (j == i || (pass == 0) != (j > i)) continue;   // block-uniform
            const int p = pair_index(i, j, F);
            const SeaKvPair& P = A.L.p[i][j];
            const int64_t po = ((int64_t)p * B + b) * D;
            __syncthreads();
            if (j > i) {
                for (int e = tid; e < D; e += nth) o[e] = A.G.oc[po + e];
            } else {
                // the source field's rows of THIS sweep: granules {value, tag} published by workgroup (j, b)
                const unsigned long long* hg = A.G.handoff + ((int64_t)b * F + j) * D;
                for (int e = tid; e < D; e += nth) {
                    float v = 0.f;
                    int it = 0;
                    for (; it < KV_SPIN_LIMIT; ++it) {
                        const unsigned long long pk = __hip_atomic_load(hg + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((uint32_t)(pk >> 32) == A.tag) {
                            v = __builtin_bit_cast(float, (uint32_t)pk);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    if (it == KV_SPIN_LIMIT) __hip_atomic_store(A.G.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    nj[e] = v;
                }
                __syncthreads();
                wg_gemv<T>(static_cast<const T*>(P.Wkv), D, D, 2 * D, nj, kv, IdentityRow(), tid, nth);
                __syncthreads();
                // bias, rotary embedding of k, append, merge this key into the head's (o, m, l)
                if (tid < D / 2) {
                    const int hh = tid / hd2, t = tid - hh * hd2;
                    const float2 c = reinterpret_cast<const float2*>(A.G.rope_cross)[(int64_t)pos * hd2 + t];
                    const int c0 = hh * hd + 2 * t;
                    float oe, oo;
                    rope_pair(kv[c0] + P.bkv[c0], kv[c0 + 1] + P.bkv[c0 + 1], c.x, c.y, oe, oo);
                    oe = round_to(oe, T());
                    oo = round_to(oo, T());
                    kv[c0] = oe;
                    kv[c0 + 1] = oo;
                    T* Kr = static_cast<T*>(P.Kc) + (((int64_t)b * H + hh) * cap + pos) * hd;
                    Kr[2 * t] = from_f32<T>(oe);
                    Kr[2 * t + 1] = from_f32<T>(oo);
                }
                for (int e = tid; e < D; e += nth) {
                    const float v = round_to(kv[D + e] + P.bkv[D + e], T());
                    kv[D + e] = v;
                    const int hh = e / hd;
                    static_cast<T*>(P.Vc)[(((int64_t)b * H + hh) * cap + pos) * hd + (e - hh * hd)] = from_f32<T>(v);
                }
                __syncthreads();
                for (int e = tid; e < D; e += nth) {
                    const int hh = e / hd;
                    float sc = 0.f;
                    for (int c = 0; c < hd; ++c) sc = fma1(A.G.qc[po + hh * hd + c], kv[hh * hd + c], sc);
                    const float* ml = A.G.ml + (((int64_t)p * B + b) * H + hh) * 2;
                    const float m0 = ml[0], l0 = ml[1];
                    const float m1 = fmaxf(m0, sc);
                    const float w0 = l0 > 0.f ? __expf(m0 - m1) : 0.f, w1 = __expf(sc - m1);
                    o[e] = (A.G.oc[po + e] * w0 + kv[D + e] * w1) / (l0 * w0 + w1);
                }
            }
            __syncthreads();
            wg_gemv<T>(static_cast<const T*>(P.Wp), D, D, D, o, g, IdentityRow(), tid, nth);
            __syncthreads();
            for (int e = tid; e < D; e += nth) gs[e] += gelu_erf(g[e]);
        }
    }
    __syncthreads();
    wg_gemv<T>(static_cast<const T*>(Fd.Wup), D, D, E, gs, y, IdentityRow(), tid, nth);
    __syncthreads();
    const float bs = (float)(F - 1);
    for (int e = tid; e < E; e += nth) {
        const float v = xs[e] + y[e] + bs * Fd.bup[e];
        xs[e] = v;
        A.G.xr[ro * E + e] = v;
    }
    __syncthreads();
    if (i == F - 1) return;
    wg_gemv<T>(static_cast<const T*>(Fd.Wdown), E, E, D, xs, y, IdentityRow(), tid, nth);
    __syncthreads();
    for (int e = tid; e < D; e += nth) y[e] += Fd.bdown[e];
    __syncthreads();
    wg_norm<T>(y, y, D, Fd.ln_cross, crow, false, red, tid, nth);
    unsigned long long* hg = A.G.handoff + ((int64_t)b * F + i) * D;
    for (int e = tid; e < D; e += nth)
        __hip_atomic_store(hg + e, ((unsigned long long)A.tag << 32) | (unsigned long long)__builtin_bit_cast(uint32_t, y[e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------ D: info-bottleneck add + AdaLN_2 + fc1 (models/temporal.py:139-145; base_blocks.py:22)
// grid (F * B, S / rows_per_wg), block 256.  LDS: xs[E] ns[E] hs[rows] red[32]
template <typename T>
__global__ __launch_bounds__(256) void kv_fc1_kernel(const KvArgs A, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = A.G.E, S = A.G.S, B = A.G.B, F = A.G.F;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int b = blockIdx.x % B, i = blockIdx.x / B, r0 = blockIdx.y * rows_per_wg;
    float* xs = sm;
    float* ns = xs + E;
    float* hs = ns + E;
    float* red = hs + rows_per_wg;
    const SeaKvField& Fd = A.L.f[i];
    const int64_t crow = (int64_t)A.pos * B + b, ro = (int64_t)b * F + i;
    const float* ibp = (A.L.ib != nullptr && A.G.ib_after_cross) ? A.L.ib + crow * E : nullptr;
    for (int e = tid; e < E; e += nth) {
        const float v = A.G.xr[ro * E + e] + (ibp != nullptr ? ibp[e] : 0.f);
        xs[e] = v;
        if (blockIdx.y == 0) A.G.xq[ro * E + e] = v;
    }
    __syncthreads();
    wg_norm<T>(xs, ns, E, Fd.ln2, crow, false, red, tid, nth);
    const int nr = r0 + rows_per_wg <= S ? rows_per_wg : S - r0;
    wg_gemv<T>(static_cast<const T*>(Fd.W1) + (int64_t)r0 * E, E, E, nr, ns, hs, IdentityRow(), tid, nth);
    __syncthreads();
    for (int e = tid; e < nr; e += nth) A.G.hbuf[ro * S + r0 + e] = hs[e] + Fd.b1[r0 + e];
}

// ------------------------------------------------------------------------------------------------ E: nn.LayerNorm(S) + GELU + fc2 + residual (base_blocks.py:23-25; temporal.py:145)
// grid (F * B, E / rows_per_wg), block 256.  LDS: hs[S] ys[rows] red[32]
template <typename T>
__global__ __launch_bounds__(256) void kv_fc2_kernel(const KvArgs A, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = A.G.E, S = A.G.S, B = A.G.B, F = A.G.F;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int b = blockIdx.x % B, i = blockIdx.x / B, r0 = blockIdx.y * rows_per_wg;
    float* hs = sm;
    float* ys = hs + S;
    float* red = ys + rows_per_wg;
    const SeaKvField& Fd = A.L.f[i];
    const int64_t ro = (int64_t)b * F + i;
    for (int e = tid * 4; e < S; e += nth * 4) *reinterpret_cast<float4*>(hs + e) = *reinterpret_cast<const float4*>(A.G.hbuf + ro * S + e);
    __syncthreads();
    SeaKvNorm nm;
    nm.gamma = Fd.lnw; nm.beta = Fd.lnb; nm.mod = nullptr; nm.ldmod = 0; nm.pad_ = 0;
    wg_norm<T>(hs, hs, S, nm, 0, true, red, tid, nth);
    const int nr = r0 + rows_per_wg <= E ? rows_per_wg : E - r0;
    wg_gemv<T>(static_cast<const T*>(Fd.W2) + (int64_t)r0 * S, S, S, nr, hs, ys, IdentityRow(), tid, nth);
    __syncthreads();
    for (int e = tid; e < nr; e += nth) A.G.x3[ro * E + r0 + e] = ys[e] + Fd.b2[r0 + e] + A.G.xq[ro * E + r0 + e];
}

// ------------------------------------------------------------------------------------------------ Fin: proj (+ the model's final norm after the last layer) (temporal.py:146, 412-415)
// grid F * B, block 512.  LDS: xs[E] y[E] red[32]
template <typename T>
__global__ __launch_bounds__(512) void kv_proj_kernel(const KvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = A.G.E, B = A.G.B, F = A.G.F;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int b = blockIdx.x % B, i = blockIdx.x / B;
    float* xs = sm;
    float* y = xs + E;
    float* red = y + E;
    const SeaKvField& Fd = A.L.f[i];
    const int64_t crow = (int64_t)A.pos * B + b, ro = (int64_t)b * F + i;
    for (int e = tid; e < E; e += nth) xs[e] = A.G.x3[ro * E + e];
    __syncthreads();
    wg_gemv<T>(static_cast<const T*>(Fd.Wproj), E, E, E, xs, y, IdentityRow(), tid, nth);
    __syncthreads();
    for (int e = tid; e < E; e += nth) y[e] += Fd.bproj[e];
    __syncthreads();
    if (A.last_layer) wg_norm<T>(y, y, E, A.G.final_ln[i], crow, false, red, tid, nth);
    for (int e = tid; e < E; e += nth) A.xout[ro * E + e] = y[e];
}

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static bool kdim_ok(int K, int epc) {
    if (K % epc) return false;
    const int kc = K / epc;
    return (kc <= 64 && pow2(kc) && kc >= 2) || kc == 128 || kc == 256 || kc == 512;
}

template <typename T>
static int run_steps(const SeaKvGlobal& G, const SeaKvLayer* layers, int pos0, int n_steps, uint32_t tag0, hipStream_t s) {
    const int F = G.F, E = G.E, D = G.D, S = G.S, H = G.H, B = G.B, Ln = G.L, cap = G.cap;
    const int hd_s = E / H, hd_c = G.exchange ? D / H : 0;
    const int lds_attn_s = (2 * E + 3 * hd_s + hd_s + 32 + 8 * hd_s + cap + 8) * 4;
    const int lds_attn_c = G.exchange ? (2 * D + 3 * hd_c + hd_c + 32 + 8 * hd_c + cap + 8) * 4 : 0;
    const int lds_b = (3 * E + 32) * 4, lds_t = (6 * D + 2 * E + 32) * 4, lds_p = (2 * E + 32) * 4;
    const int r1 = 32, r2 = S >= 2048 ? 4 : 8;
    const int lds_d = (2 * E + r1 + 32) * 4, lds_e = (S + r2 + 32) * 4;
    const int64_t slab = (int64_t)B * F * E;
#define KV_SELF(HDV)                                                                                   \
    do {                                                                                               \
        kv_self_kernel<T, HDV><<<dim3(F * B * H), dim3(512), lds_attn_s, s>>>(A);                      \
    } while (0)
#define KV_CROSS(HDV)                                                                                  \
    do {                                                                                               \
        kv_cross_kernel<T, HDV><<<dim3(F * (F - 1) * B * H), dim3(512), lds_attn_c, s>>>(A);           \
    } while (0)
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
                default: KV_SELF(64); break;
            }
            kv_oproj_kernel<T><<<dim3(F * B), dim3(512), lds_b, s>>>(A);
            if (G.exchange) {
                switch (hd_c) {
                    case 8: KV_CROSS(8); break;
                    case 16: KV_CROSS(16); break;
                    case 32: KV_CROSS(32); break;
                    default: KV_CROSS(64); break;
                }
                kv_tail_kernel<T><<<dim3(F * B), dim3(512), lds_t, s>>>(A);
            }
            kv_fc1_kernel<T><<<dim3(F * B, (S + r1 - 1) / r1), dim3(256), lds_d, s>>>(A, r1);
            kv_fc2_kernel<T><<<dim3(F * B, (E + r2 - 1) / r2), dim3(256), lds_e, s>>>(A, r2);
            kv_proj_kernel<T><<<dim3(F * B), dim3(512), lds_p, s>>>(A);
        }
    }
#undef KV_SELF
#undef KV_CROSS
    return 0;
}

}  // namespace

extern "C" int sea_kv_rollout(const SeaKvGlobal* G, const SeaKvLayer* layers, int pos0, int n_steps, uint32_t tag0, int dtype, void* stream) {
    SEA_REQUIRE(G != nullptr && layers != nullptr && n_steps >= 0 && pos0 >= 0, "sea_kv_rollout: bad arguments");
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_kv_rollout: bad dtype %d", dtype);
    const int epc = dtype == SEA_BF16 ? 8 : 4;
    const int F = G->F, E = G->E, D = G->D, S = G->S, H = G->H, B = G->B;
    SEA_REQUIRE(F >= 1 && F <= SEA_KV_MAX_FIELDS && B >= 1 && B <= 64 && H >= 1 && G->L >= 1 && G->cap >= pos0 + n_steps && G->cap <= 8192,
                "sea_kv_rollout: bad sizes (F=%d B=%d H=%d L=%d cap=%d pos0=%d n_steps=%d)", F, B, H, G->L, G->cap, pos0, n_steps);
    SEA_REQUIRE(E % H == 0 && E <= 512 && kdim_ok(E, epc) && S <= 8192 && kdim_ok(S, epc) && S % 4 == 0, "sea_kv_rollout: unsupported widths E=%d S=%d", E, S);
    const int hd_s = E / H;
    SEA_REQUIRE(hd_s == 8 || hd_s == 16 || hd_s == 32 || hd_s == 64, "sea_kv_rollout: self head dim %d (8 / 16 / 32 / 64)", hd_s);
    if (G->exchange) {
        SEA_REQUIRE(F >= 2 && D % H == 0 && D <= 512 && kdim_ok(D, epc), "sea_kv_rollout: unsupported exchange width D=%d (F=%d)", D, F);
        const int hd_c = D / H;
        SEA_REQUIRE(hd_c == 8 || hd_c == 16 || hd_c == 32 || hd_c == 64, "sea_kv_rollout: cross head dim %d (8 / 16 / 32 / 64)", hd_c);
        SEA_REQUIRE(G->nd_old && G->oc && G->qc && G->ml && G->handoff && G->rope_cross, "sea_kv_rollout: null exchange workspace");
    }
    SEA_REQUIRE(G->traj && G->att_e && G->xr && G->xq && G->x3 && G->hbuf && G->err && G->rope_self && (G->L == 1 || (G->xl[0] && G->xl[1])), "sea_kv_rollout: null workspace");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == SEA_BF16) run_steps<__bf16>(*G, layers, pos0, n_steps, tag0, s);
    else run_steps<float>(*G, layers, pos0, n_steps, tag0, s);
    SEA_CHECK_LAUNCH("sea_kv_rollout");
    return SEA_OK;
}
