// Causal flash attention backward for SEA's self- and cross-attention (gfx950): sea_attention_bwd.
//
// Scores are in LOG2 units, as in the forward (Q carries hd^-1/2 * log2 e; LSE = ref + log2 l): P = 2^(S - LSE), with "- LSE" riding in the MFMA's C
// operand, so a probability is one v_exp_f32 and nothing else.  With s2 = s ln-domain / ln 2: dS2 = ln2 * P (dP - delta), hence dQ2 = ln2 * dS K and
// dK = ln2 * dS^T Q2; the epilogues carry the ln2 (dQ additionally q_scale, the factor the QKV epilogue put on q).
// Two kernels, both recomputing P = 2^(S - LSE) from Q, K and the forward's log-sum-exp instead of storing [T, T]:
//   attn_bwd_dq_kernel   one workgroup per 64 QUERIES (lane = query), loops over key tiles:      dQ, delta = rowsum(dO . O)
//   attn_bwd_dkv_kernel  one workgroup per 64 KEYS (lane = key), loops over query tiles:         dK, dV
// No atomics, bitwise reproducible.  The same transposed formulation as the forward: every accumulator has the lane's own
// query (or key) as its MFMA column, and a score block goes accumulator -> operand with a pack (the 16 MFMA rows of a block
// are assigned to keys/queries so that a lane ends up with EPC consecutive ones).  Products that contract over the tile's
// ROW index read the row-major LDS tile through transposed fragments (ds_read_b64_tr_b16 for bf16).
// The epilogues undo RoPE (and the q scale) and write gradients of the q/k/v projections' outputs in [M, H*hd] layout.
#include "sea_common.hpp"

typedef short s16x4b __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ uint4 frag_T(const char* tile, int pitch, int m0, int c0, int lane);
template <>
__device__ __forceinline__ uint4 frag_T<__bf16>(const char* tile, int pitch, int m0, int c0, int lane) {
    const int idx = lane & 15, gg = lane >> 4, q = idx >> 2, p = idx & 3;
    const char* base = tile + (m0 + 8 * gg + q) * pitch + (c0 + 4 * p) * 2;
    typedef s16x4b __attribute__((address_space(3))) * lds_p;
    const s16x4b lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base));
    const s16x4b hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + 4 * pitch));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);   // already packed pairs: no repacking VALU
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}
template <>
__device__ __forceinline__ uint4 frag_T<float>(const char* tile, int pitch, int m0, int c0, int lane) {
    const int idx = lane & 15, gg = lane >> 4;
    const char* base = tile + (m0 + 4 * gg) * pitch + (c0 + idx) * 4;
    f32x4 v = {*reinterpret_cast<const float*>(base), *reinterpret_cast<const float*>(base + pitch),
               *reinterpret_cast<const float*>(base + 2 * pitch), *reinterpret_cast<const float*>(base + 3 * pitch)};
    return __builtin_bit_cast(uint4, v);
}

template <typename T, int HD>
struct BwdCfg {
    static constexpr int EPC = ActTraits<T>::EPC;
    static constexpr int CK = ActTraits<T>::CK;
    static constexpr int NB = CK / 16;
    static constexpr int KCH = 64 / CK;                      // chunks of the 64-row tile
    static constexpr int NCH = (HD + CK - 1) / CK;           // head-dim chunks of a contraction over d
    static constexpr int NDB = (HD + 15) / 16;               // 16-wide d blocks
    static constexpr int ROW = HD * (int)sizeof(T);
    static constexpr int PCOLS = NDB * 16 < CK ? CK : NDB * 16;     // a row holds whole 16-column blocks and at least one contraction chunk
    static constexpr int PITCH = PCOLS * (int)sizeof(T) + 16;       // (+16 B against bank conflicts); columns >= HD are zeros, written once
    static constexpr int TILE = 64 * PITCH;
    static constexpr int CPR = HD / EPC;                     // 16-byte chunks per row
    static constexpr int NR = (64 * CPR + 255) / 256;        // chunks per thread per tile
    // the two streamed tiles are double-buffered; ONE buffer pair where four tiles do not fit the 160 KiB (f32 at head dim 256: 65 KiB per tile) — the
    // next pair still travels in registers under the MFMAs, its LDS write waits for a second barrier per tile
    static constexpr int NBUF = 4 * TILE + 1024 <= 160 * 1024 ? 2 : 1;
};

template <typename T>
__device__ __forceinline__ uint4 pack_frag(const f32x4& lo, const f32x4& hi) {
    if constexpr (sizeof(T) == 2) {
        bf16x8 pv = {(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
        return __builtin_bit_cast(uint4, pv);
    } else {
        return __builtin_bit_cast(uint4, lo);
    }
}

// stage a [64 rows][HD] row-major tile (rows row0.. of a [n_rows, HD] matrix with row stride ld elements) through registers
template <typename T, int HD>
struct TileStager {
    using C = BwdCfg<T, HD>;
    uint4 r[C::NR];
    __device__ __forceinline__ void load(const T* base, int64_t ld, int row0, int n_rows, int tid) {
#pragma unroll
        for (int u = 0; u < C::NR; ++u) {
            const int idx = tid + u * 256;
            const int rr = idx / C::CPR, cc = idx - rr * C::CPR;
            r[u] = make_uint4(0, 0, 0, 0);
            if (idx < 64 * C::CPR && row0 + rr < n_rows) r[u] = *reinterpret_cast<const uint4*>(base + (int64_t)(row0 + rr) * ld + cc * C::EPC);
        }
    }
    __device__ __forceinline__ void store(char* tile, int tid) const {
#pragma unroll
        for (int u = 0; u < C::NR; ++u) {
            const int idx = tid + u * 256;
            const int rr = idx / C::CPR, cc = idx - rr * C::CPR;
            if (idx < 64 * C::CPR) *reinterpret_cast<uint4*>(tile + rr * C::PITCH + cc * 16) = r[u];
        }
    }
};

// zero columns HD .. PCOLS-1 of the n_tiles row-major tiles at `base` (the stagers only ever write the first HD columns)
template <typename T, int HD>
__device__ __forceinline__ void zero_tile_padding(char* base, int n_tiles, int tid) {
    using C = BwdCfg<T, HD>;
    constexpr int PADB = (C::PCOLS - HD) * (int)sizeof(T);
    if constexpr (PADB > 0) {
        constexpr int CH = PADB / 16;
        static_assert(PADB % 16 == 0, "padding must be whole 16-byte chunks");
        for (int i = tid; i < n_tiles * 64 * CH; i += 256) {
            const int t = i / (64 * CH), rem = i - t * 64 * CH;
            const int row = rem / CH, cc = rem - row * CH;
            *reinterpret_cast<uint4*>(base + t * C::TILE + row * C::PITCH + HD * (int)sizeof(T) + cc * 16) = make_uint4(0, 0, 0, 0);
        }
    }
}

// inverse rotation of one (even, odd) pair: the forward was (e', o') = (e c - o s, e s + o c)
__device__ __forceinline__ void unrope(float& e, float& o, const float2 cs) { unrope_pair(e, o, cs.x, cs.y); }   // scalar-lane ops, see sea_common.hpp

// Workgroup order and tile pairing (mode bits of the kernels' second argument).
//   The plain order (decode_attn_block: tile-major, heaviest tiles of ALL (trajectory, head) pairs first) has every XCD walk its 24 pairs' K / V
//   (dQ kernel) or Q / dO rows (dK/dV kernel) at once: 6 MB against a 4 MB L2 — the L2 hit rate of the streamed tiles is ~0 and the launch pair moves
//   8.5x its operand bytes (PMC, profiles/r03_train_cfg3_pmc_traffic.json).  A pair-major order keeps an XCD on a few pairs at a time, but with one
//   causal tile per workgroup (1 .. 32 tile units of work) its tail is the last pair's heaviest tile running alone (measured: 322 -> 341 us).
//   ATTNB_PAIRED: a workgroup takes tile t AND tile n - 1 - t, one after the other — every workgroup carries n + 1 units, so any order is balanced;
//   ATTNB_XCD:    XCD x (the hardware deals consecutive workgroups to the 8 XCDs in turn) owns the pairs x, x + 8, ..., and walks them pair-major:
//                 the 16 workgroups of a pair run together and their streamed rows (0.26 MB per pair) stay in that XCD's L2.
enum { ATTNB_XCD = 1, ATTNB_PAIRED = 2 };
__device__ __forceinline__ void decode_attn_block_bwd(int& tile, int& bh, int& z, const int mode) {
    const int nbz = gridDim.y * gridDim.z;
    if (!(mode & ATTNB_XCD) || (nbz & 7)) return decode_attn_block(tile, bh, z);
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int x = L & 7, j = L >> 3;
    const int pl = j / (int)gridDim.x;
    tile = j - pl * gridDim.x;
    const int pair = x + 8 * pl;
    z = pair / (int)gridDim.y;
    bh = pair - z * gridDim.y;
}

// ---------------------------------------------------------------------------------------------- dQ (+ delta)
// DROP and (per tile) MASK are compile-time: both kernels are VALU-issue-bound (PMC: VALU busy 80-100 %), so the un-dropped,
// off-diagonal tile — the common case — carries no select, no dropout factor and one v_fma + one v_exp per probability.
#ifndef SEA_ATTNB_WPE
#define SEA_ATTNB_WPE 1
#endif
template <typename T, int HD>
constexpr int attn_bwd_min_waves() { return (sizeof(T) == 2 && HD <= 32) ? SEA_ATTNB_WPE : 1; }

template <typename T, int HD, bool DROP>
__device__ __forceinline__ void attn_bwd_dq_tile(const SeaAttnBwdParams& P, char* smem, const int qt, const int bh, const int zp) {
    using C = BwdCfg<T, HD>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int b = bh / P.H, h = bh - b * P.H;
    const SeaAttnBwdProblem& pr = P.p[zp];
    const int Tq = P.Tq, Tk = P.Tk;
    const T* Qg = static_cast<const T*>(pr.Q) + (int64_t)bh * Tq * HD;
    const T* Kg = static_cast<const T*>(pr.K) + (int64_t)bh * P.cap * HD;
    const T* Vg = static_cast<const T*>(pr.V) + (int64_t)bh * P.cap * HD;
    const int q_row0 = qt * 64 + wave * 16, q_idx = q_row0 + r, q_ld = q_idx < Tq ? q_idx : Tq - 1;
    const uint32_t drop_stream = (P.drop.stream + zp) * (uint32_t)(P.B * P.H) + (uint32_t)bh;
    const float drop_sc = P.drop.thr > 0 ? drop_scale(P.drop.thr) : 1.f;
    const T* Og = static_cast<const T*>(pr.O) + ((int64_t)b * Tq + q_ld) * P.ldo + h * HD;
    const T* dOg = static_cast<const T*>(pr.dO) + ((int64_t)b * Tq + q_ld) * P.lddo + h * HD;

    uint4 qf[C::NCH], dof[C::NCH];
    float delta = 0.f;
#pragma unroll
    for (int c = 0; c < C::NCH; ++c) {
        const int d0 = c * C::CK + g * C::EPC;
        qf[c] = make_uint4(0, 0, 0, 0);
        dof[c] = make_uint4(0, 0, 0, 0);
        if (d0 < HD) {
            qf[c] = *reinterpret_cast<const uint4*>(Qg + (int64_t)q_ld * HD + d0);
            dof[c] = *reinterpret_cast<const uint4*>(dOg + d0);
            T ov[C::EPC], dv[C::EPC];
            *reinterpret_cast<uint4*>(ov) = *reinterpret_cast<const uint4*>(Og + d0);
            *reinterpret_cast<uint4*>(dv) = dof[c];
#pragma unroll
            for (int e = 0; e < C::EPC; ++e) delta += to_f32(ov[e]) * to_f32(dv[e]);
        }
    }
    delta += __shfl_xor(delta, 16);
    delta += __shfl_xor(delta, 32);
    const float lse = pr.LSE[(int64_t)bh * Tq + q_ld];
    if (g == 0 && q_idx < Tq) pr.delta[(int64_t)bh * Tq + q_idx] = delta;

    f32x4 dq[C::NDB];
#pragma unroll
    for (int d = 0; d < C::NDB; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int limit = P.q_pos0 + q_idx + P.src_len;
    int n_kt = (P.q_pos0 + qt * 64 + 63 + P.src_len) / 64 + 1;
    const int n_kt_all = (Tk + 63) / 64;
    n_kt = n_kt < n_kt_all ? n_kt : n_kt_all;
    const int wave_first = P.q_pos0 + q_row0 + P.src_len, wave_last = wave_first + 15;

    int frag_off[C::KCH][C::NB];
#pragma unroll
    for (int kc = 0; kc < C::KCH; ++kc)
#pragma unroll
        for (int be = 0; be < C::NB; ++be) frag_off[kc][be] = (kc * C::CK + (r >> 2) * (4 * C::NB) + be * 4 + (r & 3)) * C::PITCH + g * 16;

    constexpr float LN2 = 0.69314718055994531f;
    const f32x4 nlse4 = {-lse, -lse, -lse, -lse};   // P = 2^(s - lse): the subtraction is the MFMA's C operand
    auto tile = [&](const char* sK, const char* sV, int kt, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        f32x4 ds[C::KCH][C::NB];
#pragma unroll
        for (int kc = 0; kc < C::KCH; ++kc) {
#pragma unroll
            for (int be = 0; be < C::NB; ++be) {
                // without dropout the "- delta" of dS = P (dP - delta) rides in the MFMA's C operand: dP starts at -delta
                f32x4 s = nlse4, dp = DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{-delta, -delta, -delta, -delta};
#pragma unroll
                for (int c = 0; c < C::NCH; ++c) {
                    const uint4 ak = *reinterpret_cast<const uint4*>(sK + frag_off[kc][be] + c * C::CK * (int)sizeof(T));   // rows zero-padded to CK
                    const uint4 av = *reinterpret_cast<const uint4*>(sV + frag_off[kc][be] + c * C::CK * (int)sizeof(T));
                    mma16<T>(ak, qf[c], s);
                    mma16<T>(av, dof[c], dp);
                }
                float dfac[4] = {1.f, 1.f, 1.f, 1.f};  // dropout: dP = D (dO V^T), D = keep * scale
                if constexpr (DROP) {
                    const int key0 = kt * 64 + kc * C::CK + g * C::EPC + be * 4;
                    const uint32_t w = drop_word(P.drop.seed, drop_stream, (uint32_t)q_idx, (uint32_t)(key0 >> 2));
#pragma unroll
                    for (int q = 0; q < 4; ++q) dfac[q] = drop_factor(w, q, P.drop.thr, drop_sc);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float p = __builtin_amdgcn_exp2f(s[q]);
                    if constexpr (MASK) {
                        const int key = kt * 64 + kc * C::CK + g * C::EPC + be * 4 + q;
                        if (!(key <= limit && key < Tk)) p = 0.f;
                    }
                    if constexpr (DROP) ds[kc][be][q] = p * (dp[q] * dfac[q] - delta);
                    else ds[kc][be][q] = p * dp[q];
                }
            }
        }
        // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
#pragma unroll
        for (int kc = 0; kc < C::KCH; ++kc) {
            const uint4 bfrag = pack_frag<T>(ds[kc][0], ds[kc][C::NB - 1]);
#pragma unroll
            for (int d = 0; d < C::NDB; ++d) {
                const uint4 a = frag_T<T>(sK, C::PITCH, kc * C::CK, d * 16, lane);
                mma16<T>(a, bfrag, dq[d]);
            }
        }
    };

    zero_tile_padding<T, HD>(smem, 2 * C::NBUF, tid);
    TileStager<T, HD> stK, stV;
    stK.load(Kg, HD, 0, Tk, tid);
    stV.load(Vg, HD, 0, Tk, tid);
    stK.store(smem, tid);
    stV.store(smem + C::TILE, tid);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): pre-loop register loads are complete, so the per-tile MFMAs do not wait on the prefetch (see attention.hip)
    __syncthreads();
    for (int kt = 0; kt < n_kt; ++kt) {
        const char* sK = smem + (C::NBUF == 2 ? (kt & 1) : 0) * 2 * C::TILE;
        const char* sV = sK + C::TILE;
        const bool more = kt + 1 < n_kt;
        if (more) {
            stK.load(Kg, HD, (kt + 1) * 64, Tk, tid);
            stV.load(Vg, HD, (kt + 1) * 64, Tk, tid);
        }
        if (kt * 64 <= wave_last) {
            if (kt * 64 + 63 <= wave_first && kt * 64 + 63 < Tk) tile(sK, sV, kt, std::false_type{});
            else tile(sK, sV, kt, std::true_type{});
        }
        if constexpr (C::NBUF == 1) __syncthreads();   // every wave is done reading the only buffer pair
        if (more) {
            char* nxt = smem + (C::NBUF == 2 ? ((kt + 1) & 1) : 0) * 2 * C::TILE;
            stK.store(nxt, tid);
            stV.store(nxt + C::TILE, tid);
        }
        __syncthreads();
    }
    if (q_idx < Tq) {
        T* out = static_cast<T*>(pr.dQ) + ((int64_t)b * Tq + q_idx) * P.lddq + h * HD;
        const float2* rope = reinterpret_cast<const float2*>(P.rope) + (int64_t)(P.q_pos0 + q_idx) * (HD / 2);
#pragma unroll
        for (int d = 0; d < C::NDB; ++d) {
            const int d0 = d * 16 + g * 4;
            if (d0 < HD) {
                float v[4] = {dq[d][0], dq[d][1], dq[d][2], dq[d][3]};
                unrope(v[0], v[1], rope[d0 >> 1]);
                unrope(v[2], v[3], rope[(d0 >> 1) + 1]);
                const float qs = P.q_scale * LN2;
                store4(out + d0, v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs);
            }
        }
    }
}

template <typename T, int HD, bool DROP>
__global__ __launch_bounds__(256, (attn_bwd_min_waves<T, HD>())) void attn_bwd_dq_kernel(const SeaAttnBwdParams P, const int mode) {
    using C = BwdCfg<T, HD>;
    __shared__ __attribute__((aligned(16))) char smem[C::NBUF * 2 * C::TILE];  // NBUF buffers x (K tile, V tile)
    int tile_, bh, zp;
    decode_attn_block_bwd(tile_, bh, zp, mode);
    const int n_qt = (P.Tq + 63) / 64;
    attn_bwd_dq_tile<T, HD, DROP>(P, smem, n_qt - 1 - tile_, bh, zp);  // heaviest query tiles first
    // ATTNB_PAIRED (grid.x = ceil(n_qt / 2)): then the light partner (every wave has passed the tile loop's last barrier: the buffers are free)
    if ((mode & ATTNB_PAIRED) && tile_ != n_qt - 1 - tile_) attn_bwd_dq_tile<T, HD, DROP>(P, smem, tile_, bh, zp);
}

// ---------------------------------------------------------------------------------------------- dK, dV
template <typename T, int HD, bool DROP>
__device__ __forceinline__ void attn_bwd_dkv_tile(const SeaAttnBwdParams& P, char* smem, const int kb, const int bh, const int zp) {
    using C = BwdCfg<T, HD>;
    constexpr int VEC_OFF = C::NBUF * 2 * C::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int b = bh / P.H, h = bh - b * P.H;
    const SeaAttnBwdProblem& pr = P.p[zp];
    const int Tq = P.Tq, Tk = P.Tk;
    const T* Qg = static_cast<const T*>(pr.Q) + (int64_t)bh * Tq * HD;
    const T* Kg = static_cast<const T*>(pr.K) + (int64_t)bh * P.cap * HD;
    const T* Vg = static_cast<const T*>(pr.V) + (int64_t)bh * P.cap * HD;
    const T* dOg = static_cast<const T*>(pr.dO) + (int64_t)b * Tq * P.lddo + h * HD;
    const float* lse_g = pr.LSE + (int64_t)bh * Tq;
    const float* del_g = pr.delta + (int64_t)bh * Tq;
    const int k_row0 = kb * 64 + wave * 16, k_idx = k_row0 + r, k_ld = k_idx < Tk ? k_idx : Tk - 1;
    const uint32_t drop_stream = (P.drop.stream + zp) * (uint32_t)(P.B * P.H) + (uint32_t)bh;
    const float drop_sc = P.drop.thr > 0 ? drop_scale(P.drop.thr) : 1.f;

    uint4 kf[C::NCH], vf[C::NCH];
#pragma unroll
    for (int c = 0; c < C::NCH; ++c) {
        const int d0 = c * C::CK + g * C::EPC;
        kf[c] = d0 < HD ? *reinterpret_cast<const uint4*>(Kg + (int64_t)k_ld * HD + d0) : make_uint4(0, 0, 0, 0);
        vf[c] = d0 < HD ? *reinterpret_cast<const uint4*>(Vg + (int64_t)k_ld * HD + d0) : make_uint4(0, 0, 0, 0);
    }
    f32x4 dk[C::NDB], dv[C::NDB];
#pragma unroll
    for (int d = 0; d < C::NDB; ++d) {
        dk[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // query q sees key j iff j <= q_pos0 + q + src_len  <=>  q >= j - src_len - q_pos0
    const int q_min_lane = k_idx - P.src_len - P.q_pos0;          // first query that sees this lane's key
    const int q_min_blk = kb * 64 - P.src_len - P.q_pos0;         // first query that sees any key of the workgroup
    const int qt0 = q_min_blk > 0 ? q_min_blk / 64 : 0;
    const int n_qt = (Tq + 63) / 64;
    const int wave_qmin = k_row0 - P.src_len - P.q_pos0;          // first query that sees the wave's FIRST key
    const int wave_qmax_need = wave_qmin + 15;                    // queries >= this see ALL the wave's keys

    int frag_off[C::KCH][C::NB];
#pragma unroll
    for (int qc = 0; qc < C::KCH; ++qc)
#pragma unroll
        for (int be = 0; be < C::NB; ++be) frag_off[qc][be] = (qc * C::CK + (r >> 2) * (4 * C::NB) + be * 4 + (r & 3)) * C::PITCH + g * 16;

    constexpr float LN2 = 0.69314718055994531f;
    auto tile = [&](const char* sQ, const char* sO, const float* sL, int qt, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        f32x4 pp[C::KCH][C::NB], ds[C::KCH][C::NB];
#pragma unroll
        for (int qc = 0; qc < C::KCH; ++qc) {
#pragma unroll
            for (int be = 0; be < C::NB; ++be) {
                const int ql = qc * C::CK + g * C::EPC + be * 4;  // this lane's 4 consecutive queries of the tile
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + ql);        // -lse (log2 units; negated when staged): the C operand of S
                const f32x4 nd4 = *reinterpret_cast<const f32x4*>(sL + 64 + ql);      // -delta (negated when staged: no negation per score block here)
                f32x4 s = l4, dp = DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : nd4;   // "- lse" and "- delta" in the C operands, as in the dQ kernel
#pragma unroll
                for (int c = 0; c < C::NCH; ++c) {
                    const uint4 aq = *reinterpret_cast<const uint4*>(sQ + frag_off[qc][be] + c * C::CK * (int)sizeof(T));   // rows zero-padded to CK
                    const uint4 ao = *reinterpret_cast<const uint4*>(sO + frag_off[qc][be] + c * C::CK * (int)sizeof(T));
                    mma16<T>(aq, kf[c], s);
                    mma16<T>(ao, vf[c], dp);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float p = __builtin_amdgcn_exp2f(s[q]);
                    if constexpr (MASK) {
                        const int qi = qt * 64 + ql + q;
                        if (!(qi >= q_min_lane && qi < Tq && k_idx < Tk)) p = 0.f;
                    }
                    if constexpr (DROP) {
                        const uint32_t w = drop_word(P.drop.seed, drop_stream, (uint32_t)(qt * 64 + ql + q), (uint32_t)(k_idx >> 2));
                        const float dfac = drop_factor(w, k_idx & 3, P.drop.thr, drop_sc);
                        pp[qc][be][q] = p * dfac;                    // dV = (D P)^T dO
                        ds[qc][be][q] = p * (dp[q] * dfac + nd4[q]);  // dS = P (D dP - delta)
                    } else {
                        pp[qc][be][q] = p;
                        ds[qc][be][q] = p * dp[q];
                    }
                }
            }
        }
#pragma unroll
        for (int qc = 0; qc < C::KCH; ++qc) {
            const uint4 pfrag = pack_frag<T>(pp[qc][0], pp[qc][C::NB - 1]);
            const uint4 sfrag = pack_frag<T>(ds[qc][0], ds[qc][C::NB - 1]);
#pragma unroll
            for (int d = 0; d < C::NDB; ++d) {
                const uint4 ao = frag_T<T>(sO, C::PITCH, qc * C::CK, d * 16, lane);   // dO^T[d][q]
                mma16<T>(ao, pfrag, dv[d]);
                const uint4 aq = frag_T<T>(sQ, C::PITCH, qc * C::CK, d * 16, lane);   // Q^T[d][q]
                mma16<T>(aq, sfrag, dk[d]);
            }
        }
    };

    TileStager<T, HD> stQ, stO;
    float r_lse = 0.f, r_del = 0.f;
    auto load_vec = [&](int qt) {
        if (tid < 64) {
            const int q = qt * 64 + tid;
            r_lse = q < Tq ? -lse_g[q] : 0.f;
            r_del = q < Tq ? del_g[q] : 0.f;
        }
    };
    auto store_vec = [&](int buf) {
        float* sl = reinterpret_cast<float*>(smem + VEC_OFF) + buf * 128;
        if (tid < 64) {
            sl[tid] = r_lse;
            sl[64 + tid] = -r_del;
        }
    };
    zero_tile_padding<T, HD>(smem, 2 * C::NBUF, tid);
    if (qt0 < n_qt) {
        stQ.load(Qg, HD, qt0 * 64, Tq, tid);
        stO.load(dOg, P.lddo, qt0 * 64, Tq, tid);
        load_vec(qt0);
        stQ.store(smem, tid);
        stO.store(smem + C::TILE, tid);
        store_vec(0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): pre-loop register loads are complete, so the per-tile MFMAs do not wait on the prefetch (see attention.hip)
    __syncthreads();
    for (int qt = qt0; qt < n_qt; ++qt) {
        const int bi = (qt - qt0) & 1;
        const char* sQ = smem + (C::NBUF == 2 ? bi : 0) * 2 * C::TILE;
        const char* sO = sQ + C::TILE;
        const float* sL = reinterpret_cast<const float*>(smem + VEC_OFF) + bi * 128;
        const bool more = qt + 1 < n_qt;
        if (more && C::NBUF == 2) {
            stQ.load(Qg, HD, (qt + 1) * 64, Tq, tid);
            stO.load(dOg, P.lddo, (qt + 1) * 64, Tq, tid);
            load_vec(qt + 1);
        }
        if (qt * 64 + 63 >= wave_qmin) {  // some query of the tile sees some key of this wave
            const bool need_mask = !(qt * 64 >= wave_qmax_need && qt * 64 + 63 < Tq) || (k_row0 + 15 >= Tk);
            if (need_mask) tile(sQ, sO, sL, qt, std::true_type{});
            else tile(sQ, sO, sL, qt, std::false_type{});
        }
        if constexpr (C::NBUF == 1) {
            __syncthreads();   // every wave is done reading the only buffer pair
            // (this form also loads the next pair only now: with dK, dV and the K / V fragments of a 256-wide f32 head in registers there is no room
            // for two tiles in flight — it is the parity path of the shipped multiphase width, not a fast one)
            if (more) {
                stQ.load(Qg, HD, (qt + 1) * 64, Tq, tid);
                stO.load(dOg, P.lddo, (qt + 1) * 64, Tq, tid);
                load_vec(qt + 1);
            }
        }
        if (more) {
            char* nxt = smem + (C::NBUF == 2 ? (bi ^ 1) : 0) * 2 * C::TILE;
            stQ.store(nxt, tid);
            stO.store(nxt + C::TILE, tid);
            store_vec(bi ^ 1);
        }
        __syncthreads();
    }
    if (k_idx < Tk) {
        T* outk = static_cast<T*>(pr.dK) + ((int64_t)b * Tk + k_idx) * P.lddk + h * HD;
        T* outv = static_cast<T*>(pr.dV) + ((int64_t)b * Tk + k_idx) * P.lddv + h * HD;
        const float2* rope = reinterpret_cast<const float2*>(P.rope) + (int64_t)k_idx * (HD / 2);
#pragma unroll
        for (int d = 0; d < C::NDB; ++d) {
            const int d0 = d * 16 + g * 4;
            if (d0 < HD) {
                float v[4] = {dk[d][0], dk[d][1], dk[d][2], dk[d][3]};
                unrope(v[0], v[1], rope[d0 >> 1]);
                unrope(v[2], v[3], rope[(d0 >> 1) + 1]);
                store4(outk + d0, v[0] * LN2, v[1] * LN2, v[2] * LN2, v[3] * LN2);
                store4(outv + d0, dv[d][0], dv[d][1], dv[d][2], dv[d][3]);
            }
        }
    }
}

template <typename T, int HD, bool DROP>
__global__ __launch_bounds__(256, (attn_bwd_min_waves<T, HD>())) void attn_bwd_dkv_kernel(const SeaAttnBwdParams P, const int mode) {
    using C = BwdCfg<T, HD>;
    constexpr int VEC_OFF = C::NBUF * 2 * C::TILE;
    __shared__ __attribute__((aligned(16))) char smem[VEC_OFF + 2 * 2 * 64 * 4];  // NBUF x (Q tile, dO tile) + 2 x (lse, delta)
    int kb, bh, zp;  // key tile: the first key tiles are seen by the most queries -> ascending order is heaviest-first
    decode_attn_block_bwd(kb, bh, zp, mode);
    attn_bwd_dkv_tile<T, HD, DROP>(P, smem, kb, bh, zp);
    if (mode & ATTNB_PAIRED) {   // grid.x = ceil(n_kt / 2): then the light partner tile
        const int n_kt = (P.Tk + 63) / 64;
        __syncthreads();         // (a tile with no visible query tile never enters the loop and its barriers)
        if (kb != n_kt - 1 - kb) attn_bwd_dkv_tile<T, HD, DROP>(P, smem, n_kt - 1 - kb, bh, zp);
    }
}

template <typename T, int HD>
static void launch_bwd(const SeaAttnBwdParams& P, hipStream_t s) {
    const dim3 block(256), gq((P.Tq + 63) / 64, P.B * P.H, P.n_problems), gk((P.Tk + 63) / 64, P.B * P.H, P.n_problems);
    // paired tiles + XCD-local pair-major order for launches of at least 4096 workgroups at head dims >= 32 (SEA_TUNE=attnb_mode=0..3 forces the mode bits).
    // Measured at cfg3 (tools/bench_attn_bwd.py, tools/pmc_attnb.py): self (hd 32, 6144 workgroups) 323 -> 320 us with the pair's HBM reads 1599 -> 331 MB;
    // cross (hd 16, 4096 workgroups) 183 -> 191 us (387 -> 160 MB): at hd 16 a pair's rows are half the bytes and the plain order's finer-grained tail wins.
    const int forced = sea_tune("attnb_mode", -1);   // read per call (tests force every mode in one process)
    const long nq = (long)gq.x * gq.y * gq.z, nk = (long)gk.x * gk.y * gk.z;
    const int long_mode = HD >= 32 ? (ATTNB_PAIRED | ATTNB_XCD) : 0;
    const int mq = forced >= 0 ? forced : (nq >= 4096 ? long_mode : 0), mk = forced >= 0 ? forced : (nk >= 4096 ? long_mode : 0);
    const int pad = sea_tune("attnb_pad", 0);   // tuning aid: extra dynamic LDS bytes per workgroup (lowers the occupancy without touching the code)
    dim3 gq2 = gq, gk2 = gk;
    if (mq & ATTNB_PAIRED) gq2.x = (gq.x + 1) / 2;
    if (mk & ATTNB_PAIRED) gk2.x = (gk.x + 1) / 2;
    if (P.drop.thr > 0) {
        attn_bwd_dq_kernel<T, HD, true><<<gq2, block, pad, s>>>(P, mq);
        attn_bwd_dkv_kernel<T, HD, true><<<gk2, block, pad, s>>>(P, mk);
    } else {
        attn_bwd_dq_kernel<T, HD, false><<<gq2, block, pad, s>>>(P, mq);
        attn_bwd_dkv_kernel<T, HD, false><<<gk2, block, pad, s>>>(P, mk);
    }
}

template <typename T>
static int dispatch_bwd(const SeaAttnBwdParams& P, hipStream_t s) {
    switch (P.hd) {
        case 8: launch_bwd<T, 8>(P, s); break;
        case 16: launch_bwd<T, 16>(P, s); break;
        case 32: launch_bwd<T, 32>(P, s); break;
        case 64: launch_bwd<T, 64>(P, s); break;
        case 128: launch_bwd<T, 128>(P, s); break;
        case 256: launch_bwd<T, 256>(P, s); break;   // the shipped multiphase dims (embed_dim 2048 / 8 heads); f32: one LDS buffer pair (BwdCfg::NBUF)
        default: return -1;
    }
    return 0;
}

extern "C" int sea_attention_bwd(const SeaAttnBwdParams* params, int dtype, void* stream) {
    SEA_REQUIRE(params != nullptr, "sea_attention_bwd: null params");
    const SeaAttnBwdParams& P = *params;
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_attention_bwd: bad dtype %d", dtype);
    SEA_REQUIRE(P.n_problems >= 1 && P.n_problems <= SEA_MAX_ATTN_PROBLEMS, "sea_attention_bwd: n_problems=%d", P.n_problems);
    SEA_REQUIRE(P.B >= 1 && P.H >= 1 && P.Tq >= 1 && P.Tk >= 1 && P.cap >= P.Tk && P.q_pos0 >= 0 && P.src_len >= 0 && P.rope,
                "sea_attention_bwd: bad sizes B=%d H=%d Tq=%d Tk=%d cap=%d", P.B, P.H, P.Tq, P.Tk, P.cap);
    SEA_REQUIRE(P.hd == 8 || P.hd == 16 || P.hd == 32 || P.hd == 64 || P.hd == 128 || P.hd == 256, "sea_attention_bwd: unsupported head dim %d (8 .. 256, powers of two)", P.hd);
    const int epc = dtype == SEA_BF16 ? 8 : 4;
    SEA_REQUIRE(P.ldo % epc == 0 && P.lddo % epc == 0 && P.lddq % 4 == 0 && P.lddk % 4 == 0 && P.lddv % 4 == 0, "sea_attention_bwd: bad strides");
    SEA_REQUIRE((long)P.B * P.H <= 65535, "sea_attention_bwd: B*H too large for grid.y");
    for (int i = 0; i < P.n_problems; ++i) {
        const SeaAttnBwdProblem& q = P.p[i];
        SEA_REQUIRE(q.Q && q.K && q.V && q.O && q.dO && q.LSE && q.delta && q.dQ && q.dK && q.dV, "sea_attention_bwd[%d]: null pointer", i);
        SEA_REQUIRE(sea_aligned16(q.Q) && sea_aligned16(q.K) && sea_aligned16(q.V) && sea_aligned16(q.O) && sea_aligned16(q.dO) &&
                        sea_aligned16(q.dQ) && sea_aligned16(q.dK) && sea_aligned16(q.dV), "sea_attention_bwd[%d]: pointers must be 16-byte aligned", i);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rc = dtype == SEA_BF16 ? dispatch_bwd<__bf16>(P, s) : dispatch_bwd<float>(P, s);
    SEA_REQUIRE(rc == 0, "sea_attention_bwd: no kernel for hd=%d", P.hd);
    SEA_CHECK_LAUNCH("sea_attention_bwd");
    return SEA_OK;
}
