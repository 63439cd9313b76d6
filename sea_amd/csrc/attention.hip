// Causal flash attention forward for SEA's self- and cross-attention (gfx950): sea_attention_fwd.
//
// One workgroup = 4 waves = 64 query rows of one (trajectory, head); wave w owns rows 16w..16w+15.  K/V tiles of 64
// keys are staged through LDS and shared by the 4 waves.  Everything is computed TRANSPOSED so that a query is a
// lane (lane & 15) in every accumulator and no row statistic ever crosses lanes except the 4 lane groups of a
// query (two __shfl_xor):
//     S^T = K . Q^T      MFMA A = K rows (from LDS), B = Q (registers, loaded once)      C[key][query]   — in LOG2 units: Q arrives scaled by
//                        hd^-1/2 * log2(e) (sea_qkv_rope_grouped's q_scale), so P = 2^(S - max) and every exponential is one v_exp_f32
//     O^T = V^T . P^T    MFMA A = V^T rows (from LDS; V is kept transposed in HBM), B = P^T C[d][query]
// The S^T accumulator of lane (g = lane >> 4, q = lane & 15) holds, per 16-key MFMA block, rows 4g..4g+3.  MFMA row
// 4y+i of block beta of a key chunk is assigned key  y*(4*NB) + 4*beta + i  (NB = blocks per chunk), so after the
// NB blocks of a chunk lane group g holds the EPC = 4*NB CONSECUTIVE keys g*EPC .. g*EPC+EPC-1: exactly the
// B-operand fragment (k = g*EPC + j) of the O^T product — P goes from accumulator to operand with a pack, no
// shuffle, no LDS.
// Head dims 8..128 (multiples of 8); contraction padded with zero lanes when hd < CK (hd = 16 bf16, hd = 8).
#include "sea_common.hpp"

template <typename T, int HD>
struct AttnCfg {
    static constexpr int EPC = ActTraits<T>::EPC;
    static constexpr int CK = ActTraits<T>::CK;              // keys per key chunk = contraction per mma16
    static constexpr int NB = CK / 16;                       // 16-key MFMA blocks per key chunk
    static constexpr int KCH = 64 / CK;                      // key chunks per 64-key tile
    static constexpr int NCH = (HD + CK - 1) / CK;           // head-dim chunks of the S^T contraction
    static constexpr int NDB = (HD + 15) / 16;               // 16-row d blocks of O^T
    static constexpr int K_ROW = HD * (int)sizeof(T);        // bytes per key row
    static constexpr int K_COLS = HD < CK ? CK : HD;         // LDS row holds a whole contraction chunk: columns >= HD are zeros, written once
    static constexpr int K_STRIDE = K_COLS * (int)sizeof(T) + 16;   // padded LDS stride
    static constexpr int V_ROW = 64 * (int)sizeof(T);        // bytes per d row of the V^T tile
    static constexpr int V_STRIDE = V_ROW + 16;
    static constexpr int K_BYTES = 64 * K_STRIDE;
    static constexpr int V_BYTES = HD * V_STRIDE;
    static constexpr int LDS_BYTES = K_BYTES + V_BYTES;
};

// SPLIT = 2: two groups of 4 waves share a query tile and take alternate key tiles (group g: tiles g, g+2, ...), each with its
// own running max / sum / O^T; the halves are merged through LDS at the end.  This halves the serial chain of key tiles of a
// workgroup — what bounds the launch when the grid is only a few workgroups per CU (one trajectory: 768 workgroups).
// DROP is a template parameter so that the inference instantiation carries none of the dropout's select/merge moves.
// waves per SIMD the register allocation must leave room for (the second __launch_bounds__ argument): the kernel is bound by the latency of a
// wave's per-tile dependency chain, so resident waves are what it runs on
// Measured (tools/attn_ab.sh, bf16): hd 16 at 8 waves per SIMD (64 VGPRs, 4-5 spilled) 57.1 -> 56.4 us at B = 8 and — the 1024-thread SPLIT = 4 form needs
// 8 for two workgroups per CU — 17.9 -> 13.8 us at B = 1; hd 32 at 6 (80 VGPRs) 111.0 -> 107.3 us at B = 8, at 8 it spills 37 registers (185 us).
template <typename T, int HD, bool DROP>
constexpr int attn_min_waves() {
    if (DROP) return sizeof(T) == 2 && HD <= 32 ? 5 : 1;   // (the dropout form carries the mask words: it spills at the bounds below)
#ifndef SEA_ATTN_WPE16
#define SEA_ATTN_WPE16 8
#endif
#ifndef SEA_ATTN_WPE32
#define SEA_ATTN_WPE32 6
#endif
    return sizeof(T) == 2 ? (HD <= 16 ? SEA_ATTN_WPE16 : (HD == 32 ? SEA_ATTN_WPE32 : 1)) : 1;
}

template <typename T, int HD, int SPLIT>
struct AttnFwdLds {
    using C = AttnCfg<T, HD>;
    static constexpr int MERGE_BYTES = SPLIT > 1 ? (SPLIT - 1) * 256 * (2 + 4 * C::NDB) * 4 : 0;
    // per group: double-buffered K and V^T tiles; ONE buffer where two do not fit the 160 KiB (f32 at head dim 256: 133 KiB per tile pair) — the
    // next tile still travels in registers under the MFMAs, its LDS write waits for a second barrier per tile
    static constexpr int NBUF = 2 * C::LDS_BYTES <= 160 * 1024 ? 2 : 1;
    static_assert(NBUF == 2 || (SPLIT == 1 && HD >= C::CK), "single-buffer form: one wave group, no zero-padded key rows");
    static constexpr int RING_BYTES = SPLIT * NBUF * C::LDS_BYTES;
    static constexpr int BYTES = RING_BYTES > MERGE_BYTES ? RING_BYTES : MERGE_BYTES;
};

// one 64-row query tile `qt` of (trajectory, head) pair `bh` of problem `zp`
template <typename T, int HD, int SPLIT, bool DROP>
__device__ __forceinline__ void attention_fwd_tile(const SeaAttnParams& P, char* smem_all, const int qt, const int bh, const int zp) {
    using C = AttnCfg<T, HD>;
    constexpr int NBUF = AttnFwdLds<T, HD, SPLIT>::NBUF;
    const int grp = SPLIT > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;  // wave-uniform
    char* smem = smem_all + grp * NBUF * C::LDS_BYTES;

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;  // tid, wave: inside the group
    const int r = lane & 15, g = lane >> 4;
    const int b = bh / P.H, h = bh - b * P.H;
    const SeaAttnProblem& pr = P.p[zp];
    const T* Qg = static_cast<const T*>(pr.Q) + (int64_t)bh * P.Tq * HD;
    const T* Kg = static_cast<const T*>(pr.K) + (int64_t)bh * P.cap * HD;
    const T* Vg = static_cast<const T*>(pr.Vt) + (int64_t)bh * HD * P.cap;
    const int Tq = P.Tq, Tk = P.Tk, cap = P.cap;

    const int q_row0 = qt * 64 + wave * 16;
    const int q_idx = q_row0 + r;
    const int q_ld = q_idx < Tq ? q_idx : Tq - 1;
    const uint32_t drop_stream = (P.drop.stream + zp) * (uint32_t)(P.B * P.H) + (uint32_t)bh;

    // Q fragments (B operand of S^T): lane holds Q[q][c*CK + g*EPC .. +EPC)
    uint4 qf[C::NCH];
#pragma unroll
    for (int c = 0; c < C::NCH; ++c) {
        const int d0 = c * C::CK + g * C::EPC;
        qf[c] = d0 < HD ? *reinterpret_cast<const uint4*>(Qg + (int64_t)q_ld * HD + d0) : make_uint4(0, 0, 0, 0);
    }

    f32x4 oacc[C::NDB];
#pragma unroll
    for (int d = 0; d < C::NDB; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Online softmax in the exp2 domain with a LAZY reference: the scores arrive in log2 units (the QKV epilogue scaled q by hd^-1/2 * log2 e: include/sea_hip.h), the
    // running sums are l = sum 2^(s - ref), O^T = sum 2^(s - ref) v with a per-row reference `ref` that is moved only when a tile's maximum exceeds it
    // by more than REBASE_THR (or at the row's first visible tile).  "- ref" rides in the MFMA's C operand (negm4), so the common tile costs one v_exp
    // per score and nothing else: no subtraction, no rescale of O^T.  Probabilities are bounded by 2^REBASE_THR (bf16 keeps its 8 significant bits at
    // any magnitude, the sums are fp32).  The decision is per ROW (the four lanes of a row agree after group_max4), the wave-uniform branch only skips
    // work that would multiply by exactly 1 / subtract exactly 0: a row's result never depends on the other rows of its wave (bitwise causality).
    constexpr float REBASE_THR = 8.0f;
    float ref = 0.f, l_i = 0.f;
    float thr = -INFINITY;       // -inf: the row has not seen a key yet (row-level: set from the row maximum, the same in the four lanes of a row)
    f32x4 negm4 = {0.f, 0.f, 0.f, 0.f};

    const int limit = P.q_pos0 + q_idx + P.src_len;                       // keys j <= limit are visible to this lane's query
    const int blk_last = P.q_pos0 + qt * 64 + 63 + P.src_len;             // last key any row of the workgroup may see
    int n_kt = blk_last / 64 + 1;
    const int n_kt_all = (Tk + 63) / 64;
    n_kt = n_kt < n_kt_all ? n_kt : n_kt_all;
    const int wave_first = P.q_pos0 + q_row0 + P.src_len;                 // last key the wave's FIRST row may see
    const int wave_last = wave_first + 15;                                // last key any of the wave's rows may see

    // ---- K/V tile staging through registers (issue-early / write-late, cdna_hip_programming.md T14): the global loads of
    // tile kt+1 are issued before the MFMAs of tile kt and written to the OTHER LDS buffer after them; one barrier per tile.
    constexpr int K_CPR = HD / C::EPC;                       // 16-byte chunks per key row
    constexpr int V_CPR = 64 / C::EPC;                       // 16-byte chunks per d row of the V^T tile
    constexpr int NKR = (64 * K_CPR + 255) / 256;            // chunks per thread
    constexpr int NVR = (HD * V_CPR + 255) / 256;
    uint4 rk[NKR], rv[NVR];
    // per-thread chunk coordinates, fixed for the whole kernel
    int k_row[NKR], k_goff[NKR], k_soff[NKR], v_key[NVR], v_soff[NVR];
    int64_t v_goff[NVR];
#pragma unroll
    for (int u = 0; u < NKR; ++u) {
        const int idx = tid + u * 256;
        const int rr = idx / K_CPR, cc = idx - rr * K_CPR;
        k_row[u] = idx < 64 * K_CPR ? rr : -1;
        k_goff[u] = rr * HD + cc * C::EPC;
        k_soff[u] = rr * C::K_STRIDE + cc * 16;
    }
#pragma unroll
    for (int u = 0; u < NVR; ++u) {
        const int idx = tid + u * 256;
        const int d = idx / V_CPR, cc = idx - d * V_CPR;
        v_key[u] = idx < HD * V_CPR ? cc * C::EPC : -1;
        v_goff[u] = (int64_t)d * cap + cc * C::EPC;
        v_soff[u] = d * C::V_STRIDE + cc * 16;
    }
    auto load_tile = [&](int kt) {
        const T* Kt = Kg + (int64_t)kt * 64 * HD;
        const T* Vt = Vg + kt * 64;
        if ((kt + 1) * 64 <= Tk) {  // block-uniform fast path: the whole tile is inside the key range
#pragma unroll
            for (int u = 0; u < NKR; ++u)
                if (k_row[u] >= 0) rk[u] = *reinterpret_cast<const uint4*>(Kt + k_goff[u]);
#pragma unroll
            for (int u = 0; u < NVR; ++u)
                if (v_key[u] >= 0) rv[u] = *reinterpret_cast<const uint4*>(Vt + v_goff[u]);
        } else {  // last, partial tile: zero beyond Tk (0 * garbage must stay 0)
#pragma unroll
            for (int u = 0; u < NKR; ++u) {
                rk[u] = make_uint4(0, 0, 0, 0);
                if (k_row[u] >= 0 && kt * 64 + k_row[u] < Tk) rk[u] = *reinterpret_cast<const uint4*>(Kt + k_goff[u]);
            }
#pragma unroll
            for (int u = 0; u < NVR; ++u) {
                rv[u] = make_uint4(0, 0, 0, 0);
                const int key0 = kt * 64 + v_key[u];
                if (v_key[u] >= 0 && key0 < Tk) {
                    if (key0 + C::EPC <= Tk) {
                        rv[u] = *reinterpret_cast<const uint4*>(Vt + v_goff[u]);
                    } else {
                        T tmp[C::EPC];
#pragma unroll
                        for (int e = 0; e < C::EPC; ++e) tmp[e] = key0 + e < Tk ? Vt[v_goff[u] + e] : from_f32<T>(0.f);
                        rv[u] = *reinterpret_cast<const uint4*>(tmp);
                    }
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        char* dK = smem + buf * C::LDS_BYTES;
        char* dV = dK + C::K_BYTES;
#pragma unroll
        for (int u = 0; u < NKR; ++u)
            if (k_row[u] >= 0) *reinterpret_cast<uint4*>(dK + k_soff[u]) = rk[u];
#pragma unroll
        for (int u = 0; u < NVR; ++u)
            if (v_key[u] >= 0) *reinterpret_cast<uint4*>(dV + v_soff[u]) = rv[u];
    };

    // lane-constant LDS offsets of the fragments
    int k_frag_off[C::KCH][C::NB];
#pragma unroll
    for (int kc = 0; kc < C::KCH; ++kc)
#pragma unroll
        for (int be = 0; be < C::NB; ++be)
            k_frag_off[kc][be] = (kc * C::CK + (r >> 2) * (4 * C::NB) + be * 4 + (r & 3)) * C::K_STRIDE + g * 16;
    const int v_frag_off = r * C::V_STRIDE + g * 16;

    // One 64-key tile for this wave (MASK = the tile crosses the causal diagonal or the end of the keys): scores relative to the reference, the tile's
    // row maximum and — rarely — a move of the reference, probabilities, O^T += V^T P^T.
    // (Measured and not kept, tools/attn_ab.sh: a max-free tile — exponentiate as it stands, check only the row sum the matrix core delivers anyway, redo
    // with the maximum when it is huge — 16 VALU instructions fewer per tile, no faster: B = 8 107.8 / 56.3 us against 107.3 / 56.4 at head dim 32 / 16,
    // and slower at B = 1, where the second copy of the tile's code costs the 1024-thread form its 64-register budget.  The kernel is bound by the
    // latency of a wave's per-tile chain — LDS fragments -> MFMA -> max -> exp -> pack -> MFMA, one barrier per tile — not by VALU issue.)
    // (Round 3, built, parity-green and not kept: the K / V^T tiles through a 3- / 4-stage global_load_lds ring (no staging registers, two to three tiles in
    // flight, source-side swizzle, zeros for the upper half of a head-dim-16 chunk by lane select) instead of the register-staged double buffer: cross-attention
    // at cfg2 13.3 -> 14.4 us, self 17.8 -> 18.3, B = 8 forward 1.094 -> 1.111 ms.  Global latency is not what the tile loop waits for: the launch lasts as long
    // as its heaviest workgroup — the last query tile's 8 rounds of key tiles with four waves per SIMD taking turns at ~500 VALU cycles per wave and tile.)
    constexpr bool MFMA_SUM = sizeof(T) == 2 && !DROP;
    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);   // bf16 1.0 x 8: the A operand of the row sums

    auto scores = [&](const char* sK, f32x4 (&sc)[C::KCH][C::NB], int kt, auto mask_tag) {   // S'^T = K . Q^T - ref (masked entries -inf)
        constexpr bool MASK = decltype(mask_tag)::value;
#pragma unroll
        for (int kc = 0; kc < C::KCH; ++kc) {
#pragma unroll
            for (int be = 0; be < C::NB; ++be) {
                sc[kc][be] = negm4;
#pragma unroll
                for (int c = 0; c < C::NCH; ++c) {
                    // head dims below one contraction chunk: the tile rows are zero-padded in LDS, so no lane needs a select here
                    const uint4 a = *reinterpret_cast<const uint4*>(sK + k_frag_off[kc][be] + c * C::CK * (int)sizeof(T));
                    mma16<T>(a, qf[c], sc[kc][be]);
                }
                if constexpr (MASK) {   // (this lane: query q_idx, keys kt*64 + kc*CK + g*EPC + be*4 + reg)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int key = kt * 64 + kc * C::CK + g * C::EPC + be * 4 + q;
                        if (!(key <= limit && key < Tk)) sc[kc][be][q] = -INFINITY;
                    }
                }
            }
        }
    };
    auto rebase = [&](f32x4 (&sc)[C::KCH][C::NB]) {   // move the reference of the rows whose tile maximum calls for it; sc stays relative to the (new) reference
        float mx = -INFINITY;
#pragma unroll
        for (int kc = 0; kc < C::KCH; ++kc)
#pragma unroll
            for (int be = 0; be < C::NB; ++be) {
                mx = fmaxf(fmaxf(mx, sc[kc][be][0]), sc[kc][be][1]);   // (v_max3_f32)
                mx = fmaxf(fmaxf(mx, sc[kc][be][2]), sc[kc][be][3]);
            }
        mx = group_max4(mx);
        // `thr`: -inf until the row has seen a key (its first visible tile always moves the reference), REBASE_THR afterwards
        if (__builtin_amdgcn_ballot_w64(mx > thr) != 0) {          // wave-uniform; rare after a row's first tiles
            const bool need = mx > thr;
            const float delta = need ? mx : 0.f;
            const float alpha = thr == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // (first tile: l = O = 0, and 2^-delta may overflow) need false: exactly 1
            ref += delta;
            negm4 = f32x4{-ref, -ref, -ref, -ref};
            thr = mx > -INFINITY ? REBASE_THR : thr;
#pragma unroll
            for (int kc = 0; kc < C::KCH; ++kc)
#pragma unroll
                for (int be = 0; be < C::NB; ++be)
#pragma unroll
                    for (int q = 0; q < 4; ++q) sc[kc][be][q] -= delta;
#pragma unroll
            for (int d = 0; d < C::NDB; ++d) oacc[d] *= alpha;
            l_i *= alpha;
        }
    };
    // probabilities in place, packed P^T fragments, the tile's row sum (matrix core: complete; VALU: this lane group's partial, un-dropped)
    auto exp_pack = [&](f32x4 (&sc)[C::KCH][C::NB], uint4 (&pf)[C::KCH], int kt) -> float {
        float psum = 0.f;
#pragma unroll
        for (int kc = 0; kc < C::KCH; ++kc)
#pragma unroll
            for (int be = 0; be < C::NB; ++be)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float p = __builtin_amdgcn_exp2f(sc[kc][be][q]);
                    sc[kc][be][q] = p;
                    if constexpr (!MFMA_SUM) psum += p;
                }
        if constexpr (DROP) {  // dropout on the probabilities: the row sum above stays un-dropped (softmax first, then dropout)
            const float dsc = drop_scale(P.drop.thr);
#pragma unroll
            for (int kc = 0; kc < C::KCH; ++kc)
#pragma unroll
                for (int be = 0; be < C::NB; ++be) {
                    const int key0 = kt * 64 + kc * C::CK + g * C::EPC + be * 4;
                    const uint32_t w = drop_word(P.drop.seed, drop_stream, (uint32_t)q_idx, (uint32_t)(key0 >> 2));
#pragma unroll
                    for (int q = 0; q < 4; ++q) sc[kc][be][q] *= drop_factor(w, q, P.drop.thr, dsc);
                }
        }
        f32x4 lsum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < C::KCH; ++kc) {
            if constexpr (sizeof(T) == 2) {
                bf16x8 pv = {(__bf16)sc[kc][0][0], (__bf16)sc[kc][0][1], (__bf16)sc[kc][0][2], (__bf16)sc[kc][0][3],
                             (__bf16)sc[kc][1][0], (__bf16)sc[kc][1][1], (__bf16)sc[kc][1][2], (__bf16)sc[kc][1][3]};
                pf[kc] = __builtin_bit_cast(uint4, pv);
            } else {
                pf[kc] = __builtin_bit_cast(uint4, sc[kc][0]);
            }
            if constexpr (MFMA_SUM) mma16<T>(ones, pf[kc], lsum);
        }
        if constexpr (MFMA_SUM) psum = lsum[0];
        return psum;
    };
    auto process = [&](const char* sK, const char* sV, int kt, auto mask_tag) {
        f32x4 sc[C::KCH][C::NB];
        uint4 pf[C::KCH];
        scores(sK, sc, kt, mask_tag);
        rebase(sc);
        const float psum = exp_pack(sc, pf, kt);
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int kc = 0; kc < C::KCH; ++kc) {
#pragma unroll
            for (int d = 0; d < C::NDB; ++d) {
                uint4 a = make_uint4(0, 0, 0, 0);
                if (d * 16 + r < HD) a = *reinterpret_cast<const uint4*>(sV + v_frag_off + d * 16 * C::V_STRIDE + kc * C::CK * (int)sizeof(T));
                mma16<T>(a, pf[kc], oacc[d]);
            }
        }
        l_i += psum;
    };

    if constexpr (HD < C::CK) {   // zero columns HD .. CK-1 of every key row of both buffers of this group (the staging never touches them)
        constexpr int PADB = (C::CK - HD) * (int)sizeof(T);   // multiple of 16
        for (int i = tid; i < 2 * 64 * (PADB / 16); i += 256) {
            const int buf = i / (64 * (PADB / 16)), rem = i - buf * 64 * (PADB / 16);
            const int row = rem / (PADB / 16), cc = rem - row * (PADB / 16);
            *reinterpret_cast<uint4*>(smem + buf * C::LDS_BYTES + row * C::K_STRIDE + C::K_ROW + cc * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    // group g walks tiles g, g + SPLIT, ...; every thread of the workgroup executes every barrier
    const int n_it = (n_kt + SPLIT - 1) / SPLIT;
    if (grp < n_kt) {
        load_tile(grp);
        store_tile(0);
    }
    // Every register loaded before the loop (the Q fragments) is complete from here on.  Without this the compiler, seeing the path
    // that skips the conditional prologue above, keeps "Q may still be in flight" alive around the loop and waits vmcnt(0) before the
    // first MFMA of EVERY tile — i.e. on the K/V prefetch it has just issued, serialising load latency with compute.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
        const int kt = it * SPLIT + grp;
        const char* sK = smem + (NBUF == 2 ? (it & 1) : 0) * C::LDS_BYTES;
        const char* sV = sK + C::K_BYTES;
        const bool more = kt + SPLIT < n_kt;
        if (more) load_tile(kt + SPLIT);
        const int tile_last = kt * 64 + 63;
        if (kt < n_kt) {
            if (tile_last <= wave_first && tile_last < Tk) {
                process(sK, sV, kt, std::false_type{});   // fully visible to every row of the wave: no masking
            } else if (kt * 64 <= wave_last) {
                process(sK, sV, kt, std::true_type{});    // diagonal / last tile
            }                                             // else: nothing visible to this wave (wave-uniform)
        }
        if constexpr (NBUF == 1) __syncthreads();   // every wave is done reading the only buffer
        if (more) store_tile(NBUF == 2 ? (it + 1) & 1 : 0);
        __syncthreads();
    }

    if constexpr (SPLIT > 1) {
        // merge groups 1.. into group 0: m = max(m0, m1), l = l0 e^(m0-m) + l1 e^(m1-m), O likewise (flash-decoding combine)
        float* mg = reinterpret_cast<float*>(smem_all);
        constexpr int STR = 2 + 4 * C::NDB;
        if (grp != 0) {
            float* dst = mg + ((grp - 1) * 256 + tid) * STR;
            dst[0] = thr > -INFINITY ? ref : -INFINITY;   // (row-level: the f32 / dropout paths keep PARTIAL row sums per lane group, l_i == 0 says nothing about the row)
            dst[1] = l_i;
#pragma unroll
            for (int d = 0; d < C::NDB; ++d)
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[2 + 4 * d + q] = oacc[d][q];
        }
        __syncthreads();
        if (grp != 0) return;
#pragma unroll
        for (int o = 0; o < SPLIT - 1; ++o) {
            const float* src = mg + (o * 256 + tid) * STR;
            const float l1 = src[1];
            const float m1 = src[0];                          // -inf: the group saw no key of this row
            const float m = fmaxf(ref, m1);  // group 0 always owns tile 0 (key 0 is visible to every row), so ref is a real reference
            const float a0 = __builtin_amdgcn_exp2f(ref - m), a1 = __builtin_amdgcn_exp2f(m1 - m);  // m1 = -inf -> a1 = 0
            l_i = l_i * a0 + l1 * a1;
#pragma unroll
            for (int d = 0; d < C::NDB; ++d)
#pragma unroll
                for (int q = 0; q < 4; ++q) oacc[d][q] = oacc[d][q] * a0 + src[2 + 4 * d + q] * a1;
            ref = m;
        }
    }

    // ---- finalize: this lane holds O^T[d = 16*db + 4g + reg][query q_idx]
    float l = l_i;
    if constexpr (!(sizeof(T) == 2 && !DROP)) {  // VALU row sums are per lane group
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
    }
    const float inv = 1.0f / l;
    if (q_idx < Tq) {
        T* Og = static_cast<T*>(pr.O) + ((int64_t)b * Tq + q_idx) * P.ldo + h * HD;
#pragma unroll
        for (int d = 0; d < C::NDB; ++d) {
            const int d0 = d * 16 + g * 4;
            if (d0 < HD) store4(Og + d0, oacc[d][0] * inv, oacc[d][1] * inv, oacc[d][2] * inv, oacc[d][3] * inv);
        }
        if (pr.LSE != nullptr && g == 0) pr.LSE[(int64_t)bh * Tq + q_idx] = ref + __log2f(l);   // log2 units, like the scores (the backward recomputes P = 2^(S - LSE))
    }
}

// Workgroup order of the LONG launches (one wave group per query tile, >= 4096 workgroups, head dim >= 32): as in the backward (attention_bwd.hip) a workgroup
// takes query tile n - 1 - t and then tile t — n + 1 tile units each, so any order is balanced — and XCD x (the hardware deals consecutive workgroups to the
// 8 XCDs in turn) owns the (trajectory, head) pairs x, x + 8, ... and walks them pair-major: the 16 workgroups of a pair run together and the K / V^T rows they
// stream (0.26 MB per pair at head dim 32) stay in that XCD's L2.  The plain order (tile-major over all pairs) has every XCD stream 24 pairs' rows at once:
// PMC at cfg3's self launch 631 MB read for 99 MB of operands, 6.5 TB/s over its 96 us — the launch ran at the rate of the fabric.
template <typename T, int HD, int SPLIT, bool DROP>
__global__ __launch_bounds__(256 * SPLIT, (attn_min_waves<T, HD, DROP>())) void attention_fwd_kernel(const SeaAttnParams P, const int paired) {
    __shared__ __attribute__((aligned(16))) char smem_all[AttnFwdLds<T, HD, SPLIT>::BYTES];
    const int n_qt = (P.Tq + 63) / 64;
    int tile_, bh, zp;
    constexpr bool PAIRABLE = SPLIT == 1 && HD >= 32 && HD <= 64;   // (the other instantiations keep ONE copy of the tile's code: head dim 16 lives on a 64-register budget)
    if (PAIRABLE && paired && ((gridDim.y * gridDim.z) & 7) == 0) {
        const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int x = L & 7, j = L >> 3;
        const int pl = j / (int)gridDim.x;
        tile_ = j - pl * gridDim.x;
        const int pair = x + 8 * pl;
        zp = pair / (int)gridDim.y;
        bh = pair - zp * gridDim.y;
    } else {
        decode_attn_block(tile_, bh, zp);
    }
    attention_fwd_tile<T, HD, SPLIT, DROP>(P, smem_all, n_qt - 1 - tile_, bh, zp);  // heaviest (latest) query tiles first
    if constexpr (PAIRABLE) {
        if (paired && tile_ != n_qt - 1 - tile_) attention_fwd_tile<T, HD, SPLIT, DROP>(P, smem_all, tile_, bh, zp);   // (every wave has passed the tile loop's last barrier)
    }
}

template <typename T>
__device__ __forceinline__ void unpack16(const uint4& r, float (&o)[ActTraits<T>::EPC]);
template <>
__device__ __forceinline__ void unpack16<float>(const uint4& r, float (&o)[4]) {
    o[0] = __builtin_bit_cast(float, r.x); o[1] = __builtin_bit_cast(float, r.y); o[2] = __builtin_bit_cast(float, r.z); o[3] = __builtin_bit_cast(float, r.w);
}
template <>
__device__ __forceinline__ void unpack16<__bf16>(const uint4& r, float (&o)[8]) {   // bf16 -> f32 is a shift
    o[0] = __builtin_bit_cast(float, r.x << 16); o[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
    o[2] = __builtin_bit_cast(float, r.y << 16); o[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
    o[4] = __builtin_bit_cast(float, r.z << 16); o[5] = __builtin_bit_cast(float, r.z & 0xffff0000u);
    o[6] = __builtin_bit_cast(float, r.w << 16); o[7] = __builtin_bit_cast(float, r.w & 0xffff0000u);
}

// ---------------------------------------------------------------------------------------------- one query row per (trajectory, head)
// The KV-cache rollout step (Tq = 1): the tiled kernel above spends a 16-query MFMA tile and a serial walk over all key tiles of one workgroup
// on a single query.  Here a workgroup of 8 waves owns one (problem, trajectory, head):
//   phase 1  every thread scores keys tid, tid + 512, ... (q . K[key] in fp32, q pre-scaled), workgroup max and sum, probabilities to LDS;
//   phase 2  wave w owns head columns d = w, w + 8, ...; a lane walks 8-key vectors of row d of V^T (16-byte loads, consecutive lanes consecutive
//            vectors) against the probabilities in LDS, then a wave reduction per column.
// Same visibility rule as above (keys j <= q_pos0 + src_len, j < Tk); probabilities stay fp32 (the tiled kernel rounds them to the MFMA dtype).
template <typename T, int HD>
__global__ __launch_bounds__(512) void attention_row_kernel(const SeaAttnParams P) {
    constexpr int EPC = ActTraits<T>::EPC;                 // elements per 16 bytes
    constexpr int NW = 8, KPT = 16;                        // keys per thread in phase 1: up to 512 * 16 = 8192 keys
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);           // [2][NW] workgroup max / sum
    float* prob = red + 2 * NW + 16;                       // [round_up(nk, 8) + 8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, zp = blockIdx.y;
    const int b = bh / P.H, h = bh - b * P.H;
    const SeaAttnProblem& pr = P.p[zp];
    const T* Qg = static_cast<const T*>(pr.Q) + (int64_t)bh * HD;           // Tq = 1
    const T* Kg = static_cast<const T*>(pr.K) + (int64_t)bh * P.cap * HD;
    const T* Vg = static_cast<const T*>(pr.Vt) + (int64_t)bh * HD * P.cap;
    const int vis = P.q_pos0 + P.src_len + 1;
    const int nk = vis < P.Tk ? vis : P.Tk;                                  // visible keys 0 .. nk-1 (>= 1: key 0 is always visible)
    float q[HD];
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) load4(Qg + c * 4, *reinterpret_cast<float(*)[4]>(q + c * 4));
    // ---- phase 1: the rows of G keys are requested together (one memory round trip per group; all of a 2048-key cache is one group in bf16)
    constexpr int CPK = HD * (int)sizeof(T) / 16;          // 16-byte chunks per key row (hd = 8 bf16: one row is exactly 16 bytes)
    constexpr int G = CPK >= 16 ? 1 : (16 / CPK > 4 ? 4 : 16 / CPK);
    float sc[KPT];
    float mx = -INFINITY;
#pragma unroll
    for (int j0 = 0; j0 < KPT; j0 += G) {
        if (j0 * 512 >= nk) break;                         // block-uniform
        uint4 raw[G][CPK];
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            const int key = tid + (j0 + jj) * 512;
            const uint4* kr = reinterpret_cast<const uint4*>(Kg + (int64_t)(key < nk ? key : nk - 1) * HD);
#pragma unroll
            for (int c = 0; c < CPK; ++c) raw[jj][c] = kr[c];
        }
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            const int key = tid + (j0 + jj) * 512;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CPK; ++c) {
                float kv[EPC];
                unpack16<T>(raw[jj][c], kv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc += q[c * EPC + e] * kv[e];
            }
            sc[j0 + jj] = key < nk ? acc : -INFINITY;
            mx = fmaxf(mx, sc[j0 + jj]);
        }
    }
    mx = wave_max_xor(mx, lane);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int key = tid + j * 512;
        if (j * 512 >= nk) break;                          // block-uniform: sc[j] is not set beyond
        if (key < nk) {
            const float pv = __builtin_amdgcn_exp2f(sc[j] - m);   // scores are in log2 units (q carries hd^-1/2 * log2 e)
            prob[key] = pv;
            ls += pv;
        }
    }
    const int nk8 = (nk + 7) & ~7;
    if (tid < nk8 - nk) prob[nk + tid] = 0.f;              // the tail of the last 8-key vector
    ls = wave_sum_xor(ls, lane);
    if (lane == 0) red[NW + wave] = ls;
    __syncthreads();
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) l += red[NW + w];
    const float inv = 1.0f / l;
    // ---- phase 2
    const int nvec = nk8 >> 3;
    T* Og = static_cast<T*>(pr.O) + (int64_t)b * P.ldo + h * HD;              // Tq = 1: row b
    constexpr int DPW = HD / NW > 0 ? HD / NW : 1;                            // head columns per wave (hd = 8: waves 0..7 take one each)
    constexpr int VPC = 8 * (int)sizeof(T) / 16;                              // 16-byte chunks per 8-key vector of a V^T row
    float acc[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) acc[i] = 0.f;
    if (wave * DPW < HD) {
#pragma unroll 4
        for (int v = lane; v < nvec; v += 64) {
            uint4 raw[DPW][VPC];
#pragma unroll
            for (int i = 0; i < DPW; ++i) {
                const uint4* vr = reinterpret_cast<const uint4*>(Vg + (int64_t)(wave * DPW + i) * P.cap + v * 8);
#pragma unroll
                for (int c = 0; c < VPC; ++c) raw[i][c] = vr[c];
            }
            float pv[8];
            load4(prob + v * 8, *reinterpret_cast<float(*)[4]>(pv));
            load4(prob + v * 8 + 4, *reinterpret_cast<float(*)[4]>(pv + 4));
            const bool tail = v * 8 + 8 > nk;   // last vector: V^T beyond the visible keys is not data (0 * garbage must stay 0)
#pragma unroll
            for (int i = 0; i < DPW; ++i) {
                float vv[8];
#pragma unroll
                for (int c = 0; c < VPC; ++c) unpack16<T>(raw[i][c], *reinterpret_cast<float(*)[EPC]>(vv + c * EPC));
                if (tail) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) vv[e] = v * 8 + e < nk ? vv[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i] += pv[e] * vv[e];
            }
        }
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            acc[i] = wave_sum_xor(acc[i], lane);
            if (lane == 0) Og[wave * DPW + i] = from_f32<T>(acc[i] * inv);
        }
    }
    (void)EPC;
}

// The same for WIDE heads (64, and 128 / 256: the shipped widths, embed_dim 1024 / 2048 over 8 heads), where a thread cannot hold the query and a key row:
// HD / 32 adjacent lanes share a key (32 columns each, the partial dot products meet by lane shuffles), the raw scores go to LDS and are turned into
// probabilities by a second pass (no per-thread score array: any number of keys), and phase 2 walks a wave's HD / 8 value rows 8 at a time.
template <typename T, int HD>
__global__ __launch_bounds__(512) void attention_row_wide_kernel(const SeaAttnParams P) {
    constexpr int EPC = ActTraits<T>::EPC;
    constexpr int NW = 8, LPK = HD / 32, KPP = 512 / LPK, CPS = 32 / EPC;   // lanes per key, keys per pass, 16-byte chunks per 32-column slice
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);           // [2][NW] workgroup max / sum
    float* prob = red + 2 * NW + 16;                       // [round_up(nk, 8) + 8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, zp = blockIdx.y;
    const int b = bh / P.H, h = bh - b * P.H;
    const SeaAttnProblem& pr = P.p[zp];
    const T* Qg = static_cast<const T*>(pr.Q) + (int64_t)bh * HD;           // Tq = 1
    const T* Kg = static_cast<const T*>(pr.K) + (int64_t)bh * P.cap * HD;
    const T* Vg = static_cast<const T*>(pr.Vt) + (int64_t)bh * HD * P.cap;
    const int vis = P.q_pos0 + P.src_len + 1;
    const int nk = vis < P.Tk ? vis : P.Tk;
    const int part = tid % LPK, k_first = tid / LPK;
    constexpr int DPW = HD / NW;                           // value rows per wave (phase 2)
    const int nk8 = (nk + 7) & ~7, nvec = nk8 >> 3;        // 8-key vectors of a value row
    // Short caches in bf16 (<= 512 keys: one 8-key vector per lane): this lane's vector of EVERY value row of the wave is requested now, beside the query and
    // the key rows — the launch is then two memory round trips (keys, values in flight together) instead of 2 + DPW / 8 (a 100-step rollout at the multiphase
    // width: 19 us per launch at head dim 256, five dependent trips).  Addresses depend on nothing computed here.
    constexpr bool CAN_PRE = sizeof(T) == 2;
    const bool pre = CAN_PRE && nvec <= 64;                // block-uniform
    uint4 vpre[CAN_PRE ? DPW : 1];
    if (pre) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) vpre[i] = make_uint4(0u, 0u, 0u, 0u);
        if (lane < nvec) {   // only the lanes that hold a vector ask for one (a 100-key cache: 13 of 64 — the texture path's time goes with the lanes in flight)
#pragma unroll
            for (int i = 0; i < DPW; ++i) vpre[i] = *reinterpret_cast<const uint4*>(Vg + (int64_t)(wave * DPW + i) * P.cap + lane * 8);
        }
    }
    float q[32];
#pragma unroll
    for (int c = 0; c < 8; ++c) load4(Qg + part * 32 + c * 4, *reinterpret_cast<float(*)[4]>(q + c * 4));
    // ---- phase 1a: raw scores to LDS, the running maximum in registers
    float mx = -INFINITY;
    for (int key = k_first; key < nk; key += 2 * KPP) {   // two keys per iteration: both rows are requested before either is used
        uint4 raw[2][CPS];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kk = key + u * KPP;
#pragma unroll
            for (int c = 0; c < CPS; ++c) raw[u][c] = make_uint4(0u, 0u, 0u, 0u);
            if (kk < nk) {   // lanes without a key ask for nothing
                const uint4* kr = reinterpret_cast<const uint4*>(Kg + (int64_t)kk * HD + part * 32);
#pragma unroll
                for (int c = 0; c < CPS; ++c) raw[u][c] = kr[c];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kk = key + u * KPP;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CPS; ++c) {
                float kv[EPC];
                unpack16<T>(raw[u][c], kv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc += q[c * EPC + e] * kv[e];
            }
            if constexpr (LPK >= 2) acc += lane_xor<1>(acc, lane);
            if constexpr (LPK >= 4) acc += lane_xor<2>(acc, lane);
            if constexpr (LPK >= 8) acc += lane_xor<4>(acc, lane);
            if (kk < nk) {
                if (part == 0) prob[kk] = acc;
                mx = fmaxf(mx, acc);
            }
        }
    }
    mx = wave_max_xor(mx, lane);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
    // ---- phase 1b: probabilities in place
    float ls = 0.f;
    for (int key = tid; key < nk; key += 512) {
        const float pv = __builtin_amdgcn_exp2f(prob[key] - m);   // log2 units, as above
        prob[key] = pv;
        ls += pv;
    }
    if (tid < nk8 - nk) prob[nk + tid] = 0.f;              // the tail of the last 8-key vector
    ls = wave_sum_xor(ls, lane);
    if (lane == 0) red[NW + wave] = ls;
    __syncthreads();
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) l += red[NW + w];
    const float inv = 1.0f / l;
    // ---- phase 2: wave w owns value rows d = w * DPW .. + DPW - 1, eight at a time
    T* Og = static_cast<T*>(pr.O) + (int64_t)b * P.ldo + h * HD;
    constexpr int DCH = 8;
    constexpr int VPC = 8 * (int)sizeof(T) / 16;
    if constexpr (CAN_PRE) {
        if (pre) {
            float pv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // lanes beyond the last vector hold a clamped copy of it: zero weights
            if (lane < nvec) {
                load4(prob + lane * 8, *reinterpret_cast<float(*)[4]>(pv));
                load4(prob + lane * 8 + 4, *reinterpret_cast<float(*)[4]>(pv + 4));   // keys >= nk of the last vector: prob is 0 there
            }
            float acc[DPW];
#pragma unroll
            for (int i = 0; i < DPW; ++i) {
                float vv[8];
                unpack16<T>(vpre[i], vv);
                acc[i] = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i] += pv[e] * (lane * 8 + e < nk ? vv[e] : 0.f);   // 0 * garbage beyond nk must stay 0
            }
            // the DPW row sums of the wave by one halving exchange (DPW + 5 - log2 DPW shuffles instead of 6 DPW), row (lane / (64 / DPW)) left in lane `lane`:
            // one store instruction for the wave's rows instead of DPW
            const float tot = lane_scatter_sum<DPW>(acc, lane);
            constexpr int LS = 64 / DPW;
            if (lane % LS == 0) Og[wave * DPW + lane / LS] = from_f32<T>(tot * inv);
            return;
        }
    }
    for (int dc = 0; dc < DPW; dc += DCH) {
        float acc[DCH];
#pragma unroll
        for (int i = 0; i < DCH; ++i) acc[i] = 0.f;
        for (int v = lane; v < nvec; v += 64) {
            uint4 raw[DCH][VPC];
#pragma unroll
            for (int i = 0; i < DCH; ++i) {
                const uint4* vr = reinterpret_cast<const uint4*>(Vg + (int64_t)(wave * DPW + dc + i) * P.cap + v * 8);
#pragma unroll
                for (int c = 0; c < VPC; ++c) raw[i][c] = vr[c];
            }
            float pv[8];
            load4(prob + v * 8, *reinterpret_cast<float(*)[4]>(pv));
            load4(prob + v * 8 + 4, *reinterpret_cast<float(*)[4]>(pv + 4));
            const bool tail = v * 8 + 8 > nk;
#pragma unroll
            for (int i = 0; i < DCH; ++i) {
                float vv[8];
#pragma unroll
                for (int c = 0; c < VPC; ++c) unpack16<T>(raw[i][c], *reinterpret_cast<float(*)[EPC]>(vv + c * EPC));
                if (tail) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) vv[e] = v * 8 + e < nk ? vv[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i] += pv[e] * vv[e];
            }
        }
#pragma unroll
        for (int i = 0; i < DCH; ++i) {
            acc[i] = wave_sum_xor(acc[i], lane);
            if (lane == 0) Og[wave * DPW + dc + i] = from_f32<T>(acc[i] * inv);
        }
    }
}

template <typename T>
static bool launch_attention_row(const SeaAttnParams& P, hipStream_t s) {
    static const int on = sea_tune("attn_row", 1);  // tuning aid: 0 keeps the tiled kernel
    if (!on || P.Tq != 1 || P.drop.thr != 0 || P.Tk > 8192 || (P.hd != 8 && P.hd != 16 && P.hd != 32 && P.hd != 64 && P.hd != 128 && P.hd != 256)) return false;
    for (int i = 0; i < P.n_problems; ++i)
        if (P.p[i].LSE != nullptr) return false;
    const dim3 grid(P.B * P.H, P.n_problems), block(512);
    const int lds = (2 * 8 + 16 + ((P.Tk + 7) & ~7) + 8) * 4;
    switch (P.hd) {
        case 64: attention_row_wide_kernel<T, 64><<<grid, block, lds, s>>>(P); return true;   // (a thread that holds a 64-column query AND key rows spills: 879 registers in bf16)
        case 128: attention_row_wide_kernel<T, 128><<<grid, block, lds, s>>>(P); return true;
        case 256: attention_row_wide_kernel<T, 256><<<grid, block, lds, s>>>(P); return true;
        case 8: attention_row_kernel<T, 8><<<grid, block, lds, s>>>(P); break;
        case 16: attention_row_kernel<T, 16><<<grid, block, lds, s>>>(P); break;
        default: attention_row_kernel<T, 32><<<grid, block, lds, s>>>(P); break;
    }
    return true;
}

template <typename T, int SPLIT, bool DROP>
static int launch_attention_s(const SeaAttnParams& P, hipStream_t s) {
    dim3 grid((P.Tq + 63) / 64, P.B * P.H, P.n_problems);
    const dim3 block(256 * SPLIT);
    // long launches at head dims >= 32: paired causal tiles in the XCD-local order (attention_fwd_kernel).  SEA_TUNE=attn_paired=0|1 forces (read per call).
    const int forced = sea_tune("attn_paired", -1);
    const int paired = SPLIT == 1 && P.hd >= 32 && P.hd <= 64 && (forced >= 0 ? forced : (long)grid.x * grid.y * grid.z >= 4096);
    if (paired) grid.x = (grid.x + 1) / 2;
    switch (P.hd) {
        case 8: attention_fwd_kernel<T, 8, SPLIT, DROP><<<grid, block, 0, s>>>(P, paired); break;
        case 16: attention_fwd_kernel<T, 16, SPLIT, DROP><<<grid, block, 0, s>>>(P, paired); break;
        case 32: attention_fwd_kernel<T, 32, SPLIT, DROP><<<grid, block, 0, s>>>(P, paired); break;
        case 64: attention_fwd_kernel<T, 64, SPLIT, DROP><<<grid, block, 0, s>>>(P, paired); break;
        case 128: attention_fwd_kernel<T, 128, 1, DROP><<<grid, dim3(256), 0, s>>>(P, paired); break;  // LDS: one group only
        case 256: attention_fwd_kernel<T, 256, 1, DROP><<<grid, dim3(256), 0, s>>>(P, paired); break;  // the shipped multiphase dims (embed_dim 2048 / 8 heads); f32: single LDS buffer
        default: return -1;
    }
    return 0;
}

template <typename T>
static int launch_attention(const SeaAttnParams& P, hipStream_t s) {
    // few workgroups per CU and a long key range: split the key tiles of a query tile over two wave groups
    if (launch_attention_row<T>(P, s)) return 0;
    const long blocks = (long)((P.Tq + 63) / 64) * P.B * P.H * P.n_problems;
    const bool split = blocks <= 1024 && P.Tk >= 256;
    // at most two workgroups per CU: four wave groups per query tile (measured at cfg2: cross-attention 18.3 -> 17.3 us; with 768
    // workgroups the 1024-thread workgroups no longer co-reside and it is slower, 21.7 -> 24.7 us)
    static const int split4 = sea_tune("attn_split4", -1);  // tuning aid: 0 off, 1 on
    if (split && (split4 == 1 || (split4 < 0 && blocks <= 512)) && P.hd <= 32 && P.drop.thr == 0) {
        const dim3 grid((P.Tq + 63) / 64, P.B * P.H, P.n_problems), block(1024);
        if (P.hd == 32) attention_fwd_kernel<T, 32, 4, false><<<grid, block, 0, s>>>(P, 0);
        else if (P.hd == 16) attention_fwd_kernel<T, 16, 4, false><<<grid, block, 0, s>>>(P, 0);
        else attention_fwd_kernel<T, 8, 4, false><<<grid, block, 0, s>>>(P, 0);
        return 0;
    }
    if (P.drop.thr > 0) return split ? launch_attention_s<T, 2, true>(P, s) : launch_attention_s<T, 1, true>(P, s);
    return split ? launch_attention_s<T, 2, false>(P, s) : launch_attention_s<T, 1, false>(P, s);
}

extern "C" int sea_attention_fwd(const SeaAttnParams* params, int dtype, void* stream) {
    SEA_REQUIRE(params != nullptr, "sea_attention_fwd: null params");
    const SeaAttnParams& P = *params;
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_attention_fwd: bad dtype %d", dtype);
    SEA_REQUIRE(P.n_problems >= 1 && P.n_problems <= SEA_MAX_ATTN_PROBLEMS, "sea_attention_fwd: n_problems=%d", P.n_problems);
    SEA_REQUIRE(P.B >= 1 && P.H >= 1 && P.Tq >= 1 && P.Tk >= 1 && P.cap >= P.Tk && P.q_pos0 >= 0 && P.src_len >= 0,
                "sea_attention_fwd: bad sizes B=%d H=%d Tq=%d Tk=%d cap=%d q_pos0=%d src_len=%d", P.B, P.H, P.Tq, P.Tk, P.cap, P.q_pos0, P.src_len);
    SEA_REQUIRE(P.hd == 8 || P.hd == 16 || P.hd == 32 || P.hd == 64 || P.hd == 128 || P.hd == 256, "sea_attention_fwd: unsupported head dim %d (8..256, powers of two)", P.hd);
    SEA_REQUIRE(P.cap % 8 == 0, "sea_attention_fwd: cap=%d must be a multiple of 8", P.cap);
    SEA_REQUIRE(P.drop.thr >= 0 && P.drop.thr <= 255, "sea_attention_fwd: bad dropout threshold %d", P.drop.thr);
    SEA_REQUIRE(P.ldo >= P.H * P.hd && P.ldo % 4 == 0, "sea_attention_fwd: bad ldo=%d", P.ldo);
    SEA_REQUIRE((long)P.B * P.H <= 65535, "sea_attention_fwd: B*H too large for grid.y");
    for (int i = 0; i < P.n_problems; ++i) {
        SEA_REQUIRE(P.p[i].Q && P.p[i].K && P.p[i].Vt && P.p[i].O, "sea_attention_fwd[%d]: null pointer", i);
        SEA_REQUIRE(sea_aligned16(P.p[i].Q) && sea_aligned16(P.p[i].K) && sea_aligned16(P.p[i].Vt) && sea_aligned16(P.p[i].O),
                    "sea_attention_fwd[%d]: pointers must be 16-byte aligned", i);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rc = dtype == SEA_BF16 ? launch_attention<__bf16>(P, s) : launch_attention<float>(P, s);
    SEA_REQUIRE(rc == 0, "sea_attention_fwd: no kernel for hd=%d", P.hd);
    SEA_CHECK_LAUNCH("sea_attention_fwd");
    return SEA_OK;
}
