// The front of a SEA block in ONE launch (gfx950, bf16): sea_adaln_qkv.
//
//     h        = silu(cond_mlp.0(c))                         (models/base_blocks.py:337-338, 344: one scalar condition per row)
//     [w | b]  = h . W2c^T + b2c                             (cond_mlp.2)
//     y        = xhat(x) * (gamma + 1 + w) + (beta + b)      (AdaLN_0, :345-350)
//     q, k, v  = y . [Wq; Wk; Wv]^T + bias  -> rotary embedding on q / k, q scale, the attention layouts   (models/base_blocks.py:176-190, 300-324)
//
// Rounds 1-4 ran this as a silu launch (hidden rows of the condition MLP to memory), the condition GEMM (+ the normalisation as its epilogue, gemm_adaln.hip) and
// the QKV launch: 9 + 22 + 12 us at cfg2 in front of the first attention, each a chain of memory round trips on a chip that is a quarter full.  Here a workgroup
// (8 waves) owns 32 rows from the caller's fp32 rows to the attention operands:
//   - the hidden rows are GENERATED into LDS (32 x 512 silu per workgroup: each exactly once — a tiled GEMM regenerates them per column tile);
//   - layer A (K = 2E): wave w owns 32 scale and the SAME 32 shift columns of the modulation (weight rows chosen per wave), W2c through a private ring of
//     2 KiB slots per wave (LDS-DMA, counted), its part of the modulation goes to LDS as bf16 (what the two-launch form stores to memory);
//   - the row pass (16 threads per row): statistics of x by lane exchanges, y as bf16 into the operand tile of layer B;
//   - layer B (K = E): wave w owns 96 of the 3E output columns, [Wq; Wk; Wv] through the same rings; the epilogue stages the rotated / scaled values in LDS in
//     the attention layouts and the stores leave as whole rows (Q, K: 64 B of a (head, step) row per 4 lanes; V^T: 8 steps of a (head, dim) row per lane).
// Riders: 128 x 128 tiles of plain GEMMs of later launches (cond_mlp.2 of ln_cross) run on the CUs this launch leaves idle, as in chain.hip.
#include "gemm_tile.hpp"
#include "ib_rows.hpp"
#include <stdlib.h>

#define SEA_AQKV_MAX_SILU 8

struct AqkvLaunch {
    SeaAdalnQkv g[SEA_MAX_AQKV_GROUPS];
    int tile_start[SEA_MAX_AQKV_GROUPS + 1];
    int n_groups;
    float eps;
    SeaQkvCommon c;
    ChainRiderPod rg[SEA_CHAIN_MAX_RIDERS];
    int n_riders, rider_n;
    // row riders: hidden rows silu(cond_mlp.0(c)) of other modules (64 rows per workgroup and group) and the information-bottleneck rows (64 per workgroup), for
    // launches further down the step
    SeaSiluGroup sg[SEA_AQKV_MAX_SILU];
    int n_silu, silu_wgs, ib_wgs, M_r;
    const float* cond_r;
    SeaIbParams ib;
    unsigned long long* stamps;   // tuning aid (sea_aqkv_debug_stamps): 16 clock stamps (100 MHz) per workgroup, or NULL
    int probe;   // development: SEA_TUNE=aqkv_probe=n ends every workgroup after stage n (1 layer A, 2 the row pass, 3 layer B's loop); outputs are then not written
};
static_assert(sizeof(AqkvLaunch) <= 4096, "kernel arguments of sea_adaln_qkv");

static unsigned long long* g_aqkv_stamps = nullptr;
// Tuning aid, not part of the ABI proper (tools/aqkv_probe.py): a device buffer of (workgroups of the next launches) * 16 8-byte words receives time stamps of the phases.
extern "C" void sea_aqkv_debug_stamps(unsigned long long* buf) { g_aqkv_stamps = buf; }

// LDS-DMA piece: wave-uniform base (SGPR pair) + 32-bit per-lane byte offset -> 1 KiB at lds_addr (lane-linear)
__device__ __forceinline__ void glds16_aq(const void* ubase, unsigned lane_off, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(ubase), "s"(lds_addr) : "memory");
}

// wait until at most 2 * after pieces (the slots issued after the one about to be read) are outstanding; lgkmcnt(0): the fragment reads of the previous slot are retired
template <int AFTER>
__device__ __forceinline__ void aq_wait() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * AFTER) : "memory");
}
__device__ __forceinline__ void aq_wait_n(int after) {   // (constant per unrolled iteration)
    switch (after) {
        case 6: aq_wait<6>(); break;
        case 5: aq_wait<5>(); break;
        case 4: aq_wait<4>(); break;
        case 3: aq_wait<3>(); break;
        case 2: aq_wait<2>(); break;
        case 1: aq_wait<1>(); break;
        default: aq_wait<0>(); break;
    }
}

__device__ __forceinline__ f32x4 aq_mma_first(const uint4& a, const uint4& b) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}

struct AqkvCfg {
    static constexpr int E = 256, KC = 2 * E, BM = 32, NW = 8;
    static constexpr int KTA = KC / 64, KTB = E / 64;               // K-tiles of the two layers
    static constexpr int K3 = 256;                                  // optional third layer: [M, K3] generated rows x [K3, K3] (the ln_cross modulation at D = 128)
    static constexpr int HID_BYTES = KTA * BM * 128;                // 32 KiB: generated operand of layer A, K-tile major, swizzled; later the y tile (16 KiB) + rotary rows
    static constexpr int NE_OFF = 0, ROPE_OFF = KTB * BM * 128;     // y tile | (cos, sin) pairs of the workgroup's rows: hd / 2 pairs of 8 bytes per row at a pitch of + 16 bytes
    static constexpr int OUT_PAD = 16;                              // bytes added to a staged (head, row) of hd values: the accumulator layout's 16 rows then fall into different banks
    static constexpr int SLOT = 16 * 128, RSA = 8, RSB = 6;         // ring slots of a wave: layer A 8, layer B 6 (the last 4 KiB of its 16 hold its part of the modulation)
    static constexpr int RING_OFF = HID_BYTES, WAVE_RING = RSA * SLOT;
    static constexpr int MOD_OFF = RSB * SLOT;                      // inside a wave's ring region: 32 rows x (32 scale | 32 shift) bf16
    static constexpr int OUT_OFF = RING_OFF;                        // epilogue of layer B: [3][H][32 rows][hd] bf16 = 48 KiB (+ padding) over the drained rings
    static constexpr int OUT3_OFF = OUT_OFF + 3 * BM * (E * 2 + 16 * 16);   // ... and layer C's [32][K3] behind it
    static constexpr int BYTES = RING_OFF + NW * WAVE_RING;         // 160 KiB
    static_assert(BYTES <= 160 * 1024 && ROPE_OFF + BM * (16 * 8 + 16) <= HID_BYTES && MOD_OFF + BM * 128 <= WAVE_RING /* (rows at a pitch of 128 B: the spare 4 KiB hold exactly 32 of them) */ && OUT3_OFF + BM * (K3 * 2 + 16) <= BYTES && OUT_PAD == 16, "LDS plan of sea_adaln_qkv");
};

// HAS3: every group of the launch carries the optional third layer
template <bool HAS3>
__global__ __launch_bounds__(512) void adaln_qkv_kernel(const AqkvLaunch L) {
    using T = __bf16;
    using C = AqkvCfg;
    constexpr int E = C::E, KC = C::KC, BM = C::BM, NW = C::NW, KTA = C::KTA, KTB = C::KTB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int total = L.tile_start[L.n_groups];
    if ((int)blockIdx.x >= total + L.rider_n) {   // ---- row riders (block-uniform)
        const int x = (int)blockIdx.x - total - L.rider_n;
        const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
        if (x < L.silu_wgs) {   // 64 rows of one group: a wave takes 8 of them with its slice of w1 / b1 in registers (rowops.hip: silu_outer_kernel)
            const int per_g = (L.M_r + 63) >> 6;
            const int gy = x / per_g;
            const SeaSiluGroup& S = L.sg[gy];
            const int row0 = (x - gy * per_g) * 64 + wave_ * 8;
            if (row0 >= L.M_r) return;
            float cv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) cv[i] = L.cond_r[row0 + i < L.M_r ? row0 + i : L.M_r - 1];
            T* out = static_cast<T*>(S.Hid) + (int64_t)row0 * S.ld;
            for (int k = lane_ * 4; k < S.K2; k += 256) {
                float w[4], bb[4];
                load4(S.w1 + k, w);
                load4(S.b1 + k, bb);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (row0 + i < L.M_r)
                        store4(out + (int64_t)i * S.ld + k, silu_f(w[0] * cv[i] + bb[0]), silu_f(w[1] * cv[i] + bb[1]), silu_f(w[2] * cv[i] + bb[2]), silu_f(w[3] * cv[i] + bb[3]));
                }
            }
        } else {                // 64 information-bottleneck rows: a wave per row, 8 rows per wave
            const int row0 = (x - L.silu_wgs) * 64 + wave_ * 8;
            for (int i = 0; i < 8; ++i) {
                const int row = row0 + i;
                if (row < L.ib.M) ib_store_row(L.ib, L.cond_r[row], row, lane_);
            }
        }
        return;
    }
    if ((int)blockIdx.x >= total) {   // ---- a rider tile (block-uniform): four waves run it, the other four leave
        if (threadIdx.x >= 256) return;
        const int tile = (int)blockIdx.x - total;
        int ri = 0;
        while (ri + 1 < L.n_riders && tile >= L.rg[ri + 1].tile_start) ++ri;
        const ChainRiderPod& R = L.rg[ri];
        ChainRiderGroup G;
        G.A = R.A; G.W = R.W; G.bias = R.bias; G.Cact = R.Cact;
        G.lda = R.lda; G.ldw = R.ldw; G.ldcact = R.ldcact; G.M = R.M; G.N = R.N; G.K = R.K;
        const int t = tile - R.tile_start;
        const int tiles_n = (R.N + 127) / 128;
        gemm_tile_body<T, 128, 128, true, true, false>(G, t / tiles_n, t % tiles_n, smem, 0);
        return;
    }
    int gi = 0;
    while (gi + 1 < L.n_groups && (int)blockIdx.x >= L.tile_start[gi + 1]) ++gi;
    const SeaAdalnQkv& G = L.g[gi];
    const int m0 = ((int)blockIdx.x - L.tile_start[gi]) * BM, M = G.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int rl = lane >> 3, chunk = (lane & 7) ^ (rl & 7);
    unsigned long long* stp = L.stamps != nullptr ? L.stamps + (size_t)blockIdx.x * 16 : nullptr;
    int stn = 0;
    auto stamp = [&]() {
        if (stp != nullptr && tid == 0 && stn < 16) stp[stn] = wall_clock64();
        ++stn;
    };
    stamp();   // 0: entry
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const unsigned ring = lds_base + (unsigned)(C::RING_OFF + wave * C::WAVE_RING);
    const T* W2c = static_cast<const T*>(G.W2c);
    const T* Wq = static_cast<const T*>(G.Wqkv);
    const unsigned wa_lane = (unsigned)((rl * G.ldw2c + chunk * 8) * 2), wb_lane = (unsigned)((rl * G.ldw + chunk * 8) * 2);
    // layer A slot k = (kt, j): K-tile kt of the 16 modulation columns of block j of this wave — j = 0, 1: scale columns 32 w + 16 j ..; j = 2, 3: the shift
    // columns of the SAME outputs, E + 32 w + 16 (j - 2) ..
    auto slot_a = [&](int k) {
        const int kt = k >> 2, j = k & 3;
        const int row0 = (j < 2 ? 32 * wave + 16 * j : E + 32 * wave + 16 * (j - 2));
        const T* ub = W2c + (int64_t)row0 * G.ldw2c + kt * 64;   // uniform
        const unsigned dst = ring + (unsigned)((k % C::RSA) * C::SLOT);
        glds16_aq(ub, wa_lane, dst);
        glds16_aq(ub + (int64_t)8 * G.ldw2c, wa_lane, dst + 1024u);
    };
    // layer B slot k = (kt, j): K-tile kt of output columns 96 w + 16 j .. of [q | k | v]
    auto slot_b = [&](int k) {
        const int kt = k / 6, j = k - kt * 6;
        const T* ub = Wq + (int64_t)(96 * wave + 16 * j) * G.ldw + kt * 64;   // uniform
        const unsigned dst = ring + (unsigned)((k % C::RSB) * C::SLOT);
        glds16_aq(ub, wb_lane, dst);
        glds16_aq(ub + (int64_t)8 * G.ldw, wb_lane, dst + 1024u);
    };
    // optional layer C (W3 != NULL): slots behind layer B's in the same ring, k3 = (kt, j): K-tile kt of the 16 columns 32 w + 16 j .. of the third matrix
    constexpr bool has3 = HAS3;
    const T* W3 = static_cast<const T*>(G.W3);
    const unsigned wc_lane = (unsigned)((rl * G.ldw3 + chunk * 8) * 2);
    auto slot_c = [&](int k) {           // k: position in the ring sequence (layer B's 24 slots first)
        const int k3 = k - 6 * KTB, kt = k3 >> 1, j = k3 & 1;
        const T* ub = W3 + (int64_t)(32 * wave + 16 * j) * G.ldw3 + kt * 64;   // uniform
        const unsigned dst = ring + (unsigned)((k % C::RSB) * C::SLOT);
        glds16_aq(ub, wc_lane, dst);
        glds16_aq(ub + (int64_t)8 * G.ldw3, wc_lane, dst + 1024u);
    };
    // ---- the row pass's own operands (thread: row prow, 16 columns from pc0): requested FIRST, the weight slots behind them (see the hidden rows below)
    const int prow = tid >> 4, pl = tid & 15, pc0 = pl * 16;
    int mrow = m0 + prow;
    const bool rok = mrow < M;
    mrow = rok ? mrow : M - 1;
    float xv[16], gq[16], bq[16];
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
        load4(G.X + (int64_t)mrow * G.ldx + pc0 + c, *reinterpret_cast<float(*)[4]>(xv + c));
        load4(G.gamma + pc0 + c, *reinterpret_cast<float(*)[4]>(gq + c));
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[c + e] = 0.f;
        if (G.beta != nullptr) load4(G.beta + pc0 + c, *reinterpret_cast<float(*)[4]>(bq + c));
    }
    // ... and the (cos, sin) pair number pl of this row (hd / 2 pairs of 8 bytes per row, one per thread): staged in LDS by the row pass for layer B's epilogue
    const int H = L.c.H, hd = L.c.hd, Tlen = L.c.T, cap = L.c.cap, hd2 = hd >> 1;
    float2 cs_row = make_float2(1.f, 0.f);
    if (pl < hd2) {
        const int b_ = mrow / Tlen, t_ = mrow - b_ * Tlen;
        cs_row = reinterpret_cast<const float2*>(L.c.rope)[(int64_t)(L.c.pos0 + t_) * hd2 + pl];
    }
    // ... and the biases of this wave's 96 output columns of layer B (requested here they are older than every counted piece; requested at their use they were
    // a cold round trip in front of the epilogue)
    float bqv[6][4];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bqv[j][q] = 0.f;
        if (G.bqkv != nullptr) load4(G.bqkv + 96 * wave + 16 * j + 4 * g, bqv[j]);
    }
    float b2v[4][4];   // cond_mlp.2's bias for this wave's modulation columns
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) b2v[j][q] = 0.f;
        if (G.b2c != nullptr) load4(G.b2c + (j < 2 ? 32 * wave + 16 * j : E + 32 * wave + 16 * (j - 2)) + 4 * g, b2v[j]);
    }
    // layer C: this thread's 8 contraction indices (chunk tid & 31 of the 32) of w13 / b13, the condition of its two rows, the bias of this wave's 32 columns
    float w3v[8], b3v[8], c3v[2], b3o[2][4];
#pragma unroll
    for (int e = 0; e < 8; ++e) w3v[e] = b3v[e] = 0.f;
    c3v[0] = c3v[1] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) b3o[j][q] = 0.f;
    if (has3) {
        const int kc8 = tid & 31;
        load4(G.w13 + kc8 * 8, *reinterpret_cast<float(*)[4]>(w3v));
        load4(G.w13 + kc8 * 8 + 4, *reinterpret_cast<float(*)[4]>(w3v + 4));
        load4(G.b13 + kc8 * 8, *reinterpret_cast<float(*)[4]>(b3v));
        load4(G.b13 + kc8 * 8 + 4, *reinterpret_cast<float(*)[4]>(b3v + 4));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + (tid >> 5) + 16 * i;
            c3v[i] = G.cond[m < M ? m : M - 1];
        }
        if (G.b3 != nullptr) {
            load4(G.b3 + 32 * wave + 4 * g, b3o[0]);
            load4(G.b3 + 32 * wave + 16 + 4 * g, b3o[1]);
        }
    }
    // ---- the hidden rows of the condition MLP, generated: thread = 8 contraction indices (chunk kc8 of the 64) of the 4 rows 4 wave .. 4 wave + 3
    {
        const int kc8 = lane;                                  // 64 chunks of 8 = KC
        // w1 / b1 by loads the compiler does not track, the first weight slots behind them, and ONE counted wait that lets those 14 pieces fly on: a tracked load's
        // wait at its first use is a vmcnt(0) — the generation would start only when the whole ring prologue has landed (measured: 6 us from entry to the rows)
        f32x4 w1a, w1b, b1a, b1b;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w1a) : "v"(G.w1 + kc8 * 8) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w1b) : "v"(G.w1 + kc8 * 8 + 4) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b1a) : "v"(G.b1 + kc8 * 8) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b1b) : "v"(G.b1 + kc8 * 8 + 4) : "memory");
        float cvr[4];   // the condition of this wave's four rows (untracked loads as well: the compiler loads them per lane, not as scalars)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wave * 4 + i < M ? m0 + wave * 4 + i : M - 1;
            asm volatile("global_load_dword %0, %1, off" : "=v"(cvr[i]) : "v"(G.cond + m) : "memory");
        }
#pragma unroll
        for (int k = 0; k < C::RSA - 1; ++k) slot_a(k);
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(w1a), "+v"(w1b), "+v"(b1a), "+v"(b1b), "+v"(cvr[0]), "+v"(cvr[1]), "+v"(cvr[2]), "+v"(cvr[3]) : "n"(2 * (AqkvCfg::RSA - 1)) : "memory");
        const float w1[8] = {w1a[0], w1a[1], w1a[2], w1a[3], w1b[0], w1b[1], w1b[2], w1b[3]};
        const float b1[8] = {b1a[0], b1a[1], b1a[2], b1a[3], b1b[0], b1b[1], b1b[2], b1b[3]};
        const int kt = kc8 >> 3, ck = kc8 & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 4 + i;
            const float cv = cvr[i];
            bf16x8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (__bf16)silu_f(w1[e] * cv + b1[e]);
            *reinterpret_cast<bf16x8*>(smem + kt * (BM * 128) + row * 128 + ((ck ^ (row & 7)) << 4)) = hv;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the hidden rows are complete
    stamp();   // 1: hidden rows generated
    // ================================================================================================ layer A: the modulation of this wave's 32 outputs
    f32x4 acc[2][4];
    {
        uint4 af[2][2];
#pragma unroll
        for (int k = 0; k < 4 * KTA; ++k) {
            const int kt = k >> 2, j = k & 3;
            aq_wait_n(4 * KTA - 1 - k < C::RSA - 2 ? 4 * KTA - 1 - k : C::RSA - 2);
            if (k + C::RSA - 1 < 4 * KTA) slot_a(k + C::RSA - 1);
            if (j == 0) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc)
                        af[mb][kc] = *reinterpret_cast<const uint4*>(smem + kt * (BM * 128) + (mb * 16 + r) * 128 + (((kc * 4 + g) ^ (r & 7)) << 4));
            }
            const char* sl = smem + C::RING_OFF + wave * C::WAVE_RING + (k % C::RSA) * C::SLOT + r * 128;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const uint4 w = *reinterpret_cast<const uint4*>(sl + (((kc * 4 + g) ^ (r & 7)) << 4));
                if (kt == 0 && kc == 0) {
                    acc[0][j] = aq_mma_first(w, af[0][0]);
                    acc[1][j] = aq_mma_first(w, af[1][0]);
                } else {
                    mma16<T>(w, af[0][kc], acc[0][j]);
                    mma16<T>(w, af[1][kc], acc[1][j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    stamp();   // 2: layer A's loop
    // Every value the compiler loaded itself at the start is "used" HERE, where the loop's last wait has emptied the memory queue anyway: its wait for a load sits
    // at the first use and counts only the loads it issued itself — at a first use further down it would drain the counted LDS-DMA pieces in flight there
    // (seen: 1.5 us in front of the row pass, whose x / gamma / beta had landed ten microseconds earlier).
#pragma unroll
    for (int c = 0; c < 16; ++c) asm volatile("" ::"v"(xv[c]), "v"(gq[c]), "v"(bq[c]));
#pragma unroll
    for (int j = 0; j < 6; ++j) asm volatile("" ::"v"(bqv[j][0]), "v"(bqv[j][1]), "v"(bqv[j][2]), "v"(bqv[j][3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(b2v[j][0]), "v"(b2v[j][1]), "v"(b2v[j][2]), "v"(b2v[j][3]));
    asm volatile("" ::"v"(cs_row.x), "v"(cs_row.y), "v"(c3v[0]), "v"(c3v[1]));
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(w3v[e]), "v"(b3v[e]));
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(b3o[j][0]), "v"(b3o[j][1]), "v"(b3o[j][2]), "v"(b3o[j][3]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads are retired: its ring takes layer B's first slots
#pragma unroll
    for (int k = 0; k < C::RSB - 1; ++k) slot_b(k);
    // this wave's part of the modulation, + bias, as bf16 (the precision the two-launch form stores): row mb * 16 + r, columns [scale 32 | shift 32]
    {
        char* mp = smem + C::RING_OFF + wave * C::WAVE_RING + C::MOD_OFF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                store4(reinterpret_cast<T*>(mp + (mb * 16 + r) * 128 + (16 * j + 4 * g) * 2), acc[mb][j][0] + b2v[j][0], acc[mb][j][1] + b2v[j][1], acc[mb][j][2] + b2v[j][2],
                       acc[mb][j][3] + b2v[j][3]);
        }
    }
    if (L.probe == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave's modulation part is in LDS, every wave is done with the hidden rows
    stamp();   // 3: modulation parts exchanged
    // ================================================================================================ the row pass: AdaLN_0 of 32 rows, 16 threads each
    {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 16; c += 4)
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e] = add1(s4[e], xv[c + e]);
        float sm_ = add1(add1(s4[0], s4[1]), add1(s4[2], s4[3]));
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) sm_ += __shfl_xor(sm_, o);
        const float mean = sm_ * (1.0f / (float)E);
        float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 16; c += 4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d_ = xv[c + e] - mean;
                q4[e] = fma1(d_, d_, q4[e]);
            }
        float sq = add1(add1(q4[0], q4[1]), add1(q4[2], q4[3]));
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / (float)E) + L.eps);
        // output columns pc0 .. pc0 + 15 are columns 16 (pl & 1) .. of the part of wave pl >> 1
        const char* mp = smem + C::RING_OFF + (pl >> 1) * C::WAVE_RING + C::MOD_OFF + prow * 128 + (pl & 1) * 32;
        float mw[16], mb[16];
        load8(reinterpret_cast<const T*>(mp), *reinterpret_cast<float(*)[8]>(mw));
        load8(reinterpret_cast<const T*>(mp + 16), *reinterpret_cast<float(*)[8]>(mw + 8));
        load8(reinterpret_cast<const T*>(mp + 64), *reinterpret_cast<float(*)[8]>(mb));
        load8(reinterpret_cast<const T*>(mp + 80), *reinterpret_cast<float(*)[8]>(mb + 8));
#pragma unroll
        for (int c = 0; c < 16; c += 8) {
            const int col = pc0 + c, kt = col >> 6, ck = (col & 63) >> 3;
            bf16x8 pv;
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = (__bf16)((xv[c + e] - mean) * rstd * (gq[c + e] + 1.0f + mw[c + e]) + (bq[c + e] + mb[c + e]));
            *reinterpret_cast<bf16x8*>(smem + C::NE_OFF + kt * (BM * 128) + prow * 128 + ((ck ^ (prow & 7)) << 4)) = pv;
        }
        if (pl < hd2) *reinterpret_cast<float2*>(smem + C::ROPE_OFF + prow * (hd2 * 8 + 16) + pl * 8) = cs_row;
    }
    // (the hidden rows' memory now holds y: the overwrite above follows the barrier behind layer A, after which no wave reads the hidden rows)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (L.probe == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    stamp();   // 4: row pass
    // layer C's operand: the hidden rows silu(w13 * c + b13) of the third matrix, K-tile kt into the spare 4 KiB of wave kt's ring region (the modulation parts
    // the row pass has just read); consumed behind layer B's slots, after the barrier in front of slot 6 KTB
    if (has3) {
        const int kc8 = tid & 31, kt = kc8 >> 3, ck = kc8 & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 5) + 16 * i;
            bf16x8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (__bf16)silu_f(w3v[e] * c3v[i] + b3v[e]);
            *reinterpret_cast<bf16x8*>(smem + C::RING_OFF + kt * C::WAVE_RING + C::MOD_OFF + row * 128 + ((ck ^ (row & 7)) << 4)) = hv;
        }
    }
    // ================================================================================================ layer B: 96 columns of [q | k | v] per wave (+ layer C: 32 columns)
    f32x4 acc2[2][6], acc3[2][2];
    {
        uint4 af[2][2];
        constexpr int NTOT = 6 * KTB + (has3 ? 2 * (AqkvCfg::K3 / 64) : 0);
#pragma unroll
        for (int k = 0; k < NTOT; ++k) {
            const int after_ = NTOT - 1 - k < C::RSB - 2 ? NTOT - 1 - k : C::RSB - 2;
            if (k >= 6 * KTB) {   // ---- a slot of layer C
                const int k3 = k - 6 * KTB, kt = k3 >> 1, j = k3 & 1;
                aq_wait_n(after_);
                if (k3 == 0) __builtin_amdgcn_s_barrier();   // every wave's share of the generated rows is in LDS (their writes were retired by the wait above)
                if (k + C::RSB - 1 < NTOT) slot_c(k + C::RSB - 1);
                if (j == 0) {
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int kc = 0; kc < 2; ++kc)
                            af[mb][kc] = *reinterpret_cast<const uint4*>(smem + C::RING_OFF + kt * C::WAVE_RING + C::MOD_OFF + (mb * 16 + r) * 128 + (((kc * 4 + g) ^ (r & 7)) << 4));
                }
                const char* sl = smem + C::RING_OFF + wave * C::WAVE_RING + (k % C::RSB) * C::SLOT + r * 128;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const uint4 w = *reinterpret_cast<const uint4*>(sl + (((kc * 4 + g) ^ (r & 7)) << 4));
                    if (kt == 0 && kc == 0) {
                        acc3[0][j] = aq_mma_first(w, af[0][0]);
                        acc3[1][j] = aq_mma_first(w, af[1][0]);
                    } else {
                        mma16<T>(w, af[0][kc], acc3[0][j]);
                        mma16<T>(w, af[1][kc], acc3[1][j]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            const int kt = k / 6, j = k - kt * 6;
            aq_wait_n(after_);
            if (k + C::RSB - 1 < NTOT) {
                if (k + C::RSB - 1 < 6 * KTB) slot_b(k + C::RSB - 1);
                else slot_c(k + C::RSB - 1);
            }
            if (j == 0) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc)
                        af[mb][kc] = *reinterpret_cast<const uint4*>(smem + C::NE_OFF + kt * (BM * 128) + (mb * 16 + r) * 128 + (((kc * 4 + g) ^ (r & 7)) << 4));
            }
            const char* sl = smem + C::RING_OFF + wave * C::WAVE_RING + (k % C::RSB) * C::SLOT + r * 128;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const uint4 w = *reinterpret_cast<const uint4*>(sl + (((kc * 4 + g) ^ (r & 7)) << 4));
                if (kt == 0 && kc == 0) {
                    acc2[0][j] = aq_mma_first(w, af[0][0]);
                    acc2[1][j] = aq_mma_first(w, af[1][0]);
                } else {
                    mma16<T>(w, af[0][kc], acc2[0][j]);
                    mma16<T>(w, af[1][kc], acc2[1][j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    stamp();   // 5: layer B's loop (this wave)
    __syncthreads();   // every wave is done with its ring: the staged outputs go over that memory
    stamp();   // 6: ... every wave
    if (L.probe == 3) return;
    // ---- bias, rotary embedding on q / k, q scale -> bf16, staged as [part][head][row][hd]
    {
        const float q_scale = L.c.q_scale;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int nb = 96 * wave + 16 * j;                  // first column of the block in [q | k | v] (wave-uniform)
            const int part = nb >> 8, hb = nb & (E - 1);
            const int h = hd == 32 ? hb >> 5 : hb >> 4, dd = (hb & (hd - 1)) + 4 * g;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int row = mb * 16 + r;
                float v[4], o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc2[mb][j][q] + bqv[j][q];
                if (part < 2) {   // (wave-uniform)
                    const float4 cs = *reinterpret_cast<const float4*>(smem + C::ROPE_OFF + row * (hd2 * 8 + 16) + (dd >> 1) * 8);
                    rope_pair(v[0], v[1], cs.x, cs.y, o[0], o[1]);
                    rope_pair(v[2], v[3], cs.z, cs.w, o[2], o[3]);
                    if (part == 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] *= q_scale;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = v[q];
                }
                store4(reinterpret_cast<T*>(smem + C::OUT_OFF + ((part * H + h) * BM + row) * (hd * 2 + C::OUT_PAD) + dd * 2), o[0], o[1], o[2], o[3]);
            }
        }
    }
    if (has3) {   // layer C's 32 columns of this wave, + bias, bf16, behind the staged q | k | v: [32 rows][K3 columns] at a pitch of + 16 bytes
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                store4(reinterpret_cast<T*>(smem + C::OUT3_OFF + (mb * 16 + r) * (C::K3 * 2 + 16) + (32 * wave + 16 * j + 4 * g) * 2), acc3[mb][j][0] + b3o[j][0], acc3[mb][j][1] + b3o[j][1],
                       acc3[mb][j][2] + b3o[j][2], acc3[mb][j][3] + b3o[j][3]);
    }
    stamp();   // 7: staged (this wave)
    __syncthreads();
    if (L.probe == 4) return;
    stamp();   // 8: ... every wave
    // ---- Q and K rows: item = (part, head, row, 16-byte piece of the head's hd values); a wave instruction covers 64 / (hd / 8) consecutive rows of one head
    {
        const int cs_ = hd == 32 ? 2 : 1;                       // log2 of the pieces per (head, row): hd / 8
        const int n_items = (2 * H * BM) << cs_;
        T* Qo = static_cast<T*>(G.Q);
        T* Ko = static_cast<T*>(G.K);
        for (int id = tid; id < n_items; id += 512) {
            const int ph = id >> (5 + cs_), rem = id & ((32 << cs_) - 1);
            const int row = rem >> cs_, pc = rem & ((1 << cs_) - 1);
            const int m = m0 + row;
            if (m >= M) continue;
            const int part = ph >= H ? 1 : 0, h = ph - part * H;
            const int b_ = m / Tlen, t_ = m - b_ * Tlen;
            const uint4 val = *reinterpret_cast<const uint4*>(smem + C::OUT_OFF + (ph * BM + row) * (hd * 2 + C::OUT_PAD) + pc * 16);
            const uint32_t bh = (uint32_t)(b_ * H + h);
            if (part == 0) *reinterpret_cast<uint4*>(Qo + ((bh * (uint32_t)Tlen + t_) * (uint32_t)hd + pc * 8)) = val;
            else *reinterpret_cast<uint4*>(Ko + ((bh * (uint32_t)cap + (uint32_t)(L.c.pos0 + t_)) * (uint32_t)hd + pc * 8)) = val;
        }
    }
    stamp();   // 9: Q / K rows stored
    if (L.probe == 5) return;
    // ---- V^T [B, H, hd, cap]: item = (head, dim, 8 consecutive rows): the 8 values gathered from the staged tile, one 16-byte store when they are 8 steps of one
    // trajectory at an aligned position, element stores otherwise
    {
        T* Vto = static_cast<T*>(G.Vt);
        const int n_items = H * hd * (BM / 8);
        for (int id = tid; id < n_items; id += 512) {
            const int tc = id & 3, hdd = id >> 2;               // 8-row piece, (head, dim)
            const int hs_ = hd == 32 ? 5 : 4;
            const int h = hdd >> hs_, dd = hdd & (hd - 1);
            const int mA = m0 + tc * 8;
            if (mA >= M) continue;
            const char* src = smem + C::OUT_OFF + ((2 * H + h) * BM + tc * 8) * (hd * 2 + C::OUT_PAD) + dd * 2;
            alignas(16) __bf16 vals[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) vals[i] = *reinterpret_cast<const __bf16*>(src + i * (hd * 2 + C::OUT_PAD));
            const int b_ = mA / Tlen, t_ = mA - b_ * Tlen;
            const uint32_t pos = (uint32_t)(L.c.pos0 + t_);
            T* dst = Vto + ((uint32_t)(b_ * H + h) * (uint32_t)hd + dd) * (uint32_t)cap + pos;
            if (mA + 7 < M && t_ + 7 < Tlen && (pos & 7u) == 0u) {
                *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(vals);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = mA + i;
                    if (m >= M) break;
                    const int bi = m / Tlen, ti = m - bi * Tlen;
                    Vto[((uint32_t)(bi * H + h) * (uint32_t)hd + dd) * (uint32_t)cap + (uint32_t)(L.c.pos0 + ti)] = vals[i];
                }
            }
        }
    }
    if (has3) {   // the third layer's rows: 32 pieces of 16 bytes per row
        T* M3 = static_cast<T*>(G.mod3);
        for (int id = tid; id < BM * (C::K3 / 8); id += 512) {
            const int row = id >> 5, pc = id & 31;
            const int m = m0 + row;
            if (m < M) *reinterpret_cast<uint4*>(M3 + (int64_t)m * G.ldmod3 + pc * 8) = *reinterpret_cast<const uint4*>(smem + C::OUT3_OFF + row * (C::K3 * 2 + 16) + pc * 16);
        }
    }
    stamp();   // 10: V^T stored (issued)
}

extern "C" int sea_adaln_qkv(const SeaAdalnQkv* groups, int n_groups, const SeaQkvCommon* common, const SeaGemmGroup* riders, int n_riders, const SeaSiluGroup* silu, int n_silu,
                             const float* silu_c, int silu_M, const SeaIbParams* ib, float eps, int dtype, void* stream) {
    SEA_REQUIRE(groups != nullptr && common != nullptr && n_groups >= 1 && n_groups <= SEA_MAX_AQKV_GROUPS, "sea_adaln_qkv: n_groups=%d out of range", n_groups);
    const int E = groups[0].E;
    if (dtype != SEA_BF16 || E != AqkvCfg::E || common->H * common->hd != E || (common->hd != 32 && common->hd != 16)) {
        sea_set_error("sea_adaln_qkv: unsupported dtype / shape (dtype=%d E=%d H=%d hd=%d): bf16, E = 256, head dim 16 or 32, H * hd = E", dtype, E, common->H, common->hd);
        return SEA_EUNSUPPORTED;
    }
    SEA_REQUIRE(common->rope && common->T >= 1 && common->cap >= common->pos0 + common->T && common->pos0 >= 0 && common->cap % 8 == 0, "sea_adaln_qkv: bad rotary table / cache geometry (T=%d pos0=%d cap=%d)",
                common->T, common->pos0, common->cap);
    AqkvLaunch L;
    memset(&L, 0, sizeof(L));
    int total = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaAdalnQkv& G = groups[i];
        SEA_REQUIRE(G.E == E && G.M >= 1 && G.M % common->T == 0, "sea_adaln_qkv[%d]: the groups of a launch share E; M = B * T", i);
        SEA_REQUIRE(G.X && G.cond && G.w1 && G.b1 && G.W2c && G.gamma && G.Wqkv && G.Q && G.K && G.Vt, "sea_adaln_qkv[%d]: null pointer", i);
        SEA_REQUIRE(G.ldx % 4 == 0 && G.ldx >= E && G.ldw2c % 8 == 0 && G.ldw2c >= 2 * E && G.ldw % 8 == 0 && G.ldw >= E, "sea_adaln_qkv[%d]: bad strides", i);
        SEA_REQUIRE(sea_aligned16(G.X) && sea_aligned16(G.w1) && sea_aligned16(G.b1) && sea_aligned16(G.W2c) && sea_aligned16(G.b2c) && sea_aligned16(G.gamma) && sea_aligned16(G.beta) &&
                        sea_aligned16(G.Wqkv) && sea_aligned16(G.bqkv) && sea_aligned16(G.Q) && sea_aligned16(G.K) && sea_aligned16(G.Vt),
                    "sea_adaln_qkv[%d]: pointers must be 16-byte aligned", i);
        SEA_REQUIRE((int64_t)(G.M / common->T) * common->H * common->cap * common->hd < (1ll << 31), "sea_adaln_qkv[%d]: attention tensors must stay below 2^31 elements", i);
        if (G.W3 != nullptr) {
            SEA_REQUIRE(G.w13 && G.b13 && G.mod3 && G.N3 == AqkvCfg::K3 && G.ldw3 % 8 == 0 && G.ldw3 >= G.N3 && G.ldmod3 % 8 == 0 && G.ldmod3 >= G.N3 && sea_aligned16(G.w13) && sea_aligned16(G.b13) &&
                            sea_aligned16(G.W3) && sea_aligned16(G.b3) && sea_aligned16(G.mod3),
                        "sea_adaln_qkv[%d]: third layer: null / misaligned pointer, bad stride, or N3 = %d != 256", i, G.N3);
        }
        L.g[i] = G;
        L.tile_start[i] = total;
        total += (G.M + 31) / 32;
    }
    L.tile_start[n_groups] = total;
    L.n_groups = n_groups;
    L.eps = eps;
    L.c = *common;
    SEA_REQUIRE(n_riders >= 0 && n_riders <= SEA_CHAIN_MAX_RIDERS && (n_riders == 0 || riders != nullptr), "sea_adaln_qkv: bad rider arguments");
    int rider_total = 0;
    for (int i = 0; i < n_riders; ++i) {
        const SeaGemmGroup& G = riders[i];
        SEA_REQUIRE(G.A && !G.silu_c && G.W && G.Cact && !G.C32 && !G.R && !G.Z && G.act == 0 && G.drop.thr == 0 && G.n_seg == 1, "sea_adaln_qkv: rider %d is not a plain bf16 group (A, W, bias, Cact only)", i);
        SEA_REQUIRE(G.M >= 1 && G.N >= 8 && G.N % 8 == 0 && G.K >= 64 && G.K % 64 == 0 && G.lda % 8 == 0 && G.lda >= G.K && G.ldw % 8 == 0 && G.ldw >= G.K && G.ldcact % 8 == 0 && G.ldcact >= G.N,
                    "sea_adaln_qkv: rider %d: bad shape M=%d N=%d K=%d (whole 64-wide K-tiles) lda=%d ldw=%d ldcact=%d", i, G.M, G.N, G.K, G.lda, G.ldw, G.ldcact);
        SEA_REQUIRE(sea_aligned16(G.A) && sea_aligned16(G.W) && sea_aligned16(G.bias) && sea_aligned16(G.Cact), "sea_adaln_qkv: rider %d: pointers must be 16-byte aligned", i);
        ChainRiderPod& R = L.rg[i];
        R.A = G.A; R.W = G.W; R.bias = G.bias; R.Cact = G.Cact;
        R.lda = G.lda; R.ldw = G.ldw; R.ldcact = G.ldcact; R.M = G.M; R.N = G.N; R.K = G.K;
        R.tile_start = rider_total;
        rider_total += ((G.M + 127) / 128) * ((G.N + 127) / 128);
    }
    L.n_riders = n_riders;
    L.rider_n = rider_total;
    SEA_REQUIRE(n_silu >= 0 && n_silu <= SEA_AQKV_MAX_SILU && (n_silu == 0 || (silu != nullptr && silu_c != nullptr && silu_M >= 1)), "sea_adaln_qkv: bad silu rider arguments (n_silu=%d)", n_silu);
    // (the information-bottleneck rows are evaluated on silu_c too: SeaIbParams.c is not read)
    for (int i = 0; i < n_silu; ++i) {
        SEA_REQUIRE(silu[i].w1 && silu[i].b1 && silu[i].Hid && silu[i].K2 >= 4 && silu[i].K2 % 4 == 0 && silu[i].ld >= silu[i].K2 && silu[i].ld % 4 == 0 && sea_aligned16(silu[i].w1) &&
                        sea_aligned16(silu[i].b1) && sea_aligned16(silu[i].Hid),
                    "sea_adaln_qkv: silu rider %d: null / misaligned pointer or bad shape", i);
        L.sg[i] = silu[i];
    }
    L.n_silu = n_silu;
    L.cond_r = silu_c;
    L.M_r = silu_M;
    L.silu_wgs = n_silu * ((silu_M + 63) / 64);
    if (ib != nullptr) {
        SEA_REQUIRE(silu_c != nullptr && silu_M == ib->M && ib->X[0] && ib->w1 && ib->M >= 1 && ib->E >= 4 && ib->E % 4 == 0 && ib->ldx >= ib->E && ib->mode >= 0 && ib->mode <= 2 &&
                        (ib->mode != 0 || (ib->b1 && ib->lnw && ib->lnb && ib->w2 && ib->b2 && ib->h >= 1 && ib->h <= 64)),
                    "sea_adaln_qkv: bad info-bottleneck rider");
        L.ib = *ib;
        L.ib_wgs = (ib->M + 63) / 64;
    }
    static const int probe = sea_tune("aqkv_probe", 0);
    L.probe = probe;
    L.stamps = g_aqkv_stamps;
    constexpr int rider_lds = GemmMainloop<__bf16, 128, 128>::DMA_LDS_BYTES;
    constexpr int lds = AqkvCfg::BYTES > rider_lds ? AqkvCfg::BYTES : rider_lds;
    const bool has3 = groups[0].W3 != nullptr;
    for (int i = 1; i < n_groups; ++i) SEA_REQUIRE((groups[i].W3 != nullptr) == has3, "sea_adaln_qkv[%d]: the third layer in every group of a launch or in none", i);
    const dim3 grid(total + rider_total + L.silu_wgs + L.ib_wgs);
    if (has3) {
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(adaln_qkv_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)once;
        adaln_qkv_kernel<true><<<grid, dim3(512), lds, static_cast<hipStream_t>(stream)>>>(L);
    } else {
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(adaln_qkv_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)once;
        adaln_qkv_kernel<false><<<grid, dim3(512), lds, static_cast<hipStream_t>(stream)>>>(L);
    }
    SEA_CHECK_LAUNCH("sea_adaln_qkv");
    return SEA_OK;
}
