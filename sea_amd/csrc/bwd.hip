// Backward-pass kernels of the SEA temporal path other than attention (gfx950):
//   sea_wgrad_grouped      dW[N,K] += dY[M,N]^T . X[M,K], db[N] += colsum(dY)       (TN GEMM, split over M, fp32 atomics)
//   sea_transpose_weights  act-dtype W^T shadow of every weight matrix (operand of the data-gradient GEMMs)
//   sea_rownorm_bwd        backward of sea_rownorm (AdaLN / LayerNorm / LayerNorm+GELU)
//   sea_silu_outer_bwd     backward of sea_silu_outer
//   sea_ib_bwd             parameter gradients of the information-bottleneck MLP
#include "sea_common.hpp"
#include <stdlib.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ transposed fragments
// A tile stored [m][c] (contraction index m is the ROW) must feed an MFMA operand that wants, per lane, EPC consecutive
// m for one column c0 + (lane & 15).  bf16: two ds_read_b64_tr_b16 (each delivers a 4-row x 16-column block
// column-major: lane 4q+p of a 16-lane group supplies the address of row q, columns 4p..4p+3 and receives column
// (lane & 15) of the 4 rows; cdna_hip_programming.md T10).  f32: four 4-byte reads.
template <typename T>
__device__ __forceinline__ uint4 load_frag_T(const char* tile, int pitch, int m0, int c0, int lane);

template <>
__device__ __forceinline__ uint4 load_frag_T<__bf16>(const char* tile, int pitch, int m0, int c0, int lane) {
    const int idx = lane & 15, gg = lane >> 4, q = idx >> 2, p = idx & 3;
    const char* base = tile + (m0 + 8 * gg + q) * pitch + (c0 + 4 * p) * 2;
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + 4 * pitch));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);   // already packed pairs
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}
template <>
__device__ __forceinline__ uint4 load_frag_T<float>(const char* tile, int pitch, int m0, int c0, int lane) {
    const int idx = lane & 15, gg = lane >> 4;
    const char* base = tile + (m0 + 4 * gg) * pitch + (c0 + idx) * 4;
    f32x4 v = {*reinterpret_cast<const float*>(base), *reinterpret_cast<const float*>(base + pitch),
               *reinterpret_cast<const float*>(base + 2 * pitch), *reinterpret_cast<const float*>(base + 3 * pitch)};
    return __builtin_bit_cast(uint4, v);
}

// ------------------------------------------------------------------------------------------------ weight gradient (TN GEMM)
#define SEA_MAX_WGRAD_GROUPS 32
struct WgradLaunch {
    SeaWgradGroup g[SEA_MAX_WGRAD_GROUPS];
    int tile_start[SEA_MAX_WGRAD_GROUPS + 1];  // in (tile x split) units
    int splits[SEA_MAX_WGRAD_GROUPS];
    int rows_per_split[SEA_MAX_WGRAD_GROUPS];
    int n_groups;
    int xcd;   // 1: XCD-contiguous work order (see the kernel)
    int stage_out;   // 1: plain-stored full tiles leave through LDS as whole rows (SEA_TUNE=wgrad_stage=0: one dword per lane, the form of rounds 1-4)
};

// Output tile TN (n) x TK (k) per workgroup, one WT x WT tile (WT / 16 squared MFMA tiles) per wave: 64 x 64 (4 waves of 32 x 32) or 128 x 128
// (4 waves of 64 x 64).  The contraction runs over 64-row stages of dY and X held
// [m][columns] in LDS (padded row pitch), double-buffered through registers like the forward GEMM.
// LDS row padding: the transposed bf16 fragment read takes, per 16-lane group, 32 bytes of each of 4 consecutive rows — a pitch of 8 dwords mod 64
// banks puts those on 32 distinct banks; the f32 read takes 64 bytes of each of two rows 4 apart per LDS cycle: 4 dwords mod 32 per row.  (Shifting
// every 8-row block by another 128 bytes, so that the two 16-lane groups of an LDS cycle never meet, changed neither SQ_LDS_BANK_CONFLICT — what it
// counts here are the extra array cycles of the 16-byte stores — nor the time.)
template <typename T>
constexpr int WGRAD_PAD = sizeof(T) == 2 ? 32 : 16;

template <typename T, int TN, int TK, int WT>
__global__ __launch_bounds__((TN / WT) * (TK / WT) * 64) void wgrad_kernel(const WgradLaunch L) {
    constexpr int EPC = ActTraits<T>::EPC, CK = ActTraits<T>::CK;
    constexpr int WK = TK / WT, NT = (TN / WT) * WK * 64;   // waves along k, threads
    constexpr int MI = WT / 16;                    // 16 x 16 MFMA tiles per wave per dimension
    constexpr int PAD = WGRAD_PAD<T>;
    constexpr int PITCH_Y = TN * (int)sizeof(T) + PAD, PITCH_X = TK * (int)sizeof(T) + PAD;
    constexpr int CPR_Y = TN * (int)sizeof(T) / 16, CPR_X = TK * (int)sizeof(T) / 16;   // 16-byte chunks per tile row
    constexpr int NCH_Y = 64 * CPR_Y / NT, NCH_X = 64 * CPR_X / NT;                     // chunks per thread per operand
    static_assert(NCH_Y * NT == 64 * CPR_Y && NCH_X * NT == 64 * CPR_X, "whole chunks per thread");
    constexpr int TILE_Y = 64 * PITCH_Y, TILE_X = 64 * PITCH_X;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 buffers x (dY tile, X tile)

    // Work items are numbered (group, split, tn, tk) with tk fastest: neighbours share the dY column block (same tn, same rows), the next ones the
    // X block.  Each XCD takes a contiguous range of them (xcd_remap), so that what neighbours share is fetched into ONE L2 once instead of into
    // several: in the plain order (split fastest, dealt round-robin) no two workgroups of an XCD shared anything.
    const int bid = L.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    int gi = 0;
    while (gi + 1 < L.n_groups && bid >= L.tile_start[gi + 1]) ++gi;
    const SeaWgradGroup& G = L.g[gi];
    int t = bid - L.tile_start[gi];
    const int splits = L.splits[gi];
    const int tiles_k = (G.K + TK - 1) / TK, tiles_n = (G.N + TN - 1) / TN;
    int split, tn, tk;
    if (L.xcd) {
        tk = t % tiles_k;
        t /= tiles_k;
        tn = t % tiles_n;
        split = t / tiles_n;
    } else {
        split = t % splits;
        t /= splits;
        tn = t / tiles_k;
        tk = t - tn * tiles_k;
    }
    const int n0 = tn * TN, k0 = tk * TK;
    const int m_begin = split * L.rows_per_split[gi];
    int m_end = m_begin + L.rows_per_split[gi];
    m_end = m_end < G.M ? m_end : G.M;
    if (m_begin >= m_end) return;  // block-uniform

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WK, wk = wave - wn * WK;
    const T* dY = static_cast<const T*>(G.dY);
    const T* X = static_cast<const T*>(G.X);
    // Stages are requested TWO ahead into two register sets (stage s waits in set s & 1 from its request in step s - 2 to its LDS store
    // at the end of step s - 1): a 64-row stage is only 32 MFMAs of work per wave against a ~2 us memory round trip, and at two
    // waves per SIMD nothing else hides it.  Every path issues the same loads in the same order (rows clamped, columns clamped, out-of-range data
    // zeroed by a select at the store) so that the compiler can wait with vmcnt(N) for the older set alone.  A stage that lies wholly inside the
    // rows of the split, in a tile that lies wholly inside the matrix (block-uniform: every stage but the last of almost every workgroup), takes the
    // plain path: one scalar row base + a per-thread 32-bit offset per chunk, no clamps, no selects — the two waves of a SIMD share its VALU
    // issue with the MFMAs, and the clamped form spent as many issue cycles on addresses and selects as on the matrix core.
    uint4 ry[2][NCH_Y], rx[2][NCH_X];
    const bool full_tile = n0 + TN <= G.N && k0 + TK <= G.K;
    // chunk u of a thread lies u * (NT / CPR) rows below its chunk 0: one per-thread offset per operand, the rest is scalar
    constexpr int RSTEP_Y = NT / CPR_Y, RSTEP_X = NT / CPR_X;
    static_assert(RSTEP_Y * CPR_Y == NT && RSTEP_X * CPR_X == NT, "a chunk column per thread");
    const uint32_t offy0 = (uint32_t)((tid / CPR_Y) * G.lddy + n0 + (tid % CPR_Y) * EPC);
    const uint32_t offx0 = (uint32_t)((tid / CPR_X) * G.ldx + k0 + (tid % CPR_X) * EPC);
    auto load_stage = [&](int ms, auto set_tag) {
        constexpr int ST = decltype(set_tag)::value;
        if (full_tile && ms + 64 <= m_end) {
            const T* by = dY + (int64_t)ms * G.lddy;
            const T* bx = X + (int64_t)ms * G.ldx;
#pragma unroll
            for (int u = 0; u < NCH_Y; ++u) ry[ST][u] = *reinterpret_cast<const uint4*>(by + (int64_t)(u * RSTEP_Y) * G.lddy + offy0);
#pragma unroll
            for (int u = 0; u < NCH_X; ++u) rx[ST][u] = *reinterpret_cast<const uint4*>(bx + (int64_t)(u * RSTEP_X) * G.ldx + offx0);
            return;
        }
#pragma unroll
        for (int u = 0; u < NCH_Y; ++u) {
            const int idx = tid + u * NT;
            const int rr = idx / CPR_Y, cc = idx - rr * CPR_Y;
            int m = ms + rr;
            m = m < m_end ? m : m_end - 1;
            int cn = n0 + cc * EPC;
            cn = cn < G.N ? cn : G.N - EPC;
            ry[ST][u] = *reinterpret_cast<const uint4*>(dY + (int64_t)m * G.lddy + cn);
        }
#pragma unroll
        for (int u = 0; u < NCH_X; ++u) {
            const int idx = tid + u * NT;
            const int rr = idx / CPR_X, cc = idx - rr * CPR_X;
            int m = ms + rr;
            m = m < m_end ? m : m_end - 1;
            int ck = k0 + cc * EPC;
            ck = ck < G.K ? ck : G.K - EPC;
            rx[ST][u] = *reinterpret_cast<const uint4*>(X + (int64_t)m * G.ldx + ck);
        }
    };
    const int soffy0 = (tid / CPR_Y) * PITCH_Y + (tid % CPR_Y) * 16;            // LDS byte offsets of chunk 0; chunk u: + u * RSTEP rows
    const int soffx0 = TILE_Y + (tid / CPR_X) * PITCH_X + (tid % CPR_X) * 16;
    auto store_stage = [&](int buf, int ms, auto set_tag) {
        constexpr int ST = decltype(set_tag)::value;
        char* sy = smem + buf * (TILE_Y + TILE_X);
        if (full_tile && ms + 64 <= m_end) {
#pragma unroll
            for (int u = 0; u < NCH_Y; ++u) *reinterpret_cast<uint4*>(sy + soffy0 + u * RSTEP_Y * PITCH_Y) = ry[ST][u];
#pragma unroll
            for (int u = 0; u < NCH_X; ++u) *reinterpret_cast<uint4*>(sy + soffx0 + u * RSTEP_X * PITCH_X) = rx[ST][u];
            return;
        }
#pragma unroll
        for (int u = 0; u < NCH_Y; ++u) {
            const int idx = tid + u * NT;
            const int rr = idx / CPR_Y, cc = idx - rr * CPR_Y;
            const bool yok = ms + rr < m_end && n0 + cc * EPC < G.N;
            *reinterpret_cast<uint4*>(sy + soffy0 + u * RSTEP_Y * PITCH_Y) = yok ? ry[ST][u] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < NCH_X; ++u) {
            const int idx = tid + u * NT;
            const int rr = idx / CPR_X, cc = idx - rr * CPR_X;
            const bool xok = ms + rr < m_end && k0 + cc * EPC < G.K;
            *reinterpret_cast<uint4*>(sy + soffx0 + u * RSTEP_X * PITCH_X) = xok ? rx[ST][u] : make_uint4(0, 0, 0, 0);
        }
    };

    f32x4 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // db[n] = column sums of dY: on the matrix core too — the wave's dY fragments against an all-ones operand give, in every column of the result, the
    // sum over the contraction rows (wave-uniform: the k-first tile's k-first waves; 64 two-byte LDS reads and a dependent add chain per thread before)
    f32x4 bacc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = G.db != nullptr && tk == 0 && wk == 0;
    const uint32_t one_bits = sizeof(T) == 2 ? 0x3f803f80u : 0x3f800000u;
    const uint4 ones = make_uint4(one_bits, one_bits, one_bits, one_bits);

    const int n_stage = (m_end - m_begin + 63) / 64;
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    auto step = [&](int st, auto cur_tag, auto nxt_tag) {
        // request stage st + 2 (set st & 1, emptied at the end of step st - 1), multiply stage st, then move stage st + 1 to the other buffer
        const int ms2 = m_begin + (st + 2) * 64;
        load_stage(ms2 < m_end ? ms2 : m_end - 1, cur_tag);   // past the end: a harmless re-read, zeroed / never used
        if (st >= 0) {
            const char* sy = smem + (st & 1) * (TILE_Y + TILE_X);
            const char* sx = sy + TILE_Y;
#pragma unroll
            for (int mk = 0; mk < 64; mk += CK) {
                uint4 a[MI], b[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = load_frag_T<T>(sy, PITCH_Y, mk, wn * WT + i * 16, lane);
#pragma unroll
                for (int j = 0; j < MI; ++j) b[j] = load_frag_T<T>(sx, PITCH_X, mk, wk * WT + j * 16, lane);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j) mma16<T>(a[i], b[j], acc[i][j]);
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) mma16<T>(a[i], ones, bacc[i]);
                }
            }
        }
        store_stage((st + 1) & 1, m_begin + (st + 1) * 64, nxt_tag);   // stage st + 1 (requested one step ago); st = -2 stores the zeroed registers
        __syncthreads();
    };
#pragma unroll
    for (int u = 0; u < NCH_Y; ++u) ry[0][u] = ry[1][u] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < NCH_X; ++u) rx[0][u] = rx[1][u] = make_uint4(0, 0, 0, 0);
    // the pipeline fills inside the loop (steps -2, -1 only request and store), whole pairs only: the loop is entered with nothing in
    // flight and has no conditional load, the two things the compiler needs to count the loads (see attention.hip)
    int st = -2;
    for (; st + 1 < n_stage; st += 2) {
        step(st, Set0{}, Set1{});
        step(st + 1, Set1{}, Set0{});
    }
    if (st < n_stage) step(st, Set0{}, Set1{});
    // C[n][k]: lane holds column k = lane & 15, rows n = 4 (lane >> 4) + reg
    const int r = lane & 15, g = lane >> 4;
    const bool plain_store = G.overwrite != 0 && splits == 1;   // block-uniform
    // A plain-stored FULL tile leaves through LDS as whole rows, 16 bytes per lane.  In the accumulator layout a lane holds ONE column of four rows: every store
    // instruction is a dword per lane (4 rows x 64 B), 64 of them per wave — the reference's own widths write 134 MB of fp32 per field MLP matrix that way:
    // 1 M dword-store instructions per launch, ~150 of the 197 us of the multiphase mlp.fc2 gradient.  The two stage buffers are free behind the main loop's
    // last barrier.
    constexpr int CPITCH = TK * 4 + 16;
    constexpr bool CAN_STAGE = TN * CPITCH <= 2 * (TILE_Y + TILE_X);
    if (CAN_STAGE && plain_store && L.stage_out && n0 + TN <= G.N && k0 + TK <= G.K && (G.lddw & 3) == 0 && (reinterpret_cast<uintptr_t>(G.dW) & 15u) == 0) {   // block-uniform
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float*>(smem + (wn * WT + i * 16 + g * 4 + q) * CPITCH + (wk * WT + j * 16 + r) * 4) = acc[i][j][q];
        __syncthreads();
        constexpr int PPR = TK / 4;   // 16-byte pieces per tile row
        for (int idx = tid; idx < TN * PPR; idx += NT) {
            const int row = idx / PPR, c4 = idx - row * PPR;
            *reinterpret_cast<float4*>(G.dW + (int64_t)(n0 + row) * G.lddw + k0 + c4 * 4) = *reinterpret_cast<const float4*>(smem + row * CPITCH + c4 * 16);
        }
    } else
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int k = k0 + wk * WT + j * 16 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * WT + i * 16 + g * 4 + q;
                if (n < G.N && k < G.K) {
                    // the only contribution to this element in a step whose gradient buffer was zeroed (overwrite, one split): a plain store — the fp32
                    // atomics run at ~1.3 TB/s chip-wide, a quarter of the store rate, and were ~200 of the 357 us of the 2048 x 16384 MLP gradients
                    if (plain_store) G.dW[(int64_t)n * G.lddw + k] = acc[i][j][q];
                    else atomicAdd(G.dW + (int64_t)n * G.lddw + k, acc[i][j][q]);
                }
            }
        }
    if (do_bias && r == 0) {   // every column of bacc holds the same sums
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * WT + i * 16 + g * 4 + q;
                if (n < G.N) atomicAdd(G.db + n, bacc[i][q]);
            }
    }
}

extern "C" int sea_wgrad_grouped(const SeaWgradGroup* groups, int n_groups, int dtype, void* stream) {
    SEA_REQUIRE(groups && n_groups >= 1 && n_groups <= SEA_MAX_WGRAD_GROUPS, "sea_wgrad_grouped: n_groups=%d", n_groups);
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_wgrad_grouped: bad dtype %d", dtype);
    const int epc = dtype == SEA_BF16 ? 8 : 4;
    WgradLaunch L;
    memset(&L, 0, sizeof(L));
    bool big = true, long_m = true;
    for (int i = 0; i < n_groups; ++i) {
        const SeaWgradGroup& G = groups[i];
        SEA_REQUIRE(G.dY && G.X && G.dW, "sea_wgrad_grouped[%d]: null pointer", i);
        SEA_REQUIRE(G.M >= 1 && G.N >= 8 && G.K >= 8 && G.N % 8 == 0 && G.K % 8 == 0, "sea_wgrad_grouped[%d]: bad shape M=%d N=%d K=%d", i, G.M, G.N, G.K);
        SEA_REQUIRE(G.lddy % epc == 0 && G.ldx % epc == 0 && G.lddy >= G.N && G.ldx >= G.K && G.lddw >= G.K, "sea_wgrad_grouped[%d]: bad strides", i);
        SEA_REQUIRE(sea_aligned16(G.dY) && sea_aligned16(G.X), "sea_wgrad_grouped[%d]: dY/X must be 16-byte aligned", i);
        big = big && G.N % 128 == 0 && G.K % 128 == 0;
        long_m = long_m && G.M >= 2048;
    }
    static const int forced = sea_tune("wgrad_tile", 0);  // tuning aid
    long tiles128 = 0;
    for (int i = 0; i < n_groups; ++i) tiles128 += (long)((groups[i].N + 127) / 128) * ((groups[i].K + 127) / 128);
    // few 128-tiles would mean many contraction splits, i.e. many atomic passes over the same outputs: measured at cfg3 the small
    // exchange matrices (6 tiles) run 1.8x slower on the 128 tile, the MLP / condition matrices (96-192 tiles) 1.2-1.4x faster.
    // (256 x 128 / 128 x 256 tiles with 8 waves, the wide side along the smaller matrix dimension so that the larger operand is read by half as many
    // workgroups, were measured too: cond 152 against 181 us before the main loop was reworked, 131 against 132 after — not kept.)
    // Short contractions with large outputs (the reference's own widths: M = B T ~ 800 rows, the MLP matrices 2048 x 16384, configs/multiphase_flow.py:112-141):
    // a 128 x 128 tile re-reads (128 + 128) M operand elements for 128^2 outputs — 1.7 GB of L2 -> LDS traffic for mlp.fc2's gradient — a 256 x 128 tile
    // (8 waves of 64 x 64) three quarters of it (256 x 256 with 16 waves spills: 128 registers per wave).  bf16, whole tiles, at least two per CU.
    bool big256 = dtype == SEA_BF16;
    long tiles256 = 0;
    int m_lim = 0;
    for (int i = 0; i < n_groups; ++i) {
        big256 = big256 && groups[i].N % 256 == 0 && groups[i].K % 128 == 0;
        tiles256 += (long)(groups[i].N / 256) * (groups[i].K / 128);
        m_lim = groups[i].M > m_lim ? groups[i].M : m_lim;
    }
    const bool use256 = forced == 256 && big256;   // measured at the multiphase MLP gradients (M = 796): 218 us against 201 with the 128 tile — opt-in only (SEA_TUNE=wgrad_tile=256)
    (void)tiles256; (void)m_lim;
    const int tn = use256 ? 256 : (forced == 64 ? 64 : (forced == 128 ? 128 : (big && ((long_m && tiles128 >= 64) || tiles128 >= 1024) ? 128 : 64)))   /* short contractions: the 128 tile once the launch is many tiles deep (M = 796, 2048 x 16384: 299 -> 201 us) */, tk = use256 ? 128 : tn;
    long base_tiles = 0;
    for (int i = 0; i < n_groups; ++i) base_tiles += (long)((groups[i].N + tn - 1) / tn) * ((groups[i].K + tk - 1) / tk);
    // Split the contraction so that the launch fills the chip's resident-workgroup slots (2 per CU with the 128 tile's 72 KiB of LDS)
    // in WHOLE rounds, never below 256 rows per split: the launch lasts rounds x stages-per-split, so 576 workgroups on 512
    // slots (what "about 512" gave the MLP matrices at cfg3: 96 tiles x 6 splits) take two rounds of 42 stages where 480 take one of 51.  The
    // split count minimises rounds x (stages + 2) — the 2 prices a workgroup's fill and its atomic pass over the tile.
    static const int wg_target = sea_tune("wgrad_target", 0);  // tuning aid: "about this many workgroups"
    // the 64 tile sits four to a CU (40 KiB of LDS); launches of a few small matrices keep to 512 — their extra splits cost more in atomic passes over
    // the same few tiles than they fill (measured) —, a launch of many (the exchange's gradients in one launch: 184 tiles) fills all 1024
    const long slots = tn == 256 ? 256 : (tn == 64 && base_tiles >= 64 ? 1024 : 512);
    int m_max = 1;
    for (int i = 0; i < n_groups; ++i) m_max = groups[i].M > m_max ? groups[i].M : m_max;
    long best_splits = 1;
    if (wg_target > 0) {
        best_splits = (wg_target + base_tiles - 1) / base_tiles;
    } else {
        long best_cost = -1;
        const long s_max = (m_max + 255) / 256;
        for (long sp = 1; sp <= s_max && sp <= 64; ++sp) {
            const long rounds = (base_tiles * sp + slots - 1) / slots;
            const long stages = ((m_max + sp - 1) / sp + 63) / 64;
            const long cost = rounds * (stages + 2);
            if (best_cost < 0 || cost < best_cost) best_cost = cost, best_splits = sp;
        }
    }
    int total = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaWgradGroup& G = groups[i];
        long want = best_splits;
        long max_splits = (G.M + 255) / 256;
        int splits = (int)(want < 1 ? 1 : (want > max_splits ? max_splits : want));
        int rows = ((G.M + splits - 1) / splits + 63) / 64 * 64;
        splits = (G.M + rows - 1) / rows;
        L.g[i] = G;
        L.splits[i] = splits;
        L.rows_per_split[i] = rows;
        L.tile_start[i] = total;
        total += ((G.N + tn - 1) / tn) * ((G.K + tk - 1) / tk) * splits;
    }
    L.tile_start[n_groups] = total;
    L.n_groups = n_groups;
    static const bool xcd_order = sea_tune("wgrad_xcd", 1) != 0;  // tuning aid
    L.xcd = xcd_order ? 1 : 0;
    L.stage_out = sea_tune("wgrad_stage", 1) != 0;   // read per call
    hipStream_t s = static_cast<hipStream_t>(stream);
#define LAUNCH_WG(TT, TN_, TK_, WT_)                                                                                             \
    do {                                                                                                                         \
        constexpr int lds_ = 2 * 64 * ((TN_ + TK_) * (int)sizeof(TT) + 2 * WGRAD_PAD<TT>);                                                      \
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<TT, TN_, TK_, WT_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_); \
        (void)once;                                                                                                              \
        wgrad_kernel<TT, TN_, TK_, WT_><<<dim3(total), dim3((TN_ / WT_) * (TK_ / WT_) * 64), lds_, s>>>(L);                      \
    } while (0)
    if (dtype == SEA_BF16) { if (tn == 256) LAUNCH_WG(__bf16, 256, 128, 64); else if (tn == 128) LAUNCH_WG(__bf16, 128, 128, 64); else LAUNCH_WG(__bf16, 64, 64, 32); }
    else { if (tn == 128) LAUNCH_WG(float, 128, 128, 64); else LAUNCH_WG(float, 64, 64, 32); }
#undef LAUNCH_WG
    SEA_CHECK_LAUNCH("sea_wgrad_grouped");
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ W^T shadow
// One 32 x 32 tile per workgroup; desc (device memory) = n_desc x {src_off, dst_off, rows, cols} in elements.
template <typename T>
__global__ __launch_bounds__(256) void transpose_weights_kernel(const float* __restrict__ src, T* __restrict__ dst, const int64_t* __restrict__ desc,
                                                                const int* __restrict__ tile_start, int n_desc) {
    __shared__ float tile[32][33];
    const int bid = blockIdx.x;
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {  // last matrix whose first tile is <= bid
        const int mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= bid) lo = mid; else hi = mid - 1;
    }
    const int64_t* d = desc + 4 * lo;
    const int64_t so = d[0], dof = d[1];
    const int rows = (int)d[2], cols = (int)d[3];
    const int t = bid - tile_start[lo];
    const int tiles_c = (cols + 31) / 32;
    const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[so + (int64_t)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;  // dst is [cols][rows]
        if (r < rows && c < cols) dst[dof + (int64_t)c * rows + r] = from_f32<T>(tile[tx][ty + 8 * i]);
    }
}

extern "C" int sea_transpose_weights(const float* src, void* dst, int dst_dtype, const int64_t* desc, const int32_t* tile_start, int n_desc,
                                     int total_tiles, void* stream) {
    SEA_REQUIRE(src && dst && desc && tile_start && n_desc >= 1 && total_tiles >= 1, "sea_transpose_weights: bad arguments");
    SEA_REQUIRE(dst_dtype == SEA_F32 || dst_dtype == SEA_BF16, "sea_transpose_weights: bad dtype");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dst_dtype == SEA_BF16) transpose_weights_kernel<__bf16><<<dim3(total_tiles), dim3(256), 0, s>>>(src, static_cast<__bf16*>(dst), desc, tile_start, n_desc);
    else transpose_weights_kernel<float><<<dim3(total_tiles), dim3(256), 0, s>>>(src, static_cast<float*>(dst), desc, tile_start, n_desc);
    SEA_CHECK_LAUNCH("sea_transpose_weights");
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ row-norm backward
#define SEA_MAX_NORM_BWD_GROUPS 8
struct NormBwdLaunch {
    SeaNormBwdGroup g[SEA_MAX_NORM_BWD_GROUPS];
    int M, d, gelu, accumulate;
    float* ws;  // optional [n_groups][gridDim.x][2][d] partial column sums (then reduced by colsum_finish_kernel)
};

// Second stage of the column sums: out[c] += sum_b ws[(g * nblk + b) * 2 * d + c].  Hundreds of workgroups adding to the same few
// thousand addresses serialise at the memory-side atomic unit (MI355X_MICROARCH.md, Global float atomics: "every workgroup into one
// row: 14x slower"); writing per-workgroup partials and summing them here keeps it to one add per address.
struct ColsumFinish {
    float* out_a[24];
    float* out_b[24];
    int width[24];
    const float* ws;
    int nblk, stride;  // stride = floats per (group, block) slab = 2 * max width
};
__global__ __launch_bounds__(256) void colsum_finish_kernel(const ColsumFinish F) {
    const int g = blockIdx.y, w = F.width[g];
    // gridDim.z slices of the workgroup partials: 16 adders per address instead of hundreds, and enough parallelism to stream the slab
    const int per = (F.nblk + gridDim.z - 1) / gridDim.z;
    const int b0 = blockIdx.z * per, b1 = b0 + per < F.nblk ? b0 + per : F.nblk;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < 2 * w; c += gridDim.x * 256) {
        const float* src = F.ws + (int64_t)g * F.nblk * F.stride + c;
        float acc = 0.f;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) acc += src[(int64_t)b * F.stride];
        float* dst = c < w ? F.out_a[g] : F.out_b[g];
        if (dst != nullptr) atomicAdd(dst + (c < w ? c : c - w), acc);
    }
}

// grid = (nblk, n_groups); each workgroup walks rows blockIdx.x*4 + wave, += 4*gridDim.x; one wave per row.
// y = xhat*s + t, s = gamma (+1 + mod_w), t = beta (+ mod_b);  optional y = gelu(y).
// dxhat = dy*s;  dx = rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat));  dgamma += dy*xhat;  dbeta += dy;
// dmod = [dy*xhat | dy].  Column sums are accumulated in LDS (ds_add_f32) and flushed with one global atomic per column.
// KMAX > 0: d <= 256*KMAX and the column sums are accumulated in registers (lane owns columns lane*4 + 256k) and merged once at
// the end; KMAX == 0: any d <= 16384, per-element LDS atomics (slow: 2 ds_add per element).
template <typename T, bool DY_ACT, bool X_ACT, int KMAX>
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const NormBwdLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float cs[];  // [2][d]
    constexpr int KA = KMAX > 0 ? KMAX : 1;
    float accg[KA][4], accb[KA][4];
#pragma unroll
    for (int k = 0; k < KA; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) accg[k][e] = accb[k][e] = 0.f;
    const SeaNormBwdGroup& G = L.g[blockIdx.y];
    const int d = L.d, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * d; i += 256) cs[i] = 0.f;
    __syncthreads();
    using DYT = typename std::conditional<DY_ACT, T, float>::type;
    using XT = typename std::conditional<X_ACT, T, float>::type;
    const float inv_d = 1.0f / (float)d;
    for (int row = blockIdx.x * 4 + wave; row < L.M; row += 4 * gridDim.x) {
        const DYT* dy = static_cast<const DYT*>(G.dY) + (int64_t)row * G.lddy;
        const XT* x = static_cast<const XT*>(G.X) + (int64_t)row * G.ldx;
        const T* mod = G.mod ? static_cast<const T*>(G.mod) + (int64_t)row * G.ldmod : nullptr;
        T* dmod = G.dmod ? static_cast<T*>(G.dmod) + (int64_t)row * G.lddmod : nullptr;
        const float mean = G.mean[row], rstd = G.rstd[row];
        float c1 = 0.f, c2 = 0.f;
        float kdx[KA][4], kxh[KA][4];
        auto pass_a = [&](int i, int k) {
            float xv[4], dv[4], s[4], tt[4] = {0.f, 0.f, 0.f, 0.f};
            load4(x + i, xv);
            load4(dy + i, dv);
            load4(G.gamma + i, s);
            if (mod) {
                float mw[4];
                load4(mod + i, mw);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += 1.0f + mw[e];
            }
            if (L.gelu) {
                if (G.beta) load4(G.beta + i, tt);
                if (mod) {
                    float mb[4];
                    load4(mod + d + i, mb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) tt[e] += mb[e];
                }
            }
            float dyx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[e] - mean) * rstd;
                if (L.gelu) dv[e] *= gelu_grad_for<T>(xh * s[e] + tt[e]);
                const float dxh = dv[e] * s[e];
                c1 += dxh;
                c2 += dxh * xh;
                dyx[e] = dv[e] * xh;
                if constexpr (KMAX > 0) {
                    accg[k][e] += dyx[e];
                    accb[k][e] += dv[e];
                    kdx[k][e] = dxh;   // kept for the second pass: no second erf / second read of the row
                    kxh[k][e] = xh;
                } else {
                    atomicAdd(&cs[i + e], dyx[e]);
                    atomicAdd(&cs[d + i + e], dv[e]);
                }
            }
            if (dmod) {
                store4(dmod + i, dyx[0], dyx[1], dyx[2], dyx[3]);
                store4(dmod + d + i, dv[0], dv[1], dv[2], dv[3]);
            }
        };
        if constexpr (KMAX > 0) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (lane * 4 + 256 * k < d) pass_a(lane * 4 + 256 * k, k);
        } else {
            for (int i = lane * 4; i < d; i += 256) pass_a(i, 0);
        }
        c1 = wave_sum(c1) * inv_d;
        c2 = wave_sum(c2) * inv_d;
        float* dx32 = G.dX32 ? G.dX32 + (int64_t)row * G.lddx32 : nullptr;
        T* dxa = G.dXact ? static_cast<T*>(G.dXact) + (int64_t)row * G.lddxact : nullptr;
        auto emit = [&](int i, float (&o)[4]) {
            if (dx32) {
                if (L.accumulate) {
                    float old[4];
                    load4(dx32 + i, old);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += old[e];
                }
                store4(dx32 + i, o[0], o[1], o[2], o[3]);
            }
            if (dxa) store4(dxa + i, o[0], o[1], o[2], o[3]);
        };
        if constexpr (KMAX > 0) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int i = lane * 4 + 256 * k;
                if (i < d) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rstd * (kdx[k][e] - c1 - kxh[k][e] * c2);
                    emit(i, o);
                }
            }
            continue;
        }
        for (int i = lane * 4; i < d; i += 256) {
            float xv[4], dv[4], s[4], tt[4] = {0.f, 0.f, 0.f, 0.f};
            load4(x + i, xv);
            load4(dy + i, dv);
            load4(G.gamma + i, s);
            if (mod) {
                float mw[4];
                load4(mod + i, mw);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += 1.0f + mw[e];
            }
            if (L.gelu) {
                if (G.beta) load4(G.beta + i, tt);
                if (mod) {
                    float mb[4];
                    load4(mod + d + i, mb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) tt[e] += mb[e];
                }
            }
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[e] - mean) * rstd;
                if (L.gelu) dv[e] *= gelu_grad_for<T>(xh * s[e] + tt[e]);
                o[e] = rstd * (dv[e] * s[e] - c1 - xh * c2);
            }
            emit(i, o);
        }
    }
    if constexpr (KMAX > 0) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = lane * 4 + 256 * k;
            if (i < d) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    atomicAdd(&cs[i + e], accg[k][e]);
                    atomicAdd(&cs[d + i + e], accb[k][e]);
                }
            }
        }
    }
    __syncthreads();
    if (L.ws != nullptr) {
        float* dst = L.ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * d;
        for (int i = threadIdx.x; i < 2 * d; i += 256) dst[i] = cs[i];
        return;
    }
    for (int i = threadIdx.x; i < d; i += 256) {
        if (G.dgamma) atomicAdd(G.dgamma + i, cs[i]);
        if (G.dbeta) atomicAdd(G.dbeta + i, cs[d + i]);
    }
}

// Wide rows (d > 1024: the MLP hidden width, 2048 at embed_dim 256, 8192 / 16384 at the shipped widths): ONE WORKGROUP per row instead of one
// wave.  A wave per row needs ~120 VGPRs of per-row state and leaves 4 waves per SIMD waiting on their own loads (measured: 51 % of wave time in
// s_waitcnt, 36 % SIMD utilisation); with the row spread over NT threads x KCH chunks of 4 columns each lane owns few chunks, the row's loads are
// spread over many requesters, and every thread keeps the column sums of ITS columns in registers for all the rows of its workgroup — no LDS
// atomics, no second pass over the row (the per-element LDS-atomic form this replaces for d > 2048 re-read the row, evaluated GELU' twice and ran
// at 0.4 TB/s: 200 us for the 78 MB of the shipped cylinder width).  Row statistics cross the waves through LDS.  Column sums leave as plain
// stores into the two-stage workspace (or as atomics when the caller gave none).
// a piece of PW = 4 or 8 consecutive columns (f32: 16 / 2 x 16 bytes; bf16: 8 / 16 bytes)
template <typename U, int PW>
__device__ __forceinline__ void loadp(const U* src, float (&o)[PW]) {
    if constexpr (PW == 4) load4(src, o);
    else load8(src, o);
}
__device__ __forceinline__ void storep(float* dst, const float (&v)[4]) { store4(dst, v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void storep(__bf16* dst, const float (&v)[4]) { store4(dst, v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void storep(float* dst, const float (&v)[8]) {
    store4(dst, v[0], v[1], v[2], v[3]);
    store4(dst + 4, v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void storep(__bf16* dst, const float (&v)[8]) { store8(dst, v); }

// PW = columns per piece: 4, or 8 where every row operand is bf16 (16-byte accesses: the LayerNorm + GELU of the MLP)
template <typename T, bool DY_ACT, bool X_ACT, int NT, int KCH, int PW>
__global__ __launch_bounds__(NT) void rownorm_bwd_wide_kernel(const NormBwdLaunch L) {
    constexpr int NW = NT / 64;
    __shared__ float red[2 * NW];
    const SeaNormBwdGroup& G = L.g[blockIdx.y];
    const int d = L.d, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    using DYT = typename std::conditional<DY_ACT, T, float>::type;
    using XT = typename std::conditional<X_ACT, T, float>::type;
    const float inv_d = 1.0f / (float)d;
    // (more than 8 columns per thread, rows of up to 16384 columns on 1024 threads: 128 registers per thread — the row-independent gain / shift are re-read per row from L2
    // instead of kept, and the second pass recomputes xhat from x instead of keeping it)
    constexpr bool KEEP = KCH * PW <= 8;
    float accg[KCH][PW], accb[KCH][PW], s[KCH][PW], tt[KCH][PW];
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
        const int i = tid * PW + NT * PW * k;
#pragma unroll
        for (int e = 0; e < PW; ++e) accg[k][e] = accb[k][e] = s[k][e] = tt[k][e] = 0.f;
        if (KEEP && i < d && G.mod == nullptr) {  // row-independent gain / shift: loaded once
            loadp(G.gamma + i, s[k]);
            if (L.gelu && G.beta) loadp(G.beta + i, tt[k]);
        }
    }
    for (int row = blockIdx.x; row < L.M; row += gridDim.x) {
        const DYT* dy = static_cast<const DYT*>(G.dY) + (int64_t)row * G.lddy;
        const XT* x = static_cast<const XT*>(G.X) + (int64_t)row * G.ldx;
        const T* mod = G.mod ? static_cast<const T*>(G.mod) + (int64_t)row * G.ldmod : nullptr;
        T* dmod = G.dmod ? static_cast<T*>(G.dmod) + (int64_t)row * G.lddmod : nullptr;
        const float mean = G.mean[row], rstd = G.rstd[row];
        float* dx32 = G.dX32 ? G.dX32 + (int64_t)row * G.lddx32 : nullptr;
        T* dxa = G.dXact ? static_cast<T*>(G.dXact) + (int64_t)row * G.lddxact : nullptr;
        auto gain_shift = [&](int i, float (&sv)[PW], float (&tv)[PW]) {   // this chunk's s = gamma (+ 1 + mod_w), t = beta (+ mod_b) (t only feeds GELU')
            loadp(G.gamma + i, sv);
            for (int e = 0; e < PW; ++e) tv[e] = 0.f;
            if (L.gelu && G.beta) loadp(G.beta + i, tv);
            if (mod) {
                float mw[PW];
                loadp(mod + i, mw);
#pragma unroll
                for (int e = 0; e < PW; ++e) sv[e] += 1.0f + mw[e];
                if (L.gelu) {
                    float mb[PW];
                    loadp(mod + d + i, mb);
#pragma unroll
                    for (int e = 0; e < PW; ++e) tv[e] += mb[e];
                }
            }
        };
        auto emit = [&](int i, float (&o)[PW]) {
            if (dx32) {
                if (L.accumulate) {
                    float old[PW];
                    loadp(dx32 + i, old);
#pragma unroll
                    for (int e = 0; e < PW; ++e) o[e] += old[e];
                }
                storep(dx32 + i, o);
            }
            if (dxa) storep(dxa + i, o);
        };
        auto row_stats = [&](float& c1, float& c2) {   // sums over the row of dxhat and dxhat * xhat, divided by d
            c1 = wave_sum(c1);
            c2 = wave_sum(c2);
            if (lane == 0) {
                red[wave] = c1;
                red[NW + wave] = c2;
            }
            __syncthreads();
            c1 = c2 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                c1 += red[w];
                c2 += red[NW + w];
            }
            c1 *= inv_d;
            c2 *= inv_d;
            __syncthreads();
        };
        float c1 = 0.f, c2 = 0.f;
        if constexpr (KEEP) {
            float xv[KCH][PW], dv[KCH][PW];
#pragma unroll
            for (int k = 0; k < KCH; ++k) {  // all of the row's loads first
                const int i = tid * PW + NT * PW * k;
#pragma unroll
                for (int e = 0; e < PW; ++e) xv[k][e] = dv[k][e] = 0.f;
                if (i < d) {
                    loadp(x + i, xv[k]);
                    loadp(dy + i, dv[k]);
                    if (mod) gain_shift(i, s[k], tt[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < KCH; ++k) {   // xv becomes xhat, dv becomes dxhat = dy (gelu') s — in place: the second pass needs exactly these two
                const int i = tid * PW + NT * PW * k;
                if (i < d) {
                    float dyx[PW], dvg[PW];
#pragma unroll
                    for (int e = 0; e < PW; ++e) {
                        const float xh = (xv[k][e] - mean) * rstd;
                        dvg[e] = L.gelu ? dv[k][e] * gelu_grad_for<T>(xh * s[k][e] + tt[k][e]) : dv[k][e];
                        const float dxh = dvg[e] * s[k][e];
                        c1 += dxh;
                        c2 += dxh * xh;
                        dyx[e] = dvg[e] * xh;
                        accg[k][e] += dyx[e];
                        accb[k][e] += dvg[e];
                        dv[k][e] = dxh;
                        xv[k][e] = xh;
                    }
                    if (dmod) {
                        storep(dmod + i, dyx);
                        storep(dmod + d + i, dvg);
                    }
                }
            }
            row_stats(c1, c2);
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                const int i = tid * PW + NT * PW * k;
                if (i < d) {
                    float o[PW];
#pragma unroll
                    for (int e = 0; e < PW; ++e) o[e] = rstd * (dv[k][e] - c1 - xv[k][e] * c2);
                    emit(i, o);
                }
            }
        } else {
            // rows of more than 8192 columns: 4 chunks per thread do not fit the 128 registers a 1024-thread workgroup leaves a thread, so the second pass
            // reads the row again (64 KB the first pass has just pulled into L2) and evaluates GELU' a second time instead of keeping dxhat / xhat
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                const int i = tid * PW + NT * PW * k;
                if (i < d) {
                    float xv[PW], dv[PW], sv[PW], tv[PW], dyx[PW], dvg[PW];
                    loadp(x + i, xv);
                    loadp(dy + i, dv);
                    gain_shift(i, sv, tv);
#pragma unroll
                    for (int e = 0; e < PW; ++e) {
                        const float xh = (xv[e] - mean) * rstd;
                        dvg[e] = L.gelu ? dv[e] * gelu_grad_for<T>(xh * sv[e] + tv[e]) : dv[e];
                        const float dxh = dvg[e] * sv[e];
                        c1 += dxh;
                        c2 += dxh * xh;
                        dyx[e] = dvg[e] * xh;
                        accg[k][e] += dyx[e];
                        accb[k][e] += dvg[e];
                    }
                    if (dmod) {
                        storep(dmod + i, dyx);
                        storep(dmod + d + i, dvg);
                    }
                }
            }
            row_stats(c1, c2);
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                const int i = tid * PW + NT * PW * k;
                if (i < d) {
                    float xv[PW], dv[PW], sv[PW], tv[PW], o[PW];
                    loadp(x + i, xv);
                    loadp(dy + i, dv);
                    gain_shift(i, sv, tv);
#pragma unroll
                    for (int e = 0; e < PW; ++e) {
                        const float xh = (xv[e] - mean) * rstd;
                        const float dvg = L.gelu ? dv[e] * gelu_grad_for<T>(xh * sv[e] + tv[e]) : dv[e];
                        o[e] = rstd * (dvg * sv[e] - c1 - xh * c2);
                    }
                    emit(i, o);
                }
            }
        }
    }
    // a thread owns its columns: plain stores of the workgroup's partial sums (two-stage), or one atomic per column
    float* ws = L.ws != nullptr ? L.ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * d : nullptr;
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
        const int i = tid * PW + NT * PW * k;
        if (i < d) {
            if (ws != nullptr) {
                storep(ws + i, accg[k]);
                storep(ws + d + i, accb[k]);
            } else {
#pragma unroll
                for (int e = 0; e < PW; ++e) {
                    if (G.dgamma) atomicAdd(G.dgamma + i + e, accg[k][e]);
                    if (G.dbeta) atomicAdd(G.dbeta + i + e, accb[k][e]);
                }
            }
        }
    }
}

extern "C" int sea_rownorm_bwd(const SeaNormBwdGroup* groups, int n_groups, int M, int d, int dy_is_act, int x_is_act, int gelu,
                               int accumulate, int dtype, float* ws, int64_t ws_floats, void* stream) {
    SEA_REQUIRE(groups && n_groups >= 1 && n_groups <= SEA_MAX_NORM_BWD_GROUPS, "sea_rownorm_bwd: n_groups=%d", n_groups);
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_rownorm_bwd: bad dtype %d", dtype);
    SEA_REQUIRE(M >= 1 && d >= 4 && d % 4 == 0 && d <= 16384, "sea_rownorm_bwd: bad M=%d d=%d", M, d);
    NormBwdLaunch L;
    memset(&L, 0, sizeof(L));
    for (int i = 0; i < n_groups; ++i) {
        const SeaNormBwdGroup& G = groups[i];
        SEA_REQUIRE(G.dY && G.X && G.gamma && G.mean && G.rstd && (G.dX32 || G.dXact), "sea_rownorm_bwd[%d]: null pointer", i);
        SEA_REQUIRE(G.lddy % 4 == 0 && G.ldx % 4 == 0 && G.lddy >= d && G.ldx >= d, "sea_rownorm_bwd[%d]: bad input strides", i);
        SEA_REQUIRE((!G.mod || (G.ldmod % 4 == 0 && G.ldmod >= 2 * d)) && (!G.dmod || (G.lddmod % 4 == 0 && G.lddmod >= 2 * d)) &&
                        (!G.dX32 || (G.lddx32 % 4 == 0 && G.lddx32 >= d)) && (!G.dXact || (G.lddxact % 4 == 0 && G.lddxact >= d)),
                    "sea_rownorm_bwd[%d]: bad strides", i);
        SEA_REQUIRE(sea_aligned16(G.dY) && sea_aligned16(G.X) && sea_aligned16(G.mod) && sea_aligned16(G.gamma) && sea_aligned16(G.beta) &&
                        sea_aligned16(G.dX32) && sea_aligned16(G.dXact) && sea_aligned16(G.dmod), "sea_rownorm_bwd[%d]: pointers must be 16-byte aligned", i);
        L.g[i] = G;
    }
    L.M = M; L.d = d; L.gelu = gelu; L.accumulate = accumulate;
    const bool wide = d > 1024;  // one workgroup per row: 256 threads up to 2048 columns, 512 up to 4096, 1024 beyond (2 / 2 / 2-4 chunks of 4 columns per thread)
    int nblk = wide ? M : (M + 3) / 4;
    static const int nblk_cap = sea_tune("normbwd_blocks", 512);  // tuning aid
    if (nblk > nblk_cap) nblk = nblk_cap;
    if (wide) {
        // all workgroups of the launch resident at once (5 per CU at the wide kernel's 90 VGPRs): a second, partial round costs more than
        // the extra parallelism returns (measured at cfg3, 3 groups: 384 per group 171 us, 512 207 us, 448 220 us)
        const int per_cu = d <= 2048 ? 5 : (d <= 4096 ? 2 : 1);
        int fit = (per_cu * 256 / n_groups) / 64 * 64;
        if (d > 2048 && fit < 64) fit = 64;
        if (fit >= 64 && nblk > fit) nblk = fit;
    }
    // pieces of 8 columns (16-byte accesses) where every row operand is bf16 and nothing is modulated: the MLP's LayerNorm + GELU
    bool wide8 = wide && dtype == SEA_BF16 && dy_is_act && x_is_act && d % 8 == 0 && sea_tune("normbwd_wide8", 1) != 0;
    for (int i = 0; i < n_groups && wide8; ++i)
        wide8 = !groups[i].mod && !groups[i].dmod && !groups[i].dX32 && groups[i].lddy % 8 == 0 && groups[i].ldx % 8 == 0 && groups[i].lddxact % 8 == 0;
    const bool two_stage = ws != nullptr && ws_floats >= (int64_t)n_groups * nblk * 2 * d && nblk > 8;
    L.ws = two_stage ? ws : nullptr;
    const dim3 grid(nblk, n_groups), block(256);
    const size_t lds = (size_t)(2 * d + 8) * sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define LAUNCH_NBK(TT, DYA, XA, KM)                                                                                        \
    do {                                                                                                                   \
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rownorm_bwd_kernel<TT, DYA, XA, KM>),      \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
        rownorm_bwd_kernel<TT, DYA, XA, KM><<<grid, block, lds, s>>>(L);                                                   \
    } while (0)
#define LAUNCH_NB(TT, DYA, XA)                         \
    do {                                               \
        if (d <= 256) LAUNCH_NBK(TT, DYA, XA, 1);      \
        else if (d <= 512) LAUNCH_NBK(TT, DYA, XA, 2); \
        else if (d <= 1024) LAUNCH_NBK(TT, DYA, XA, 4);\
        else if (wide8 && d <= 2048) rownorm_bwd_wide_kernel<TT, DYA, XA, 256, 1, 8><<<grid, dim3(256), 0, s>>>(L);\
        else if (wide8 && d <= 4096) rownorm_bwd_wide_kernel<TT, DYA, XA, 512, 1, 8><<<grid, dim3(512), 0, s>>>(L);\
        else if (wide8 && d <= 8192) rownorm_bwd_wide_kernel<TT, DYA, XA, 1024, 1, 8><<<grid, dim3(1024), 0, s>>>(L);\
        else if (wide8) rownorm_bwd_wide_kernel<TT, DYA, XA, 1024, 2, 8><<<grid, dim3(1024), 0, s>>>(L);\
        else if (d <= 2048) rownorm_bwd_wide_kernel<TT, DYA, XA, 256, 2, 4><<<grid, dim3(256), 0, s>>>(L);\
        else if (d <= 4096) rownorm_bwd_wide_kernel<TT, DYA, XA, 512, 2, 4><<<grid, dim3(512), 0, s>>>(L);\
        else if (d <= 8192) rownorm_bwd_wide_kernel<TT, DYA, XA, 1024, 2, 4><<<grid, dim3(1024), 0, s>>>(L);\
        else rownorm_bwd_wide_kernel<TT, DYA, XA, 1024, 4, 4><<<grid, dim3(1024), 0, s>>>(L);\
    } while (0)
    if (dtype == SEA_BF16) {
        if (dy_is_act && x_is_act) LAUNCH_NB(__bf16, true, true);
        else if (dy_is_act) LAUNCH_NB(__bf16, true, false);
        else if (x_is_act) LAUNCH_NB(__bf16, false, true);
        else LAUNCH_NB(__bf16, false, false);
    } else {
        LAUNCH_NB(float, false, false);
    }
#undef LAUNCH_NB
#undef LAUNCH_NBK
    if (two_stage) {
        ColsumFinish F;
        memset(&F, 0, sizeof(F));
        for (int i = 0; i < n_groups; ++i) {
            F.out_a[i] = groups[i].dgamma;
            F.out_b[i] = groups[i].dbeta;
            F.width[i] = d;
        }
        F.ws = ws; F.nblk = nblk; F.stride = 2 * d;
        colsum_finish_kernel<<<dim3((2 * d + 255) / 256, n_groups, 16), dim3(256), 0, s>>>(F);
    }
    SEA_CHECK_LAUNCH("sea_rownorm_bwd");
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ silu outer backward
#define SEA_MAX_SILU_BWD_GROUPS 24
struct SiluBwdLaunch {
    SeaSiluBwdGroup g[SEA_MAX_SILU_BWD_GROUPS];
    const float* c;
    int M;
    float* ws;   // optional [n_groups][gridDim.x][2 * maxk]
    int maxk;
};

// Column sums in registers: lane owns columns lane*4 + 256k (k < 8, K2 <= 2048), w1/b1 of those columns are loaded once.
template <typename T, int KM>
__global__ __launch_bounds__(256) void silu_outer_bwd_kernel(const SiluBwdLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float cs[];  // [2][K2]
    const SeaSiluBwdGroup& G = L.g[blockIdx.y];
    const int K2 = G.K2, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * K2; i += 256) cs[i] = 0.f;
    __syncthreads();
    float aw[KM][4], ab[KM][4], w[KM][4], bb[KM][4];   // KM = ceil(max K2 / 256): registers scale with the row width
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        const int i = lane * 4 + 256 * k;
#pragma unroll
        for (int e = 0; e < 4; ++e) aw[k][e] = ab[k][e] = w[k][e] = bb[k][e] = 0.f;
        if (i < K2) {
            load4(G.w1 + i, w[k]);
            load4(G.b1 + i, bb[k]);
        }
    }
    // RB rows per trip with their loads requested together.  Measured at cfg3 (K2 = 512): RB = 1 111 us, 2 123 us, 4 129 us — the extra
    // registers cost more occupancy than the extra loads in flight return, so one row per trip it is.
    constexpr int RB = 1;
    const int rstep = 4 * gridDim.x;
    for (int row0 = blockIdx.x * 4 + wave; row0 < L.M; row0 += RB * rstep) {
        float cv[RB], dv[RB][KM][4];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int row = row0 + u * rstep;
            const int rc = row < L.M ? row : L.M - 1;
            cv[u] = L.c[rc];
            const T* dh = static_cast<const T*>(G.dHid) + (int64_t)rc * G.ld;
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                const int i = lane * 4 + 256 * k;
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[u][k][e] = 0.f;
                if (i < K2) load4(dh + i, dv[u][k]);
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const bool live = row0 + u * rstep < L.M;
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                const int i = lane * 4 + 256 * k;
                if (i < K2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pre = w[k][e] * cv[u] + bb[k][e];
                        const float sg = 1.0f / (1.0f + __expf(-pre));
                        const float dpre = live ? dv[u][k][e] * sg * (1.0f + pre * (1.0f - sg)) : 0.f;
                        aw[k][e] += dpre * cv[u];
                        ab[k][e] += dpre;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        const int i = lane * 4 + 256 * k;
        if (i < K2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(&cs[i + e], aw[k][e]);
                atomicAdd(&cs[K2 + i + e], ab[k][e]);
            }
        }
    }
    __syncthreads();
    if (L.ws != nullptr) {
        float* dst = L.ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * L.maxk;
        for (int i = threadIdx.x; i < K2; i += 256) {
            dst[i] = cs[i];
            dst[K2 + i] = cs[K2 + i];
        }
        return;
    }
    for (int i = threadIdx.x; i < K2; i += 256) {
        atomicAdd(G.dw1 + i, cs[i]);
        atomicAdd(G.db1 + i, cs[K2 + i]);
    }
}

// Column-block form: a workgroup owns 256 columns of one group and a slice of the rows; lane = 4 consecutive columns, the four waves take rows
// w, w + 4, ... of the slice (four rows of a wave in flight), partial sums meet in LDS and leave as ONE plain store per column into the two-stage
// workspace.  No LDS atomics, no per-workgroup slab of the whole row width: the row form above writes n_blocks x 2 K2 floats of partials per group —
// with the 798 rows of a shipped cylinder batch that was more bytes than the input (140 us for 23 MB) — and walks its rows one dependent load at a time.
struct SiluColsLaunch {
    SeaSiluBwdGroup g[SEA_MAX_SILU_BWD_GROUPS];
    int cb_start[SEA_MAX_SILU_BWD_GROUPS + 1];   // first column block of each group
    int n_groups;
    const float* c;
    int M, rows_per;
    float* ws;      // [n_groups][gridDim.y][2 * maxk]
    int maxk;
};

template <typename T>
__global__ __launch_bounds__(256) void silu_outer_bwd_cols_kernel(const SiluColsLaunch L) {
    __shared__ float red[3][8][64];
    int gi = 0;
    while (gi + 1 < L.n_groups && (int)blockIdx.x >= L.cb_start[gi + 1]) ++gi;
    const SeaSiluBwdGroup& G = L.g[gi];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K2 = G.K2;
    const int col = ((int)blockIdx.x - L.cb_start[gi]) * 256 + lane * 4;
    const bool act = col < K2;
    const int cc = act ? col : K2 - 4;
    float w[4], bb[4], aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    load4(G.w1 + cc, w);
    load4(G.b1 + cc, bb);
    const int r0 = blockIdx.y * L.rows_per, r1 = min(L.M, r0 + L.rows_per);
    const T* dh0 = static_cast<const T*>(G.dHid) + cc;
    for (int row = r0 + wave; row < r1; row += 16) {
        float cv[4], dv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // four rows requested together
            const int rr = row + 4 * u;
            const int rc = rr < r1 ? rr : row;
            cv[u] = L.c[rc];
            load4(dh0 + (int64_t)rc * G.ld, dv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool live = act && row + 4 * u < r1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pre = w[e] * cv[u] + bb[e];
                const float sg = 1.0f / (1.0f + __expf(-pre));
                const float dpre = live ? dv[u][e] * sg * (1.0f + pre * (1.0f - sg)) : 0.f;
                aw[e] += dpre * cv[u];
                ab[e] += dpre;
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[wave - 1][e][lane] = aw[e];
            red[wave - 1][4 + e][lane] = ab[e];
        }
    }
    __syncthreads();
    if (wave == 0 && act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            aw[e] += (red[0][e][lane] + red[1][e][lane]) + red[2][e][lane];
            ab[e] += (red[0][4 + e][lane] + red[1][4 + e][lane]) + red[2][4 + e][lane];
        }
        if (L.ws != nullptr) {
            float* dst = L.ws + ((int64_t)gi * gridDim.y + blockIdx.y) * 2 * L.maxk;
            store4(dst + col, aw[0], aw[1], aw[2], aw[3]);
            store4(dst + K2 + col, ab[0], ab[1], ab[2], ab[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(G.dw1 + col + e, aw[e]);
                atomicAdd(G.db1 + col + e, ab[e]);
            }
        }
    }
}

extern "C" int sea_silu_outer_bwd(const SeaSiluBwdGroup* groups, int n_groups, const float* c, int M, int dtype, float* ws, int64_t ws_floats,
                                  void* stream) {
    SEA_REQUIRE(groups && c && n_groups >= 1 && n_groups <= SEA_MAX_SILU_BWD_GROUPS && M >= 1, "sea_silu_outer_bwd: bad arguments");
    SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_BF16, "sea_silu_outer_bwd: bad dtype %d", dtype);
    SiluBwdLaunch L;
    memset(&L, 0, sizeof(L));
    int maxk = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaSiluBwdGroup& G = groups[i];
        SEA_REQUIRE(G.dHid && G.w1 && G.b1 && G.dw1 && G.db1 && G.K2 >= 4 && G.K2 % 4 == 0 && G.K2 <= 2048 && G.ld >= G.K2 && G.ld % 4 == 0, "sea_silu_outer_bwd[%d]: bad group", i);
        SEA_REQUIRE(sea_aligned16(G.dHid) && sea_aligned16(G.w1) && sea_aligned16(G.b1), "sea_silu_outer_bwd[%d]: pointers must be 16-byte aligned", i);
        L.g[i] = G;
        maxk = G.K2 > maxk ? G.K2 : maxk;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const int cols_form = sea_tune("silubwd_cols", 1);   // tuning aid: 0 = the one-wave-per-row kernel
    if (cols_form && ws != nullptr) {
        SiluColsLaunch Q;
        memset(&Q, 0, sizeof(Q));
        int ncb = 0;
        for (int i = 0; i < n_groups; ++i) {
            Q.g[i] = groups[i];
            Q.cb_start[i] = ncb;
            ncb += (groups[i].K2 + 255) / 256;
        }
        Q.cb_start[n_groups] = ncb;
        Q.n_groups = n_groups; Q.c = c; Q.M = M; Q.maxk = maxk;
        // row splits: about four workgroups per CU in all, at least 16 rows (one trip of a workgroup) per split, and what the workspace holds
        int rs = (1024 + ncb - 1) / ncb;
        const int rs_rows = (M + 15) / 16;
        rs = rs > rs_rows ? rs_rows : rs;
        const int64_t rs_ws = ws_floats / ((int64_t)n_groups * 2 * maxk);
        rs = rs > rs_ws ? (int)rs_ws : rs;
        if (rs >= 1) {
            Q.rows_per = (M + rs - 1) / rs;
            rs = (M + Q.rows_per - 1) / Q.rows_per;
            Q.ws = ws;
            if (dtype == SEA_BF16) silu_outer_bwd_cols_kernel<__bf16><<<dim3(ncb, rs), dim3(256), 0, s>>>(Q);
            else silu_outer_bwd_cols_kernel<float><<<dim3(ncb, rs), dim3(256), 0, s>>>(Q);
            ColsumFinish F;
            memset(&F, 0, sizeof(F));
            for (int i = 0; i < n_groups; ++i) {
                F.out_a[i] = groups[i].dw1;
                F.out_b[i] = groups[i].db1;
                F.width[i] = groups[i].K2;
            }
            F.ws = ws; F.nblk = rs; F.stride = 2 * maxk;
            colsum_finish_kernel<<<dim3((2 * maxk + 255) / 256, n_groups, rs < 16 ? rs : 16), dim3(256), 0, s>>>(F);
            SEA_CHECK_LAUNCH("sea_silu_outer_bwd");
            return SEA_OK;
        }
    }
    L.c = c; L.M = M;
    int nblk = (M + 3) / 4;
    // every workgroup of the launch resident at once (about 6 per CU): measured at cfg3 (12 groups) 128 per group 76 us, 160 90 us, 256 112 us
    static const int sb_cap = sea_tune("silubwd_blocks", 0);  // tuning aid
    int cap = sb_cap > 0 ? sb_cap : (6 * 256 / n_groups) / 32 * 32;
    cap = cap < 32 ? 32 : (cap > 256 ? 256 : cap);
    if (nblk > cap) nblk = cap;
    const bool two_stage = ws != nullptr && ws_floats >= (int64_t)n_groups * nblk * 2 * maxk && nblk > 8;
    L.ws = two_stage ? ws : nullptr;
    L.maxk = maxk;
    const size_t lds = (size_t)2 * maxk * sizeof(float);
#define LAUNCH_SB(TT, KMV)                                                                                                            \
    do {                                                                                                                              \
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(silu_outer_bwd_kernel<TT, KMV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        silu_outer_bwd_kernel<TT, KMV><<<dim3(nblk, n_groups), dim3(256), lds, s>>>(L);                                                \
    } while (0)
#define LAUNCH_SB_T(TT)                                              \
    do {                                                              \
        if (maxk <= 256) LAUNCH_SB(TT, 1);                            \
        else if (maxk <= 512) LAUNCH_SB(TT, 2);                       \
        else if (maxk <= 1024) LAUNCH_SB(TT, 4);                      \
        else LAUNCH_SB(TT, 8);                                        \
    } while (0)
    if (dtype == SEA_BF16) LAUNCH_SB_T(__bf16); else LAUNCH_SB_T(float);
#undef LAUNCH_SB_T
#undef LAUNCH_SB
    if (two_stage) {
        ColsumFinish F;
        memset(&F, 0, sizeof(F));
        for (int i = 0; i < n_groups; ++i) {
            F.out_a[i] = groups[i].dw1;
            F.out_b[i] = groups[i].db1;
            F.width[i] = groups[i].K2;
        }
        F.ws = ws; F.nblk = nblk; F.stride = 2 * maxk;
        colsum_finish_kernel<<<dim3((2 * maxk + 255) / 256, n_groups, 16), dim3(256), 0, s>>>(F);
    }
    SEA_CHECK_LAUNCH("sea_silu_outer_bwd");
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ information bottleneck backward
// x_f += ib(c) leaves dx unchanged; this kernel produces the PARAMETER gradients of ib = W2 gelu(LN_h(w1 c + b1)) + b2 from
// dib = sum_f dX_f.  One wave per row; LDS accumulators for db2 [E] and dW2 [E][h]; the h-wide hidden state lives in lanes 0..h-1.
__global__ __launch_bounds__(256) void ib_bwd_kernel(const SeaIbBwdParams P) {
    extern __shared__ __attribute__((aligned(16))) float cs[];  // [E] db2, [E*h] dW2
    const int E = P.E, h = P.h, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s_b2 = cs;
    float* s_w2 = cs + E;
    for (int i = threadIdx.x; i < E * (h + 1); i += 256) cs[i] = 0.f;
    __syncthreads();
    const bool act = lane < h;
    const float w1 = act ? P.w1[lane] : 0.f, b1 = act ? P.b1[lane] : 0.f;
    const float lw = act ? P.lnw[lane] : 0.f, lb = act ? P.lnb[lane] : 0.f;
    float a_w1 = 0.f, a_b1 = 0.f, a_lw = 0.f, a_lb = 0.f;  // lane k accumulates the gradients of hidden unit k over its rows
    for (int row = blockIdx.x * 4 + wave; row < P.M; row += 4 * gridDim.x) {
        const float cv = P.c[row];
        const float pre = act ? w1 * cv + b1 : 0.f;
        const float mean = wave_sum(pre) / (float)h;
        const float cen = act ? pre - mean : 0.f;
        const float var = wave_sum(cen * cen) / (float)h;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const float xh = cen * rstd;
        const float u = xh * lw + lb;
        const float hid = act ? gelu_erf(u) : 0.f;
        float dhid = 0.f;  // lane k will hold d(hid_k)
        for (int e0 = lane * 4; e0 < E; e0 += 256) {
            float dib[4] = {0.f, 0.f, 0.f, 0.f};
            for (int f = 0; f < P.n_fields; ++f) {
                float v[4];
                load4(P.dX[f] + (int64_t)row * P.ldx + e0, v);
                if (P.drop.thr > 0) {
                    const uint32_t w = drop_word(P.drop.seed, P.drop.stream + f, (uint32_t)row, (uint32_t)(e0 >> 2));
                    const float sc = drop_scale(P.drop.thr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= drop_factor(w, e, P.drop.thr, sc);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dib[e] += v[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(&s_b2[e0 + e], dib[e]);
            for (int k = 0; k < h; ++k) {
                const float hk = __shfl(hid, k);
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(&s_w2[(e0 + e) * h + k], dib[e] * hk);
            }
        }
        // dhid_k = sum_e dib[e] W2[e][k]: every lane re-walks its columns per k (E*h is small), then a wave sum per k
        for (int k = 0; k < h; ++k) {
            float part = 0.f;
            for (int e0 = lane * 4; e0 < E; e0 += 256) {
                float dib[4] = {0.f, 0.f, 0.f, 0.f};
                for (int f = 0; f < P.n_fields; ++f) {
                    float v[4];
                    load4(P.dX[f] + (int64_t)row * P.ldx + e0, v);
                    if (P.drop.thr > 0) {
                        const uint32_t w = drop_word(P.drop.seed, P.drop.stream + f, (uint32_t)row, (uint32_t)(e0 >> 2));
                        const float sc = drop_scale(P.drop.thr);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= drop_factor(w, e, P.drop.thr, sc);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) dib[e] += v[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) part += dib[e] * P.w2[(int64_t)(e0 + e) * h + k];
            }
            part = wave_sum(part);
            if (lane == k) dhid = part;
        }
        const float du = act ? dhid * gelu_erf_grad(u) : 0.f;
        a_lw += du * xh;
        a_lb += du;
        const float dxh = du * lw;
        const float c1 = wave_sum(dxh) / (float)h;
        const float c2 = wave_sum(dxh * xh) / (float)h;
        const float dpre = act ? rstd * (dxh - c1 - xh * c2) : 0.f;
        a_w1 += dpre * cv;
        a_b1 += dpre;
    }
    if (act) {
        atomicAdd(P.dw1 + lane, a_w1);
        atomicAdd(P.db1 + lane, a_b1);
        atomicAdd(P.dlnw + lane, a_lw);
        atomicAdd(P.dlnb + lane, a_lb);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < E; i += 256) atomicAdd(P.db2 + i, s_b2[i]);
    for (int i = threadIdx.x; i < E * h; i += 256) atomicAdd(P.dw2 + i, s_w2[i]);
}

// Fast path (h <= 8, E <= 256*KE): dib, db2 and dW2 partial sums live in registers (lane owns columns lane*4 + 256k); one pass over
// dX per row; the partial sums are merged through LDS once per workgroup.
template <int KE>
__global__ __launch_bounds__(256) void ib_bwd_fast_kernel(const SeaIbBwdParams P) {
    extern __shared__ __attribute__((aligned(16))) float cs[];  // [E] db2, [E*h] dW2
    constexpr int HM = 8;
    const int E = P.E, h = P.h, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s_b2 = cs;
    float* s_w2 = cs + E;
    for (int i = threadIdx.x; i < E * (h + 1); i += 256) cs[i] = 0.f;
    __syncthreads();
    const bool act = lane < h;
    const float w1 = act ? P.w1[lane] : 0.f, b1 = act ? P.b1[lane] : 0.f;
    const float lw = act ? P.lnw[lane] : 0.f, lb = act ? P.lnb[lane] : 0.f;
    float a_w1 = 0.f, a_b1 = 0.f, a_lw = 0.f, a_lb = 0.f;
    float ab2[KE][4], aw2[KE][4][HM];
#pragma unroll
    for (int k = 0; k < KE; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ab2[k][e] = 0.f;
#pragma unroll
            for (int j = 0; j < HM; ++j) aw2[k][e][j] = 0.f;
        }
    float w2r[KE][4][HM];   // the lane's rows of W2: row-invariant, loaded once
#pragma unroll
    for (int k = 0; k < KE; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < HM; ++j) {
                const int col = lane * 4 + 256 * k + e;
                w2r[k][e][j] = (col < E && j < h) ? P.w2[(int64_t)col * h + j] : 0.f;
            }
    for (int row = blockIdx.x * 4 + wave; row < P.M; row += 4 * gridDim.x) {
        const float cv = P.c[row];
        const float pre = act ? w1 * cv + b1 : 0.f;
        const float mean = wave_sum(pre) / (float)h;
        const float cen = act ? pre - mean : 0.f;
        const float var = wave_sum(cen * cen) / (float)h;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const float xh = cen * rstd;
        const float u = xh * lw + lb;
        const float hid = act ? gelu_erf(u) : 0.f;
        float hk[HM], part[HM];
#pragma unroll
        for (int j = 0; j < HM; ++j) {
            hk[j] = __shfl(hid, j);
            part[j] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < KE; ++k) {
            const int e0 = lane * 4 + 256 * k;
            if (e0 < E) {
                float dib[4] = {0.f, 0.f, 0.f, 0.f};
                for (int f = 0; f < P.n_fields; ++f) {
                    float v[4];
                    load4(P.dX[f] + (int64_t)row * P.ldx + e0, v);
                    if (P.drop.thr > 0) {
                        const uint32_t w = drop_word(P.drop.seed, P.drop.stream + f, (uint32_t)row, (uint32_t)(e0 >> 2));
                        const float sc = drop_scale(P.drop.thr);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= drop_factor(w, e, P.drop.thr, sc);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) dib[e] += v[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ab2[k][e] += dib[e];
#pragma unroll
                    for (int j = 0; j < HM; ++j) {
                        aw2[k][e][j] += dib[e] * hk[j];
                        part[j] += dib[e] * w2r[k][e][j];
                    }
                }
            }
        }
        float dhid = 0.f;
#pragma unroll
        for (int j = 0; j < HM; ++j) {
            const float tot = wave_sum(part[j]);
            if (lane == j) dhid = tot;
        }
        const float du = act ? dhid * gelu_erf_grad(u) : 0.f;
        a_lw += du * xh;
        a_lb += du;
        const float dxh = du * lw;
        const float c1 = wave_sum(dxh) / (float)h;
        const float c2 = wave_sum(dxh * xh) / (float)h;
        const float dpre = act ? rstd * (dxh - c1 - xh * c2) : 0.f;
        a_w1 += dpre * cv;
        a_b1 += dpre;
    }
    if (act) {
        atomicAdd(P.dw1 + lane, a_w1);
        atomicAdd(P.db1 + lane, a_b1);
        atomicAdd(P.dlnw + lane, a_lw);
        atomicAdd(P.dlnb + lane, a_lb);
    }
#pragma unroll
    for (int k = 0; k < KE; ++k) {
        const int e0 = lane * 4 + 256 * k;
        if (e0 < E) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(&s_b2[e0 + e], ab2[k][e]);
#pragma unroll
                for (int j = 0; j < HM; ++j)
                    if (j < h) atomicAdd(&s_w2[(e0 + e) * h + j], aw2[k][e][j]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < E; i += 256) atomicAdd(P.db2 + i, s_b2[i]);
    for (int i = threadIdx.x; i < E * h; i += 256) atomicAdd(P.dw2 + i, s_w2[i]);
}

// ---- column-block form (h <= 8): three launches.
//  (1) ib_bwd_cols_kernel, grid (E / 256 column blocks, row splits): lane = 4 consecutive columns, the four waves take rows w, w + 4, ... of the split.
//      Per row: dib = sum_f dX_f (dropout as in the forward), db2 / dW2 partial sums in registers (the row's hidden vector comes from lanes 0 .. h-1),
//      and the row's d hidden_j = sum_e dib_e W2[e][j]: 8 partial dot products per lane, summed over the wave by a transposing butterfly (10 exchanges
//      instead of 8 x 6) and stored (one column block) or added (several) into dhid [M, 8].  The partial sums leave through LDS as plain stores
//      into the workspace [row split][E | E h].
//  (2) ib_bwd_rows_kernel, one thread per row: the scalar chain hidden -> GELU -> LayerNorm_h -> w1 c + b1 backwards, 4 h sums -> 4 h atomics per workgroup.
//  (3) ib_bwd_finish_kernel: db2 / dW2 += sum over the row splits.
// The one-wave-per-row kernels above keep dW2 for all the wave's columns in registers (272 of them at E = 1024: spilled) and end in n_blocks x E (1 + h)
// atomics onto the same addresses: 80 us for the 6.5 MB of a shipped cylinder batch.
__device__ __forceinline__ void ib_hidden(float cv, float w1, float b1, float lw, float lb, int h, bool act, float& xh, float& u, float& rstd) {
    const float pre = act ? w1 * cv + b1 : 0.f;
    const float mean = wave_sum(pre) / (float)h;
    const float cen = act ? pre - mean : 0.f;
    const float var = wave_sum(cen * cen) / (float)h;
    rstd = 1.0f / sqrtf(var + 1e-5f);
    xh = cen * rstd;
    u = xh * lw + lb;
}

__global__ __launch_bounds__(256) void ib_bwd_cols_kernel(const SeaIbBwdParams P, int rows_per, int n_cb) {
    constexpr int HM = 8;
    __shared__ float red[3][4 * (HM + 1)][64];
    const int E = P.E, h = P.h, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = blockIdx.x * 256 + lane * 4;
    const bool actc = col < E;
    const int cc = actc ? col : E - 4;
    const bool act = lane < h;
    const float w1 = act ? P.w1[lane] : 0.f, b1 = act ? P.b1[lane] : 0.f, lw = act ? P.lnw[lane] : 0.f, lb = act ? P.lnb[lane] : 0.f;
    float w2r[4][HM], ab2[4], aw2[4][HM];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ab2[e] = 0.f;
#pragma unroll
        for (int j = 0; j < HM; ++j) {
            aw2[e][j] = 0.f;
            w2r[e][j] = (actc && j < h) ? P.w2[(int64_t)(col + e) * h + j] : 0.f;
        }
    }
    const float dsc = P.drop.thr > 0 ? drop_scale(P.drop.thr) : 1.f;
    const int r0 = blockIdx.y * rows_per, r1 = min(P.M, r0 + rows_per);
    for (int row = r0 + wave; row < r1; row += 4) {
        float dib[4] = {0.f, 0.f, 0.f, 0.f};
        for (int f = 0; f < P.n_fields; ++f) {
            float v[4];
            load4(P.dX[f] + (int64_t)row * P.ldx + cc, v);
            if (P.drop.thr > 0) {
                const uint32_t wd = drop_word(P.drop.seed, P.drop.stream + f, (uint32_t)row, (uint32_t)(cc >> 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= drop_factor(wd, e, P.drop.thr, dsc);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dib[e] += v[e];
        }
        if (!actc) dib[0] = dib[1] = dib[2] = dib[3] = 0.f;
        float xh, u, rstd;
        ib_hidden(P.c[row], w1, b1, lw, lb, h, act, xh, u, rstd);
        const float hid = act ? gelu_erf(u) : 0.f;
        float part[HM];
#pragma unroll
        for (int j = 0; j < HM; ++j) {
            const float hj = __shfl(hid, j);
            part[j] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                aw2[e][j] += dib[e] * hj;
                part[j] += dib[e] * w2r[e][j];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ab2[e] += dib[e];
        // wave sum of the 8 partial dot products: after exchanges at distances 1, 2, 4 every lane holds ONE of them summed over its 8-lane group
        // (which one: j = 4 b0 + 2 b1 + b2 of its lane bits), three more exchanges sum the groups
        float q4[4], q2[2], q1;
        {
            const bool b0 = lane & 1, b1_ = lane & 2, b2 = lane & 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float send = b0 ? part[i] : part[i + 4], keep = b0 ? part[i + 4] : part[i];
                q4[i] = keep + __shfl_xor(send, 1);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float send = b1_ ? q4[i] : q4[i + 2], keep = b1_ ? q4[i + 2] : q4[i];
                q2[i] = keep + __shfl_xor(send, 2);
            }
            const float send = b2 ? q2[0] : q2[1], keep = b2 ? q2[1] : q2[0];
            q1 = keep + __shfl_xor(send, 4);
            q1 += __shfl_xor(q1, 8);
            q1 += __shfl_xor(q1, 16);
            q1 += __shfl_xor(q1, 32);
        }
        if (lane < 8) {
            const int j = ((lane & 1) << 2) | (lane & 2) | ((lane >> 2) & 1);
            if (n_cb == 1) P.dhid[(int64_t)row * HM + j] = q1;
            else atomicAdd(P.dhid + (int64_t)row * HM + j, q1);
        }
    }
    // the four waves' partial sums of this column block: waves 1 .. 3 through LDS, wave 0 stores
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[wave - 1][e][lane] = ab2[e];
#pragma unroll
            for (int j = 0; j < HM; ++j) red[wave - 1][4 + e * HM + j][lane] = aw2[e][j];
        }
    }
    __syncthreads();
    if (wave == 0 && actc) {
        float* dst = P.ws + (int64_t)blockIdx.y * E * (1 + h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dst[col + e] = ab2[e] + (red[0][e][lane] + red[1][e][lane]) + red[2][e][lane];
#pragma unroll
            for (int j = 0; j < HM; ++j)
                if (j < h) dst[E + (int64_t)(col + e) * h + j] = aw2[e][j] + (red[0][4 + e * HM + j][lane] + red[1][4 + e * HM + j][lane]) + red[2][4 + e * HM + j][lane];
        }
    }
}

__global__ __launch_bounds__(256) void ib_bwd_rows_kernel(const SeaIbBwdParams P, int rezero) {
    constexpr int HM = 8;
    __shared__ float red[4][4 * HM];
    const int h = P.h, row = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[4][HM];   // d w1 | d b1 | d lnw | d lnb of hidden unit j, this thread's row
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < HM; ++j) acc[q][j] = 0.f;
    if (row < P.M) {
        const float cv = P.c[row];
        float pre[HM], mean = 0.f, var = 0.f;
#pragma unroll
        for (int j = 0; j < HM; ++j) {
            pre[j] = j < h ? P.w1[j] * cv + P.b1[j] : 0.f;
            mean += pre[j];
        }
        mean /= (float)h;
#pragma unroll
        for (int j = 0; j < HM; ++j) {
            pre[j] = j < h ? pre[j] - mean : 0.f;
            var += pre[j] * pre[j];
        }
        const float rstd = 1.0f / sqrtf(var / (float)h + 1e-5f);
        float dxh[HM], xh[HM], c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < HM; ++j) {
            xh[j] = pre[j] * rstd;
            const float lwj = j < h ? P.lnw[j] : 0.f, lbj = j < h ? P.lnb[j] : 0.f;
            const float u = xh[j] * lwj + lbj;
            const float du = j < h ? P.dhid[(int64_t)row * HM + j] * gelu_erf_grad(u) : 0.f;
            if (rezero) P.dhid[(int64_t)row * HM + j] = 0.f;   // several column blocks ADD into dhid: leave it zero for the next step
            acc[2][j] = du * xh[j];
            acc[3][j] = du;
            dxh[j] = du * lwj;
            c1 += dxh[j];
            c2 += dxh[j] * xh[j];
        }
        c1 /= (float)h;
        c2 /= (float)h;
#pragma unroll
        for (int j = 0; j < HM; ++j) {
            const float dpre = j < h ? rstd * (dxh[j] - c1 - xh[j] * c2) : 0.f;
            acc[0][j] = dpre * cv;
            acc[1][j] = dpre;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < HM; ++j) {
            const float t = wave_sum(acc[q][j]);
            if (lane == 0) red[wave][q * HM + j] = t;
        }
    __syncthreads();
    if (threadIdx.x < 4 * HM) {
        const int q = threadIdx.x / HM, j = threadIdx.x % HM;
        if (j < h) {
            const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            float* dst = q == 0 ? P.dw1 : (q == 1 ? P.db1 : (q == 2 ? P.dlnw : P.dlnb));
            atomicAdd(dst + j, t);
        }
    }
}

// gridDim.y slices of the row splits (as colsum_finish_kernel: 16 adders per address, and enough workgroups to stream the partials — 9 workgroups walking 1024
// splits one after the other took as long as the column kernel itself, 34 us at cfg3)
__global__ __launch_bounds__(256) void ib_bwd_finish_kernel(const SeaIbBwdParams P, int rs) {
    const int n = P.E * (1 + P.h);
    const int per = (rs + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = b0 + per < rs ? b0 + per : rs;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float acc = 0.f;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) acc += P.ws[(int64_t)b * n + i];
        if (b0 < b1) atomicAdd(i < P.E ? P.db2 + i : P.dw2 + (i - P.E), acc);
    }
}

// ib = nn.Linear(1, E): dw[e] += sum_m c[m] g[m, e], db[e] += sum_m g[m, e] with g = sum over the fields of dX_f.  grid = (E / 64 column blocks, row chunks);
// lane = column, the four waves of a workgroup take rows w, w + 4, ... of the chunk, partial sums meet in LDS, one atomic per column per workgroup.
__global__ __launch_bounds__(256) void ib_linear_bwd_kernel(const SeaIbBwdParams P) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int rows_per = (P.M + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(P.M, r0 + rows_per);
    float aw = 0.f, ab = 0.f;
    if (col < P.E) {
        for (int m = r0 + wave; m < r1; m += 4) {
            float gsum = 0.f;
            for (int f = 0; f < P.n_fields; ++f) gsum += P.dX[f][(int64_t)m * P.ldx + col];
            aw = fmaf(P.c[m], gsum, aw);
            ab += gsum;
        }
    }
    red[0][wave][lane] = aw;
    red[1][wave][lane] = ab;
    __syncthreads();
    if (wave == 0 && col < P.E) {
        atomicAdd(P.dw1 + col, (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]));
        atomicAdd(P.db1 + col, (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]));
    }
}

extern "C" int sea_ib_bwd(const SeaIbBwdParams* params, void* stream) {
    SEA_REQUIRE(params != nullptr, "sea_ib_bwd: null params");
    const SeaIbBwdParams& P = *params;
    if (P.mode == 1) {
        SEA_REQUIRE(P.n_fields >= 1 && P.n_fields <= 8 && P.M >= 1 && P.E >= 1 && P.ldx >= P.E && P.c && P.dw1 && P.db1, "sea_ib_bwd (linear): bad arguments");
        for (int f = 0; f < P.n_fields; ++f) SEA_REQUIRE(P.dX[f] != nullptr, "sea_ib_bwd (linear): dX[%d] null", f);
        const int chunks = P.M >= 4096 ? 64 : (P.M + 63) / 64;
        ib_linear_bwd_kernel<<<dim3((P.E + 63) / 64, chunks), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(P);
        SEA_CHECK_LAUNCH("sea_ib_bwd");
        return SEA_OK;
    }
    SEA_REQUIRE(P.mode == 0, "sea_ib_bwd: mode %d", P.mode);
    SEA_REQUIRE(P.n_fields >= 1 && P.n_fields <= 8 && P.M >= 1 && P.E >= 4 && P.E % 4 == 0 && P.h >= 1 && P.h <= 64 && P.ldx >= P.E && P.ldx % 4 == 0,
                "sea_ib_bwd: bad sizes");
    SEA_REQUIRE(P.c && P.w1 && P.b1 && P.lnw && P.lnb && P.w2 && P.dw1 && P.db1 && P.dlnw && P.dlnb && P.dw2 && P.db2, "sea_ib_bwd: null pointer");
    for (int f = 0; f < P.n_fields; ++f) SEA_REQUIRE(P.dX[f] && sea_aligned16(P.dX[f]), "sea_ib_bwd: dX[%d] null or misaligned", f);
    hipStream_t s0 = static_cast<hipStream_t>(stream);
    static const int cols_form = sea_tune("ibbwd_cols", 1);   // tuning aid: 0 = the one-wave-per-row kernels
    if (cols_form && P.h <= 8 && P.ws != nullptr && P.dhid != nullptr && P.ws_floats >= (int64_t)P.E * (1 + P.h)) {
        const int n_cb = (P.E + 255) / 256;
        int rs = (1024 + n_cb - 1) / n_cb;                       // about four workgroups per CU in all
        const int rs_rows = (P.M + 15) / 16;                     // ... of at least 16 rows
        const int64_t rs_ws = P.ws_floats / ((int64_t)P.E * (1 + P.h));
        rs = rs > rs_rows ? rs_rows : rs;
        rs = rs > rs_ws ? (int)rs_ws : rs;
        const int rows_per = (P.M + rs - 1) / rs;
        rs = (P.M + rows_per - 1) / rows_per;
        ib_bwd_cols_kernel<<<dim3(n_cb, rs), dim3(256), 0, s0>>>(P, rows_per, n_cb);
        ib_bwd_rows_kernel<<<dim3((P.M + 255) / 256), dim3(256), 0, s0>>>(P, n_cb > 1 ? 1 : 0);
        const int n = P.E * (1 + P.h);
        ib_bwd_finish_kernel<<<dim3((n + 255) / 256 < 256 ? (n + 255) / 256 : 256, rs >= 64 ? 16 : 1), dim3(256), 0, s0>>>(P, rs);
        SEA_CHECK_LAUNCH("sea_ib_bwd");
        return SEA_OK;
    }
    const size_t lds = (size_t)P.E * (P.h + 1) * sizeof(float);
    SEA_REQUIRE(lds <= 160 * 1024, "sea_ib_bwd: E*(h+1) too large for LDS");
    if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ib_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int nblk = (P.M + 3) / 4;
    static const int ib_cap = sea_tune("ibbwd_blocks", 256);  // tuning aid
    if (nblk > ib_cap) nblk = ib_cap;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (P.h <= 8 && P.E <= 256) ib_bwd_fast_kernel<1><<<dim3(nblk), dim3(256), lds, s>>>(P);
    else if (P.h <= 8 && P.E <= 1024) ib_bwd_fast_kernel<4><<<dim3(nblk), dim3(256), lds, s>>>(P);
    else ib_bwd_kernel<<<dim3(nblk), dim3(256), lds, s>>>(P);
    SEA_CHECK_LAUNCH("sea_ib_bwd");
    return SEA_OK;
}
