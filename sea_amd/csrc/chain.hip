// Row-local chains of the state-exchange block in ONE launch per chain (gfx950, bf16): sea_row_chain.
//
// Everything between two attention launches of SEABlockTemporal (reference models/temporal.py:126-192) is ROW-LOCAL: a row of the residual stream goes
// through a few Linear layers, a GELU, a row norm and the rotary epilogue of the next attention's operands without ever looking at another row.  At one
// trajectory (M = 2024 rows per field) each of those layers is a launch of a few microseconds that is mostly launch, pipeline fill and drain; the chain
// here keeps a workgroup's 16 or 32 rows on its CU from the attention output to the NEXT attention's Q / K / V^T operands:
//
//   form A (n_seg = 0, after the self-attention):   x = Xin + a2 . W2^T (+ b2)                       self-attention out-projection + residual  (base_blocks.py:201, temporal.py:136)
//   form B (n_seg >= 1, a field's exchange tail):   g = sum_s gelu(att_s . Wp_s^T);  x = Xin + g . W2^T + S b2   projections + GELU, cross_up of the sum + residual  (temporal.py:183-191)
//   then, when has_down:                            y = norm(x . Wd^T + bd)                           cross_down + ln_cross of the rows as they are NOW  (temporal.py:177-181)
//   then, for each of n_proj projection entries:    [q | k | v] columns of  y . Wq/k/v^T + b  -> rotary embedding, q scale, the attention layouts
//                                                                                                      (base_blocks.py:271-280; every q of the field, and the k / v of
//                                                                                                      the pairs that read THIS field's rows at this point of the Gauss-Seidel sweep)
//
// With it the cfg2 forward has no cross-attention QKV launch, no out-projection launch and no down + norm launch: 18 launches -> 14.
//
// A workgroup = 4 waves owns BM = 16 MI rows (MI = 1: short launches; MI = 2 when 16-row workgroups would not fit the chip in one round); wave w owns the column
// quarter w of every layer's output (transposed MFMA tile of gemm_core.hpp: lane & 15 = row, lane >> 4 = column quad).  Every weight matrix goes L2 -> LDS by
// global_load_lds bursts of whole 128-byte lines (no VGPRs); LDS holds two 64 KiB weight halves that the layers take turns in — a half is refilled as soon as
// the layer that read it is done, so a burst lands under the layer in front of its consumer — plus the A tile / bf16 x tile, the g / y tile, the rotary rows
// and projection biases of the workgroup's rows, and the statistics scratch.
#include "norm_epilogue.hpp"
#include "gemm_tile.hpp"
#include "ib_rows.hpp"
#include <stdlib.h>

// Riders: work of LATER launches that depends on nothing this launch computes, run by extra workgroups of this launch on the CUs its chain workgroups leave idle
// (a chain launch at one trajectory is 127-192 workgroups on 256 CUs).  Two kinds: 128 x 128 tiles of plain grouped GEMMs (cond_mlp.2 of the AdaLN modules the
// field MLP and the final norm read, models/base_blocks.py:339,344: their operand, silu(cond_mlp.0), comes from the silu launch in front) and rows of the
// information-bottleneck MLP (models/temporal.py:111-116) stored for the pass that adds them.  grid.y = n_groups is the rider row.
struct ChainLaunch {
    SeaRowChain p[SEA_CHAIN_MAX_GROUPS];   // grid.y = group (field)
    SeaQkvCommon c;
    float eps;
    int n_groups;
    unsigned long long* stamps;   // tuning aid (sea_chain_debug_stamps): 16 clock stamps (100 MHz) per workgroup, or NULL
    ChainRiderPod rg[SEA_CHAIN_MAX_RIDERS];
    int n_riders, rider_t0, rider_n, ib_wgs;   // this launch runs rider tiles [rider_t0, rider_t0 + rider_n) and ib_wgs workgroups of info-bottleneck rows
    SeaIbParams ib;
};
static_assert(sizeof(ChainLaunch) <= 4096, "kernel arguments of sea_row_chain");

static unsigned long long* g_chain_stamps = nullptr;
// Tuning aid, not part of the ABI proper: a device buffer of (workgroups of the next launches) * 16 8-byte words receives time stamps of the chain's phases.
extern "C" void sea_chain_debug_stamps(unsigned long long* buf) { g_chain_stamps = buf; }

template <int D, int E, int MI>
struct ChainCfg {
    static constexpr int BM = 16 * MI, BKB = 128, BK = 64;
    static constexpr int KTD = D / BK, KTE = E / BK;
    static constexpr int AT_BYTES = BM * E * 2;            // A tile: att_s (n_seg D <= E columns) or a2 (E columns); later the bf16 x tile (A of the down-projection)
    static constexpr int YT_BYTES = BM * D * 2;            // g (A of the up-projection, form B); later the normalised rows y (A of the projections)
    static constexpr int WH_BYTES = E * D * 2;             // one weight half: n_seg Wp | W2 [E, D] | half the K-tiles of W2 [E, E] | Wd [D, E] | projection rows [<= 2 D, D]
    static constexpr int ROPE_BYTES = 2048;                // (cos, sin) pairs of the workgroup's rows: BM * hd / 2 * 8 bytes
    static constexpr int BIAS_BYTES = 4096;                // projection biases: sum of N <= 1024 floats
    static constexpr int RED_BYTES = MI * 512;
    static constexpr int LDS_BYTES = AT_BYTES + YT_BYTES + 2 * WH_BYTES + ROPE_BYTES + BIAS_BYTES + RED_BYTES;
    static constexpr int SMAX = E / D;                     // segments of form B: their Wp share one half, their att tiles the A tile
    static constexpr int NI1 = D / 64, NI2 = E / 64, NI3 = D / 64, NI4 = 2 * D / 64;   // 16-column blocks per wave: stage 1 (D columns), 2 (E), 3 (D), 4 (up to 2 D)
    // every LDS-DMA destination lies inside the request: A tile <= AT_BYTES (n_seg D <= E, K2 <= E), a weight burst <= WH_BYTES (checked on the host per entry)
    static_assert(E == 2 * D && LDS_BYTES <= 160 * 1024 && SMAX * D * D * 2 <= WH_BYTES && (KTE / 2) * E * BKB <= WH_BYTES && D * E * 2 <= WH_BYTES, "LDS plan of sea_row_chain");
};

// One wave-instruction of LDS-DMA per (address, LDS destination) pair, four in one asm block (M0 saved / restored once per block).
__device__ __forceinline__ void glds16_x4(const void* p0, const void* p1, const void* p2, const void* p3, unsigned l0, unsigned l1, unsigned l2, unsigned l3) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
        "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
        "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(l0), "s"(l1), "s"(l2), "s"(l3) : "memory");
}

// n_rows rows (a multiple of 32) of a K-contiguous bf16 WEIGHT matrix -> K-tiled LDS block: K-tile kt (64 columns from k0 + 64 kt) at dst + kt * n_rows * 128, row i
// at + i * 128, 16-byte chunk c of a row at chunk position c ^ (i & 7) (swizzle on the source side).  All 4 waves cooperate: wave w takes the 8-row slabs
// w, w + 4, ...; addresses advance by pointer increments, four pieces per asm block (issuing a piece used to cost ~110 cycles of 64-bit multiplies, M0 moves
// and loop control: a third of a burst's time, tools/chain_probe.py stamps).
__device__ __forceinline__ void chain_dma_w(const __bf16* src, int ld, int n_rows, int k0, int n_kt, unsigned dst, int wv, int lane) {
    const int rl = lane >> 3;
    const int chunk = (lane & 7) ^ (rl & 7);
    const char* p = reinterpret_cast<const char*>(src + (int64_t)(wv * 8 + rl) * ld + k0 + chunk * 8);
    const int64_t step = (int64_t)ld * 64;        // 32 rows (4 waves x 8) in bytes
    const int n4 = n_rows >> 7;                   // groups of four slabs per wave
    const int rem = (n_rows >> 5) & 3;            // slabs left over (n_rows / 32 mod 4)
    for (int kt = 0; kt < n_kt; ++kt) {
        const char* q = p + kt * 128;
        unsigned l = dst + (unsigned)(kt * n_rows * 128 + wv * 1024);
        for (int i = 0; i < n4; ++i) {
            glds16_x4(q, q + step, q + 2 * step, q + 3 * step, l, l + 4096u, l + 8192u, l + 12288u);
            q += 4 * step;
            l += 16384u;
        }
        for (int i = 0; i < rem; ++i) {
            glds16_gn(q, l);
            q += step;
            l += 4096u;
        }
    }
}

// the workgroup's BM rows [m0, m0 + BM) of an activation matrix (rows beyond row_last repeat it) as n_kt K-tiles of BM rows
__device__ __forceinline__ void chain_dma_a(const __bf16* src, int ld, int m0, int row_last, int bm, int n_kt, unsigned dst, int wv, int lane) {
    const int rl = lane >> 3;
    const int chunk = (lane & 7) ^ (rl & 7);
    for (int u = wv; u < (bm >> 3); u += 4) {
        int row = m0 + u * 8 + rl;
        row = row < row_last ? row : row_last;
        const char* q = reinterpret_cast<const char*>(src + (int64_t)row * ld + chunk * 8);
        for (int kt = 0; kt < n_kt; ++kt) glds16_gn(q + kt * 128, dst + (unsigned)(kt * bm * 128 + u * 1024));
    }
}

// workgroup barrier for LDS traffic only: the wave's LDS operations are retired, its global stores and DMA bursts stay in flight (__syncthreads() would wait for them)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// acc[i][j] += A tile (K-tiles of BM rows at sA, this lane's row r of row block i) . W rows (K-tiles of w_rows rows at sW; this wave's 16-row blocks j from wrow0)
template <int MI, int NI>
__device__ __forceinline__ void chain_mma(const char* sA, int a_kts, const char* sW, int w_kts, int wrow0, int n_kt, int ni, int r, int g, f32x4 (&acc)[MI][NI]) {
    for (int kt = 0; kt < n_kt; ++kt) {
        const char* pa = sA + kt * a_kts + r * 128;
        const char* pw = sW + kt * w_kts + (wrow0 + r) * 128;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int off = ((kc * 4 + g) ^ (r & 7)) << 4;
            uint4 af[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const uint4*>(pa + i * 16 * 128 + off);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (j < ni) {
                    const uint4 wf = *reinterpret_cast<const uint4*>(pw + j * 16 * 128 + off);
#pragma unroll
                    for (int i = 0; i < MI; ++i) mma16<__bf16>(wf, af[i], acc[i][j]);
                }
            }
        }
    }
}

// a use of a register the compiler can see: its wait for the load that produced `v` is placed HERE (while nothing else is in flight) instead of at the first
// arithmetic use, where it would drain the LDS-DMA bursts issued in between (hipcc waits vmcnt(0) for its own loads: it does not count the asm-issued DMA)
__device__ __forceinline__ void touch(float v) { asm volatile("" ::"v"(v)); }

// The group's SeaRowChain in REGISTERS: dword k in lane k & 63 of v[k >> 6] (four vector loads per lane at kernel entry), a field = one v_readlane.
// Read from the kernel-argument segment at its use, a field is a scalar load with its full latency in front of every basic block that uses one: the first build of this
// kernel issued 97 of them, ~30 inside every round of the projection loop (tools/chain_probe.py stamps: 2.4-2.8 us per round for 16 MFMAs per wave).
struct ChainParams {
    uint32_t v[4];
};
static_assert(sizeof(SeaRowChain) <= 4 * 64 * 4 && sizeof(SeaRowChain) % 4 == 0, "the parameter block of a group fits four dwords per lane");
template <int DW>
__device__ __forceinline__ uint32_t cp_u32(const ChainParams& p) {
    return (uint32_t)__builtin_amdgcn_readlane((int)p.v[DW >> 6], DW & 63);
}
template <int DW>
__device__ __forceinline__ uint64_t cp_u64(const ChainParams& p) {
    return (uint64_t)cp_u32<DW>(p) | ((uint64_t)cp_u32<DW + 1>(p) << 32);
}
__device__ __forceinline__ uint32_t cp_dyn_u32(const ChainParams& p, int dw) {   // dw wave-uniform
    const int sel = dw >> 6;
    const uint32_t src = sel == 0 ? p.v[0] : (sel == 1 ? p.v[1] : (sel == 2 ? p.v[2] : p.v[3]));
    return (uint32_t)__builtin_amdgcn_readlane((int)src, dw & 63);
}
__device__ __forceinline__ uint64_t cp_dyn_u64(const ChainParams& p, int dw) { return (uint64_t)cp_dyn_u32(p, dw) | ((uint64_t)cp_dyn_u32(p, dw + 1) << 32); }
#define CP_DW(field) ((int)(offsetof(SeaRowChain, field) / 4))
#define CP_I32(field) ((int)cp_u32<CP_DW(field)>(pl))
#define CP_F32(field) (__uint_as_float(cp_u32<CP_DW(field)>(pl)))
#define CP_PTR(type, field) (reinterpret_cast<type>(cp_u64<CP_DW(field)>(pl)))
#define QK_DW(field) ((int)(offsetof(SeaQkvGroup, field) / 4))

// what NormEpilogue reads of a SeaGemmNormGroup, for the chain's down-projection: no residual, no pre-norm outputs, no info-bottleneck term (compile-time nulls)
template <int D, int E>
struct ChainNormGroup {
    static constexpr int N = D, K = E, n_seg = 1;
    static constexpr float bias_scale = 1.0f;
    static constexpr const float* R = nullptr;
    static constexpr float* C32 = nullptr;
    static constexpr void* Cact = nullptr;
    static constexpr const float* ib_c = nullptr;
    static constexpr const float* ib_w1 = nullptr;
    static constexpr const float* ib_b1 = nullptr;
    static constexpr const float* ib_lnw = nullptr;
    static constexpr const float* ib_lnb = nullptr;
    static constexpr const float* ib_w2 = nullptr;
    static constexpr const float* ib_b2 = nullptr;
    static constexpr int ib_h = 0, ldr = 0, ldc32 = 0, ldcact = 0;
    int M, ldmod, ldy32, ldyact;
    const float* bias;
    const void* mod;
    const float* gamma;
    const float* beta;
    float* Y32;
    void* Yact;
    float* mean;
    float* rstd;
};

template <int D, int E, int MI>
__global__ __launch_bounds__(256) void row_chain_kernel(const ChainLaunch L) {
    using T = __bf16;
    using C = ChainCfg<D, E, MI>;
    constexpr int BM = C::BM, KTD = C::KTD, KTE = C::KTE, NI1 = C::NI1, NI2 = C::NI2, NI3 = C::NI3, NI4 = C::NI4, SMAX = C::SMAX;
    constexpr int A_KTS = BM * 128;   // bytes per K-tile of a row tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    if ((int)blockIdx.y == L.n_groups) {   // ---- the rider row (block-uniform)
        const int x = blockIdx.x;
        if (x < L.rider_n) {
            const int tile = L.rider_t0 + x;
            int gi = 0;
            while (gi + 1 < L.n_riders && tile >= L.rg[gi + 1].tile_start) ++gi;
            const ChainRiderPod& R = L.rg[gi];
            ChainRiderGroup G;
            G.A = R.A; G.W = R.W; G.bias = R.bias; G.Cact = R.Cact;
            G.lda = R.lda; G.ldw = R.ldw; G.ldcact = R.ldcact; G.M = R.M; G.N = R.N; G.K = R.K;
            const int t = tile - R.tile_start;
            const int tiles_n = (R.N + 127) / 128;
            // the LDS-DMA ring main loop (4 stages of A | W K-tiles in flight): a rider workgroup is ALONE on its CU (the launch's LDS request is the chain's), so
            // nothing else hides its memory latency — the single-buffered form took 15 us per tile here, generated operand included (round 4, measured)
            gemm_tile_body<T, 128, 128, true, true, false>(G, t / tiles_n, t % tiles_n, smem, 0);
        } else if (x < L.rider_n + L.ib_wgs) {
            const int row0 = (x - L.rider_n) * 64 + wave * 16;   // a wave per row, 16 rows per wave
            for (int i = 0; i < 16; ++i) {
                const int row = row0 + i;
                if (row < L.ib.M) ib_store_row(L.ib, L.ib.c[row], row, lane);
            }
        }
        return;
    }
    // ---- the group's parameter block: four dwords per lane, straight from the kernel-argument segment
    ChainParams pl;
    {
        const uint32_t* raw = reinterpret_cast<const uint32_t*>(&L.p[0]) + (size_t)blockIdx.y * (sizeof(SeaRowChain) / 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dw = k * 64 + lane;
            pl.v[k] = dw < (int)(sizeof(SeaRowChain) / 4) ? raw[dw] : 0u;
        }
    }
    const int M = CP_I32(M), S = CP_I32(n_seg);
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;   // block-uniform (groups of different row counts share the grid)
    unsigned long long* stp = L.stamps != nullptr ? L.stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 : nullptr;
    int stn = 0;
    auto stamp = [&]() {
        if (stp != nullptr && tid == 0 && stn < 16) stp[stn] = wall_clock64();
        ++stn;
    };
    stamp();   // 0: entry
    char* AT = smem;
    char* YT = AT + C::AT_BYTES;
    char* WA = YT + C::YT_BYTES;
    char* WB = WA + C::WH_BYTES;
    float* s_rope = reinterpret_cast<float*>(WB + C::WH_BYTES);
    float* s_bias = reinterpret_cast<float*>(reinterpret_cast<char*>(s_rope) + C::ROPE_BYTES);
    float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(s_bias) + C::BIAS_BYTES);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const unsigned at_l = lds0, wa_l = lds0 + C::AT_BYTES + C::YT_BYTES, wb_l = wa_l + C::WH_BYTES;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const T* W2 = CP_PTR(const T*, W2);
    const int ldw2 = CP_I32(ldw2);
    const bool has_down = CP_I32(has_down) != 0;
    const int n_proj = has_down ? CP_I32(n_proj) : 0;
    const T* Wd = CP_PTR(const T*, down.W);
    const int ldwd = CP_I32(down.ldw);

    // ---- burst 0: the A tile and the weights of stages 1 and 2
    if (S > 0) {
        const int ldatt = CP_I32(ldatt), ldwp = CP_I32(ldwp);
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            if (s < S) {
                const T* att = reinterpret_cast<const T*>(cp_dyn_u64(pl, CP_DW(att) + 2 * s));
                const T* Wp = reinterpret_cast<const T*>(cp_dyn_u64(pl, CP_DW(Wp) + 2 * s));
                chain_dma_a(att, ldatt, m0, M - 1, BM, KTD, at_l + (unsigned)(s * KTD * A_KTS), wv, lane);
                chain_dma_w(Wp, ldwp, D, 0, KTD, wa_l + (unsigned)(s * D * D * 2), wv, lane);
            }
        }
        chain_dma_w(W2, ldw2, E, 0, KTD, wb_l, wv, lane);
    } else {
        chain_dma_a(CP_PTR(const T*, a2), CP_I32(lda2), m0, M - 1, BM, KTE, at_l, wv, lane);
        chain_dma_w(W2, ldw2, E, 0, KTE / 2, wa_l, wv, lane);
        chain_dma_w(W2, ldw2, E, (KTE / 2) * 64, KTE / 2, wb_l, wv, lane);
    }
    stamp();   // 1: burst 0 requested
    // ---- epilogue operands of stages 2 and 3 (registers), rotary rows and projection biases of stage 4 (LDS): ordinary loads, all requested now
    int mrow[MI];
    bool mok[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + i * 16 + r;
        mok[i] = m < M;
        mrow[i] = mok[i] ? m : M - 1;
    }
    const float* b2 = CP_PTR(const float*, b2);
    const float* Xin = CP_PTR(const float*, Xin);
    const int ldxin = CP_I32(ldxin);
    const float bias_scale = CP_F32(bias_scale);
    float bv2[NI2][4], rv2[MI][NI2][4];
#pragma unroll
    for (int j = 0; j < NI2; ++j) {
        const int n = wave * (E / 4) + j * 16 + g * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) bv2[j][q] = 0.f;
        if (b2 != nullptr) load4(b2 + n, bv2[j]);
#pragma unroll
        for (int i = 0; i < MI; ++i) load4(Xin + (int64_t)mrow[i] * ldxin + n, rv2[i][j]);
    }
    // the down-projection's norm group as NormEpilogue reads it: the fields the chain uses from the parameter block, everything else a compile-time constant
    ChainNormGroup<D, E> Gd;
    Gd.M = M;
    Gd.bias = CP_PTR(const float*, down.bias);
    Gd.mod = CP_PTR(const void*, down.mod);
    Gd.ldmod = CP_I32(down.ldmod);
    Gd.gamma = CP_PTR(const float*, down.gamma);
    Gd.beta = CP_PTR(const float*, down.beta);
    Gd.Y32 = CP_PTR(float*, down.Y32);
    Gd.ldy32 = CP_I32(down.ldy32);
    Gd.Yact = CP_PTR(void*, down.Yact);
    Gd.ldyact = CP_I32(down.ldyact);
    Gd.mean = CP_PTR(float*, down.mean);
    Gd.rstd = CP_PTR(float*, down.rstd);
    NormEpilogue<T, NI3, true, true, true> epi3[MI];
    if (has_down) {
#pragma unroll
        for (int i = 0; i < MI; ++i) epi3[i].prefetch(Gd, mrow[i], wave * (D / 4), g);
    }
    const int hd = L.c.hd, hd2 = hd >> 1, Tlen = L.c.T, pos0 = L.c.pos0;
    if (n_proj > 0) {
        // rotary rows: (cos, sin) pairs [hd2] of row m0 + i at s_rope[i * hd + ...] (positions pos0 + t; rows of different trajectories in one tile keep their own t)
        for (int idx = tid; idx < BM * hd2; idx += 256) {
            const int i = idx / hd2, k = idx - i * hd2;
            int m = m0 + i;
            m = m < M ? m : M - 1;
            const int t = m % Tlen;
            const float2 cs = reinterpret_cast<const float2*>(L.c.rope)[(int64_t)(pos0 + t) * hd2 + k];
            reinterpret_cast<float2*>(s_rope)[idx] = cs;
        }
        int boff = 0;
        for (int e = 0; e < n_proj; ++e) {
            const int base = CP_DW(proj) + e * (int)(sizeof(SeaQkvGroup) / 4);
            const float* bias = reinterpret_cast<const float*>(cp_dyn_u64(pl, base + QK_DW(bias)));
            const int N = (int)cp_dyn_u32(pl, base + QK_DW(N));
            for (int n = tid; n < N; n += 256) s_bias[boff + n] = bias != nullptr ? bias[n] : 0.f;
            boff += N;
        }
    }
#pragma unroll
    for (int j = 0; j < NI2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            touch(bv2[j][q]);
#pragma unroll
            for (int i = 0; i < MI; ++i) touch(rv2[i][j][q]);
        }
    if (has_down) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI3; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    touch(epi3[i].bv[j][q]); touch(epi3[i].gm[j][q]); touch(epi3[i].bt[j][q]); touch(epi3[i].mw[j][q]); touch(epi3[i].mb[j][q]);
                }
    }
    stamp();   // 2: epilogue operands here (the compiler's waits drain the DMA too)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA bursts are not tracked by the compiler
    __syncthreads();
    stamp();   // 3: burst 0 landed

    f32x4 acc2[MI][NI2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (S > 0) {
        // ---- stage 1: g = sum_s gelu(att_s . Wp_s^T), this wave's D/4 columns; the GELU outputs are summed in fp32 (cross_up is linear and shared by the segments)
        float gsum[MI][NI1][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI1; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) gsum[i][j][q] = 0.f;
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            if (s < S) {
                f32x4 acc1[MI][NI1];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI1; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                chain_mma<MI, NI1>(AT + s * KTD * A_KTS, A_KTS, WA + s * D * D * 2, D * 128, wave * (D / 4), KTD, NI1, r, g, acc1);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI1; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) gsum[i][j][q] += gelu_erf(acc1[i][j][q]);
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI1; ++j) {
                const int kk = wave * (D / 4) + j * 16 + g * 4;   // this lane's 4 consecutive contraction indices of stage 2
                store4(reinterpret_cast<T*>(YT + (kk >> 6) * A_KTS + (i * 16 + r) * 128 + ((((kk & 63) >> 3) ^ (r & 7)) << 4) + (kk & 7) * 2), gsum[i][j][0], gsum[i][j][1], gsum[i][j][2],
                       gsum[i][j][3]);
            }
        lds_barrier();   // g is in place; nobody reads the Wp half or the att tile any more
        if (has_down) chain_dma_w(Wd, ldwd, D, 0, KTE, wa_l, wv, lane);   // lands under stage 2
        chain_mma<MI, NI2>(YT, A_KTS, WB, E * 128, wave * (E / 4), KTD, NI2, r, g, acc2);
    } else {
        // ---- stage 2 over the two halves of W2's K-tiles; the first half is refilled with Wd as soon as it has been read
        chain_mma<MI, NI2>(AT, A_KTS, WA, E * 128, wave * (E / 4), KTE / 2, NI2, r, g, acc2);
        lds_barrier();
        if (has_down) chain_dma_w(Wd, ldwd, D, 0, KTE, wa_l, wv, lane);
        chain_mma<MI, NI2>(AT + (KTE / 2) * A_KTS, A_KTS, WB, E * 128, wave * (E / 4), KTE / 2, NI2, r, g, acc2);
    }
    lds_barrier();   // every wave is done with the A tile (the x tile takes its place) and with the second weight half
    stamp();   // 4: stage 2 multiplied
    // ---- the first projection burst takes the free half: it lands under the stage-2 epilogue and stage 3
    int e_next = 0;        // first entry not yet requested
    auto burst = [&](unsigned half_l, int& e_lo, int& e_hi) {   // request entries e_next .. while they fit one half; returns the range
        e_lo = e_next;
        int bytes = 0;
        while (e_next < n_proj) {
            const int base = CP_DW(proj) + e_next * (int)(sizeof(SeaQkvGroup) / 4);
            const int N = (int)cp_dyn_u32(pl, base + QK_DW(N));
            if (bytes + N * D * 2 > C::WH_BYTES) break;
            chain_dma_w(reinterpret_cast<const T*>(cp_dyn_u64(pl, base + QK_DW(W))), (int)cp_dyn_u32(pl, base + QK_DW(ldw)), N, 0, KTD, half_l + (unsigned)bytes, wv, lane);
            bytes += N * D * 2;
            ++e_next;
        }
        e_hi = e_next;
    };
    int b_lo[2] = {0, 0}, b_hi[2] = {0, 0};   // entry ranges resident in (or on their way to) half B (index 0) and half A (index 1)
    if (n_proj > 0) burst(wb_l, b_lo[0], b_hi[0]);
    // ---- stage-2 epilogue: bias, residual; the new x as fp32 rows and as the bf16 A tile of stage 3
    float v2[MI][NI2][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI2; ++j) {
            const int n = wave * (E / 4) + j * 16 + g * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) v2[i][j][q] = acc2[i][j][q] + bv2[j][q] * bias_scale + rv2[i][j][q];
            store4(reinterpret_cast<T*>(AT + (n >> 6) * A_KTS + (i * 16 + r) * 128 + ((((n & 63) >> 3) ^ (r & 7)) << 4) + (n & 7) * 2), v2[i][j][0], v2[i][j][1], v2[i][j][2], v2[i][j][3]);
        }
    // Wd (and the first projection burst) have landed before anything else of this wave is put into the memory pipeline (stores count in vmcnt too)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();   // 5: Wd / first projection burst landed
    {
        float* X = CP_PTR(float*, X);
        T* Xact = CP_PTR(T*, Xact);
        const int ldx = CP_I32(ldx), ldxact = CP_I32(ldxact);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (!mok[i]) continue;
            float* xr_ = X + (int64_t)(m0 + i * 16 + r) * ldx + wave * (E / 4) + g * 4;
            T* xa_ = Xact != nullptr ? Xact + (int64_t)(m0 + i * 16 + r) * ldxact + wave * (E / 4) + g * 4 : nullptr;
#pragma unroll
            for (int j = 0; j < NI2; ++j) {
                store4(xr_ + j * 16, v2[i][j][0], v2[i][j][1], v2[i][j][2], v2[i][j][3]);
                if (xa_ != nullptr) store4(xa_ + j * 16, v2[i][j][0], v2[i][j][1], v2[i][j][2], v2[i][j][3]);
            }
        }
    }
    stamp();   // 6: x stored
    if (!has_down) return;   // block-uniform
    lds_barrier();
    // ---- stage 3: y = norm(x . Wd^T + bd); the normalised rows also go to LDS as the A tile of the projections
    {
        f32x4 acc3[MI][NI3];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI3; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        chain_mma<MI, NI3>(AT, A_KTS, WA, D * 128, wave * (D / 4), KTE, NI3, r, g, acc3);
#pragma unroll
        for (int i = 0; i < MI; ++i)
            epi3[i].finish(Gd, acc3[i], m0 + i * 16 + r, wave * (D / 4), r, g, wave, L.eps, red + i * 128, n_proj > 0 ? YT + i * 16 * 128 : nullptr, A_KTS);
    }
    stamp();   // 7: stage 3 done
    if (n_proj == 0) return;
    // ---- stage 4: the projections of y, burst by burst; the burst after the one being computed is in flight in the other half.
    // hd is a power of two >= 16 and every column offset a multiple of 16 (host): a 16-column block lies inside one head of one of q / k / v, so part, head and
    // the branch are WAVE-uniform (scalar), and every address is a per-row offset (once per row block) plus a per-block scalar, in 32 bits.
    const int Hh = L.c.H, cap = L.c.cap, Ea = Hh * hd;
    const int hd_sh = 31 - __builtin_clz((unsigned)hd);
    uint32_t offq[MI], offk[MI], offv[MI];   // element offsets of this lane's row: Q [b, ., t, .], K [b, ., pos, .], V^T [b, ., ., pos]
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int b_ = mrow[i] / Tlen;
        const int t_ = mrow[i] - b_ * Tlen;
        offq[i] = (uint32_t)((b_ * Hh * Tlen + t_) << hd_sh);
        offk[i] = (uint32_t)((b_ * Hh * cap + pos0 + t_) << hd_sh);
        offv[i] = (uint32_t)(b_ * Hh * hd * cap + pos0 + t_);
    }
    const float q_scale = L.c.q_scale;
    int cur = 0;           // half of the burst to compute next: 0 = half B, 1 = half A
    int boff = 0;          // offset of the current entry's biases in s_bias
    for (;;) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();   // the burst in half `cur` has landed (and, first round, y is complete); everybody is done with the other half
        stamp();   // 8, 10, ..: burst landed
        if (e_next < n_proj) burst(cur ? wb_l : wa_l, b_lo[cur ^ 1], b_hi[cur ^ 1]);
        else b_lo[cur ^ 1] = b_hi[cur ^ 1] = 0;
        const char* Wh = cur ? WA : WB;
        int woff = 0;
        for (int e = b_lo[cur]; e < b_hi[cur]; ++e) {
            // Two kinds of entries (host-checked): a q projection (N = D, col0 = 0: every column is q) and a k | v projection (N = 2 D, col0 = D: this wave's
            // quarter lies in k for waves 0 / 1, in v for waves 2 / 3) — one wave-uniform branch per entry, straight-line epilogues (the general per-block form
            // compiled into ~2600 instructions of branches and register copies per round: 4.7 us for 16 MFMAs, tools/chain_probe.py stamps)
            const int base = CP_DW(proj) + e * (int)(sizeof(SeaQkvGroup) / 4);
            const int N = (int)cp_dyn_u32(pl, base + QK_DW(N));
            if (N == D) {
                constexpr int NQ = D / 64;
                f32x4 acc4[MI][NQ];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NQ; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                chain_mma<MI, NQ>(YT, A_KTS, Wh + woff, D * 128, wave * (D / 4), KTD, NQ, r, g, acc4);
                T* dst = reinterpret_cast<T*>(cp_dyn_u64(pl, base + QK_DW(Qout)));
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const int nb = wv * (D / 4) + j * 16;                          // wave-uniform block start, inside one head
                    const int dd = (nb & (hd - 1)) + g * 4;
                    const uint32_t hoff = (uint32_t)((nb >> hd_sh) * Tlen) << hd_sh;
                    const float4 bq = *reinterpret_cast<const float4*>(s_bias + boff + nb + g * 4);
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const float4 cs = *reinterpret_cast<const float4*>(s_rope + ((i * 16 + r) * hd2 + (dd >> 1)) * 2);   // two (cos, sin) pairs
                        float o[4];
                        rope_pair(acc4[i][j][0] + bq.x, acc4[i][j][1] + bq.y, cs.x, cs.y, o[0], o[1]);
                        rope_pair(acc4[i][j][2] + bq.z, acc4[i][j][3] + bq.w, cs.z, cs.w, o[2], o[3]);
                        if (mok[i]) store4(dst + (offq[i] + hoff + (uint32_t)dd), o[0] * q_scale, o[1] * q_scale, o[2] * q_scale, o[3] * q_scale);
                    }
                }
            } else {
                constexpr int NK = 2 * D / 64;
                f32x4 acc4[MI][NK];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NK; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                chain_mma<MI, NK>(YT, A_KTS, Wh + woff, 2 * D * 128, wave * (D / 2), KTD, NK, r, g, acc4);
                if (wv < 2) {   // k: rotary embedding, row-major cache rows
                    T* dst = reinterpret_cast<T*>(cp_dyn_u64(pl, base + QK_DW(Kout)));
#pragma unroll
                    for (int j = 0; j < NK; ++j) {
                        const int nb = wv * (D / 2) + j * 16;                      // column inside k
                        const int dd = (nb & (hd - 1)) + g * 4;
                        const uint32_t hoff = (uint32_t)((nb >> hd_sh) * cap) << hd_sh;
                        const float4 bq = *reinterpret_cast<const float4*>(s_bias + boff + nb + g * 4);
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            const float4 cs = *reinterpret_cast<const float4*>(s_rope + ((i * 16 + r) * hd2 + (dd >> 1)) * 2);
                            float o[4];
                            rope_pair(acc4[i][j][0] + bq.x, acc4[i][j][1] + bq.y, cs.x, cs.y, o[0], o[1]);
                            rope_pair(acc4[i][j][2] + bq.z, acc4[i][j][3] + bq.w, cs.z, cs.w, o[2], o[3]);
                            if (mok[i]) store4(dst + (offk[i] + hoff + (uint32_t)dd), o[0], o[1], o[2], o[3]);
                        }
                    }
                } else {        // v: transposed (keys contiguous), and the optional row-major copy
                    T* vt = reinterpret_cast<T*>(cp_dyn_u64(pl, base + QK_DW(Vtout)));
                    T* vo = reinterpret_cast<T*>(cp_dyn_u64(pl, base + QK_DW(Vout)));
#pragma unroll
                    for (int j = 0; j < NK; ++j) {
                        const int nb = (wv - 2) * (D / 2) + j * 16;                // column inside v
                        const int dd = (nb & (hd - 1)) + g * 4;
                        const int h = nb >> hd_sh;
                        const uint32_t hoffv = (uint32_t)(((h << hd_sh) + dd) * cap);
                        const float4 bq = *reinterpret_cast<const float4*>(s_bias + boff + D + nb + g * 4);
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            if (!mok[i]) continue;
                            const float v0 = acc4[i][j][0] + bq.x, v1 = acc4[i][j][1] + bq.y, v2q = acc4[i][j][2] + bq.z, v3 = acc4[i][j][3] + bq.w;
                            T* d_ = vt + (offv[i] + hoffv);
                            d_[0] = (T)v0;
                            d_[(uint32_t)cap] = (T)v1;
                            d_[2u * (uint32_t)cap] = (T)v2q;
                            d_[3u * (uint32_t)cap] = (T)v3;
                            if (vo != nullptr) store4(vo + (offk[i] + ((uint32_t)(h * cap) << hd_sh) + (uint32_t)dd), v0, v1, v2q, v3);
                        }
                    }
                }
            }
            woff += N * D * 2;
            boff += N;
        }
        stamp();   // 9, 11, ..: burst computed
        if (b_hi[cur ^ 1] == b_lo[cur ^ 1]) break;   // nothing was requested for the other half: done
        cur ^= 1;
    }
}

// (Round 4, measured and removed — tools/chain_probe.py, gpurun_out/r04i/stamps2.txt: a second form of this kernel streamed 16 KiB weight slabs through a six-slot LDS
// ring with two loader waves (global_load_lds only in their vmcnt queue, counted waits, FULL / FREE words in LDS) to four consumer waves that each owned 16 complete
// rows of a 64-row workgroup, so that no consumer ever met another at a barrier.  It was correct and 2x SLOWER: 31 us per launch against 15.  A slab's round trip on
// this path is ~2.5 us (the weights are cold in the XCD's L2 at every step), the ring holds 6 slabs, so 24 slabs are four to five dependent round trips however
// the waves are organised — the burst form above has the same number of round trips with 4-8x the bytes in each.)

template <typename K>
static int set_lds_chain(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : -1;
}

static int row_chain_launch(const SeaRowChain* params, int n_groups, const SeaQkvCommon* common, const SeaGemmGroup* riders, int n_riders, int tile0, int n_tiles,
                            const SeaIbParams* ib, float eps, int dtype, void* stream);

extern "C" int sea_row_chain(const SeaRowChain* params, int n_groups, const SeaQkvCommon* common, float eps, int dtype, void* stream) {
    return row_chain_launch(params, n_groups, common, nullptr, 0, 0, 0, nullptr, eps, dtype, stream);
}

extern "C" int sea_row_chain_riders(const SeaRowChain* params, int n_groups, const SeaQkvCommon* common, const SeaGemmGroup* riders, int n_riders, int tile0, int n_tiles,
                                    const SeaIbParams* ib, float eps, int dtype, void* stream) {
    return row_chain_launch(params, n_groups, common, riders, n_riders, tile0, n_tiles, ib, eps, dtype, stream);
}

static int row_chain_launch(const SeaRowChain* params, int n_groups, const SeaQkvCommon* common, const SeaGemmGroup* riders, int n_riders, int tile0, int n_tiles,
                            const SeaIbParams* ib, float eps, int dtype, void* stream) {
    SEA_REQUIRE(params != nullptr && n_groups >= 1 && n_groups <= SEA_CHAIN_MAX_GROUPS, "sea_row_chain: n_groups=%d out of range", n_groups);
    ChainLaunch L;
    memset(&L, 0, sizeof(L));
    L.eps = eps;
    L.n_groups = n_groups;
    L.stamps = g_chain_stamps;
    // ---- riders (validated like sea_gemm_grouped's generated-operand groups)
    SEA_REQUIRE(n_riders >= 0 && n_riders <= SEA_CHAIN_MAX_RIDERS && (n_riders == 0 || riders != nullptr) && tile0 >= 0 && n_tiles >= 0, "sea_row_chain_riders: bad rider arguments");
    int rider_total = 0;
    for (int i = 0; i < n_riders; ++i) {
        const SeaGemmGroup& G = riders[i];
        SEA_REQUIRE(dtype == SEA_BF16 && G.A && !G.silu_c && G.W && G.Cact && !G.C32 && !G.R && !G.Z && G.act == 0 && G.drop.thr == 0 && G.n_seg == 1,
                    "sea_row_chain_riders[%d]: a rider is a plain bf16 group (A, W, bias, Cact only)", i);
        SEA_REQUIRE(G.M >= 1 && G.N >= 8 && G.N % 8 == 0 && G.K >= 64 && G.K % 64 == 0 && G.lda % 8 == 0 && G.lda >= G.K && G.ldw % 8 == 0 && G.ldw >= G.K && G.ldcact % 8 == 0 && G.ldcact >= G.N,
                    "sea_row_chain_riders[%d]: bad shape M=%d N=%d K=%d (whole 64-wide K-tiles) lda=%d ldw=%d ldcact=%d", i, G.M, G.N, G.K, G.lda, G.ldw, G.ldcact);
        SEA_REQUIRE(sea_aligned16(G.A) && sea_aligned16(G.W) && sea_aligned16(G.bias) && sea_aligned16(G.Cact), "sea_row_chain_riders[%d]: pointers must be 16-byte aligned", i);
        ChainRiderPod& R = L.rg[i];
        R.A = G.A; R.W = G.W; R.bias = G.bias; R.Cact = G.Cact;
        R.lda = G.lda; R.ldw = G.ldw; R.ldcact = G.ldcact; R.M = G.M; R.N = G.N; R.K = G.K;
        R.tile_start = rider_total;
        rider_total += ((G.M + 127) / 128) * ((G.N + 127) / 128);
    }
    SEA_REQUIRE(tile0 + n_tiles <= rider_total, "sea_row_chain_riders: tiles [%d, %d) of %d", tile0, tile0 + n_tiles, rider_total);
    L.n_riders = n_riders; L.rider_t0 = tile0; L.rider_n = n_tiles;
    if (ib != nullptr) {
        SEA_REQUIRE(ib->X[0] && ib->c && ib->w1 && ib->M >= 1 && ib->E >= 4 && ib->E % 4 == 0 && ib->ldx >= ib->E && ib->mode >= 0 && ib->mode <= 2 &&
                        (ib->mode != 0 || (ib->b1 && ib->lnw && ib->lnb && ib->w2 && ib->b2 && ib->h >= 1 && ib->h <= 64)),
                    "sea_row_chain_riders: bad info-bottleneck rider");
        L.ib = *ib;
        L.ib_wgs = (ib->M + 63) / 64;
    }
    const int D = params[0].D, E = params[0].E;
    int m_max = 0, any_proj = 0;
    for (int gi = 0; gi < n_groups; ++gi) {
        const SeaRowChain& P = params[gi];
        const bool shape_ok = (P.D == 128 && P.E == 256) || (P.D == 64 && P.E == 128);
        if (dtype != SEA_BF16 || !shape_ok || P.n_seg < 0 || P.n_seg * P.D > P.E) {
            sea_set_error("sea_row_chain: unsupported dtype / shape (dtype=%d D=%d E=%d n_seg=%d): bf16, (D,E) in {(128,256),(64,128)}, n_seg*D <= E", dtype, P.D, P.E, P.n_seg);
            return SEA_EUNSUPPORTED;
        }
        SEA_REQUIRE(P.D == D && P.E == E, "sea_row_chain[%d]: the groups of a launch share their widths", gi);
        SEA_REQUIRE(P.M >= 1 && P.W2 && P.Xin && P.X && sea_aligned16(P.W2) && sea_aligned16(P.b2) && sea_aligned16(P.Xin) && sea_aligned16(P.X) && sea_aligned16(P.Xact),
                    "sea_row_chain[%d]: null / misaligned pointer", gi);
        const int K2 = P.n_seg > 0 ? P.D : P.E;
        SEA_REQUIRE(P.ldw2 % 8 == 0 && P.ldw2 >= K2 && P.ldxin % 4 == 0 && P.ldxin >= P.E && P.ldx % 4 == 0 && P.ldx >= P.E && (!P.Xact || (P.ldxact % 4 == 0 && P.ldxact >= P.E)),
                    "sea_row_chain[%d]: bad strides", gi);
        if (P.n_seg == 0) SEA_REQUIRE(P.a2 && sea_aligned16(P.a2) && P.lda2 % 8 == 0 && P.lda2 >= P.E, "sea_row_chain[%d]: a2: null / misaligned / short rows", gi);
        else SEA_REQUIRE(P.ldatt % 8 == 0 && P.ldatt >= P.D && P.ldwp % 8 == 0 && P.ldwp >= P.D, "sea_row_chain[%d]: bad segment strides", gi);
        for (int s = 0; s < P.n_seg; ++s)
            SEA_REQUIRE(P.att[s] && sea_aligned16(P.att[s]) && P.Wp[s] && sea_aligned16(P.Wp[s]), "sea_row_chain[%d]: segment %d: null / misaligned operand", gi, s);
        L.p[gi] = P;
        SEA_REQUIRE(P.n_proj >= 0 && P.n_proj <= SEA_CHAIN_MAX_PROJ && (P.n_proj == 0 || P.has_down), "sea_row_chain[%d]: n_proj=%d (projections read the normalised rows: has_down)", gi, P.n_proj);
        if (P.has_down) {
            const SeaGemmNormGroup& G = P.down;
            SEA_REQUIRE(G.W && G.gamma && (G.Y32 || G.Yact || P.n_proj > 0) && G.ldw % 8 == 0 && G.ldw >= P.E, "sea_row_chain[%d]: down: null pointer or bad ldw", gi);
            SEA_REQUIRE((!G.Y32 || (G.ldy32 % 4 == 0 && G.ldy32 >= P.D)) && (!G.Yact || (G.ldyact % 4 == 0 && G.ldyact >= P.D)) && (!G.mod || (G.ldmod % 4 == 0 && G.ldmod >= 2 * P.D)),
                        "sea_row_chain[%d]: down: bad output / modulation strides", gi);
            SEA_REQUIRE(sea_aligned16(G.W) && sea_aligned16(G.bias) && sea_aligned16(G.mod) && sea_aligned16(G.gamma) && sea_aligned16(G.beta) && sea_aligned16(G.Y32) && sea_aligned16(G.Yact),
                        "sea_row_chain[%d]: down: pointers must be 16-byte aligned", gi);
            SeaGemmNormGroup& Gd = L.p[gi].down;   // the fields NormEpilogue reads besides the pointers checked above
            Gd.M = P.M; Gd.N = P.D; Gd.K = P.E; Gd.n_seg = 1; Gd.bias_scale = 1.0f;
            Gd.R = nullptr; Gd.C32 = nullptr; Gd.Cact = nullptr; Gd.ib_c = nullptr;
        }
        int nsum = 0;
        for (int e = 0; e < P.n_proj; ++e) {
            const SeaQkvGroup& Q = P.proj[e];
            SEA_REQUIRE(common != nullptr, "sea_row_chain: projections need the SeaQkvCommon");
            const int Ea = common->H * common->hd;
            SEA_REQUIRE(Q.W && sea_aligned16(Q.W) && sea_aligned16(Q.bias) && Q.ldw % 8 == 0 && Q.ldw >= P.D && Q.K == P.D && (Q.N == P.D / 2 || Q.N == P.D || Q.N == 2 * P.D) && Q.N % 64 == 0,
                        "sea_row_chain[%d]: projection %d: W [N, D] with N in {D, 2D} (multiples of 64), K = D (N=%d K=%d ldw=%d)", gi, e, Q.N, Q.K, Q.ldw);
            SEA_REQUIRE((Q.N == P.D && Q.col0 == 0) || (Q.N == 2 * P.D && Q.col0 == P.D), "sea_row_chain[%d]: projection %d: a q entry (N = D, col0 = 0) or a k | v entry (N = 2 D, col0 = D), not N=%d col0=%d", gi, e, Q.N, Q.col0);
            SEA_REQUIRE(Q.col0 >= 0 && Q.col0 % 16 == 0 && Q.col0 + Q.N <= 3 * Ea && Ea == P.D, "sea_row_chain[%d]: projection %d: columns [%d,%d) outside [0,%d) / H hd != D", gi, e, Q.col0, Q.col0 + Q.N, 3 * Ea);
            const bool hasq = Q.col0 < Ea, hask = Q.col0 < 2 * Ea && Q.col0 + Q.N > Ea, hasv = Q.col0 + Q.N > 2 * Ea;
            SEA_REQUIRE((!hasq || Q.Qout) && (!hask || Q.Kout) && (!hasv || Q.Vtout) && sea_aligned16(Q.Qout) && sea_aligned16(Q.Kout) && sea_aligned16(Q.Vtout) && sea_aligned16(Q.Vout),
                        "sea_row_chain[%d]: projection %d: missing / misaligned output pointer", gi, e);
            nsum += Q.N;
            any_proj = 1;
        }
        SEA_REQUIRE(nsum * 4 <= 4096, "sea_row_chain[%d]: the projections' output columns sum to %d (at most 1024)", gi, nsum);
        m_max = P.M > m_max ? P.M : m_max;
    }
    if (any_proj) {
        const SeaQkvCommon& c = *common;
        if (c.hd < 16 || (c.hd & (c.hd - 1)) != 0) {
            sea_set_error("sea_row_chain: unsupported head dim %d (a power of two >= 16: a 16-column block of a projection lies inside one head)", c.hd);
            return SEA_EUNSUPPORTED;
        }
        SEA_REQUIRE(c.rope != nullptr && c.H >= 1 && c.hd >= 4 && c.hd % 4 == 0 && c.T >= 1 && c.pos0 >= 0 && c.cap >= c.pos0 + c.T && m_max % c.T == 0,
                    "sea_row_chain: bad common H=%d hd=%d T=%d pos0=%d cap=%d (M=%d)", c.H, c.hd, c.T, c.pos0, c.cap, m_max);
        SEA_REQUIRE((int64_t)(m_max / c.T + 1) * c.H * c.cap * c.hd < ((int64_t)1 << 31) && (int64_t)(c.pos0 + c.T) * (c.hd / 2) < ((int64_t)1 << 30),
                    "sea_row_chain: attention tensors too large for 32-bit element offsets");
        L.c = c;
    }
    // 16-row workgroups while they fit the chip in ONE round (a workgroup owns most of a CU's LDS), 32-row ones beyond; SEA_TUNE=chain_rows=16|32 forces
    const int forced = sea_tune("chain_rows", 0);   // read per call (tests force both forms in one process)
    int mi = ((m_max + 15) / 16) * n_groups <= 256 ? 1 : 2;
    if (forced == 16) mi = 1;
    if (forced == 32) mi = 2;
    if (any_proj && 16 * mi * (L.c.hd / 2) * 8 > 2048) {
        if (mi == 2 && 16 * (L.c.hd / 2) * 8 <= 2048) mi = 1;
        else {
            sea_set_error("sea_row_chain: head dim %d: the rotary rows of a workgroup do not fit their LDS block", L.c.hd);
            return SEA_EUNSUPPORTED;
        }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int chain_x = (m_max + 16 * mi - 1) / (16 * mi), rider_x = L.rider_n + L.ib_wgs;
    const dim3 grid(chain_x > rider_x ? chain_x : rider_x, n_groups + (rider_x > 0 ? 1 : 0));
    // a rider tile runs the 4-stage LDS-DMA ring of the 128 x 128 GEMM: the launch requests the larger of the two footprints
    constexpr int rider_lds = GemmMainloop<__bf16, 128, 128>::DMA_LDS_BYTES;
#define LAUNCH_CH(DD, EE, MM)                                                                                   \
    do {                                                                                                        \
        constexpr int own_ = ChainCfg<DD, EE, MM>::LDS_BYTES;                                                   \
        constexpr int max_ = own_ > rider_lds ? own_ : rider_lds;                                               \
        static int once = set_lds_chain(row_chain_kernel<DD, EE, MM>, max_);                                    \
        (void)once;                                                                                             \
        row_chain_kernel<DD, EE, MM><<<grid, dim3(256), (L.rider_n > 0 ? max_ : own_), s>>>(L);                 \
    } while (0)
    if (D == 128) { if (mi == 1) LAUNCH_CH(128, 256, 1); else LAUNCH_CH(128, 256, 2); }
    else { if (mi == 1) LAUNCH_CH(64, 128, 1); else LAUNCH_CH(64, 128, 2); }
#undef LAUNCH_CH
    SEA_CHECK_LAUNCH("sea_row_chain");
    return SEA_OK;
}
