// Weight-stationary streaming GEMM for the SHORT contractions of the long launches (gfx950, bf16): C[M, N] = A[M, K] . W[N, K]^T + bias, K = 256 or 512.
//
// Why another form (tools/shortk_probe.py): at K <= 512 every tiled form of sea_gemm_grouped lands on the same time (mlp.fc1 at
// cfg3, 3 x 16192 x 2048 x 256: 96.7 us on 128 x 128 tiles, 100.6 on 256 x 256, 100.8 on the LDS-DMA ring) because a workgroup does operands in -> MFMA ->
// tile out in SEQUENCE and moves its operand panels through the L2 -> LDS path once per tile (780 MB for that launch at the ~8 TB/s the path sustains).
// Here the weights never pass through LDS at all and the activation rows pass once per 256 output columns:
//
//   * a workgroup (8 waves) is PERSISTENT and owns a panel of 256 output columns for a run of 32-row tiles; each MULTIPLYING wave holds the W fragments of its
//     columns for the whole contraction in REGISTERS (128 VGPRs: 64 columns x K = 256, or 32 columns x K = 512), loaded once per panel;
//   * the 32 x K activation tiles stream through a 3-slot LDS ring by global_load_lds (8-row x 128-byte pieces, source-side XOR swizzle as gemm256.hip), two
//     tiles ahead; a multiplying wave reads all fragments of the tile (2 row blocks x K / 32 steps: 0.25 / 0.5 LDS reads per MFMA);
//   * the output tile (32 x 256 bf16) goes through one of TWO staging buffers and leaves as whole 512-byte rows at the START of the next iteration, i.e. its
//     stores are in flight under the next tile's MFMAs; the barrier that publishes a ring slot also publishes the staging rows;
//   * who issues the DMA pieces and the stores is a matter of ROLES (GemmWsCfg): whatever a multiplying wave does besides its MFMAs idles its SIMD's matrix pipe.
//
// Per iteration a CU moves 16 KB (K = 256) in and 16 KB out for 4.2 MFLOP: 12 k iterations for mlp.fc1 = 47 per CU.
// Plain launches only (bias at most, activation-dtype output): sea_gemm_grouped routes here when every group qualifies (sea_gemm_ws_try).
#include "gemm_core.hpp"
#include <stdlib.h>

struct GemmWsLaunch {
    SeaGemmGroup g[SEA_MAX_GROUPS];
    int it_start[SEA_MAX_GROUPS + 1];   // first iteration (panel-major: panel, then row tile) of each group
    int n_groups;
    int chunk;                          // iterations per workgroup
    unsigned long long* probe;          // development (SEA_TUNE=gemm_ws_probe=1): s_memtime stamps of workgroup 0, [iteration 8 .. 15][wave][5 points], or NULL
};

__device__ __forceinline__ void glds16_ws(const void* ubase, unsigned lane_off, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(ubase), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ void ws_stamp(const GemmWsLaunch& L, int j, int wave, int lane, int k) {
    if (L.probe != nullptr && blockIdx.x == 0 && j >= 8 && j < 16 && lane == 0) {
        L.probe[((j - 8) * 12 + wave) * 5 + k] = __builtin_readcyclecounter();
        if (wave == 0 && k == 0) L.probe[8 * 12 * 5 + (j - 8)] = wall_clock64();   // the 100 MHz device clock beside it: what a "cycle" of the stamps is worth
    }
}

template <int KT>
struct GemmWsCfg {
    // Roles (in-kernel stamps, tools/ws_probe.py).  The tile's 16 KB leave a CU at ~20 B/clk: its 16 stores of 1 KB take 700 - 850 cycles to ISSUE whichever waves
    // issue them (every CU stores at once; the chip's write path sets the pace); a DMA piece costs its issuing wave 60 - 100 cycles; and a wave that waits for a
    // ring slot by vmcnt also waits for its own stores (vmcnt counts both; only loads return in order among themselves).  Whatever a MULTIPLYING wave does besides
    // its MFMAs is time the matrix pipe of its SIMD idles (two such waves per SIMD feed it), so:
    //   K = 256: eight waves — waves 0 .. 3 own 64 columns each (W fragments: 128 registers) and do nothing but fragments -> MFMAs -> staging, ONE per SIMD with
    //            its MFMAs back to back (fragment reads two steps ahead of the MFMAs that use them); waves 4 .. 7, their SIMD partners, move the bytes: the ring's DMA
    //            pieces (four per wave and slot) and the staged tile's stores — their vmcnt wait also covers their stores of the iteration before, long gone by then.
    //            (Twelve waves with eight 32-column multipliers measured 73.7 us on mlp.fc1: all eight read every fragment of the tile — a 500-cycle burst of LDS
    //            reads in front of each tile's MFMAs — and the younger multiplier of a SIMD finishes 500 cycles behind the older.)
    //   K = 512: the W fragments of 32 columns take 128 registers — waves 0 .. 3 issue the DMA pieces, waves 4 .. 7 the stores, all eight multiply 32 columns.
    static constexpr int NCB = KT == 4 ? 4 : 2;                 // 16-column blocks per multiplying wave
    static constexpr int NMW = KT == 4 ? 4 : 8;                 // multiplying waves
    static constexpr int FDW = KT == 4 ? 4 : 0;                 // first of the four DMA waves
    static constexpr int NDW = 4;
    static constexpr int FSW = 4;                               // first of the four store waves
    static constexpr int NWAVES = 8;
};

template <int KT>   // K = 64 KT
__global__ __launch_bounds__(GemmWsCfg<KT>::NWAVES * 64) void gemm_ws_kernel(const GemmWsLaunch L) {
    using T = __bf16;
    using Cf = GemmWsCfg<KT>;
    constexpr int BM = 32, BN = 256, BKB = 128, NS = 3;
    constexpr int SUB = BM * BKB;                 // one K-tile of a slot: 32 rows x 128 B
    constexpr int SLOT = KT * SUB;                // 16 KiB (K = 256) / 32 KiB (K = 512)
    constexpr int PPS = KT * (BM / 8);            // DMA pieces (8 rows x 128 B) per slot
    constexpr int NDW = Cf::NDW, FDW = Cf::FDW, FSW = Cf::FSW, NMW = Cf::NMW;
    constexpr int PW = PPS / NDW;                 // pieces per DMA wave and slot: 4 / 8
    constexpr int SP = BN * 2 + 16;               // staging row pitch (bytes)
    constexpr int STG = BM * SP;
    constexpr int STG_OFF = NS * SLOT;
    constexpr int KS = 2 * KT;                    // 32-wide contraction steps
    static_assert(PPS % NDW == 0 && STG_OFF + 2 * STG <= 160 * 1024 && NMW * Cf::NCB * 16 == BN && (PW == 4 || PW == 8), "layout of gemm_ws_kernel");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int rl = lane >> 3, chunk = (lane & 7) ^ (rl & 7);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const int total = L.it_start[L.n_groups];
    int it = (int)blockIdx.x * L.chunk;
    const int it_end = it + L.chunk < total ? it + L.chunk : total;

    while (it < it_end) {
        // ---- the segment: iterations [it, it + n_it) of ONE (group, column panel)
        int gi = 0;
        while (gi + 1 < L.n_groups && it >= L.it_start[gi + 1]) ++gi;
        const SeaGemmGroup& G = L.g[gi];
        const int M = G.M, N = G.N;
        const int tiles_m = (M + BM - 1) / BM;
        const int local = it - L.it_start[gi];
        const int panel = local / tiles_m, rt0 = local - panel * tiles_m;
        int n_it = tiles_m - rt0;
        n_it = n_it < it_end - it ? n_it : it_end - it;
        const int n0 = panel * BN;
        const T* A = static_cast<const T*>(G.A);
        const T* W = static_cast<const T*>(G.W);
        T* C = static_cast<T*>(G.Cact);
        const int lda = G.lda, ldc = G.ldcact;

        auto segment = [&](auto mul_tag, auto dma_tag, auto store_tag) {
            constexpr bool MUL = decltype(mul_tag)::value, DMA = decltype(dma_tag)::value, STORE = decltype(store_tag)::value;
            constexpr int NCB = MUL ? Cf::NCB : 1;   // 16-column blocks of the wave (a mover declares one and never touches it)
            const int c0 = wave * (16 * Cf::NCB);    // the wave's first column inside the panel
            // W fragments of the wave's columns, all of K, in registers: wf[cb][ks] = W[n0 + c0 + 16 cb + r][32 ks + 8 g .. + 8)  (rows past N clamped: never stored)
            uint4 wf[NCB][KS];
            float bv[NCB][4];
            if constexpr (MUL) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    int n = n0 + c0 + cb * 16 + r;
                    n = n < N ? n : N - 1;
                    const T* wr = W + (int64_t)n * G.ldw + g * 8;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) wf[cb][ks] = *reinterpret_cast<const uint4*>(wr + ks * 32);
                    const int nb = n0 + c0 + cb * 16 + g * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) bv[cb][q] = 0.f;
                    if (G.bias != nullptr && nb < N) load4(G.bias + nb, bv[cb]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) bv[cb][q] *= G.bias_scale;
                }
            }
            // slot j of the segment = row tile rt0 + j, ring position j % NS; DMA wave d (0 .. 3) issues pieces d, d + 4, ... (piece p: K-tile p / 4, rows 8 (p % 4) ..)
            auto dma_slot = [&](int j) {
                const unsigned base = lds_base + (unsigned)((j % NS) * SLOT);
                const int m0 = (rt0 + j) * BM;
#pragma unroll
                for (int u = 0; u < PW; ++u) {
                    const int p = u * NDW + (wave - FDW);
                    const int kt = p >> 2, r8 = p & 3;
                    int row = m0 + r8 * 8 + rl;
                    row = row < M ? row : M - 1;
                    glds16_ws(A + kt * 64, (unsigned)(((int64_t)row * lda + chunk * 8) * 2), base + (unsigned)(kt * SUB + r8 * 8 * BKB));
                }
            };
            // the staged tile jt of the segment -> C, by the four store waves: 32 rows x 512 B = 1024 pieces of 16 B, four per lane
            auto copy_out = [&](int jt) {
                const char* st = smem + STG_OFF + (jt & 1) * STG;
                const int m0p = (rt0 + jt) * BM;
                const int t4 = tid - FSW * 64;
                if (m0p + BM <= M && n0 + BN <= N) {   // workgroup-uniform: the full tile without predicates — four LDS reads, then four stores
                    uint4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = t4 + u * 256;
                        v[u] = *reinterpret_cast<const uint4*>(st + (idx >> 5) * SP + (idx & 31) * 16);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = t4 + u * 256;
                        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));   // (nt: the rows are not read again soon; 79.6 -> 77.2 us on mlp.fc1)
                        __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v[u]), reinterpret_cast<u32x4_t*>(C + (int64_t)(m0p + (idx >> 5)) * ldc + n0 + (idx & 31) * 8));
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = t4 + u * 256;
                        const int row = idx >> 5, cc = idx & 31;
                        const int m = m0p + row, n = n0 + cc * 8;
                        if (m < M && n < N) *reinterpret_cast<uint4*>(C + (int64_t)m * ldc + n) = *reinterpret_cast<const uint4*>(st + row * SP + cc * 16);
                    }
                }
            };
            // (the previous segment's last staging rows were copied out behind a barrier: ring and staging are free here)
            if constexpr (DMA) {
                dma_slot(0);
                if (n_it > 1) dma_slot(1);
            }
            for (int j = 0; j < n_it; ++j) {
                if constexpr (DMA) {
                    // slot j has landed: at most the PW pieces of slot j + 1 (issued after it) may still fly — loads return in order.  (K = 256: the wave's stores of
                    // iteration j - 1 count too and must be done — they have been for long: they were the first thing it issued an iteration ago.)
                    if (j + 1 < n_it) {
                        if constexpr (PW == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    }
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                ws_stamp(L, j, wave, lane, 0);
                __builtin_amdgcn_s_barrier();   // every DMA wave's pieces of slot j; every wave's staging rows of tile j - 1; nobody still reads slot j - 1
                ws_stamp(L, j, wave, lane, 1);
                if constexpr (STORE) {
                    if (j >= 1) copy_out(j - 1);         // tile j - 1 leaves, in flight under this tile's MFMAs
                }
                if constexpr (DMA) {
                    if (j + 2 < n_it) dma_slot(j + 2);   // ring position (j + 2) % 3 = (j - 1) % 3: read during iteration j - 1, which every wave has left
                }
                ws_stamp(L, j, wave, lane, 2);
                if constexpr (MUL) {
                    f32x4 acc[2][NCB];
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = f32x4{bv[cb][0], bv[cb][1], bv[cb][2], bv[cb][3]};
                    const char* sA = smem + (j % NS) * SLOT + r * BKB;
                    auto frag = [&](int ks, int rb) {
                        const int kt = ks >> 1, kc = ks & 1;
                        return *reinterpret_cast<const uint4*>(sA + kt * SUB + rb * 16 * BKB + (((kc * 4 + g) ^ (r & 7)) << 4));
                    };
                    if constexpr (NCB == 4) {
                        // one multiplier per SIMD: batches of two steps (4 fragments, 16 MFMAs), the next batch requested before this batch's MFMAs — 256 cycles of
                        // matrix work cover an LDS round trip, the pipe runs back to back
                        uint4 af[2][2][2];
#pragma unroll
                        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb) af[0][s_][rb] = frag(s_, rb);
#pragma unroll
                        for (int b_ = 0; b_ < KS / 2; ++b_) {
                            if (b_ + 1 < KS / 2) {
#pragma unroll
                                for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                                    for (int rb = 0; rb < 2; ++rb) af[(b_ + 1) & 1][s_][rb] = frag(2 * (b_ + 1) + s_, rb);
                                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");   // batch b_ has landed, batch b_ + 1 may still fly
                            } else {
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            }
#pragma unroll
                            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                                    for (int cb = 0; cb < NCB; ++cb) mma16<T>(wf[cb][2 * b_ + s_], af[b_ & 1][s_][rb], acc[rb][cb]);
                        }
                    } else {
                        // two multipliers per SIMD: the tile's fragments in batches of 8 steps (64 registers), all requested before the batch's first MFMA — left to
                        // itself the compiler reads two fragments, waits, issues four MFMAs: an LDS round trip per step (stamps: 1350 cycles per 32 MFMAs; batched 1100)
                        constexpr int PF = 8;
#pragma unroll
                        for (int k0 = 0; k0 < KS; k0 += PF) {
                            uint4 af[PF][2];
#pragma unroll
                            for (int s_ = 0; s_ < PF; ++s_)
#pragma unroll
                                for (int rb = 0; rb < 2; ++rb) af[s_][rb] = frag(k0 + s_, rb);
                            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the first half has landed; the rest lands under the first MFMAs
#pragma unroll
                            for (int s_ = 0; s_ < PF / 2; ++s_)
#pragma unroll
                                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                                    for (int cb = 0; cb < NCB; ++cb) mma16<T>(wf[cb][k0 + s_], af[s_][rb], acc[rb][cb]);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                            for (int s_ = PF / 2; s_ < PF; ++s_)
#pragma unroll
                                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                                    for (int cb = 0; cb < NCB; ++cb) mma16<T>(wf[cb][k0 + s_], af[s_][rb], acc[rb][cb]);
                        }
                    }
                    ws_stamp(L, j, wave, lane, 3);
                    // acc[rb][cb][q] = C[m0 + 16 rb + r][n0 + c0 + 16 cb + 4 g + q] (bias rode in as the accumulators' start)
                    char* st = smem + STG_OFF + (j & 1) * STG;
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb)
                            store4(reinterpret_cast<T*>(st + (rb * 16 + r) * SP) + (c0 + cb * 16 + g * 4), acc[rb][cb][0], acc[rb][cb][1], acc[rb][cb][2], acc[rb][cb][3]);
                }
                ws_stamp(L, j, wave, lane, 4);
            }
            // the segment's last tile
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (STORE) copy_out(n_it - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // staging rows read: the next segment may write them (and its first slots may land in the ring)
        };
        if constexpr (KT == 4) {
            if (wave < NMW) segment(std::true_type{}, std::false_type{}, std::false_type{});
            else segment(std::false_type{}, std::true_type{}, std::true_type{});
        } else {
            if (wave < NDW) segment(std::true_type{}, std::true_type{}, std::false_type{});
            else segment(std::true_type{}, std::false_type{}, std::true_type{});
        }
        it += n_it;
    }
}

template <int KT>
static void launch_ws(GemmWsLaunch& L, long total, int cus, int probe, unsigned long long* probe_buf, hipStream_t s) {
    L.chunk = (int)((total + cus - 1) / cus);
    const int grid = (int)((total + L.chunk - 1) / L.chunk);
    L.probe = probe ? probe_buf : nullptr;
    if (L.probe) (void)hipMemsetAsync(L.probe, 0, (8 * 12 * 5 + 8) * 8, s);
    constexpr int lds = 3 * KT * 32 * 128 + 2 * 32 * (256 * 2 + 16);
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<KT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)once;
    gemm_ws_kernel<KT><<<dim3(grid), dim3(GemmWsCfg<KT>::NWAVES * 64), lds, s>>>(L);
    if (L.probe && probe == 2) {   // print: [iteration][wave] cycles from the iteration's first stamp of wave 0: arrive, past the barrier, bytes issued, MFMAs done, staged
        unsigned long long h[8 * 12 * 5 + 8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, L.probe, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "ws probe: iterations 8 -> 15 of workgroup 0: %lld stamp cycles, %lld ticks of the 100 MHz clock = %.0f cycles and %.3f us per iteration (%.2f GHz)\n",
                (long long)(h[(7 * 12) * 5] - h[0]), (long long)(h[8 * 12 * 5 + 7] - h[8 * 12 * 5]), (double)(h[(7 * 12) * 5] - h[0]) / 7.0,
                (double)(h[8 * 12 * 5 + 7] - h[8 * 12 * 5]) / 7.0 / 100.0, (double)(h[(7 * 12) * 5] - h[0]) / ((double)(h[8 * 12 * 5 + 7] - h[8 * 12 * 5]) * 10.0));
        for (int j = 0; j < 8; ++j)
            for (int w = 0; w < GemmWsCfg<KT>::NWAVES; ++w) {
                const unsigned long long t0 = h[(j * 12 + 0) * 5 + 0];
                fprintf(stderr, "ws probe it %2d wave %2d: arrive %6lld  barrier %6lld  issued %6lld  mfma %6lld  staged %6lld\n", j + 8, w, (long long)(h[(j * 12 + w) * 5 + 0] - t0),
                        (long long)(h[(j * 12 + w) * 5 + 1] - t0), (long long)(h[(j * 12 + w) * 5 + 2] - t0), (long long)(h[(j * 12 + w) * 5 + 3] - t0), (long long)(h[(j * 12 + w) * 5 + 4] - t0));
            }
    }
}

// Launches the streaming kernel when every group of the launch qualifies — one launch per contraction length present (K = 256 and K = 512 groups of one call, the
// condition GEMMs of cfg3, go as two); returns false (nothing launched) otherwise.  SEA_TUNE=gemm_ws=0 keeps the tiled kernels, =2 forces short launches here too.
bool sea_gemm_ws_try(const SeaGemmGroup* groups, int n_groups, hipStream_t s) {
    const int on = sea_tune("gemm_ws", 1);   // read per call (tests force the kernel on short launches: 2)
    if (!on) return false;
    GemmWsLaunch L4, L8;
    memset(&L4, 0, sizeof(L4));
    memset(&L8, 0, sizeof(L8));
    long t4 = 0, t8 = 0;
    for (int i = 0; i < n_groups; ++i) {
        const SeaGemmGroup& G = groups[i];
        if ((G.K != 256 && G.K != 512) || G.silu_c != nullptr || G.n_seg != 1 || G.drop.thr != 0 || G.act != 0 || G.A == nullptr) return false;
        if (G.Cact == nullptr || G.C32 != nullptr || G.R != nullptr || G.Z != nullptr) return false;
        if (G.N % 8 != 0 || G.ldcact % 8 != 0 || G.lda % 8 != 0 || G.ldw % 8 != 0) return false;
        if (G.M < 4096 && on != 2) return false;                      // short launches: a panel's weight load is not amortised
        if ((int64_t)G.M * G.lda * 2 >= (1ll << 32)) return false;   // 32-bit operand offsets
        GemmWsLaunch& L = G.K == 256 ? L4 : L8;
        long& t = G.K == 256 ? t4 : t8;
        L.g[L.n_groups] = G;
        L.it_start[L.n_groups] = (int)t;
        t += (long)((G.N + 255) / 256) * ((G.M + 31) / 32);
        if (t >= (1l << 30)) return false;
        L.it_start[++L.n_groups] = (int)t;
    }
    const int cus = sea_cu_count();
    if (t4 + t8 < 8L * cus && on != 2) return false;                  // fewer than eight tiles per CU: the tiled kernels' shorter pipeline fill wins
    static unsigned long long* probe_buf = nullptr;
    const int probe = sea_tune("gemm_ws_probe", 0);
    if (probe && probe_buf == nullptr && hipMalloc(reinterpret_cast<void**>(&probe_buf), (8 * 12 * 5 + 8) * 8) != hipSuccess) probe_buf = nullptr;
    if (t8 > 0) launch_ws<8>(L8, t8, cus, probe_buf ? probe : 0, probe_buf, s);
    if (t4 > 0) launch_ws<4>(L4, t4, cus, probe_buf ? probe : 0, probe_buf, s);
    return true;
}
