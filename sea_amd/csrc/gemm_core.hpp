// LDS-tiled MFMA main loop shared by the grouped GEMM kernels (gfx950).
//
// C[BM x BN] tile of  sum_s A_s[M,K] . W[N,K]^T  — both operands are K-contiguous ("NT" GEMM, the nn.Linear
// layout), so A and B fragments are both 16-byte loads along K.
//
// Block = 256 threads = 4 waves in a WM x WN grid (2 x 2 unless stated); each wave owns (BM/WM) x (BN/WN) as MI x NI tiles of 16 x 16.
// LDS: per K-tile every row holds 128 bytes of K (64 bf16 / 32 f32) as eight 16-byte chunks; chunk c of row r is
// stored at chunk position c ^ (r & 7): the ds_read_b128 of a fragment (16 rows x the same chunk) then covers all
// 16 bank slots of the 256-byte bank row (conflict-free; MI355X_MICROARCH.md §LDS, cdna_hip_programming.md T2)
// and the ds_write_b128 of the staging pass (8 lanes per row) stays a permutation of one row's 32 banks.
// The MFMA computes the TRANSPOSED 16x16 tile (W fragment as the A operand, activation fragment as B): in the C/D map
// (col = lane & 15, row = 4*(lane >> 4) + reg) the lane's column is then an output ROW m and its 4 registers are 4
// CONSECUTIVE output columns n, so every epilogue access is a 16-byte (f32) or 8-byte (bf16) vector and a RoPE pair sits
// in one lane.  acc[i][j][q] = C[m0 + wm*WTM + 16 i + (lane & 15)][n0 + wn*WTN + 16 j + 4 (lane >> 4) + q].
// Staging is through registers (global_load_dwordx4 -> ds_write_b128) with the split of T14: the loads of tile
// k+1 are issued before the MFMAs of tile k and written to the other LDS buffer after them; one barrier per
// K-tile.
#pragma once
#include "sea_common.hpp"

template <typename T, int BM_, int BN_, int WM_ = 2>
struct GemmCfg {
    static constexpr int BM = BM_, BN = BN_;
    static constexpr int WM = WM_, WN = 4 / WM_;            // wave grid (rows x columns of waves); 4 x 1: a wave owns whole tile rows
    static constexpr int BKB = 128;                         // bytes of K per row per K-tile
    static constexpr int EPC = ActTraits<T>::EPC;           // elements per 16-byte chunk
    static constexpr int BK = BKB / (int)sizeof(T);         // K elements per K-tile
    static constexpr int WTM = BM / WM, WTN = BN / WN;      // wave tile
    static constexpr int MI = WTM / 16, NI = WTN / 16;      // 16x16 tiles per wave
    static constexpr int A_CH = BM * 8 / 256;               // 16-byte chunks staged per thread (A)
    static constexpr int B_CH = BN * 8 / 256;               // (W)
    static constexpr int BUF_BYTES = (BM + BN) * BKB;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
};

template <typename T, int BM, int BN, int WM = 2>
struct GemmMainloop {
    using C = GemmCfg<T, BM, BN, WM>;

    const T* A;
    const T* W;
    int64_t a_seg_stride;
    int lda, ldw, M, N, K, n_seg, m0, n0;
    // nsplit > 0: the tile's BN rows of W are TWO blocks of BN / 2 — rows n0 .. and rows nsplit + n0 .. (gemm_adaln.hip: the scale and the shift half of an AdaLN
    // modulation for the same output columns land in one tile, so the normalisation is that tile's epilogue)
    int nsplit = 0;
    __device__ __forceinline__ int w_row(int q) const { return nsplit > 0 ? (q < BN / 2 ? n0 + q : nsplit + n0 + q - BN / 2) : n0 + q; }

    uint4 ra[C::A_CH], rb[C::B_CH];

    // Optional GENERATED A operand (the AdaLN condition MLP, models/base_blocks.py:337-338,344): A[m, k] = silu(w1[k] * c[m] + b1[k]) is a
    // function of one scalar per row, so the tile is computed on the VALU (under the other waves' MFMAs) instead of being written to HBM by
    // a launch of its own and read back: no [M, 2d] hidden matrix, one launch less.  w1 / b1 are staged in LDS once per workgroup.
    const float* silu_c = nullptr;   // f32 [M]
    const float* s_w1 = nullptr;     // LDS copies, f32 [K]
    const float* s_b1 = nullptr;
    float cvals[C::A_CH];

    __device__ __forceinline__ void init_silu(int tid) {
        const int rr = tid >> 3;
#pragma unroll
        for (int i = 0; i < C::A_CH; ++i) {
            int row = m0 + rr + 32 * i;
            row = row < M ? row : M - 1;
            cvals[i] = silu_c[row];
        }
    }

    template <bool SILU = false>
    __device__ __forceinline__ void load_tile(int kt, int tid) {
        const int c = tid & 7;
        const int rr = tid >> 3;
        const int kk = kt * C::BK + c * C::EPC;
        const int k_total = K * n_seg;
        const bool kvalid = kk < k_total;
        int seg = 0, kin = kk;
        if (n_seg > 1) {
            seg = kk / K;
            kin = kk - seg * K;
        }
        const T* w_base = W + kin;
        if constexpr (SILU) {
            float w[C::EPC], b[C::EPC];
#pragma unroll
            for (int e = 0; e < C::EPC; e += 4) {
                const float4 wv = kvalid ? *reinterpret_cast<const float4*>(s_w1 + kk + e) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 bv = kvalid ? *reinterpret_cast<const float4*>(s_b1 + kk + e) : make_float4(0.f, 0.f, 0.f, 0.f);
                w[e] = wv.x; w[e + 1] = wv.y; w[e + 2] = wv.z; w[e + 3] = wv.w;
                b[e] = bv.x; b[e + 1] = bv.y; b[e + 2] = bv.z; b[e + 3] = bv.w;
            }
#pragma unroll
            for (int i = 0; i < C::A_CH; ++i) {
                T v[C::EPC];
#pragma unroll
                for (int e = 0; e < C::EPC; ++e) v[e] = from_f32<T>(silu_f(w[e] * cvals[i] + b[e]));
                ra[i] = kvalid ? *reinterpret_cast<const uint4*>(v) : make_uint4(0, 0, 0, 0);
            }
        } else {
            const T* a_base = A + seg * a_seg_stride + kin;
#pragma unroll
            for (int i = 0; i < C::A_CH; ++i) {
                int row = m0 + rr + 32 * i;
                row = row < M ? row : M - 1;
                ra[i] = kvalid ? *reinterpret_cast<const uint4*>(a_base + (int64_t)row * lda) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < C::B_CH; ++i) {
            int row = w_row(rr + 32 * i);
            row = row < N ? row : N - 1;
            rb[i] = kvalid ? *reinterpret_cast<const uint4*>(w_base + (int64_t)row * ldw) : make_uint4(0, 0, 0, 0);
        }
    }

    __device__ __forceinline__ void store_tile(char* buf, int tid) {
        const int c = tid & 7;
        const int rr = tid >> 3;
        char* sA = buf;
        char* sB = buf + BM * C::BKB;
#pragma unroll
        for (int i = 0; i < C::A_CH; ++i) {
            const int row = rr + 32 * i;
            *reinterpret_cast<uint4*>(sA + row * C::BKB + ((c ^ (row & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < C::B_CH; ++i) {
            const int row = rr + 32 * i;
            *reinterpret_cast<uint4*>(sB + row * C::BKB + ((c ^ (row & 7)) << 4)) = rb[i];
        }
    }

    __device__ __forceinline__ void compute_tile(const char* buf, int wm, int wn, int lane, f32x4 (&acc)[C::MI][C::NI]) {
        const int r = lane & 15, g = lane >> 4;
        const char* sA = buf + (wm * C::WTM + r) * C::BKB;
        const char* sB = buf + BM * C::BKB + (wn * C::WTN + r) * C::BKB;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int off = ((kc * 4 + g) ^ (r & 7)) << 4;  // (row & 7) == (r & 7): tile rows start at multiples of 16
            uint4 af[C::MI], bf[C::NI];
#pragma unroll
            for (int i = 0; i < C::MI; ++i) af[i] = *reinterpret_cast<const uint4*>(sA + i * 16 * C::BKB + off);
#pragma unroll
            for (int j = 0; j < C::NI; ++j) bf[j] = *reinterpret_cast<const uint4*>(sB + j * 16 * C::BKB + off);
#pragma unroll
            for (int i = 0; i < C::MI; ++i)
#pragma unroll
                for (int j = 0; j < C::NI; ++j) mma16<T>(bf[j], af[i], acc[i][j]);  // W rows as MFMA-A: C^T, see below
        }
    }

    // ------------------------------------------------------------------------------------------------------------------
    // LDS-DMA pipeline (K % BK == 0): K-tiles go HBM -> LDS directly with global_load_lds_dwordx4 (no VGPRs, no ds_write) into a
    // ring of NS stages; NS-1 stages are in flight while one is computed, so a K-tile's HBM/L2 latency is covered by NS-2 compute
    // phases instead of one.  One wave-instruction writes 1 KiB = 8 tile rows x 128 B, lane-linear in LDS, so the XOR swizzle is
    // applied to the per-lane SOURCE address (cdna_hip_programming.md rule 21): lane (row r = lane >> 3, position p = lane & 7)
    // fetches chunk p ^ (r & 7).  The loads are issued from inline asm so that hipcc neither counts them nor drains them with a
    // vmcnt(0) in front of every ds_read; completion is tracked by hand: counted s_waitcnt vmcnt(N), then a raw s_barrier, then
    // the reads (one barrier per K-tile).
    static constexpr int NS = 4;
    static constexpr int A_DMA = BM / 32, B_DMA = BN / 32;   // wave-instructions per wave per stage
    static constexpr int LPS = A_DMA + B_DMA;
    static constexpr int DMA_LDS_BYTES = NS * C::BUF_BYTES;
    // every LDS-DMA destination of dma_stage — slot (kt % NS), 8-row slab r0 of the A rows [0, BM) or of the W rows [BM, BM + BN), 1 KiB each — ends inside
    // the NS * BUF_BYTES the launch requests: the last A slab at (BM / 8 - 1 + 1) KiB = BM * BKB, the last W slab at BUF_BYTES
    static_assert(A_DMA * 4 * 8 == BM && B_DMA * 4 * 8 == BN && (BM + BN) * C::BKB == C::BUF_BYTES && DMA_LDS_BYTES <= 160 * 1024, "LDS-DMA ring of the tiled GEMM");

    static __device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
    }

    template <int NSX = NS>
    __device__ __forceinline__ void dma_stage(int kt, unsigned lds_base, int wave, int lane) const {
        const int kk = kt * C::BK;
        int seg = 0, kin = kk;
        if (n_seg > 1) {
            seg = kk / K;
            kin = kk - seg * K;
        }
        const int rl = lane >> 3;                                   // row inside the 8-row slab
        const int chunk = (lane & 7) ^ (rl & 7);                    // swizzle on the source side
        const T* a_base = A + seg * a_seg_stride + kin + chunk * C::EPC;
        const T* w_base = W + kin + chunk * C::EPC;
        const unsigned stage = lds_base + (unsigned)((kt % NSX) * C::BUF_BYTES);
#pragma unroll
        for (int u = 0; u < A_DMA; ++u) {
            const int r0 = (u * 4 + wave) * 8;
            int row = m0 + r0 + rl;
            row = row < M ? row : M - 1;
            glds16(a_base + (int64_t)row * lda, stage + (unsigned)(r0 * C::BKB));
        }
#pragma unroll
        for (int u = 0; u < B_DMA; ++u) {
            const int r0 = (u * 4 + wave) * 8;
            int row = w_row(r0 + rl);
            row = row < N ? row : N - 1;
            glds16(w_base + (int64_t)row * ldw, stage + (unsigned)(BM * C::BKB + r0 * C::BKB));
        }
    }

    __device__ __forceinline__ void run_dma(char* smem, f32x4 (&acc)[C::MI][C::NI]) { run_dma_n<NS>(smem, acc); }

    // NSX stages: 4 = three K-tiles in flight beside the one being multiplied (128 KiB at 128 x 128: one workgroup per CU); 2 = one in flight (64 KiB: TWO
    // workgroups per CU, whose waves fill each other's fragment-read and barrier waits — for launches with enough tiles to put two on every CU)
    template <int NSX>
    __device__ __forceinline__ void run_dma_n(char* smem, f32x4 (&acc)[C::MI][C::NI]) {
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave / C::WN, wn = wave % C::WN;
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nk = (K * n_seg) / C::BK;
        const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
        for (int s = 0; s < NSX - 1 && s < nk; ++s) dma_stage<NSX>(s, lds_base, wave, lane);
        for (int kt = 0; kt < nk; ++kt) {
            // stage kt has landed once at most the loads of the stages issued after it are outstanding
            const int newer = (nk - 1 - kt) < (NSX - 2) ? (nk - 1 - kt) : (NSX - 2);
            // lgkmcnt(0): this wave's fragment reads of stage kt-1 are RETIRED before it arrives at the barrier.  The buffer they read is refilled right
            // behind the barrier (one phase after its last read), and nothing but a retired read orders an LDS-DMA write against an earlier ds_read
            // (cdna_hip_programming.md, "Read a staged buffer ..." / WAR).  Found in round 2 with a sibling of this loop (mlp_fused.hip): with two
            // workgroups per CU rare 8-row pieces of a tile were multiplied against the NEXT stage's data.
            if (newer == NSX - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSX - 2) * LPS) : "memory");
            else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave's part of stage kt is in LDS; everyone is done reading stage kt-1
            if (kt + NSX - 1 < nk) dma_stage<NSX>(kt + NSX - 1, lds_base, wave, lane);  // refill the buffer stage kt-1 used
            compute_tile(smem + (kt % NSX) * C::BUF_BYTES, wm, wn, lane, acc);
        }
        __syncthreads();
    }

    // Single LDS buffer: half the LDS per workgroup, hence twice the resident workgroups per CU; costs a second barrier per K-tile.
    // For short contractions a tile is mostly prologue / epilogue VALU and memory waits (PMC: 7 VALU per MFMA, MFMA pipe 17 % busy at
    // K = 256), which other resident workgroups can fill — more of them beats a deeper pipeline inside one.
    template <bool SILU = false>
    __device__ __forceinline__ void run_single(char* smem, f32x4 (&acc)[C::MI][C::NI]) {
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = tid >> 6;
        const int wm = wave / C::WN, wn = wave % C::WN;
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nk = (K * n_seg + C::BK - 1) / C::BK;
        load_tile<SILU>(0, tid);
        for (int kt = 0; kt < nk; ++kt) {
            store_tile(smem, tid);
            __syncthreads();
            if (kt + 1 < nk) load_tile<SILU>(kt + 1, tid);
            compute_tile(smem, wm, wn, lane, acc);
            __syncthreads();
        }
    }

    __device__ __forceinline__ void run(char* smem, f32x4 (&acc)[C::MI][C::NI]) {
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = tid >> 6;
        const int wm = wave / C::WN, wn = wave % C::WN;
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nk = (K * n_seg + C::BK - 1) / C::BK;
        load_tile(0, tid);
        store_tile(smem, tid);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            char* cur = smem + (kt & 1) * C::BUF_BYTES;
            char* nxt = smem + ((kt + 1) & 1) * C::BUF_BYTES;
            const bool more = kt + 1 < nk;
            if (more) load_tile(kt + 1, tid);
            compute_tile(cur, wm, wn, lane, acc);
            if (more) store_tile(nxt, tid);
            __syncthreads();
        }
    }
};
