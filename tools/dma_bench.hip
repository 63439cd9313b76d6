// Microbenchmark (gfx950): how fast can a workgroup stream an L2-resident matrix into its LDS ring?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA) into a ring of NS stages, counted s_waitcnt vmcnt + one s_barrier per stage (the main loop of
//           sea_mlp_fc1_ln_gelu / GemmMainloop::run_dma without the MFMAs)
//   mode 1: global_load_dwordx4 into registers only (no LDS), same amount in flight
//   mode 2: global_load_dwordx4 into registers, then ds_write_b128 into the ring, one barrier per stage
//   mode 3: mode 0 with the address pattern of sea_mlp_fc1_ln_gelu: W1 [2048, 256] bf16 row-major (512-byte rows), a stage = 64 rows x 4 K-tiles,
//           one wave-instruction = 8 rows x 128 bytes
//   mode 4: mode 0 with the address pattern of fc2 on 64x64 tiles: [rows, 2048] bf16 row-major (4096-byte rows), a stage = 128 fixed rows x one
//           128-byte K-tile that advances by 128 bytes per stage
// Every workgroup reads one of 3 matrices of 1 MiB (like the three W1 of the cfg2 field MLPs), `passes` times.
//   hipcc --offload-arch=gfx950 -O3 -o dma_bench tools/dma_bench.hip && ./dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

constexpr int SPAN = 1 << 20;   // bytes of one matrix

// NW waves, NS ring stages, LPS 1-KiB pieces per wave per stage (stage = NW * LPS KiB)
template <int NW, int NS, int LPS, int MODE>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const char* src, int nstages, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = NW * LPS * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)(blockIdx.x % 3) * SPAN;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    unsigned acc = 0;
    if constexpr (MODE == 0 || MODE == 3 || MODE == 4) {
        const int rl = lane >> 3, cl = lane & 7;
        auto issue = [&](int s) {
            const unsigned off = (unsigned)(((size_t)s * STAGE) & (SPAN - 1));
#pragma unroll
            for (int i = 0; i < LPS; ++i) {
                const int p = i * NW + wave;
                const char* g;
                if constexpr (MODE == 0) g = base + off + p * 1024 + lane * 16;
                else if constexpr (MODE == 3) {   // NW * LPS = 32 pieces: K-tile kt = p >> 3, row group u = p & 7; 64 rows per stage, 32 stages per matrix
                    const int kt = p >> 3, u = p & 7;
                    g = base + (size_t)(((s & 31) * 64 + u * 8 + rl) * 512 + kt * 128 + cl * 16);
                } else {                          // rows (blockIdx.x * 16 + p) * 8 + rl of a [256, 2048] bf16 matrix (= 1 MiB), K-tile s & 31
                    const int row = ((blockIdx.x * (NW * LPS) + p) * 8 + rl) & 255;
                    g = base + (size_t)(row * 4096 + (s & 31) * 128 + cl * 16);
                }
                glds16(g, lds_base + (unsigned)((s % NS) * STAGE + p * 1024));
            }
        };
        for (int s = 0; s < NS - 1 && s < nstages; ++s) issue(s);
        for (int s = 0; s < nstages; ++s) {
            if (s + NS - 2 < nstages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + NS - 1 < nstages) issue(s + NS - 1);
            acc += *reinterpret_cast<const unsigned*>(smem + (s % NS) * STAGE + tid * 4);   // one ds_read per stage keeps the data "used"
        }
    } else {
        // register path: LPS * (NS - 1) pieces in flight per wave, rotated by hand
        uint4 r[NS - 1][LPS];
        auto issue = [&](int s, uint4 (&dst)[LPS]) {
            const unsigned off = (unsigned)(((size_t)s * STAGE) & (SPAN - 1));
#pragma unroll
            for (int i = 0; i < LPS; ++i) {
                const int p = i * NW + wave;
                dst[i] = *reinterpret_cast<const uint4*>(base + off + p * 1024 + lane * 16);
            }
        };
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue(s, r[s]);
        for (int s0 = 0; s0 < nstages; s0 += NS - 1) {
#pragma unroll
            for (int k = 0; k < NS - 1; ++k) {
                const int s = s0 + k;
                if (s >= nstages) break;
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < LPS; ++i)
                        *reinterpret_cast<uint4*>(smem + (s % NS) * STAGE + (i * NW + wave) * 1024 + lane * 16) = r[k][i];
                    __syncthreads();
                    acc += *reinterpret_cast<const unsigned*>(smem + (s % NS) * STAGE + tid * 4);
                } else {
#pragma unroll
                    for (int i = 0; i < LPS; ++i) acc += r[k][i].x ^ r[k][i].y ^ r[k][i].z ^ r[k][i].w;
                }
                if (s + NS - 1 < nstages) issue(s + NS - 1, r[k]);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int NW, int NS, int LPS, int MODE>
static void run(const char* src, unsigned* sink, int wgs, int passes) {
    constexpr int STAGE = NW * LPS * 1024;
    const int lds = MODE == 1 ? 0 : NS * STAGE;
    if (MODE == 2) return;   // the compiler drains every load before its ds_write: 15 GB/s per CU, not a useful data point
    if (lds > 160 * 1024) return;
    const int nstages = passes * (SPAN / STAGE);
    auto k = stream_kernel<NW, NS, LPS, MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0));
        k<<<dim3(wgs), dim3(NW * 64), lds>>>(src, nstages, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    const double bytes = (double)wgs * nstages * STAGE;
    const int cus = wgs < 256 ? wgs : 256;
    printf("mode %d waves %d ring %d x %2d KiB (in flight %3d KiB, lds %3d KiB) wgs %4d: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip  %5.0f ns per stage\n", MODE, NW, NS,
           STAGE / 1024, (NS - 1) * STAGE / 1024, lds / 1024, wgs, best * 1e3, bytes / (best * 1e-3) / cus / 1e9, bytes / (best * 1e-3) / 1e12,
           best * 1e6 / nstages / ((wgs + 255) / 256));
    fflush(stdout);
}

int main() {
    char* src;
    unsigned* sink;
    CK(hipMalloc(&src, 3 * SPAN));
    CK(hipMemset(src, 1, 3 * SPAN));
    CK(hipMalloc(&sink, 4));
    const int P = 8;   // passes over the matrix per workgroup
    for (int wgs : {32, 128, 192, 256, 512}) {
        printf("---- %d workgroups\n", wgs);
        // LDS-DMA, 8 waves: stage 8 / 16 / 32 KiB, ring depth 2 .. 8
        run<8, 2, 4, 0>(src, sink, wgs, P);
        run<8, 3, 4, 0>(src, sink, wgs, P);
        run<8, 4, 4, 0>(src, sink, wgs, P);
        run<8, 5, 4, 0>(src, sink, wgs, P);
        run<8, 4, 2, 0>(src, sink, wgs, P);
        run<8, 8, 2, 0>(src, sink, wgs, P);
        run<8, 8, 1, 0>(src, sink, wgs, P);
        run<8, 16, 1, 0>(src, sink, wgs, P);
        // LDS-DMA, 4 waves (the GEMM kernels): stage 16 KiB (64x64 tile), 32 KiB (128x128)
        run<4, 4, 4, 0>(src, sink, wgs, P);
        run<4, 5, 4, 0>(src, sink, wgs, P);
        run<4, 8, 4, 0>(src, sink, wgs, P);
        run<4, 4, 8, 0>(src, sink, wgs, P);
        run<8, 4, 4, 3>(src, sink, wgs, P);
        run<4, 4, 4, 4>(src, sink, wgs, P);
        // register path
        run<8, 4, 4, 1>(src, sink, wgs, P);
        run<8, 4, 2, 1>(src, sink, wgs, P);
        run<4, 4, 4, 1>(src, sink, wgs, P);
        run<8, 4, 4, 2>(src, sink, wgs, P);
        run<8, 4, 2, 2>(src, sink, wgs, P);
        run<4, 4, 4, 2>(src, sink, wgs, P);
    }
    return 0;
}
