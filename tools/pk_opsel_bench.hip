// Characterisation of the packed-fp32 defect behind the qkv_rope_kernel<bf16, 64, 64> wrong values: ONE packed instruction
// with pinned physical registers in a loop, its two result halves compared bit for bit with scalar v_fma_f32 / v_mul_f32 / v_add_f32 of the
// same operands; mismatches counted per quarter of the wave (16 lanes).  Other waves of the workgroup optionally run an MFMA / LDS / global
// load loop beside it (the failing launch has co-resident workgroups in their main loops).
//   hipcc --offload-arch=gfx950 -O2 tools/pk_opsel_bench.hip -o tools/isa_out/pk_opsel_bench ; ./pk_opsel_bench [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define N_CASES 14

// operands: a = src0 pair, b = src1 pair, c = src2 pair (registers pinned by the clobber list); results read back from the dst pair
#define PK_CASE(ID, S0, S1, S2, D, D0, D1, R0a, R0b, R1a, R1b, R2a, R2b, INSTR)                                                         \
    if (cs == ID) {                                                                                                                   \
        asm volatile("v_mov_b32 " R0a ", %2\n\tv_mov_b32 " R0b ", %3\n\tv_mov_b32 " R1a ", %4\n\tv_mov_b32 " R1b ", %5\n\t"           \
                     "v_mov_b32 " R2a ", %6\n\tv_mov_b32 " R2b ", %7\n\ts_nop 4\n\t" INSTR "\n\ts_nop 4\n\t"                           \
                     "v_mov_b32 %0, " D0 "\n\tv_mov_b32 %1, " D1 "\n\t"                                                                \
                     : "=v"(lo), "=v"(hi)                                                                                             \
                     : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)                                                           \
                     : "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");              \
    }

__device__ __forceinline__ float sfma(float a, float b, float c) {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float smul(float a, float b) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sadd(float a, float b) {
    float r;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float rnd(uint32_t s) {   // in [-2, 2), never denormal
    return ((float)(mix(s) >> 8) * (1.0f / 4194304.0f) - 2.0f) + 0.0009765625f;
}

__global__ __launch_bounds__(256) void pk_bench(int cs, int mode, int iters, unsigned long long* bad, const float* gsrc, float* gsink) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    const bool tester = mode == 0 || wave == 0;
    if (!tester) {
        // co-resident load: MFMAs (mode 1), + LDS reads and global loads (mode 2)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
        float s = 0.f;
        for (int it = 0; it < iters * 4; ++it) {
            if (mode == 1 || mode == 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);
            }
            if (mode == 3) {   // plain VALU beside the tester, no matrix core
#pragma unroll
                for (int u = 0; u < 8; ++u) s = sfma(s, 1.0001f, acc[u & 3]);
            }
            if (mode == 4) {   // LDS reads only
                const float4 v = *reinterpret_cast<const float4*>(&lds[((it * 64 + lane) * 4) & 4092]);
                s += v.x + v.w;
            }
            if (mode == 2) {
                const float4 v = *reinterpret_cast<const float4*>(&lds[((it * 64 + lane) * 4) & 4092]);
                s += v.x + v.w + gsrc[(blockIdx.x * 256 + threadIdx.x + it * 4099) & 0xfffff];
            }
        }
        if (acc[0] + acc[1] + acc[2] + acc[3] + s == 123.456f) gsink[0] = s;
        return;
    }
    unsigned long long nb_lo = 0, nb_hi = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t seed = (uint32_t)it * 0x9E3779B1u + (blockIdx.x * 256u + threadIdx.x) * 0x85EBCA77u;
        const float a0 = rnd(seed + 1), a1 = rnd(seed + 2), b0 = rnd(seed + 3), b1 = rnd(seed + 4), c0 = rnd(seed + 5), c1 = rnd(seed + 6);
        float lo = 0.f, hi = 0.f, elo = 0.f, ehi = 0.f;
        // regs: A = v[24:25] (banks 0,1), B = v[16:17] (0,1), C = v[22:23] (2,3), D = v[14:15]   — the failing instruction's own assignment
        PK_CASE(0, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_fma_f32 v[14:15], v[24:25], v[16:17], v[22:23] op_sel:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]")
        PK_CASE(1, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_fma_f32 v[14:15], v[24:25], v[16:17], v[22:23] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]")
        PK_CASE(2, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_fma_f32 v[14:15], v[24:25], v[16:17], v[22:23] op_sel:[0,1,0]")
        PK_CASE(3, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_fma_f32 v[14:15], v[24:25], v[16:17], v[22:23] op_sel:[1,0,0]")
        PK_CASE(4, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_fma_f32 v[14:15], v[24:25], v[16:17], v[22:23] op_sel:[0,0,1]")
        PK_CASE(5, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_mul_f32 v[14:15], v[24:25], v[16:17] op_sel:[0,1]")
        PK_CASE(6, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_mul_f32 v[14:15], v[24:25], v[16:17] op_sel:[1,0] op_sel_hi:[0,0]")
        PK_CASE(7, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_add_f32 v[14:15], v[24:25], v[16:17] op_sel:[0,1]")
        // other register banks: A = v[26:27] (2,3), B = v[16:17] (0,1), C = v[20:21] (0,1)
        PK_CASE(8, , , , , "v14", "v15", "v26", "v27", "v16", "v17", "v20", "v21",
                "v_pk_fma_f32 v[14:15], v[26:27], v[16:17], v[20:21] op_sel:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]")
        // B in banks (2,3): A = v[24:25] (0,1), B = v[18:19] (2,3), C = v[22:23] (2,3)
        PK_CASE(9, , , , , "v14", "v15", "v24", "v25", "v18", "v19", "v22", "v23",
                "v_pk_fma_f32 v[14:15], v[24:25], v[18:19], v[22:23] op_sel:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]")
        PK_CASE(10, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_fma_f32 v[14:15], v[24:25], v[16:17], v[22:23] op_sel:[0,1,0] op_sel_hi:[1,0,1]")
        PK_CASE(11, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_fma_f32 v[14:15], v[24:25], v[16:17], v[22:23]")
        // v_pk_mov_b32: D.lo = S0[op_sel[0]], D.hi = S1[op_sel[1]]  (the only other packed 64-bit form hipcc emits in this library: op_sel:[1,0])
        PK_CASE(12, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_mov_b32 v[14:15], v[24:25], v[16:17] op_sel:[1,0]")
        PK_CASE(13, , , , , "v14", "v15", "v24", "v25", "v16", "v17", "v22", "v23",
                "v_pk_mov_b32 v[14:15], v[24:25], v[16:17] op_sel:[0,1]")
        switch (cs) {   // lo = f(s0[x0], s1[x1], s2[x2]); hi = f(s0[y0], s1[y1], s2[y2]); op_sel defaults 0, op_sel_hi defaults 1
            case 0: case 8: case 9: elo = sfma(a0, b1, -c0); ehi = sfma(a1, b1, -c1); break;
            case 1: elo = sfma(a0, b0, -c0); ehi = sfma(a1, b0, -c1); break;
            case 2: elo = sfma(a0, b1, c0); ehi = sfma(a1, b1, c1); break;
            case 3: elo = sfma(a1, b0, c0); ehi = sfma(a1, b1, c1); break;
            case 4: elo = sfma(a0, b0, c1); ehi = sfma(a1, b1, c1); break;
            case 5: elo = smul(a0, b1); ehi = smul(a1, b1); break;
            case 6: elo = smul(a1, b0); ehi = smul(a0, b0); break;
            case 7: elo = sadd(a0, b1); ehi = sadd(a1, b1); break;
            case 10: elo = sfma(a0, b1, c0); ehi = sfma(a1, b0, c1); break;
            case 12: elo = a1; ehi = b0; break;
            case 13: elo = a0; ehi = b1; break;
            default: elo = sfma(a0, b0, c0); ehi = sfma(a1, b1, c1); break;
        }
        nb_lo += __float_as_uint(lo) != __float_as_uint(elo);
        nb_hi += __float_as_uint(hi) != __float_as_uint(ehi);
    }
    if (nb_lo) atomicAdd(&bad[(cs * 4 + (lane >> 4)) * 2 + 0], nb_lo);
    if (nb_hi) atomicAdd(&bad[(cs * 4 + (lane >> 4)) * 2 + 1], nb_hi);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned long long* bad;
    float *gsrc, *gsink;
    hipMalloc(&bad, N_CASES * 8 * sizeof(unsigned long long));
    hipMalloc(&gsrc, (1 << 20) * sizeof(float));
    hipMalloc(&gsink, 16);
    hipMemset(gsrc, 0, (1 << 20) * sizeof(float));
    const char* names[N_CASES] = {"pk_fma op_sel:[0,1,0] neg (the failing form, its registers)", "pk_fma op_sel_hi:[1,0,1] neg (control: form of the first row block)",
                                  "pk_fma op_sel:[0,1,0]", "pk_fma op_sel:[1,0,0]", "pk_fma op_sel:[0,0,1]", "pk_mul op_sel:[0,1]",
                                  "pk_mul op_sel:[1,0] op_sel_hi:[0,0]", "pk_add op_sel:[0,1]", "pk_fma op_sel:[0,1,0] neg, src0 in banks 2,3 / src2 in 0,1",
                                  "pk_fma op_sel:[0,1,0] neg, src1 in banks 2,3", "pk_fma op_sel:[0,1,0] op_sel_hi:[1,0,1]", "pk_fma (no selects)",
                                  "pk_mov_b32 op_sel:[1,0]", "pk_mov_b32 op_sel:[0,1]"};
    for (int mode = 0; mode < 5; ++mode) {
        hipMemset(bad, 0, N_CASES * 8 * sizeof(unsigned long long));
        for (int cs = 0; cs < N_CASES; ++cs) pk_bench<<<dim3(2048), dim3(256), 0, 0>>>(cs, mode, iters, bad, gsrc, gsink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        unsigned long long h[N_CASES * 8];
        hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
        const double per_q = (double)iters * 16.0 * 2048.0 * (mode == 0 ? 4 : 1);
        printf("mode %d (%s): %.3g evaluations per lane quarter\n", mode, mode == 0 ? "all waves test" : mode == 1 ? "wave 0 tests, waves 1-3 MFMA" : mode == 2 ? "wave 0 tests, waves 1-3 MFMA + LDS + global loads" : mode == 3 ? "wave 0 tests, waves 1-3 scalar VALU only" : "wave 0 tests, waves 1-3 LDS reads only", per_q);
        for (int cs = 0; cs < N_CASES; ++cs) {
            printf("  case %2d %-75s lo bad per quarter [%llu %llu %llu %llu]  hi bad [%llu %llu %llu %llu]\n", cs, names[cs], h[cs * 8 + 0], h[cs * 8 + 2], h[cs * 8 + 4],
                   h[cs * 8 + 6], h[cs * 8 + 1], h[cs * 8 + 3], h[cs * 8 + 5], h[cs * 8 + 7]);
        }
        fflush(stdout);
    }
    return 0;
}
