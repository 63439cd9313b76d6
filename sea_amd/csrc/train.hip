// Loss, metric and optimizer kernels of the SEA temporal train step (gfx950): sea_mse_fwd_bwd, sea_relative_mse,
// sea_adamw_flat.  All HBM-bandwidth-bound streaming passes: 16-byte accesses, grid-stride, fp32 arithmetic.
#include "sea_common.hpp"

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    const float total = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return total;
}

// ---------------------------------------------------------------------------------------------- MSE
// pass 1: partial[b] = sum over the block's elements of (out - tgt)^2 ; dout = (out - tgt) * gscale2 (gscale2 = 2*scale/n)
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ out, const float* __restrict__ tgt, float* __restrict__ dout,
                                                          float* __restrict__ partial, int64_t n4, float gscale2) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(out)[i];
        const float4 b = reinterpret_cast<const float4*>(tgt)[i];
        const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
        acc += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        if (dout != nullptr) reinterpret_cast<float4*>(dout)[i] = make_float4(d.x * gscale2, d.y * gscale2, d.z * gscale2, d.w * gscale2);
    }
    const float total = block_sum_256(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// pass 2 (one block): loss = sum(partial) / n, summed in a fixed order (deterministic)
__global__ __launch_bounds__(256) void mse_final_kernel(const float* __restrict__ partial, int n_partial, float* __restrict__ loss, float inv_n) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) acc += partial[i];
    const float total = block_sum_256(acc, red);
    if (threadIdx.x == 0) loss[0] = total * inv_n;
}

extern "C" int sea_mse_fwd_bwd(const float* out, const float* tgt, float* dout, float* loss, float* partial, int n_partial_cap,
                               int64_t n, float grad_scale, void* stream) {
    SEA_REQUIRE(out && tgt && loss && partial, "sea_mse_fwd_bwd: null pointer");
    SEA_REQUIRE(n >= 4 && n % 4 == 0, "sea_mse_fwd_bwd: n=%lld must be a positive multiple of 4", (long long)n);
    SEA_REQUIRE(sea_aligned16(out) && sea_aligned16(tgt) && sea_aligned16(dout), "sea_mse_fwd_bwd: pointers must be 16-byte aligned");
    SEA_REQUIRE(n_partial_cap >= 1, "sea_mse_fwd_bwd: partial workspace too small");
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks > n_partial_cap) blocks = n_partial_cap;
    hipStream_t s = static_cast<hipStream_t>(stream);
    mse_partial_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(out, tgt, dout, partial, n4, 2.0f * grad_scale / (float)n);
    mse_final_kernel<<<dim3(1), dim3(256), 0, s>>>(partial, (int)blocks, loss, 1.0f / (float)n);
    SEA_CHECK_LAUNCH("sea_mse_fwd_bwd");
    return SEA_OK;
}

// ---------------------------------------------------------------------------------------------- relative MSE
// one wave per row of the last dimension: y[row] = sum (p - t)^2 / (sum t^2 + 1e-8)
__global__ __launch_bounds__(256) void relative_mse_kernel(const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ y,
                                                           int64_t rows, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = p + row * d;
    const float* tr = t + row * d;
    float num4[4] = {0.f, 0.f, 0.f, 0.f}, den4[4] = {0.f, 0.f, 0.f, 0.f};   // scalar-lane FMAs: see sea_common.hpp (packed horizontal adds trip the build's ISA check)
    for (int i = lane * 4; i < d; i += 256) {
        float a[4], b[4];
        load4(pr + i, a);
        load4(tr + i, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float df = a[e] - b[e];
            num4[e] = fma1(df, df, num4[e]);
            den4[e] = fma1(b[e], b[e], den4[e]);
        }
    }
    const float num = wave_sum(add1(add1(num4[0], num4[1]), add1(num4[2], num4[3])));
    const float den = wave_sum(add1(add1(den4[0], den4[1]), add1(den4[2], den4[3])));
    if (lane == 0) y[row] = num / (den + 1e-8f);
}

extern "C" int sea_relative_mse(const float* pred, const float* truth, float* y, int64_t rows, int d, void* stream) {
    SEA_REQUIRE(pred && truth && y && rows >= 1 && d >= 4 && d % 4 == 0, "sea_relative_mse: bad arguments rows=%lld d=%d", (long long)rows, d);
    SEA_REQUIRE(sea_aligned16(pred) && sea_aligned16(truth), "sea_relative_mse: pointers must be 16-byte aligned");
    SEA_REQUIRE((rows + 3) / 4 <= 0x7fffffffLL, "sea_relative_mse: too many rows");
    relative_mse_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(pred, truth, y, rows, d);
    SEA_CHECK_LAUNCH("sea_relative_mse");
    return SEA_OK;
}

// ---------------------------------------------------------------------------------------------- AdamW over the flat buffer
template <typename T>
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, T* __restrict__ shadow, int64_t n4, float lr, float beta1,
                                                         float beta2, float eps, float decay, float inv_bc1, float inv_sqrt_bc2,
                                                         float grad_scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float pp[4], gg[4], mm[4], vv[4];
        load4(p + 4 * i, pp);
        load4(g + 4 * i, gg);
        load4(m + 4 * i, mm);
        load4(v + 4 * i, vv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * grad_scale;
            pp[e] *= decay;                                   // decoupled weight decay: p *= 1 - lr * wd
            mm[e] = beta1 * mm[e] + (1.0f - beta1) * gr;
            vv[e] = beta2 * vv[e] + (1.0f - beta2) * gr * gr;
            const float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
            pp[e] -= (lr * inv_bc1) * (mm[e] / denom);
        }
        store4(p + 4 * i, pp[0], pp[1], pp[2], pp[3]);
        store4(m + 4 * i, mm[0], mm[1], mm[2], mm[3]);
        store4(v + 4 * i, vv[0], vv[1], vv[2], vv[3]);
        if (shadow != nullptr) store4(shadow + 4 * i, pp[0], pp[1], pp[2], pp[3]);
    }
}

extern "C" int sea_adamw_flat(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    SEA_REQUIRE(p && g && m && v, "sea_adamw_flat: null pointer");
    SEA_REQUIRE(n >= 4 && n % 4 == 0 && step >= 1, "sea_adamw_flat: n=%lld must be a positive multiple of 4, step >= 1", (long long)n);
    SEA_REQUIRE(sea_aligned16(p) && sea_aligned16(g) && sea_aligned16(m) && sea_aligned16(v) && sea_aligned16(shadow), "sea_adamw_flat: pointers must be 16-byte aligned");
    SEA_REQUIRE(!shadow || shadow_dtype == SEA_BF16 || shadow_dtype == SEA_F32, "sea_adamw_flat: bad shadow dtype");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float decay = 1.0f - lr * weight_decay;
    if (shadow != nullptr && shadow_dtype == SEA_BF16)
        adamw_flat_kernel<__bf16><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p, g, m, v, static_cast<__bf16*>(shadow), n4, lr, beta1, beta2, eps, decay,
                                                                              (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale);
    else
        adamw_flat_kernel<float><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p, g, m, v, static_cast<float*>(nullptr), n4, lr, beta1, beta2, eps, decay,
                                                                             (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale);
    SEA_CHECK_LAUNCH("sea_adamw_flat");
    return SEA_OK;
}
