// ABI plumbing of libsea_hip.so: version, thread-local error string, device info, MFMA fragment-map self-test.
#include "sea_common.hpp"
#include <stdlib.h>

static thread_local char g_err[512] = "";

void sea_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int sea_tune(const char* key, int dflt) {
    const char* e = getenv("SEA_TUNE");
    if (e == nullptr) return dflt;
    const size_t n = strlen(key);
    for (const char* p = e; *p;) {
        while (*p == ',' || *p == ' ') ++p;
        if (strncmp(p, key, n) == 0 && p[n] == '=') return atoi(p + n + 1);
        while (*p && *p != ',') ++p;
    }
    return dflt;
}

// CUs of the CURRENT device (one process per GPU: rank r's device is not device 0), cached per device; 256 when the query fails
int sea_cu_count() {
    static int cu_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cu_of[dev] == 0) {
        int n = 0;
        cu_of[dev] = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : 256;
    }
    return cu_of[dev];
}

extern "C" int sea_abi_version(void) { return SEA_ABI_VERSION; }
extern "C" const char* sea_last_error(void) { return g_err; }

// sizeof of every ABI struct, in header order, so that a binding can verify its own layout (returns the count).
extern "C" int sea_struct_sizes(int* out, int cap) {
    const int sizes[] = {(int)sizeof(SeaGemmGroup), (int)sizeof(SeaQkvGroup), (int)sizeof(SeaQkvCommon), (int)sizeof(SeaAttnProblem),
                         (int)sizeof(SeaAttnParams), (int)sizeof(SeaNormGroup), (int)sizeof(SeaSiluGroup), (int)sizeof(SeaIbParams),
                         (int)sizeof(SeaWgradGroup), (int)sizeof(SeaNormBwdGroup), (int)sizeof(SeaSiluBwdGroup), (int)sizeof(SeaIbBwdParams),
                         (int)sizeof(SeaAttnBwdProblem), (int)sizeof(SeaAttnBwdParams), (int)sizeof(SeaDropout), (int)sizeof(SeaLaunchRec),
                         (int)sizeof(SeaGemmNormGroup), (int)sizeof(SeaExchangeTail), (int)sizeof(SeaMlpGroup), (int)sizeof(SeaMlp2Group), (int)sizeof(SeaKvNorm), (int)sizeof(SeaKvField),
                         (int)sizeof(SeaKvPair), (int)sizeof(SeaKvLayer), (int)sizeof(SeaKvGlobal), (int)sizeof(SeaStepPatch), (int)sizeof(SeaRowChain), (int)sizeof(SeaAdalnGroup), (int)sizeof(SeaAdalnQkv), (int)sizeof(SeaSplitkGroup)};
    const int n = (int)(sizeof(sizes) / sizeof(sizes[0]));
    for (int i = 0; i < n && i < cap; ++i) out[i] = sizes[i];
    return n;
}

extern "C" int sea_run_list(const SeaLaunchRec* recs, int n_recs, void* stream) {
    SEA_REQUIRE(recs != nullptr && n_recs >= 0, "sea_run_list: bad arguments");
    for (int i = 0; i < n_recs; ++i) {
        const SeaLaunchRec& R = recs[i];
        int rc;
        switch (R.op) {
            case SEA_OP_GEMM: rc = sea_gemm_grouped(static_cast<const SeaGemmGroup*>(R.p0), R.n, R.dtype, stream); break;
            case SEA_OP_QKV: rc = sea_qkv_rope_grouped(static_cast<const SeaQkvGroup*>(R.p0), R.n, static_cast<const SeaQkvCommon*>(R.p1), R.dtype, stream); break;
            case SEA_OP_ATTN: rc = sea_attention_fwd(static_cast<const SeaAttnParams*>(R.p0), R.dtype, stream); break;
            case SEA_OP_NORM: rc = sea_rownorm(static_cast<const SeaNormGroup*>(R.p0), R.n, R.i0, R.i1, R.i2, R.i3, R.f0, R.dtype, stream); break;
            case SEA_OP_SILU: rc = sea_silu_outer_ib(static_cast<const SeaSiluGroup*>(R.p0), R.n, static_cast<const float*>(R.p1), R.i0, R.dtype,
                                                     reinterpret_cast<const SeaIbParams*>(static_cast<intptr_t>(R.l0)), (int)R.l1, stream); break;
            case SEA_OP_IB: rc = sea_ib_add(static_cast<const SeaIbParams*>(R.p0), stream); break;
            case SEA_OP_CONVERT: rc = sea_convert_f32_to_act(static_cast<const float*>(R.p0), R.l0, const_cast<void*>(R.p1), R.l1, R.l2, R.l3, R.dtype, stream); break;
            case SEA_OP_GEMM_NORM: rc = sea_gemm_rownorm(static_cast<const SeaGemmNormGroup*>(R.p0), R.n, R.f0, R.dtype, stream); break;
            case SEA_OP_XTAIL: rc = sea_exchange_tail(static_cast<const SeaExchangeTail*>(R.p0), R.n, R.f0, R.dtype, stream); break;
            case SEA_OP_MLP1: rc = sea_mlp_fc1_ln_gelu(static_cast<const SeaMlpGroup*>(R.p0), R.n, R.f0, R.dtype, stream); break;
            case SEA_OP_MLP2: rc = sea_mlp_fc2_proj_norm(static_cast<const SeaMlp2Group*>(R.p0), R.n, R.f0, R.dtype, stream); break;
            case SEA_OP_GEMM_FEW: rc = sea_gemm_fewrows(static_cast<const SeaGemmGroup*>(R.p0), static_cast<const SeaNormGroup*>(R.p1), R.n, R.i0, R.i1, R.f0, R.dtype, stream); break;
            case SEA_OP_QKV_FEW: rc = sea_qkv_rope_fewrows(static_cast<const SeaQkvGroup*>(R.p0), reinterpret_cast<const SeaNormGroup*>(R.l0), R.n, static_cast<const SeaQkvCommon*>(R.p1), R.f0,
                                                           R.dtype, stream); break;
            case SEA_OP_CHAIN: rc = sea_row_chain_riders(static_cast<const SeaRowChain*>(R.p0), R.n, static_cast<const SeaQkvCommon*>(R.p1), reinterpret_cast<const SeaGemmGroup*>(static_cast<intptr_t>(R.l0)),
                                                         R.i0, R.i1, R.i2, reinterpret_cast<const SeaIbParams*>(static_cast<intptr_t>(R.l1)), R.f0, R.dtype, stream); break;
            case SEA_OP_ADALN: rc = sea_gemm_adaln(static_cast<const SeaAdalnGroup*>(R.p0), R.n, R.f0, R.dtype, stream); break;
            case SEA_OP_SPLITK: rc = sea_splitk_finish(static_cast<const SeaSplitkGroup*>(R.p0), R.n, R.dtype, stream); break;
            case SEA_OP_MLPB: rc = sea_mlp_block(static_cast<const SeaMlpGroup*>(R.p0), static_cast<const SeaMlp2Group*>(R.p1), R.n, R.f0, R.dtype, stream); break;
            case SEA_OP_AQKV: rc = sea_adaln_qkv(static_cast<const SeaAdalnQkv*>(R.p0), R.n, static_cast<const SeaQkvCommon*>(R.p1), reinterpret_cast<const SeaGemmGroup*>(static_cast<intptr_t>(R.l0)), R.i0,
                                                 reinterpret_cast<const SeaSiluGroup*>(static_cast<intptr_t>(R.l1)), R.i1, reinterpret_cast<const float*>(static_cast<intptr_t>(R.l2)), R.i2,
                                                 reinterpret_cast<const SeaIbParams*>(static_cast<intptr_t>(R.l3)), R.f0, R.dtype, stream); break;
            default: sea_set_error("sea_run_list[%d]: unknown op %d", i, R.op); return SEA_EINVAL;
        }
        if (rc != SEA_OK) return rc;   // sea_last_error() already names the entry point; the caller maps i back to its record
    }
    return SEA_OK;
}

extern "C" int sea_run_list_steps(const SeaLaunchRec* recs, int n_recs, const SeaStepPatch* patches, int n_patches, int step0, int n_steps, void* stream) {
    SEA_REQUIRE(recs != nullptr && n_recs >= 0 && n_patches >= 0 && (n_patches == 0 || patches != nullptr) && step0 >= 0 && n_steps >= 0, "sea_run_list_steps: bad arguments");
    for (int i = 0; i < n_patches; ++i)
        SEA_REQUIRE(patches[i].addr != nullptr && (patches[i].kind == 0 || patches[i].kind == 1), "sea_run_list_steps: patch %d: null address or bad kind %d", i, patches[i].kind);
    for (int s = step0; s < step0 + n_steps; ++s) {
        for (int i = 0; i < n_patches; ++i) {
            const SeaStepPatch& P = patches[i];
            const int64_t v = P.base + (int64_t)s * P.stride;
            if (P.kind == 0) *static_cast<int32_t*>(P.addr) = (int32_t)v;
            else *static_cast<int64_t*>(P.addr) = v;
        }
        const int rc = sea_run_list(recs, n_recs, stream);
        if (rc != SEA_OK) return rc;
    }
    return SEA_OK;
}

extern "C" int sea_device_info(int* cu_count, char* arch, int arch_len) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        sea_set_error("sea_device_info: no HIP device");
        return SEA_ELAUNCH;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return SEA_OK;
}

// ------------------------------------------------------------------------------------------------ MFMA self-test
// C[16x16] = A[16xCK] . B[CKx16] with exact small integers through mma16<T>, operands gathered with the lane maps
// the library assumes (A[row = lane&15][k = (lane>>4)*EPC + j], B[k][col = lane&15], C: col = lane&15,
// row = 4*(lane>>4) + reg).  A and B are asymmetric so that a transposed map cannot pass.
template <typename T>
__global__ void mfma_selftest_kernel(const T* A, const T* Bm, float* Cout) {
    constexpr int EPC = ActTraits<T>::EPC, CK = ActTraits<T>::CK;
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    T af[EPC], bf[EPC];
    for (int j = 0; j < EPC; ++j) {
        af[j] = A[r * CK + g * EPC + j];       // A[row r][k]
        bf[j] = Bm[(g * EPC + j) * 16 + r];    // B[k][col r]
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    mma16<T>(*reinterpret_cast<const uint4*>(af), *reinterpret_cast<const uint4*>(bf), acc);
    for (int q = 0; q < 4; ++q) Cout[(g * 4 + q) * 16 + r] = acc[q];
}

template <typename T>
static int run_selftest(const char* name) {
    constexpr int CK = ActTraits<T>::CK;
    float hA[16 * CK], hB[CK * 16], hC[256], ref[256];
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < CK; ++k) hA[i * CK + k] = (float)((i * 3 + k * 5) % 7 - 3);
    for (int k = 0; k < CK; ++k)
        for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (float)((k * 2 + j * 7 + 1) % 5 - 2);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float s = 0.f;
            for (int k = 0; k < CK; ++k) s += hA[i * CK + k] * hB[k * 16 + j];
            ref[i * 16 + j] = s;
        }
    T tA[16 * CK], tB[CK * 16];
    for (int i = 0; i < 16 * CK; ++i) { tA[i] = (T)hA[i]; tB[i] = (T)hB[i]; }
    T *dA = nullptr, *dB = nullptr;
    float* dC = nullptr;
    if (hipMalloc(&dA, sizeof(tA)) != hipSuccess || hipMalloc(&dB, sizeof(tB)) != hipSuccess || hipMalloc(&dC, sizeof(hC)) != hipSuccess) {
        sea_set_error("sea_selftest_mfma: hipMalloc failed");
        return SEA_ELAUNCH;
    }
    (void)hipMemcpy(dA, tA, sizeof(tA), hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, tB, sizeof(tB), hipMemcpyHostToDevice);
    mfma_selftest_kernel<T><<<1, 64>>>(dA, dB, dC);
    hipError_t e = hipDeviceSynchronize();
    (void)hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    if (e != hipSuccess) {
        sea_set_error("sea_selftest_mfma(%s): %s", name, hipGetErrorString(e));
        return SEA_ELAUNCH;
    }
    for (int i = 0; i < 256; ++i)
        if (hC[i] != ref[i]) {
            sea_set_error("sea_selftest_mfma(%s): C[%d][%d] = %g, expected %g — fragment map differs from the documented one", name, i / 16, i % 16, hC[i], ref[i]);
            return SEA_EUNSUPPORTED;
        }
    return SEA_OK;
}

extern "C" int sea_selftest_mfma(void) {
    int rc = run_selftest<__bf16>("bf16 16x16x32");
    if (rc != SEA_OK) return rc;
    return run_selftest<float>("f32 16x16x4");
}
