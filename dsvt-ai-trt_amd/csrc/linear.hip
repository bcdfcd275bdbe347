// linear.hip -- DsvtLinearPlugin: y = epilogue(A' * W^T + b) on the matrix cores of gfx950.
//
// Where the reference calls TensorRT's addFullyConnected (src/dsvt-ai-trt.cpp:283, 328-330,
// 448, 476, 490, 506, 525) followed by separate elementwise / LayerNorm / GELU plugins, this
// op runs the FC on MFMA and fuses what surrounds it:
//   prologue  A' = A + A2 for the first `add_cols` output columns (q = k = x + pos, v = x:
//             plugins/src/getValueByIndex.cu:299-301, hoisted from set slots to voxel rows)
//   epilogue  + bias; ReLU (:144) or tanh-GELU (plugins/src/gelu.cu:208-209); then up to three
//             chained "add residual, LayerNorm" stages (plugins/src/layerNorm.cu:261-402 and the
//             ElementWise SUMs around it, src/dsvt-ai-trt.cpp:669-697, 750-756).
// Rows are limited by a device-side count (count * row_mult), like every reference plugin.
//
// fp32 path: v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate).  Tile: 64 rows x
// 192 columns per 256-thread workgroup, K streamed through LDS in chunks of 32; wave w owns rows
// 16w..16w+15 and all 192 columns, so a LayerNorm row never leaves its wavefront.
// LDS rows are padded to 40 floats and lane group g reads k-chunks {g, g+4}: the two
// ds_read_b128 per fragment are bank-conflict free (checked by enumeration).
#include "plugin_base.h"
#include "device_utils.h"
#include "linear.h"

namespace dsvt {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 192, BK = 32, LDS_LD = 40, NT = BN / 16;


__device__ __forceinline__ float geluFast(float x) {
    // tanh-GELU of the reference (gelu.cu:208-209) in fp32
    const float B = 0.7978845608028654f, C = 0.035677408136300125f;
    return (0.5f + 0.5f * tanhf(x * (C * x * x + B))) * x;
}

// sum over the 16 lanes that share a row group (lanes differing in bits 0..3)
__device__ __forceinline__ float rowSum16(float v) {
    v += __shfl_xor(v, 1, kWave); v += __shfl_xor(v, 2, kWave);
    v += __shfl_xor(v, 4, kWave); v += __shfl_xor(v, 8, kWave);
    return v;
}

// Epilogue shared by the fp32 and fp16 MFMA kernels.  `acc` holds a 16-row x 192-column strip in
// the MFMA C/D layout: lane (r, g) owns rows rbase + i (i = 0..3, rbase already includes 4g) and
// columns n0 + t*16 + r.
__device__ __forceinline__ void linearEpilogue(floatx4 (&acc)[NT], const LinearArgs& a, int n0, int rbase, int r, int M, int N)
{
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int col = n0 + t * 16 + r;
        float b = (a.bias && col < N) ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = acc[t][i] + b;
            if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (a.act == ACT_GELU) v = geluFast(v);
            acc[t][i] = v;
        }
    }
    for (int s = 0; s < a.n_ln; ++s) {       // y = LayerNorm_s(y + res_s); needs N <= BN (checked on the host)
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int col = t * 16 + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = rbase + i;
                float v = acc[t][i];
                if (col < N && row < M) v += a.res[s][(size_t)row * N + col]; else if (col >= N) v = 0.f;
                acc[t][i] = v; sum[i] += v;
            }
        }
        float mean[4], den[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) mean[i] = rowSum16(sum[i]) / N;                 // layerNorm.cu:304-308
        float sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t * 16 + r < N)
#pragma unroll
                for (int i = 0; i < 4; ++i) { float d = acc[t][i] - mean[i]; sq[i] += d * d; }
#pragma unroll
        for (int i = 0; i < 4; ++i) den[i] = sqrtf(rowSum16(sq[i]) / N + a.eps);     // :333-337, :274
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int col = t * 16 + r;
            float gm = col < N ? a.gamma[s][col] : 0.f, bt = col < N ? a.beta[s][col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i] = (acc[t][i] - mean[i]) / den[i] * gm + bt;   // :274-276
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int col = n0 + t * 16 + r;
        if (col < N)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = rbase + i;
                if (row < M) a.out[(size_t)row * a.out_ld + col] = acc[t][i];
            }
    }
}

template <bool VEC>
__global__ void __launch_bounds__(256)
linear_f32_kernel(LinearArgs a)
{
    __shared__ __attribute__((aligned(16))) float sA[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float sW[BN * LDS_LD];
    uint32_t cnt = *a.count;
    long long Mll = (long long)cnt * a.row_mult;
    const int M = (int)(Mll < a.max_rows ? Mll : a.max_rows);
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int K = a.K, N = a.N;

    for (int n0 = 0; n0 < N; n0 += BN) {
        floatx4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
        const bool add = n0 < a.add_cols;
        const int ntiles = (N - n0 + 15) / 16 < NT ? (N - n0 + 15) / 16 : NT;

        for (int k0 = 0; k0 < K; k0 += BK) {
            __syncthreads();
            // ---- stage A (64 x 32) and W (192 x 32) into LDS ------------------------------
            for (int i = tid; i < BM * (BK / 4); i += 256) {
                int row = i >> 3, c4 = (i & 7) * 4, gr = m0 + row, k = k0 + c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gr < M) {
                    if (VEC) {
                        if (k < K) {
                            v = *reinterpret_cast<const float4*>(a.A + (size_t)gr * K + k);
                            if (add) {
                                float4 w = *reinterpret_cast<const float4*>(a.A2 + (size_t)gr * K + k);
                                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                            }
                        }
                    } else {
                        float t[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            t[j] = (k + j < K) ? a.A[(size_t)gr * K + k + j] + (add ? a.A2[(size_t)gr * K + k + j] : 0.f) : 0.f;
                        }
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
                *reinterpret_cast<float4*>(&sA[row * LDS_LD + c4]) = v;
            }
            for (int i = tid; i < BN * (BK / 4); i += 256) {
                int n = i >> 3, c4 = (i & 7) * 4, gn = n0 + n, k = k0 + c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gn < N) {
                    if (VEC) {
                        if (k < K) v = *reinterpret_cast<const float4*>(a.W + (size_t)gn * K + k);
                    } else {
                        float t[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[j] = (k + j < K) ? a.W[(size_t)gn * K + k + j] : 0.f;
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
                *reinterpret_cast<float4*>(&sW[n * LDS_LD + c4]) = v;
            }
            __syncthreads();
            // ---- MFMA: lane (r, g) holds k = {4g..4g+3} and {16+4g..16+4g+3} of the chunk --
            const float* pa = &sA[(wave * 16 + r) * LDS_LD + g * 4];
            const float4 a0 = *reinterpret_cast<const float4*>(pa);
            const float4 a1 = *reinterpret_cast<const float4*>(pa + 16);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int t4 = 0; t4 < NT; t4 += 4) {
                if (t4 < ntiles) {
                    float bv[4][8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* pb = &sW[((t4 + j) * 16 + r) * LDS_LD + g * 4];
                        const float4 b0 = *reinterpret_cast<const float4*>(pb);
                        const float4 b1 = *reinterpret_cast<const float4*>(pb + 16);
                        bv[j][0] = b0.x; bv[j][1] = b0.y; bv[j][2] = b0.z; bv[j][3] = b0.w;
                        bv[j][4] = b1.x; bv[j][5] = b1.y; bv[j][6] = b1.z; bv[j][7] = b1.w;
                    }
#pragma unroll
                    for (int s = 0; s < 8; ++s)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[t4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[j][s], acc[t4 + j], 0, 0, 0);
                }
            }
        }

        linearEpilogue(acc, a, n0, m0 + wave * 16 + g * 4, r, M, N);
    }
}

// -------------------------------------------------------------------------------------
// fp16-MFMA variant (v_mfma_f32_16x16x32_f16, fp32 accumulate; everything outside the products --
// bias, GELU, residuals, LayerNorm -- stays fp32).  Activations stay fp32 in HBM and are rounded
// to fp16 once, on their way into the A fragment.
//   * W chunk (192 output columns x K) lives in LDS for the whole workgroup: [192][K+16] halfs,
//     the 32-byte row pad makes the per-lane ds_read_b128 of a B fragment conflict-free;
//   * A never touches LDS: in the 16x16x32 A layout a lane needs 8 consecutive k of one row, i.e.
//     32 contiguous bytes of an fp32 row, and the four lane groups of a row cover one full 128-byte
//     line -- a direct global load is already perfectly coalesced and each element is used by one
//     wave only;
//   * a wave owns 32 rows (two 16-row MFMA tiles) so every B fragment read from LDS feeds two
//     MFMAs: one ds_read_b128 per MFMA would need 240 B/clk/CU of LDS bandwidth (peak 256).
// Workgroup = 4 waves = 128 rows x 192 columns; N > 192 loops over column chunks.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int BM16 = 128;

__device__ __forceinline__ half8 toHalf8(float4 x, float4 y) {
    half8 h;
    h[0] = (_Float16)x.x; h[1] = (_Float16)x.y; h[2] = (_Float16)x.z; h[3] = (_Float16)x.w;
    h[4] = (_Float16)y.x; h[5] = (_Float16)y.y; h[6] = (_Float16)y.z; h[7] = (_Float16)y.w;
    return h;
}

template <int KT>
__global__ void __launch_bounds__(256, 2)
linear_f16_kernel(LinearArgs a, const _Float16* __restrict__ Wh)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 sWh[];
    uint32_t cnt = *a.count;
    long long Mll = (long long)cnt * a.row_mult;
    const int M = (int)(Mll < a.max_rows ? Mll : a.max_rows);
    const int m0 = blockIdx.x * BM16;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int K = KT ? KT : a.K, N = a.N, LDW = K + 16, KC = K / 8;

    for (int n0 = 0; n0 < N; n0 += BN) {
        __syncthreads();
        for (int i = tid; i < BN * KC; i += 256) {
            int n = i / KC, c = i % KC;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (n0 + n < N) v = *reinterpret_cast<const uint4*>(Wh + (size_t)(n0 + n) * K + c * 8);
            *reinterpret_cast<uint4*>(&sWh[n * LDW + c * 8]) = v;
        }
        __syncthreads();
        floatx4 acc0[NT], acc1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { acc0[t] = floatx4{0.f, 0.f, 0.f, 0.f}; acc1[t] = floatx4{0.f, 0.f, 0.f, 0.f}; }
        const bool add = n0 < a.add_cols;
        const int ntiles = (N - n0 + 15) / 16 < NT ? (N - n0 + 15) / 16 : NT;
        int row0 = m0 + wave * 32 + r, row1 = row0 + 16;
        row0 = row0 < M ? row0 : M - 1; row1 = row1 < M ? row1 : M - 1;        // clamp: rows >= M are never stored
        const float* pa0 = a.A + (size_t)row0 * K + g * 8;
        const float* pa1 = a.A + (size_t)row1 * K + g * 8;
        const float* pb0 = add ? a.A2 + (size_t)row0 * K + g * 8 : nullptr;
        const float* pb1 = add ? a.A2 + (size_t)row1 * K + g * 8 : nullptr;
        // software pipeline: the global loads of k-step s+1 are in flight while step s runs on the MFMAs
        float4 x0 = *reinterpret_cast<const float4*>(pa0), x1 = *reinterpret_cast<const float4*>(pa0 + 4);
        float4 y0 = *reinterpret_cast<const float4*>(pa1), y1 = *reinterpret_cast<const float4*>(pa1 + 4);
        float4 p0, p1, q0, q1;
        if (add) {
            p0 = *reinterpret_cast<const float4*>(pb0); p1 = *reinterpret_cast<const float4*>(pb0 + 4);
            q0 = *reinterpret_cast<const float4*>(pb1); q1 = *reinterpret_cast<const float4*>(pb1 + 4);
        }
#pragma unroll 1
        for (int k0 = 0; k0 < K; k0 += 32) {
            if (add) {
                x0.x += p0.x; x0.y += p0.y; x0.z += p0.z; x0.w += p0.w; x1.x += p1.x; x1.y += p1.y; x1.z += p1.z; x1.w += p1.w;
                y0.x += q0.x; y0.y += q0.y; y0.z += q0.z; y0.w += q0.w; y1.x += q1.x; y1.y += q1.y; y1.z += q1.z; y1.w += q1.w;
            }
            const half8 af0 = toHalf8(x0, x1), af1 = toHalf8(y0, y1);
            const int kn = k0 + 32 < K ? k0 + 32 : k0;             // last step re-reads its own chunk (harmless)
            x0 = *reinterpret_cast<const float4*>(pa0 + kn); x1 = *reinterpret_cast<const float4*>(pa0 + kn + 4);
            y0 = *reinterpret_cast<const float4*>(pa1 + kn); y1 = *reinterpret_cast<const float4*>(pa1 + kn + 4);
            if (add) {
                p0 = *reinterpret_cast<const float4*>(pb0 + kn); p1 = *reinterpret_cast<const float4*>(pb0 + kn + 4);
                q0 = *reinterpret_cast<const float4*>(pb1 + kn); q1 = *reinterpret_cast<const float4*>(pb1 + kn + 4);
            }
            const _Float16* pw = &sWh[r * LDW + k0 + g * 8];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < ntiles) {
                    const half8 bf = *reinterpret_cast<const half8*>(pw + t * 16 * LDW);
                    acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af0, bf, acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1, bf, acc1[t], 0, 0, 0);
                }
            }
        }
        linearEpilogue(acc0, a, n0, m0 + wave * 32 + g * 4, r, M, N);
        linearEpilogue(acc1, a, n0, m0 + wave * 32 + 16 + g * 4, r, M, N);
    }
}

int launchLinearF16(const LinearArgs& a, const _Float16* Wh, hipStream_t stream) {
    static bool attr_done = false;
    const size_t lds = sizeof(_Float16) * (size_t)BN * (a.K + 16);
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_f16_kernel<192>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_f16_kernel<384>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_f16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    dim3 grid(cdiv(a.max_rows, BM16)), block(256);
    if (a.K == 192) hipLaunchKernelGGL(linear_f16_kernel<192>, grid, block, lds, stream, a, Wh);
    else if (a.K == 384) hipLaunchKernelGGL(linear_f16_kernel<384>, grid, block, lds, stream, a, Wh);
    else hipLaunchKernelGGL(linear_f16_kernel<0>, grid, block, lds, stream, a, Wh);
    return lastError();
}

int launchLinearF32(const LinearArgs& a, hipStream_t stream) {
    dim3 grid(cdiv(a.max_rows, BM)), block(256);
    if (a.K % 4 == 0) hipLaunchKernelGGL(linear_f32_kernel<true>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(linear_f32_kernel<false>, grid, block, 0, stream, a);
    return lastError();
}

// -------------------------------------------------------------------------------------
class DsvtLinearPlugin : public Plugin {
public:
    int max_rows_, K_, N_, row_mult_, act_, add_cols_, n_ln_; float eps_;
    int compute_type_;                 // 0: fp32 MFMA (exact fp32 products)   1: fp16 MFMA, fp32 accumulate
    std::vector<float> w_, b_, g_, be_;
    float *w_dev_ = nullptr, *b_dev_ = nullptr, *g_dev_ = nullptr, *be_dev_ = nullptr;
    _Float16* wh_dev_ = nullptr;
    bool ok_ = false;
    bool useF16() const { return compute_type_ == 1 && K_ % 32 == 0 && K_ <= 384; }
    DsvtLinearPlugin(int max_rows, int K, int N, int row_mult, int act, int add_cols, int n_ln, float eps, int compute_type,
                     const float* w, const float* b, const float* g, const float* be)
        : max_rows_(max_rows), K_(K), N_(N), row_mult_(row_mult), act_(act), add_cols_(add_cols), n_ln_(n_ln), eps_(eps),
          compute_type_(compute_type), w_(w, w + (size_t)N * K) {
        if (b) b_.assign(b, b + N);
        if (n_ln) { g_.assign(g, g + (size_t)n_ln * N); be_.assign(be, be + (size_t)n_ln * N); }
        auto up = [](const std::vector<float>& h, float** d) {
            if (h.empty()) { *d = nullptr; return true; }
            if (hipMalloc(d, sizeof(float) * h.size()) != hipSuccess) return false;
            return hipMemcpy(*d, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice) == hipSuccess;
        };
        ok_ = up(w_, &w_dev_) && up(b_, &b_dev_) && up(g_, &g_dev_) && up(be_, &be_dev_);
        if (ok_ && useF16()) {
            std::vector<_Float16> wh(w_.size());
            for (size_t i = 0; i < w_.size(); ++i) wh[i] = (_Float16)w_[i];
            ok_ = hipMalloc(&wh_dev_, sizeof(_Float16) * wh.size()) == hipSuccess &&
                  hipMemcpy(wh_dev_, wh.data(), sizeof(_Float16) * wh.size(), hipMemcpyHostToDevice) == hipSuccess;
        }
    }
    ~DsvtLinearPlugin() override {
        for (float* p : {w_dev_, b_dev_, g_dev_, be_dev_}) if (p) (void)hipFree(p);
        if (wh_dev_) (void)hipFree(wh_dev_);
    }
    const char* type() const override { return "DsvtLinearPlugin"; }
    int nbOutputs() const override { return 1; }
    int nbInputs() const { return 2 + (add_cols_ > 0 ? 1 : 0) + n_ln_; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i != 0) return -1;
        *out = dims3(in[0].d[0], max_rows_, N_); return 0;
    }
    int outputType(int, const int32_t* t, int) const override { return t[0]; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        if (io[pos].format != DSVT_FORMAT_LINEAR) return false;
        return pos == 1 ? io[pos].type == DSVT_INT32 : io[pos].type == DSVT_FLOAT;
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (!ok_) return static_cast<int>(hipErrorOutOfMemory);
        LinearArgs a{};
        int idx = 0;
        a.A = static_cast<const float*>(in[idx++]);
        a.count = static_cast<const uint32_t*>(in[idx++]);
        a.A2 = add_cols_ > 0 ? static_cast<const float*>(in[idx++]) : nullptr;
        for (int s = 0; s < n_ln_; ++s) {
            a.res[s] = static_cast<const float*>(in[idx++]);
            a.gamma[s] = g_dev_ + (size_t)s * N_; a.beta[s] = be_dev_ + (size_t)s * N_;
        }
        a.W = w_dev_; a.bias = b_dev_; a.out = static_cast<float*>(out[0]);
        a.row_mult = row_mult_; a.max_rows = max_rows_; a.K = K_; a.N = N_; a.add_cols = add_cols_; a.act = act_;
        a.n_ln = n_ln_; a.eps = eps_; a.out_ld = N_;
        if (zeroFill) DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(float) * (size_t)max_rows_ * N_, stream));
        return useF16() ? launchLinearF16(a, wh_dev_, stream) : launchLinearF32(a, stream);
    }
    size_t serializationSize() const override {
        return 7 * sizeof(int) + sizeof(float) + 2 * sizeof(int) + sizeof(float) * (w_.size() + b_.size() + g_.size() + be_.size());
    }
    void serialize(void* buf) const override {
        char* d = static_cast<char*>(buf);
        wr<int>(d, max_rows_); wr<int>(d, K_); wr<int>(d, N_); wr<int>(d, row_mult_); wr<int>(d, act_); wr<int>(d, add_cols_);
        wr<int>(d, n_ln_); wr<float>(d, eps_); wr<int>(d, b_.empty() ? 0 : 1); wr<int>(d, compute_type_);
        for (const std::vector<float>* v : {&w_, &b_, &g_, &be_}) { memcpy(d, v->data(), sizeof(float) * v->size()); d += sizeof(float) * v->size(); }
    }
    Plugin* clone() const override {
        return new DsvtLinearPlugin(max_rows_, K_, N_, row_mult_, act_, add_cols_, n_ln_, eps_, compute_type_, w_.data(),
                                    b_.empty() ? nullptr : b_.data(), g_.data(), be_.data());
    }
};

static Plugin* linNew(int max_rows, int K, int N, int row_mult, int act, int add_cols, int n_ln, float eps, int compute_type,
                      const float* w, const float* b, const float* g, const float* be) {
    if (max_rows <= 0 || K <= 0 || N <= 0 || row_mult <= 0 || !w) return nullptr;
    if (compute_type < 0 || compute_type > 1) return nullptr;
    if (act < 0 || act > 2 || n_ln < 0 || n_ln > 3) return nullptr;
    if (n_ln > 0 && (N > BN || !g || !be)) return nullptr;                  // a LayerNorm row must fit one tile
    if (add_cols < 0 || add_cols > N || (add_cols % BN != 0 && add_cols != N)) return nullptr;
    DsvtLinearPlugin* p = new DsvtLinearPlugin(max_rows, K, N, row_mult, act, add_cols, n_ln, eps, compute_type, w, b, g, be);
    return p;
}
static Plugin* linCreate(const DsvtPluginFieldCollection* fc) {
    const DsvtPluginField* w = findField(fc, "weight"); const DsvtPluginField* b = findField(fc, "bias");
    const DsvtPluginField* g = findField(fc, "ln_weights"); const DsvtPluginField* be = findField(fc, "ln_bias");
    int K = fieldInt(fc, "in_features"), N = fieldInt(fc, "out_features"), n_ln = fieldInt(fc, "num_layer_norms");
    if (!w || !w->data || w->length != K * N) return nullptr;
    if (b && b->data && b->length != N) return nullptr;
    if (n_ln > 0 && (!g || !be || g->length != n_ln * N || be->length != n_ln * N)) return nullptr;
    return linNew(fieldInt(fc, "max_rows"), K, N, fieldInt(fc, "row_mult", 1), fieldInt(fc, "activation"),
                  fieldInt(fc, "add_cols"), n_ln, fieldFloat(fc, "ln_eps", 0.f), fieldInt(fc, "compute_type", 0),
                  static_cast<const float*>(w->data),
                  (b && b->data) ? static_cast<const float*>(b->data) : nullptr,
                  g ? static_cast<const float*>(g->data) : nullptr, be ? static_cast<const float*>(be->data) : nullptr);
}
static Plugin* linDeser(const void* data, size_t len) {
    if (len < 9 * sizeof(int) + sizeof(float)) return nullptr;
    const char* d = static_cast<const char*>(data);
    int max_rows = rd<int>(d), K = rd<int>(d), N = rd<int>(d), row_mult = rd<int>(d), act = rd<int>(d), add_cols = rd<int>(d);
    int n_ln = rd<int>(d); float eps = rd<float>(d); int has_b = rd<int>(d); int ctype = rd<int>(d);
    if (K <= 0 || N <= 0 || n_ln < 0 || n_ln > 3) return nullptr;
    size_t need = (size_t)K * N + (has_b ? N : 0) + 2 * (size_t)n_ln * N;
    if (len < 9 * sizeof(int) + sizeof(float) + need * sizeof(float)) return nullptr;
    std::vector<float> all(need);
    memcpy(all.data(), d, need * sizeof(float));
    const float* w = all.data(); const float* b = has_b ? w + (size_t)K * N : nullptr;
    const float* g = w + (size_t)K * N + (has_b ? N : 0); const float* be = g + (size_t)n_ln * N;
    return linNew(max_rows, K, N, row_mult, act, add_cols, n_ln, eps, ctype, w, b, g, be);
}
static Creator g_linCreator{"DsvtLinearPlugin",
    {{"max_rows", DSVT_FIELD_INT32}, {"in_features", DSVT_FIELD_INT32}, {"out_features", DSVT_FIELD_INT32},
     {"row_mult", DSVT_FIELD_INT32}, {"activation", DSVT_FIELD_INT32}, {"add_cols", DSVT_FIELD_INT32},
     {"num_layer_norms", DSVT_FIELD_INT32}, {"ln_eps", DSVT_FIELD_FLOAT32}, {"compute_type", DSVT_FIELD_INT32},
     {"weight", DSVT_FIELD_FLOAT32},
     {"bias", DSVT_FIELD_FLOAT32}, {"ln_weights", DSVT_FIELD_FLOAT32}, {"ln_bias", DSVT_FIELD_FLOAT32}},
    linCreate, linDeser, {}, {}};
static Registrar g_linReg(&g_linCreator);

}  // namespace dsvt
