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

        // ---- epilogue: lane holds rows m0 + wave*16 + 4g + i (i = 0..3), cols n0 + t*16 + r ----
        const int rbase = m0 + wave * 16 + g * 4;
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
    std::vector<float> w_, b_, g_, be_;
    float *w_dev_ = nullptr, *b_dev_ = nullptr, *g_dev_ = nullptr, *be_dev_ = nullptr;
    bool ok_ = false;
    DsvtLinearPlugin(int max_rows, int K, int N, int row_mult, int act, int add_cols, int n_ln, float eps,
                     const float* w, const float* b, const float* g, const float* be)
        : max_rows_(max_rows), K_(K), N_(N), row_mult_(row_mult), act_(act), add_cols_(add_cols), n_ln_(n_ln), eps_(eps),
          w_(w, w + (size_t)N * K) {
        if (b) b_.assign(b, b + N);
        if (n_ln) { g_.assign(g, g + (size_t)n_ln * N); be_.assign(be, be + (size_t)n_ln * N); }
        auto up = [](const std::vector<float>& h, float** d) {
            if (h.empty()) { *d = nullptr; return true; }
            if (hipMalloc(d, sizeof(float) * h.size()) != hipSuccess) return false;
            return hipMemcpy(*d, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice) == hipSuccess;
        };
        ok_ = up(w_, &w_dev_) && up(b_, &b_dev_) && up(g_, &g_dev_) && up(be_, &be_dev_);
    }
    ~DsvtLinearPlugin() override {
        for (float* p : {w_dev_, b_dev_, g_dev_, be_dev_}) if (p) (void)hipFree(p);
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
        return launchLinearF32(a, stream);
    }
    size_t serializationSize() const override {
        return 7 * sizeof(int) + sizeof(float) + sizeof(int) + sizeof(float) * (w_.size() + b_.size() + g_.size() + be_.size());
    }
    void serialize(void* buf) const override {
        char* d = static_cast<char*>(buf);
        wr<int>(d, max_rows_); wr<int>(d, K_); wr<int>(d, N_); wr<int>(d, row_mult_); wr<int>(d, act_); wr<int>(d, add_cols_);
        wr<int>(d, n_ln_); wr<float>(d, eps_); wr<int>(d, b_.empty() ? 0 : 1);
        for (const std::vector<float>* v : {&w_, &b_, &g_, &be_}) { memcpy(d, v->data(), sizeof(float) * v->size()); d += sizeof(float) * v->size(); }
    }
    Plugin* clone() const override {
        return new DsvtLinearPlugin(max_rows_, K_, N_, row_mult_, act_, add_cols_, n_ln_, eps_, w_.data(),
                                    b_.empty() ? nullptr : b_.data(), g_.data(), be_.data());
    }
};

static Plugin* linNew(int max_rows, int K, int N, int row_mult, int act, int add_cols, int n_ln, float eps,
                      const float* w, const float* b, const float* g, const float* be) {
    if (max_rows <= 0 || K <= 0 || N <= 0 || row_mult <= 0 || !w) return nullptr;
    if (act < 0 || act > 2 || n_ln < 0 || n_ln > 3) return nullptr;
    if (n_ln > 0 && (N > BN || !g || !be)) return nullptr;                  // a LayerNorm row must fit one tile
    if (add_cols < 0 || add_cols > N || (add_cols % BN != 0 && add_cols != N)) return nullptr;
    DsvtLinearPlugin* p = new DsvtLinearPlugin(max_rows, K, N, row_mult, act, add_cols, n_ln, eps, w, b, g, be);
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
                  fieldInt(fc, "add_cols"), n_ln, fieldFloat(fc, "ln_eps", 0.f), static_cast<const float*>(w->data),
                  (b && b->data) ? static_cast<const float*>(b->data) : nullptr,
                  g ? static_cast<const float*>(g->data) : nullptr, be ? static_cast<const float*>(be->data) : nullptr);
}
static Plugin* linDeser(const void* data, size_t len) {
    if (len < 8 * sizeof(int) + sizeof(float)) return nullptr;
    const char* d = static_cast<const char*>(data);
    int max_rows = rd<int>(d), K = rd<int>(d), N = rd<int>(d), row_mult = rd<int>(d), act = rd<int>(d), add_cols = rd<int>(d);
    int n_ln = rd<int>(d); float eps = rd<float>(d); int has_b = rd<int>(d);
    if (K <= 0 || N <= 0 || n_ln < 0 || n_ln > 3) return nullptr;
    size_t need = (size_t)K * N + (has_b ? N : 0) + 2 * (size_t)n_ln * N;
    if (len < 8 * sizeof(int) + sizeof(float) + need * sizeof(float)) return nullptr;
    std::vector<float> all(need);
    memcpy(all.data(), d, need * sizeof(float));
    const float* w = all.data(); const float* b = has_b ? w + (size_t)K * N : nullptr;
    const float* g = w + (size_t)K * N + (has_b ? N : 0); const float* be = g + (size_t)n_ln * N;
    return linNew(max_rows, K, N, row_mult, act, add_cols, n_ln, eps, w, b, g, be);
}
static Creator g_linCreator{"DsvtLinearPlugin",
    {{"max_rows", DSVT_FIELD_INT32}, {"in_features", DSVT_FIELD_INT32}, {"out_features", DSVT_FIELD_INT32},
     {"row_mult", DSVT_FIELD_INT32}, {"activation", DSVT_FIELD_INT32}, {"add_cols", DSVT_FIELD_INT32},
     {"num_layer_norms", DSVT_FIELD_INT32}, {"ln_eps", DSVT_FIELD_FLOAT32}, {"weight", DSVT_FIELD_FLOAT32},
     {"bias", DSVT_FIELD_FLOAT32}, {"ln_weights", DSVT_FIELD_FLOAT32}, {"ln_bias", DSVT_FIELD_FLOAT32}},
    linCreate, linDeser, {}, {}};
static Registrar g_linReg(&g_linCreator);

}  // namespace dsvt
