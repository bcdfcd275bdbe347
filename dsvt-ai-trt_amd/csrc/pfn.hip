// pfn.hip -- DsvtPillarFeatureNetPlugin: the two PFN layers + both scatter-max reductions of the voxel feature
// encoder in two launches that never write a per-point activation.
//
// Reference wiring (src/dsvt-ai-trt.cpp:565-589): x0 = ReLU(BN(FC0(f))) [Nk,96] -> TorchScatterMax -> concat
// [x0 | max_pillar(x0)] [Nk,192] -> x1 = ReLU(BN(FC1(cat))) [Nk,192] -> TorchScatterMax -> pillar features [P,192].
// As separate launches that is ~1 GB of traffic per 180k-point frame (six kernels, 0.33 ms).  Two facts remove it:
//   * FC1 is linear in the two halves of its input:  FC1(cat) = W1a x0 + W1b max_pillar(x0) + b1, and the second
//     term is per PILLAR.  With t_p = W1b m_p + b1 (an ordinary [P,96] x [96,192] linear, DsvtLinearPlugin),
//     x1(point) = ReLU(W1a x0(point) + t_pillar(point)).
//   * x0 costs 10 MACs per output: cheaper to recompute than to store.
// So:  pass 0:  m_p = max over the pillar's points of x0                      -> [P, 96]
//      (linear: t = W1b' m + b1')                                             -> [P,192]
//      pass 1:  vfeat_p = max over the pillar's points of ReLU(W1a' x0 + t_p) -> [P,192] (fp32 and an fp16 copy)
// BatchNorm is folded into the weights by the caller.  The compact point ids of a pillar are consecutive
// (Points2Features' canonical order: pillar-major, then slot), so a workgroup that owns 32 consecutive pillars owns
// a contiguous range of point rows: no pillar straddles workgroups, the per-pillar maxima live in an LDS table
// (ds_max_u32 on the bit pattern: values are >= 0 after the ReLU), and nothing is atomic in global memory.
//
// Arithmetic: layer 0 on v_mfma_f32_16x16x4_f32 (fp32: the inputs are raw metric coordinates), layer 1 on
// v_mfma_f32_16x16x32_f16 with the layer-0 tile chained through registers (k-permuted W1a, see mlp.hip); maxima fp32.
#include "plugin_base.h"
#include "device_utils.h"
#include <cstdlib>

namespace dsvt {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int PF_IN = 10, PF_C0 = 96, PF_C1 = 192;
constexpr int PF_PB = 32;              // pillars per workgroup
constexpr int PF_NW = 4;               // waves per workgroup: 64 point rows per iteration

struct PfnArgs {
    const float* feat;                 // [Nk, 10]
    const uint32_t* pidx; int T;       // [P, T] compact point ids (first entry = the pillar's first row)
    const uint32_t* pcnt;              // [P]
    const uint32_t* pillar_num; int max_pillars;
    const float* w0; const float* b0;  // [96][12] (k padded with zeros), [96]
    const _Float16* w1a;               // layer 1: fragment-ordered [3 k-steps][12 tiles][64 lanes][8] (k-permuted); nullptr in pass 0
    const float* t;                    // [P,192] per-pillar term of layer 1 (pass 1)
    float* out; _Float16* out16;       // pass 0: m [P,96] (out16 unused); pass 1: vfeat [P,192] + fp16 copy
};

template <bool LAYER1>
__global__ void __launch_bounds__(64 * PF_NW)
pfn_kernel(PfnArgs a)
{
    constexpr int NC = LAYER1 ? PF_C1 : PF_C0, NT = NC / 16;
    __shared__ uint32_t sMax[PF_PB * NC];                              // 24 KB / 12 KB
    __shared__ uint32_t sStart[PF_PB + 1];
    __shared__ __attribute__((aligned(16))) _Float16 sW1[LAYER1 ? 3 * 12 * 512 : 8];     // 36 KB
    uint32_t P = *a.pillar_num; if (P > (uint32_t)a.max_pillars) P = a.max_pillars;
    const uint32_t pb0 = blockIdx.x * PF_PB;
    if (pb0 >= P) return;
    const int npil = P - pb0 < (uint32_t)PF_PB ? (int)(P - pb0) : PF_PB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;

    for (int i = tid; i < PF_PB * NC; i += 64 * PF_NW) sMax[i] = 0u;
    if (tid <= npil) {
        // first row of pillar pb0 + tid; the sentinel entry is one past the last pillar's rows
        const uint32_t p = pb0 + (tid < npil ? tid : npil - 1);
        const uint32_t s = a.pidx[(size_t)p * a.T];
        sStart[tid] = tid < npil ? s : s + a.pcnt[p];
    }
    if (LAYER1) {
        for (int i = tid; i < 3 * 12 * 64; i += 64 * PF_NW)
            *reinterpret_cast<uint4*>(&sW1[i * 8]) = *reinterpret_cast<const uint4*>(a.w1a + (size_t)i * 8);
    }
    // layer-0 weights as MFMA A fragments: lane (r, g) holds W0[16t + r][4ks + g]
    float w0f[6][3], b0f[6][4];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) w0f[t][ks] = a.w0[(16 * t + r) * 12 + 4 * ks + g];
        const float4 b = *reinterpret_cast<const float4*>(a.b0 + 16 * t + 4 * g);
        b0f[t][0] = b.x; b0f[t][1] = b.y; b0f[t][2] = b.z; b0f[t][3] = b.w;
    }
    __syncthreads();
    const uint32_t row0 = sStart[0], rowEnd = sStart[npil];

    for (uint32_t base = row0; base < rowEnd; base += 16 * PF_NW) {
        const uint32_t row = base + 16 * wave + r;
        const bool valid = row < rowEnd;
        // pillar of this row: the last j with sStart[j] <= row
        int j = 0;
#pragma unroll
        for (int step = 16; step > 0; step >>= 1) { const int c = j + step; if (c < npil && sStart[c] <= row) j = c; }
        // ---- layer 0: x0^T tile = W0 f^T; B fragment: lane (r, g) holds f[row][4ks + g] ---------------------------
        const float* fr = a.feat + (size_t)(valid ? row : row0) * PF_IN;
        float fb[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { const int k = 4 * ks + g; fb[ks] = k < PF_IN ? fr[k] : 0.f; }
        floatx4 x0[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            x0[t] = floatx4{b0f[t][0], b0f[t][1], b0f[t][2], b0f[t][3]};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) x0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0f[t][ks], fb[ks], x0[t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) x0[t][i] = fmaxf(x0[t][i], 0.f);                 // ReLU (:144)
        }
        if (!LAYER1) {
            if (valid) {
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) atomicMax(&sMax[j * NC + 16 * t + 4 * g + i], __float_as_uint(x0[t][i]) & 0x7fffffffu);      // (-0 -> +0)
            }
        } else {
            // ---- layer 1: x1^T = W1a x0^T (+ t_pillar); the layer-0 tile is the B operand (k-step s = tiles 2s, 2s+1) ---
            half8 f1[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                half8 h;
#pragma unroll
                for (int i = 0; i < 4; ++i) { h[i] = (_Float16)x0[2 * s][i]; h[4 + i] = (_Float16)x0[2 * s + 1][i]; }
                f1[s] = h;
            }
            const float* tr = a.t + (size_t)(pb0 + j) * PF_C1;
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const float4 tv = *reinterpret_cast<const float4*>(tr + 16 * t + 4 * g);
                floatx4 acc = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&sW1[((s * 12 + t) * 64 + lane) * 8]), f1[s], acc, 0, 0, 0);
                if (valid) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) atomicMax(&sMax[j * NC + 16 * t + 4 * g + i], __float_as_uint(fmaxf(acc[i], 0.f)) & 0x7fffffffu);
                }
            }
        }
    }
    __syncthreads();
    // ---- the 32 pillar rows of this workgroup ----------------------------------------------------------------------
    for (int i = tid; i < npil * (NC / 4); i += 64 * PF_NW) {
        const int pj = i / (NC / 4), c4 = (i % (NC / 4)) * 4;
        const uint4 v = *reinterpret_cast<const uint4*>(&sMax[pj * NC + c4]);
        const float4 f = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        *reinterpret_cast<float4*>(a.out + (size_t)(pb0 + pj) * NC + c4) = f;
        if (LAYER1 && a.out16) {
            half4 h; h[0] = (_Float16)f.x; h[1] = (_Float16)f.y; h[2] = (_Float16)f.z; h[3] = (_Float16)f.w;
            *reinterpret_cast<half4*>(a.out16 + (size_t)(pb0 + pj) * NC + c4) = h;
        }
    }
}

static inline int pfnPermuteK(int p) {       // see mlp.hip: position p of a permuted weight row holds column k(p)
    const int s = p / 32, q = p % 32, g = q / 8, j = q % 8;
    return 32 * s + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
}

// fields: max_pillars_num, layer (0 | 1), weight (layer 0: [96][10]; layer 1: W1a = [192][96], the x0 half of the folded FC1),
// bias (layer 0: [96]).  Inputs: feat [1,Nk,10] f32, pidx [1,P,T] i32, pcnt [1,P,1] i32, pillar_num [1] (, t [1,P,192] f32 for layer 1).
// Outputs: layer 0: m [1,P,96] f32; layer 1: vfeat [1,P,192] f32 + fp16 copy.
class DsvtPillarFeatureNetPlugin : public Plugin {
public:
    int max_pillars_, layer_;
    std::vector<float> w_, b_;
    float* w0_dev_ = nullptr; float* b0_dev_ = nullptr; _Float16* w1_dev_ = nullptr;
    // layer 1 re-computes layer 0, so it carries both weight sets
    std::vector<float> w0_, b0_;
    bool ok_ = false;
    DsvtPillarFeatureNetPlugin(int mp, int layer, const float* w0, const float* b0, const float* w1a)
        : max_pillars_(mp), layer_(layer), w0_(w0, w0 + PF_C0 * PF_IN), b0_(b0, b0 + PF_C0) {
        if (layer) w_.assign(w1a, w1a + (size_t)PF_C1 * PF_C0);
        std::vector<float> w0p((size_t)PF_C0 * 12, 0.f);
        for (int n = 0; n < PF_C0; ++n) for (int k = 0; k < PF_IN; ++k) w0p[n * 12 + k] = w0_[n * PF_IN + k];
        ok_ = hipMalloc(&w0_dev_, sizeof(float) * w0p.size()) == hipSuccess &&
              hipMemcpy(w0_dev_, w0p.data(), sizeof(float) * w0p.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc(&b0_dev_, sizeof(float) * PF_C0) == hipSuccess &&
              hipMemcpy(b0_dev_, b0_.data(), sizeof(float) * PF_C0, hipMemcpyHostToDevice) == hipSuccess;
        if (ok_ && layer) {
            std::vector<_Float16> wp((size_t)3 * 12 * 512);
            for (int s = 0; s < 3; ++s)
                for (int t = 0; t < 12; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j)
                            wp[(((size_t)s * 12 + t) * 64 + lane) * 8 + j] =
                                (_Float16)w_[(size_t)(16 * t + (lane & 15)) * PF_C0 + pfnPermuteK(32 * s + 8 * (lane >> 4) + j)];
            ok_ = hipMalloc(&w1_dev_, sizeof(_Float16) * wp.size()) == hipSuccess &&
                  hipMemcpy(w1_dev_, wp.data(), sizeof(_Float16) * wp.size(), hipMemcpyHostToDevice) == hipSuccess;
        }
    }
    ~DsvtPillarFeatureNetPlugin() override {
        if (w0_dev_) (void)hipFree(w0_dev_); if (b0_dev_) (void)hipFree(b0_dev_); if (w1_dev_) (void)hipFree(w1_dev_);
    }
    const char* type() const override { return "DsvtPillarFeatureNetPlugin"; }
    int nbOutputs() const override { return layer_ ? 2 : 1; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i < 0 || i >= nbOutputs()) return -1;
        *out = dims3(in[0].d[0], max_pillars_, layer_ ? PF_C1 : PF_C0); return 0;
    }
    int outputType(int i, const int32_t*, int) const override { return i == 0 ? DSVT_FLOAT : DSVT_HALF; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int nbIn, int) const override {
        if (io[pos].format != DSVT_FORMAT_LINEAR) return false;
        if (pos >= 1 && pos <= 3) return io[pos].type == DSVT_INT32;
        if (pos < nbIn) return io[pos].type == DSVT_FLOAT;
        return io[pos].type == (pos == nbIn ? DSVT_FLOAT : DSVT_HALF);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (!ok_) return static_cast<int>(hipErrorOutOfMemory);
        PfnArgs a{};
        a.feat = static_cast<const float*>(in[0]); a.pidx = static_cast<const uint32_t*>(in[1]);
        a.T = inDesc ? inDesc[1].dims.d[inDesc[1].dims.nbDims - 1] : 48;
        a.pcnt = static_cast<const uint32_t*>(in[2]); a.pillar_num = static_cast<const uint32_t*>(in[3]); a.max_pillars = max_pillars_;
        a.w0 = w0_dev_; a.b0 = b0_dev_; a.w1a = w1_dev_; a.t = layer_ ? static_cast<const float*>(in[4]) : nullptr;
        a.out = static_cast<float*>(out[0]); a.out16 = layer_ ? static_cast<_Float16*>(out[1]) : nullptr;
        const int NC = layer_ ? PF_C1 : PF_C0;
        if (zeroFill) {
            DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(float) * (size_t)max_pillars_ * NC, stream));
            if (layer_) DSVT_CHECK(hipMemsetAsync(out[1], 0, sizeof(_Float16) * (size_t)max_pillars_ * NC, stream));
        }
        const dim3 grid(cdiv(max_pillars_, PF_PB));
        if (layer_) hipLaunchKernelGGL(pfn_kernel<true>, grid, dim3(64 * PF_NW), 0, stream, a);
        else hipLaunchKernelGGL(pfn_kernel<false>, grid, dim3(64 * PF_NW), 0, stream, a);
        return lastError();
    }
    size_t serializationSize() const override { return 2 * sizeof(int) + sizeof(float) * (w0_.size() + b0_.size() + w_.size()); }
    void serialize(void* buf) const override {
        char* d = static_cast<char*>(buf);
        wr<int>(d, max_pillars_); wr<int>(d, layer_);
        for (const std::vector<float>* v : {&w0_, &b0_, &w_}) { memcpy(d, v->data(), sizeof(float) * v->size()); d += sizeof(float) * v->size(); }
    }
    Plugin* clone() const override { return new DsvtPillarFeatureNetPlugin(max_pillars_, layer_, w0_.data(), b0_.data(), w_.empty() ? nullptr : w_.data()); }
};
static Plugin* pfnCreate(const DsvtPluginFieldCollection* fc) {
    const int mp = fieldInt(fc, "max_pillars_num"), layer = fieldInt(fc, "layer");
    const DsvtPluginField* w0 = findField(fc, "weight0"); const DsvtPluginField* b0 = findField(fc, "bias0");
    const DsvtPluginField* w1 = findField(fc, "weight1");
    if (mp <= 0 || layer < 0 || layer > 1 || !w0 || !w0->data || w0->length != PF_C0 * PF_IN || !b0 || !b0->data || b0->length != PF_C0) return nullptr;
    if (layer && (!w1 || !w1->data || w1->length != PF_C1 * PF_C0)) return nullptr;
    return new DsvtPillarFeatureNetPlugin(mp, layer, static_cast<const float*>(w0->data), static_cast<const float*>(b0->data),
                                          layer ? static_cast<const float*>(w1->data) : nullptr);
}
static Plugin* pfnDeser(const void* data, size_t len) {
    if (len < 2 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    const int mp = rd<int>(d), layer = rd<int>(d);
    if (mp <= 0 || layer < 0 || layer > 1) return nullptr;
    const size_t n = (size_t)PF_C0 * PF_IN + PF_C0 + (layer ? (size_t)PF_C1 * PF_C0 : 0);
    if (len < 2 * sizeof(int) + n * sizeof(float)) return nullptr;
    std::vector<float> all(n); memcpy(all.data(), d, n * sizeof(float));
    return new DsvtPillarFeatureNetPlugin(mp, layer, all.data(), all.data() + PF_C0 * PF_IN, layer ? all.data() + PF_C0 * PF_IN + PF_C0 : nullptr);
}
static Creator g_pfnCreator{"DsvtPillarFeatureNetPlugin",
    {{"max_pillars_num", DSVT_FIELD_INT32}, {"layer", DSVT_FIELD_INT32}, {"weight0", DSVT_FIELD_FLOAT32}, {"bias0", DSVT_FIELD_FLOAT32},
     {"weight1", DSVT_FIELD_FLOAT32}},
    pfnCreate, pfnDeser, {}, {}};
static Registrar g_pfnReg(&g_pfnCreator);

}  // namespace dsvt
