// pillar_ops.hip -- the HBM-bound row-wise plugins for gfx950:
//   TorchScatterMaxPlugin     plugins/src/torchScatterMax.cu:201-309
//   GetValueByIndexPlugin     plugins/src/getValueByIndex.cu:282-355
//   MapSetFeature2VoxelPlugin plugins/src/mapSetFeature2voxel.cu:258-320
//   LayerNormPlugin           plugins/src/layerNorm.cu:261-402
//   GeluPlugin                plugins/src/gelu.cu:201-250
//   Map2BevPlugin             plugins/src/map2bev.cu:250-310
//   FilterBoxByScorePlugin    plugins/src/filterBoxByScore.cu:266-379
// The reference runs most of these one thread per pillar with stride-C (uncoalesced) row walks
// and a 200-float per-thread scratch array; here every row is read/written as consecutive
// float4 by consecutive lanes, reductions are wavefront shuffles, and the device-side valid
// counts (P, S) are honoured without visiting the host.
#include "plugin_base.h"
#include "device_utils.h"

namespace dsvt {

static bool f32Linear(const DsvtPluginTensorDesc& t) { return t.type == DSVT_FLOAT && t.format == DSVT_FORMAT_LINEAR; }
static bool i32Linear(const DsvtPluginTensorDesc& t) { return t.type == DSVT_INT32 && t.format == DSVT_FORMAT_LINEAR; }

// =====================================================================================
// TorchScatterMax
// =====================================================================================
// G = C/4 lanes own one pillar (one float4 of channels each); a 256-thread workgroup
// holds 256/G pillars.  Per point the G lanes read one full feature row (coalesced).
__global__ void __launch_bounds__(256)
scatter_max_kernel(const float4* __restrict__ feat, const uint32_t* __restrict__ pidx, const uint32_t* __restrict__ pcnt,
                   const uint32_t* __restrict__ pillar_num, int T, int G, int pillars_per_block,
                   float4* __restrict__ max_point, float4* __restrict__ max_voxel)
{
    int g = threadIdx.x / G, c = threadIdx.x % G;
    if (g >= pillars_per_block) return;
    uint32_t p = blockIdx.x * pillars_per_block + g;
    if (p >= *pillar_num) return;
    const uint32_t* idx = pidx + (size_t)p * T;
    uint32_t n = pcnt[p];
    float4 m = make_float4(-1000000.0f, -1000000.0f, -1000000.0f, -1000000.0f);      // torchScatterMax.cu:213-216
    for (uint32_t i = 0; i < n; ++i) {
        float4 v = feat[(size_t)idx[i] * G + c];
        m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y;                  // :226-236 (strict >)
        m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
    }
    max_voxel[(size_t)p * G + c] = m;                                                // :240-243
    for (uint32_t i = 0; i < n; ++i) max_point[(size_t)idx[i] * G + c] = m;          // :246-257
}

class TorchScatterMaxPlugin : public Plugin {
public:
    int max_points_num_, max_pillars_num_, feature_num_;
    TorchScatterMaxPlugin(int a, int b, int c) : max_points_num_(a), max_pillars_num_(b), feature_num_(c) {}
    const char* type() const override { return "TorchScatterMaxPlugin"; }
    int nbOutputs() const override { return 2; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i == 0) { *out = dims3(in[0].d[0], max_points_num_, feature_num_); return 0; }
        if (i == 1) { *out = dims3(in[0].d[0], max_pillars_num_, feature_num_); return 0; }
        return -1;
    }
    int outputType(int, const int32_t* t, int) const override { return t[0]; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        if (pos == 0 || pos == 4 || pos == 5) return f32Linear(io[pos]);
        return pos >= 1 && pos <= 3 && i32Linear(io[pos]);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out,
                void*, hipStream_t stream) override {
        int T = inDesc ? inDesc[1].dims.d[inDesc[1].dims.nbDims - 1] : 48;           // POINTS_NUM_PER_VOXEL in the reference
        int G = feature_num_ / 4, ppb = 256 / G;
        if (zeroFill) {                                                             // :300-301
            DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(float) * (size_t)max_points_num_ * feature_num_, stream));
            DSVT_CHECK(hipMemsetAsync(out[1], 0, sizeof(float) * (size_t)max_pillars_num_ * feature_num_, stream));
        }
        hipLaunchKernelGGL(scatter_max_kernel, dim3(cdiv(max_pillars_num_, ppb)), dim3(256), 0, stream,
                           static_cast<const float4*>(in[0]), static_cast<const uint32_t*>(in[1]),
                           static_cast<const uint32_t*>(in[2]), static_cast<const uint32_t*>(in[3]), T, G, ppb,
                           static_cast<float4*>(out[0]), static_cast<float4*>(out[1]));
        return lastError();
    }
    size_t serializationSize() const override { return 3 * sizeof(int); }
    void serialize(void* b) const override { char* d = static_cast<char*>(b); wr<int>(d, max_points_num_); wr<int>(d, max_pillars_num_); wr<int>(d, feature_num_); }
    Plugin* clone() const override { return new TorchScatterMaxPlugin(max_points_num_, max_pillars_num_, feature_num_); }
};
static Plugin* smNew(int a, int b, int c) {
    return (a > 0 && b > 0 && c > 0 && c % 4 == 0 && c / 4 <= 256) ? new TorchScatterMaxPlugin(a, b, c) : nullptr;
}
static Plugin* smCreate(const DsvtPluginFieldCollection* fc) {
    return smNew(fieldInt(fc, "max_points_num"), fieldInt(fc, "max_pillars_num"), fieldInt(fc, "feature_num"));
}
static Plugin* smDeser(const void* data, size_t len) {
    if (len < 3 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    int a = rd<int>(d), b = rd<int>(d), c = rd<int>(d);
    return smNew(a, b, c);
}
static Creator g_smCreator{"TorchScatterMaxPlugin",
    {{"max_points_num", DSVT_FIELD_INT32}, {"max_pillars_num", DSVT_FIELD_INT32}, {"feature_num", DSVT_FIELD_INT32}},   // :376-378
    smCreate, smDeser, {}, {}};
static Registrar g_smReg(&g_smCreator);

// =====================================================================================
// GetValueByIndex / MapSetFeature2Voxel
// =====================================================================================
__global__ void __launch_bounds__(256)
get_value_kernel(const float4* __restrict__ feat, const float4* __restrict__ pos, const uint32_t* __restrict__ inds,
                 const uint32_t* __restrict__ set_num, int L, int G, float4* __restrict__ q, float4* __restrict__ k,
                 float4* __restrict__ v)
{
    size_t total = (size_t)(*set_num) * L * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t row = i / G; int c = (int)(i % G);
        uint32_t vid = inds[row];                                                    // getValueByIndex.cu:294
        float4 f = feat[(size_t)vid * G + c], p = pos[(size_t)vid * G + c];
        float4 s = make_float4(f.x + p.x, f.y + p.y, f.z + p.z, f.w + p.w);          // :299-301
        q[i] = s; k[i] = s; v[i] = f;
    }
}

class GetValueByIndexPlugin : public Plugin {
public:
    int max_win_num_, voxel_num_set_, channel_num_, axis_id_;
    GetValueByIndexPlugin(int mw, int l, int c, int a) : max_win_num_(mw), voxel_num_set_(l), channel_num_(c), axis_id_(a) {}
    const char* type() const override { return "GetValueByIndexPlugin"; }
    int nbOutputs() const override { return 3; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i < 0 || i > 2) return -1;
        *out = dims4(in[0].d[0], max_win_num_, voxel_num_set_, channel_num_); return 0;
    }
    int outputType(int, const int32_t* t, int) const override { return t[0]; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        if (pos == 2 || pos == 3) return i32Linear(io[pos]);
        return pos >= 0 && pos <= 6 && f32Linear(io[pos]);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        size_t bytes = sizeof(float) * (size_t)max_win_num_ * voxel_num_set_ * channel_num_;
        if (zeroFill) for (int i = 0; i < 3; ++i) DSVT_CHECK(hipMemsetAsync(out[i], 0, bytes, stream));       // :347-349
        const uint32_t* inds = static_cast<const uint32_t*>(in[2]) + (size_t)axis_id_ * max_win_num_ * voxel_num_set_;   // :292
        hipLaunchKernelGGL(get_value_kernel, dim3(2048), dim3(256), 0, stream, static_cast<const float4*>(in[0]),
                           static_cast<const float4*>(in[1]), inds, static_cast<const uint32_t*>(in[3]), voxel_num_set_,
                           channel_num_ / 4, static_cast<float4*>(out[0]), static_cast<float4*>(out[1]), static_cast<float4*>(out[2]));
        return lastError();
    }
    size_t serializationSize() const override { return 4 * sizeof(int); }
    void serialize(void* b) const override {                                          // getValueByIndex.cu:112-119
        char* d = static_cast<char*>(b); wr<int>(d, voxel_num_set_); wr<int>(d, max_win_num_); wr<int>(d, channel_num_); wr<int>(d, axis_id_);
    }
    Plugin* clone() const override { return new GetValueByIndexPlugin(max_win_num_, voxel_num_set_, channel_num_, axis_id_); }
};
static Plugin* gvNew(int mw, int l, int c, int a) {
    return (mw > 0 && l > 0 && c > 0 && c % 4 == 0 && (a == 0 || a == 1)) ? new GetValueByIndexPlugin(mw, l, c, a) : nullptr;
}
static Plugin* gvCreate(const DsvtPluginFieldCollection* fc) {
    return gvNew(fieldInt(fc, "max_win_num"), fieldInt(fc, "voxel_num_set"), fieldInt(fc, "channel_num"), fieldInt(fc, "axis_id"));
}
static Plugin* gvDeser(const void* data, size_t len) {
    if (len < 4 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    int l = rd<int>(d), mw = rd<int>(d), c = rd<int>(d), a = rd<int>(d);
    return gvNew(mw, l, c, a);
}
static Creator g_gvCreator{"GetValueByIndexPlugin",
    {{"max_win_num", DSVT_FIELD_INT32}, {"voxel_num_set", DSVT_FIELD_INT32}, {"channel_num", DSVT_FIELD_INT32}, {"axis_id", DSVT_FIELD_INT32}},
    gvCreate, gvDeser, {}, {}};
static Registrar g_gvReg(&g_gvCreator);

__global__ void __launch_bounds__(256)
map_set_kernel(const float4* __restrict__ set_feat, const uint32_t* __restrict__ inds, const uint32_t* __restrict__ set_num,
               int L, int G, float4* __restrict__ out)
{
    size_t total = (size_t)(*set_num) * L * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t row = i / G; int c = (int)(i % G);
        // a voxel that fills several slots of a set repeats its index in consecutive slots (getSet.cu:346);
        // the reference lets every slot write (benign race, rows are equal up to which slot wins).
        // Here only the LAST slot of each run writes, which is what serial execution of the
        // reference produces (mapSetFeature2voxel.cu:271-273).
        int slot = (int)(row % L);
        if (slot + 1 < L && inds[row + 1] == inds[row]) continue;
        out[(size_t)inds[row] * G + c] = set_feat[i];
    }
}

class MapSetFeature2VoxelPlugin : public Plugin {
public:
    int max_win_num_, voxel_num_set_, channel_num_, axis_id_, max_pillars_num_;
    MapSetFeature2VoxelPlugin(int mw, int l, int c, int a, int mp) : max_win_num_(mw), voxel_num_set_(l), channel_num_(c), axis_id_(a), max_pillars_num_(mp) {}
    const char* type() const override { return "MapSetFeature2VoxelPlugin"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i != 0) return -1;
        *out = dims3(in[0].d[0], max_pillars_num_, channel_num_); return 0;
    }
    int outputType(int, const int32_t* t, int) const override { return t[0]; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        if (pos == 1 || pos == 2) return i32Linear(io[pos]);
        return (pos == 0 || pos == 3) && f32Linear(io[pos]);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (zeroFill) DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(float) * (size_t)max_pillars_num_ * channel_num_, stream));   // :314
        const uint32_t* inds = static_cast<const uint32_t*>(in[1]) + (size_t)axis_id_ * max_win_num_ * voxel_num_set_;          // :265
        hipLaunchKernelGGL(map_set_kernel, dim3(2048), dim3(256), 0, stream, static_cast<const float4*>(in[0]), inds,
                           static_cast<const uint32_t*>(in[2]), voxel_num_set_, channel_num_ / 4, static_cast<float4*>(out[0]));
        return lastError();
    }
    size_t serializationSize() const override { return 5 * sizeof(int); }
    void serialize(void* b) const override {                                          // mapSetFeature2voxel.cu:112-120
        char* d = static_cast<char*>(b);
        wr<int>(d, voxel_num_set_); wr<int>(d, max_win_num_); wr<int>(d, channel_num_); wr<int>(d, max_pillars_num_); wr<int>(d, axis_id_);
    }
    Plugin* clone() const override { return new MapSetFeature2VoxelPlugin(max_win_num_, voxel_num_set_, channel_num_, axis_id_, max_pillars_num_); }
};
static Plugin* msNew(int mw, int l, int c, int a, int mp) {
    return (mw > 0 && l > 0 && c > 0 && c % 4 == 0 && (a == 0 || a == 1) && mp > 0) ? new MapSetFeature2VoxelPlugin(mw, l, c, a, mp) : nullptr;
}
static Plugin* msCreate(const DsvtPluginFieldCollection* fc) {
    return msNew(fieldInt(fc, "max_win_num"), fieldInt(fc, "voxel_num_set"), fieldInt(fc, "channel_num"), fieldInt(fc, "axis_id"),
                 fieldInt(fc, "max_pillars_num"));
}
static Plugin* msDeser(const void* data, size_t len) {
    if (len < 5 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    int l = rd<int>(d), mw = rd<int>(d), c = rd<int>(d), mp = rd<int>(d), a = rd<int>(d);
    return msNew(mw, l, c, a, mp);
}
static Creator g_msCreator{"MapSetFeature2VoxelPlugin",
    {{"max_win_num", DSVT_FIELD_INT32}, {"voxel_num_set", DSVT_FIELD_INT32}, {"channel_num", DSVT_FIELD_INT32},
     {"axis_id", DSVT_FIELD_INT32}, {"max_pillars_num", DSVT_FIELD_INT32}},           // :393-397
    msCreate, msDeser, {}, {}};
static Registrar g_msReg(&g_msCreator);

// =====================================================================================
// LayerNorm
// =====================================================================================
// one wavefront per row; a lane owns channels lane*4 .. lane*4+3 (+256 per extra trip)
template <int TRIPS>
__global__ void __launch_bounds__(256)
layer_norm_kernel(const float* __restrict__ x, const uint32_t* __restrict__ voxel_num, int C, float eps,
                  const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y)
{
    const int lane = laneId();
    uint32_t row = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (row >= *voxel_num) return;
    const float* xr = x + (size_t)row * C;
    float4 v[TRIPS];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        int c = (t * kWave + lane) * 4;
        v[t] = c < C ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
    }
    float mean = waveSum(s) / C;                                                     // layerNorm.cu:304-308
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        int c = (t * kWave + lane) * 4;
        if (c < C) {
            float a = v[t].x - mean, b = v[t].y - mean, d = v[t].z - mean, e = v[t].w - mean;
            q += (a * a + b * b) + (d * d + e * e);
        }
    }
    float var = waveSum(q) / C;                                                      // :333-337 (biased)
    float den = sqrtf(var + eps);                                                    // :274
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        int c = (t * kWave + lane) * 4;
        if (c < C) {
            float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c), o;
            o.x = (v[t].x - mean) / den * g.x + b.x;                                 // :274-276
            o.y = (v[t].y - mean) / den * g.y + b.y;
            o.z = (v[t].z - mean) / den * g.z + b.z;
            o.w = (v[t].w - mean) / den * g.w + b.w;
            *reinterpret_cast<float4*>(y + (size_t)row * C + c) = o;
        }
    }
}

class LayerNormPlugin : public Plugin {
public:
    int max_pillars_num_, channel_num_, weights_size_; float eps_;
    std::vector<float> gamma_, beta_;
    float *gamma_dev_ = nullptr, *beta_dev_ = nullptr;
    LayerNormPlugin(int mp, int c, int ws, float eps, const float* g, const float* b)
        : max_pillars_num_(mp), channel_num_(c), weights_size_(ws), eps_(eps), gamma_(g, g + ws), beta_(b, b + ws) {
        // the plugin owns its device weights (layerNorm.cu:150-155)
        if (dsvtMalloc(&gamma_dev_, sizeof(float) * ws) != hipSuccess || dsvtMalloc(&beta_dev_, sizeof(float) * ws) != hipSuccess) {
            gamma_dev_ = beta_dev_ = nullptr; return;
        }
        (void)hipMemcpy(gamma_dev_, gamma_.data(), sizeof(float) * ws, hipMemcpyHostToDevice);
        (void)hipMemcpy(beta_dev_, beta_.data(), sizeof(float) * ws, hipMemcpyHostToDevice);
    }
    ~LayerNormPlugin() override { if (gamma_dev_) (void)dsvtFree(gamma_dev_); if (beta_dev_) (void)dsvtFree(beta_dev_); }    // :432-444
    const char* type() const override { return "LayerNormPlugin"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i != 0) return -1;
        *out = dims3(in[0].d[0], max_pillars_num_, channel_num_); return 0;
    }
    int outputType(int, const int32_t* t, int) const override { return t[0]; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {                     // :218-236
        return pos == 1 ? i32Linear(io[pos]) : (pos == 0 || pos == 2) && f32Linear(io[pos]);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (!gamma_dev_) return static_cast<int>(hipErrorOutOfMemory);
        if (zeroFill) DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(float) * (size_t)max_pillars_num_ * channel_num_, stream));   // :395
        dim3 grid(cdiv(max_pillars_num_, 4)), block(256);
        const float* x = static_cast<const float*>(in[0]);
        const uint32_t* n = static_cast<const uint32_t*>(in[1]);
        float* y = static_cast<float*>(out[0]);
        int trips = cdiv(channel_num_, 256);
        if (trips == 1) hipLaunchKernelGGL(layer_norm_kernel<1>, grid, block, 0, stream, x, n, channel_num_, eps_, gamma_dev_, beta_dev_, y);
        else if (trips == 2) hipLaunchKernelGGL(layer_norm_kernel<2>, grid, block, 0, stream, x, n, channel_num_, eps_, gamma_dev_, beta_dev_, y);
        else hipLaunchKernelGGL(layer_norm_kernel<4>, grid, block, 0, stream, x, n, channel_num_, eps_, gamma_dev_, beta_dev_, y);
        return lastError();
    }
    size_t serializationSize() const override { return 3 * sizeof(int) + sizeof(float) + 2 * sizeof(float) * weights_size_; }
    void serialize(void* b) const override {                                          // :446-470
        char* d = static_cast<char*>(b);
        wr<int>(d, max_pillars_num_); wr<int>(d, channel_num_); wr<int>(d, weights_size_); wr<float>(d, eps_);
        for (int i = 0; i < weights_size_; ++i) wr<float>(d, gamma_[i]);
        for (int i = 0; i < weights_size_; ++i) wr<float>(d, beta_[i]);
    }
    Plugin* clone() const override { return new LayerNormPlugin(max_pillars_num_, channel_num_, weights_size_, eps_, gamma_.data(), beta_.data()); }
};
static Plugin* lnNew(int mp, int c, int ws, float eps, const float* g, const float* b) {
    return (mp > 0 && c > 0 && c % 4 == 0 && c <= 1024 && ws == c && g && b) ? new LayerNormPlugin(mp, c, ws, eps, g, b) : nullptr;
}
static Plugin* lnCreate(const DsvtPluginFieldCollection* fc) {
    // The reference creator advertises the field as "pes" (layerNorm.cu:497) while its
    // createPlugin reads "eps" (:558) and its factory only forwards names the creator
    // advertises (plugin_helper.h:527) -- so eps never arrives and stays 0.  Same here:
    // "pes" is advertised, "eps" is honoured if a caller does pass it.
    const int ws = fieldInt(fc, "weights_size");
    return lnNew(fieldInt(fc, "max_pillars_num"), fieldInt(fc, "channel_num"), ws,
                 fieldFloat(fc, "eps", 0.0f), fieldFloatArray(fc, "weights", ws), fieldFloatArray(fc, "bias", ws));
}
static Plugin* lnDeser(const void* data, size_t len) {
    if (len < 3 * sizeof(int) + sizeof(float)) return nullptr;
    const char* d = static_cast<const char*>(data);
    int mp = rd<int>(d), c = rd<int>(d), ws = rd<int>(d); float eps = rd<float>(d);
    if (ws <= 0 || len < 3 * sizeof(int) + sizeof(float) + 2 * sizeof(float) * (size_t)ws) return nullptr;
    std::vector<float> g(ws), b(ws);
    memcpy(g.data(), d, sizeof(float) * ws); memcpy(b.data(), d + sizeof(float) * ws, sizeof(float) * ws);
    return lnNew(mp, c, ws, eps, g.data(), b.data());
}
static Creator g_lnCreator{"LayerNormPlugin",
    {{"max_pillars_num", DSVT_FIELD_INT32}, {"channel_num", DSVT_FIELD_INT32}, {"weights_size", DSVT_FIELD_INT32},
     {"pes", DSVT_FIELD_FLOAT32}, {"weights", DSVT_FIELD_FLOAT32}, {"bias", DSVT_FIELD_FLOAT32}},                // :494-500
    lnCreate, lnDeser, {}, {}};
static Registrar g_lnReg(&g_lnCreator);

// =====================================================================================
// GeLU
// =====================================================================================
__device__ __forceinline__ float geluRef(float x) {
    // gelu.cu:208-209 with the params.h:75-77 macros: double literals => double arithmetic
    const double A = 0.5, B = 0.7978845608028654, Cc = 0.035677408136300125;
    return (float)((A + A * tanh(x * (Cc * x * x + B))) * x);
}
__global__ void __launch_bounds__(256)
gelu_kernel(const float4* __restrict__ x, const uint32_t* __restrict__ voxel_num, int G, float4* __restrict__ y)
{
    size_t total = (size_t)(*voxel_num) * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = x[i];
        y[i] = make_float4(geluRef(v.x), geluRef(v.y), geluRef(v.z), geluRef(v.w));
    }
}
class GeluPlugin : public Plugin {
public:
    int max_pillars_num_, channel_num_;
    GeluPlugin(int mp, int c) : max_pillars_num_(mp), channel_num_(c) {}
    const char* type() const override { return "GeluPlugin"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i != 0) return -1;
        *out = dims3(in[0].d[0], max_pillars_num_, channel_num_); return 0;
    }
    int outputType(int, const int32_t* t, int) const override { return t[0]; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        return pos == 1 ? i32Linear(io[pos]) : (pos == 0 || pos == 2) && f32Linear(io[pos]);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (zeroFill) DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(float) * (size_t)max_pillars_num_ * channel_num_, stream));   // :245
        hipLaunchKernelGGL(gelu_kernel, dim3(2048), dim3(256), 0, stream, static_cast<const float4*>(in[0]),
                           static_cast<const uint32_t*>(in[1]), channel_num_ / 4, static_cast<float4*>(out[0]));
        return lastError();
    }
    size_t serializationSize() const override { return 2 * sizeof(int); }
    void serialize(void* b) const override { char* d = static_cast<char*>(b); wr<int>(d, max_pillars_num_); wr<int>(d, channel_num_); }
    Plugin* clone() const override { return new GeluPlugin(max_pillars_num_, channel_num_); }
};
static Plugin* geNew(int mp, int c) { return (mp > 0 && c > 0 && c % 4 == 0) ? new GeluPlugin(mp, c) : nullptr; }
static Plugin* geCreate(const DsvtPluginFieldCollection* fc) { return geNew(fieldInt(fc, "max_pillars_num"), fieldInt(fc, "channel_num")); }
static Plugin* geDeser(const void* data, size_t len) {
    if (len < 2 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data); int mp = rd<int>(d), c = rd<int>(d); return geNew(mp, c);
}
static Creator g_geCreator{"GeluPlugin", {{"max_pillars_num", DSVT_FIELD_INT32}, {"channel_num", DSVT_FIELD_INT32}}, geCreate, geDeser, {}, {}};
static Registrar g_geReg(&g_geCreator);

// =====================================================================================
// Map2Bev
// =====================================================================================
__global__ void __launch_bounds__(256)
map2bev_kernel(const float4* __restrict__ feat, const uint4* __restrict__ coords, const uint32_t* __restrict__ voxel_num,
               int G, int gx, int gy, int frames, float4* __restrict__ bev, uint4* __restrict__ save_coords, unsigned long long* __restrict__ save_state)
{
    size_t total = (size_t)(*voxel_num) * G;
    if (save_state && blockIdx.x == 0 && threadIdx.x == 0) {                          // (persistent_output: which cells of WHICH buffer the next call may clear)
        save_state[0] = *voxel_num; save_state[1] = reinterpret_cast<unsigned long long>(bev);
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i / G; int c = (int)(i % G);
        uint4 co = coords[p];                                                        // map2bev.cu:259-261: y = .z, x = .w
        if (save_coords && c == 0) save_coords[p] = co;
        if (co.x >= (uint32_t)frames) continue;                                      // (.x = frame index of a multi-frame voxelizer, 0 otherwise)
        bev[(((size_t)co.x * gy + co.z) * gx + co.w) * G + c] = feat[i];             // :264
    }
}
// split precision (round 3): fp32 rows in, fp16 [hi | lo | hi] planes out (3 C channels per cell: the operand layout of the fp32-grade
// convolutions, see conv.hip ConvArgs::split_out)
typedef _Float16 mb_half4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
map2bev_split_kernel(const float4* __restrict__ feat, const uint4* __restrict__ coords, const uint32_t* __restrict__ voxel_num,
                     int G, int gx, int gy, int frames, mb_half4* __restrict__ bev, int x8, uint4* __restrict__ save_coords, unsigned long long* __restrict__ save_state)
{
    size_t total = (size_t)(*voxel_num) * G;
    if (save_state && blockIdx.x == 0 && threadIdx.x == 0) { save_state[0] = *voxel_num; save_state[1] = reinterpret_cast<unsigned long long>(bev); }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i / G; int c = (int)(i % G);
        uint4 co = coords[p];
        if (save_coords && c == 0) save_coords[p] = co;
        if (co.x >= (uint32_t)frames) continue;
        const float4 v = feat[i];
        const float f[4] = {v.x, v.y, v.z, v.w};
        mb_half4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hi[k] = (_Float16)fminf(fmaxf(f[k], -65504.f), 65504.f);
            lo[k] = (_Float16)fminf(fmaxf(f[k] - (float)hi[k], -65504.f), 65504.f);
        }
        mb_half4* cell = bev + (((size_t)co.x * gy + co.z) * gx + co.w) * 3 * G;
        mb_half4* o = cell + c;
        o[0] = hi; o[G] = lo;
        if (x8 > 0) {       // third plane = the fp8 operands of the correction terms (conv.hip ConvArgs::x8_out): lo8 = e4m3(2^11 lo), hi8 = e4m3(v)
            unsigned char* xp = reinterpret_cast<unsigned char*>(cell + 2 * G) + x8Offset(4 * c);
            *reinterpret_cast<unsigned*>(xp) = packE4m3((f[0] - (float)hi[0]) * 2048.f, (f[1] - (float)hi[1]) * 2048.f, (f[2] - (float)hi[2]) * 2048.f, (f[3] - (float)hi[3]) * 2048.f);
            *reinterpret_cast<unsigned*>(xp + 16) = packE4m3(f[0], f[1], f[2], f[3]);
        } else if (x8 == 0) o[2 * G] = hi;          // (x8 < 0: [hi | lo | -], the third plane is left alone -- its readers alias plane 0: split_output 3)
    }
}
// persistent_output (round 4): the dense map is zero everywhere but at <= P cells, so when the caller keeps the SAME output buffer from call to call (this
// pipeline's buffers are static) a call needs to zero only the cells the call before it wrote -- 40 MB per frame instead of the 126 / 252 / 504 MB fill
// (fp16 / fp32 / triple map), which at four frames per launch was 133 of the plugin's 201 us.  Contract of the opt-in field: between two calls nobody
// but this plugin writes the buffer.  Whether the incremental clear applies is decided ON THE DEVICE, at execution time (round 5; until then the host
// compared the output address at ENQUEUE time, which is wrong as soon as enqueue order and execution order differ -- a call recorded under stream capture
// but not run yet, then an eager call: the eager call would have cleared "the previous cells" of a map nobody ever zeroed): the scatter kernel leaves
// {cell count, address of the map it filled} in `state`, and the clear kernel takes the short path only if that address is the one it was handed;
// otherwise (first call, another buffer) it zeroes the whole map itself.
__global__ void __launch_bounds__(256)
map2bev_clear_kernel(const uint4* __restrict__ prev_coords, const unsigned long long* __restrict__ state, int CH, int gx, int gy, int frames,
                     uint4* __restrict__ bev, size_t whole)                          // CH = 16-byte chunks per cell, whole = 16-byte chunks of the map
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    if (state[1] != reinterpret_cast<unsigned long long>(bev)) {                      // (uniform over the grid)
        for (size_t i = tid; i < whole; i += nth) bev[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const size_t total = (size_t)state[0] * CH;
    for (size_t i = tid; i < total; i += nth) {
        const size_t p = i / CH; const int c = (int)(i % CH);
        const uint4 co = prev_coords[p];
        if (co.x >= (uint32_t)frames) continue;
        bev[(((size_t)co.x * gy + co.z) * gx + co.w) * CH + c] = make_uint4(0u, 0u, 0u, 0u);
    }
}
class Map2BevPlugin : public Plugin {
public:
    int max_pillars_num_, channel_num_, gx_, gy_, frames_ = 1;      // frames_ > 1 (field "frames"): coords.x selects one of `frames` stacked BEV maps
    int split_ = 0;                                                 // field "split_output": fp32 rows -> fp16 [hi | lo | hi] planes (1), [hi | lo | x8] (2) or [hi | lo | -] (3: the third plane is not written), 3 C channels per cell
    int persistent_ = 0;                                            // field "persistent_output": see map2bev_clear_kernel
    uint4* prev_coords_ = nullptr; unsigned long long* prev_state_ = nullptr;      // device: the cells of the last EXECUTED call, {count, address of its map}
    Map2BevPlugin(int mp, int c, int gx, int gy, int frames = 1, int split = 0, int persistent = 0)
        : max_pillars_num_(mp), channel_num_(c), gx_(gx), gy_(gy), frames_(frames), split_(split), persistent_(persistent) {
        if (persistent_) {
            if (dsvtMalloc(&prev_coords_, sizeof(uint4) * (size_t)mp) != hipSuccess || dsvtMalloc(&prev_state_, 2 * sizeof(unsigned long long)) != hipSuccess ||
                hipMemset(prev_state_, 0, 2 * sizeof(unsigned long long)) != hipSuccess) { persistent_ = 0; }
        }
    }
    ~Map2BevPlugin() override { if (prev_coords_) (void)dsvtFree(prev_coords_); if (prev_state_) (void)dsvtFree(prev_state_); }
    // zero the map: everything, or (persistent_output, and the device state says this buffer holds the previous call's cells and nothing else) those cells
    int clearMap(void* out, size_t bytes, int cellBytes, hipStream_t stream) {
        if (persistent_ && cellBytes % 16 == 0 && bytes % 16 == 0) {
            hipLaunchKernelGGL(map2bev_clear_kernel, dim3(2048), dim3(256), 0, stream, prev_coords_, prev_state_, cellBytes / 16, gx_, gy_, frames_,
                               static_cast<uint4*>(out), bytes / 16);
            return lastError();
        }
        DSVT_CHECK(hipMemsetAsync(out, 0, bytes, stream));
        return 0;
    }
    // (the scatter kernels record their cells only when the clear kernel can use them)
    bool saves(int cellBytes) const { return persistent_ && cellBytes % 16 == 0; }
    const char* type() const override { return "Map2BevPlugin"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i != 0) return -1;
        *out = dims4(frames_ > 1 ? frames_ : in[0].d[0], gx_, gy_, (split_ ? 3 : 1) * channel_num_); return 0;      // map2bev.cu: [1, gx, gy, C] (used as [y][x][C])
    }
    int outputType(int, const int32_t* t, int) const override { return split_ ? DSVT_HALF : t[0]; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        if (pos == 1 || pos == 2) return i32Linear(io[pos]);
        if (split_) return (pos == 0 || pos == 3) && io[pos].format == DSVT_FORMAT_LINEAR && io[pos].type == (pos == 0 ? DSVT_FLOAT : DSVT_HALF);
        // fp32 like the reference, or fp16 rows (pure 16-byte moves either way)
        return (pos == 0 || pos == 3) && io[pos].format == DSVT_FORMAT_LINEAR && (io[pos].type == DSVT_FLOAT || io[pos].type == DSVT_HALF)
               && io[pos].type == io[0].type;
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (split_) {
            if (int rc = clearMap(out[0], (size_t)2 * gx_ * gy_ * 3 * channel_num_ * frames_, 2 * 3 * channel_num_, stream)) return rc;
            hipLaunchKernelGGL(map2bev_split_kernel, dim3(2048), dim3(256), 0, stream, static_cast<const float4*>(in[0]),
                               static_cast<const uint4*>(in[1]), static_cast<const uint32_t*>(in[2]), channel_num_ / 4, gx_, gy_, frames_,
                               static_cast<mb_half4*>(out[0]), split_ == 2 ? 1 : split_ == 3 ? -1 : 0, saves(6 * channel_num_) ? prev_coords_ : nullptr, saves(6 * channel_num_) ? prev_state_ : nullptr);
            return lastError();
        }
        const int esz = (inDesc && inDesc[0].type == DSVT_HALF) ? 2 : 4;
        if ((channel_num_ * esz) % 16 != 0) return -3;
        // the dense map must be zero wherever no pillar lands, so this fill is not optional (:303)
        if (int rc = clearMap(out[0], (size_t)esz * gx_ * gy_ * channel_num_ * frames_, esz * channel_num_, stream)) return rc;
        hipLaunchKernelGGL(map2bev_kernel, dim3(2048), dim3(256), 0, stream, static_cast<const float4*>(in[0]),
                           static_cast<const uint4*>(in[1]), static_cast<const uint32_t*>(in[2]), channel_num_ * esz / 16, gx_, gy_, frames_,
                           static_cast<float4*>(out[0]), persistent_ ? prev_coords_ : nullptr, persistent_ ? prev_state_ : nullptr);
        return lastError();
    }
    // trailing ints: [frames [split [persistent]]], each present when it or a later one is not the default
    int nTrail() const { return persistent_ ? 3 : split_ ? 2 : frames_ > 1 ? 1 : 0; }
    size_t serializationSize() const override { return (4 + nTrail()) * sizeof(int); }
    void serialize(void* b) const override {
        char* d = static_cast<char*>(b); wr<int>(d, max_pillars_num_); wr<int>(d, channel_num_); wr<int>(d, gx_); wr<int>(d, gy_);
        if (nTrail() >= 1) wr<int>(d, frames_);
        if (nTrail() >= 2) wr<int>(d, split_);
        if (nTrail() >= 3) wr<int>(d, persistent_);
    }
    Plugin* clone() const override { return new Map2BevPlugin(max_pillars_num_, channel_num_, gx_, gy_, frames_, split_, persistent_); }
};
static Plugin* mbNew(int mp, int c, int gx, int gy, int frames = 1, int split = 0, int persistent = 0) {
    if (split < 0 || split > 3 || (split == 2 && c % 32 != 0)) return nullptr;                 // 2: [hi | lo | x8], the x8 plane in 32-channel groups; 3: [hi | lo | -] (third plane untouched)
    return (mp > 0 && c > 0 && c % 4 == 0 && gx > 0 && gy > 0 && frames >= 1) ? new Map2BevPlugin(mp, c, gx, gy, frames, split, persistent != 0) : nullptr;
}
static Plugin* mbCreate(const DsvtPluginFieldCollection* fc) {
    return mbNew(fieldInt(fc, "max_pillars_num"), fieldInt(fc, "channel_num"), fieldInt(fc, "grid_size_x"), fieldInt(fc, "grid_size_y"), fieldInt(fc, "frames", 1),
                 fieldInt(fc, "split_output", 0), fieldInt(fc, "persistent_output", 0));
}
static Plugin* mbDeser(const void* data, size_t len) {
    const int extra = trailingInts(len, 4 * sizeof(int), 3);
    if (extra < 0) return nullptr;
    const char* d = static_cast<const char*>(data); int mp = rd<int>(d), c = rd<int>(d), gx = rd<int>(d), gy = rd<int>(d);
    const int frames = extra >= 1 ? rd<int>(d) : 1, split = extra >= 2 ? rd<int>(d) : 0, persistent = extra >= 3 ? rd<int>(d) : 0;
    if (extra >= 3 && persistent != 1) return nullptr;               // (a seventh int is only ever written as 1: a zero-padded six-int blob is not a seven-int one)
    return mbNew(mp, c, gx, gy, frames, split, persistent);
}
static Creator g_mbCreator{"Map2BevPlugin",
    {{"max_pillars_num", DSVT_FIELD_INT32}, {"channel_num", DSVT_FIELD_INT32}, {"grid_size_x", DSVT_FIELD_INT32}, {"grid_size_y", DSVT_FIELD_INT32}},
    mbCreate, mbDeser, {}, {}};
static Registrar g_mbReg(&g_mbCreator);

// =====================================================================================
// FilterBoxByScore
// =====================================================================================
struct FBParams { int max_top_k; float min_x, max_x, min_y, max_y, min_z, max_z, vx, vy, vz, thr; };

// One workgroup; candidates are visited in rank order 64 at a time and survivors are
// compacted with a wave64 ballot, so row order = candidate rank (the reference's order is
// whatever atomicAdd produces, filterBoxByScore.cu:295).  Also fixes the reference's
// out-of-bounds tail (512 threads for 500 rows, :272-273, :319).
__global__ void __launch_bounds__(64)
filter_box_kernel(const float* __restrict__ scores, const uint32_t* __restrict__ classes, const uint32_t* __restrict__ xs,
                  const uint32_t* __restrict__ ys, const float* __restrict__ center, const float* __restrict__ center_z,
                  const float* __restrict__ angle, const float* __restrict__ dim, FBParams p, bool zero_fill,
                  float* __restrict__ out, uint32_t* __restrict__ valid_num)
{
    const int lane = threadIdx.x;
    {   // blockIdx.x = frame of a stack of frames (every tensor has K rows per frame)
        const size_t b = blockIdx.x, K = (size_t)p.max_top_k;
        scores += b * K; classes += b * K; xs += b * K; ys += b * K; center += b * K * 2; center_z += b * K; angle += b * K; dim += b * K * 3;
        out += b * K * 9; valid_num += b;
    }
    uint32_t base = 0;
    for (int i0 = 0; i0 < p.max_top_k; i0 += kWave) {
        int i = i0 + lane;
        bool keep = false;
        float nx = 0, ny = 0, cz = 0, sc = 0;
        if (i < p.max_top_k) {
            sc = scores[i];
            nx = (float)xs[i] + center[i * 2 + 0];                                   // :278-279
            ny = (float)ys[i] + center[i * 2 + 1];
            nx = nx * p.vx + p.min_x;                                                // :280-281
            ny = ny * p.vy + p.min_y;
            cz = center_z[i];
            keep = (nx >= p.min_x && nx < p.max_x && ny >= p.min_y && ny < p.max_y && cz >= p.min_z && cz < p.max_z)   // :287-291
                   && sc >= p.thr;                                                   // :293
        }
        unsigned long long m = __ballot(keep);
        if (keep) {
            uint32_t r = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            float* o = out + (size_t)r * 9;
            o[0] = nx; o[1] = ny; o[2] = cz;                                         // :297-306
            o[3] = dim[i * 3 + 0]; o[4] = dim[i * 3 + 1]; o[5] = dim[i * 3 + 2];
            o[6] = angle[i]; o[7] = (float)classes[i]; o[8] = sc;
        }
        base += (uint32_t)__popcll(m);
    }
    if (zero_fill)
        for (int j = (int)base * 9 + lane; j < p.max_top_k * 9; j += kWave) out[j] = 0.f;   // :361
    if (lane == 0) *valid_num = base;
}

class FilterBoxByScorePlugin : public Plugin {
public:
    FBParams p_;
    explicit FilterBoxByScorePlugin(const FBParams& p) : p_(p) {}
    const char* type() const override { return "FilterBoxByScorePlugin"; }
    bool handlesBatch() const override { return true; }              // one wavefront per frame of a stack
    int nbOutputs() const override { return 2; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i == 0) { *out = dims3(in[0].d[0], p_.max_top_k, 9); return 0; }         // LAST_DIMS = 9
        if (i == 1) { *out = dims1(in[0].d[0]); return 0; }
        return -1;
    }
    int outputType(int i, const int32_t*, int) const override { return i == 0 ? DSVT_FLOAT : DSVT_INT32; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        if (pos == 1 || pos == 2 || pos == 3 || pos == 9) return i32Linear(io[pos]);
        return pos >= 0 && pos <= 8 && f32Linear(io[pos]);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        const int nb = (inDesc && inDesc[0].dims.nbDims >= 1 && inDesc[0].dims.d[0] > 1) ? inDesc[0].dims.d[0] : 1;
        hipLaunchKernelGGL(filter_box_kernel, dim3(nb), dim3(64), 0, stream, static_cast<const float*>(in[0]),
                           static_cast<const uint32_t*>(in[1]), static_cast<const uint32_t*>(in[2]), static_cast<const uint32_t*>(in[3]),
                           static_cast<const float*>(in[4]), static_cast<const float*>(in[5]), static_cast<const float*>(in[6]),
                           static_cast<const float*>(in[7]), p_, true, static_cast<float*>(out[0]), static_cast<uint32_t*>(out[1]));
        return lastError();
    }
    size_t serializationSize() const override { return sizeof(int) + 10 * sizeof(float); }
    void serialize(void* b) const override {                                          // filterBoxByScore.cu:420-435
        char* d = static_cast<char*>(b);
        wr<int>(d, p_.max_top_k); wr<float>(d, p_.min_x); wr<float>(d, p_.max_x); wr<float>(d, p_.min_y); wr<float>(d, p_.max_y);
        wr<float>(d, p_.min_z); wr<float>(d, p_.max_z); wr<float>(d, p_.vx); wr<float>(d, p_.vy); wr<float>(d, p_.vz); wr<float>(d, p_.thr);
    }
    Plugin* clone() const override { return new FilterBoxByScorePlugin(p_); }
};
static Plugin* fbCreate(const DsvtPluginFieldCollection* fc) {
    FBParams p{};
    float r[6], v[3];
    p.max_top_k = fieldInt(fc, "max_top_k");
    fieldFloats(fc, "point_cloud_range", r, 6);   // (xmin,xmax,ymin,ymax,zmin,zmax): plugin_helper.h:627-632 -- NOT the Points2Features order
    fieldFloats(fc, "voxel_size", v, 3);
    p.min_x = r[0]; p.max_x = r[1]; p.min_y = r[2]; p.max_y = r[3]; p.min_z = r[4]; p.max_z = r[5];
    p.vx = v[0]; p.vy = v[1]; p.vz = v[2]; p.thr = fieldFloat(fc, "score_threshold");
    return p.max_top_k > 0 ? new FilterBoxByScorePlugin(p) : nullptr;
}
static Plugin* fbDeser(const void* data, size_t len) {
    if (len < sizeof(int) + 10 * sizeof(float)) return nullptr;
    const char* d = static_cast<const char*>(data);
    FBParams p{};
    p.max_top_k = rd<int>(d); p.min_x = rd<float>(d); p.max_x = rd<float>(d); p.min_y = rd<float>(d); p.max_y = rd<float>(d);
    p.min_z = rd<float>(d); p.max_z = rd<float>(d); p.vx = rd<float>(d); p.vy = rd<float>(d); p.vz = rd<float>(d); p.thr = rd<float>(d);
    return p.max_top_k > 0 ? new FilterBoxByScorePlugin(p) : nullptr;
}
static Creator g_fbCreator{"FilterBoxByScorePlugin",
    {{"max_top_k", DSVT_FIELD_INT32}, {"point_cloud_range", DSVT_FIELD_FLOAT32}, {"voxel_size", DSVT_FIELD_FLOAT32},
     {"score_threshold", DSVT_FIELD_FLOAT32}},                                        // :459-462
    fbCreate, fbDeser, {}, {}};
static Registrar g_fbReg(&g_fbCreator);

}  // namespace dsvt
