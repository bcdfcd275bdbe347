// voxel_pool.hip -- stage reduction of a true 3-D voxel DSVT (BASELINE configs[4]).
//
// The reference has no voxel path at all (its voxel z index is forced to 0: plugins/src/points2Features.cu:689-690,755), so nothing here restates
// reference code; the spec is upstream DSVT's multi-stage 3-D backbone, whose stages are joined by an attention-style pooling
// ("Stage_ReductionAtt_Block"): the voxels of stage s are grouped by  coords // stride  (stride (1, 1, 4) on the 468 x 468 x 32 grid: up to four
// z-neighbours per pooled voxel), and every pooled voxel becomes
//     x_j   = its children's rows in the slots  j = (x % sx) sy sz + (y % sy) sz + (z % sz), zeros in the empty slots
//     src   = max_j x_j                                  (MaxPool1d over ALL pool_volume slots, the zero rows of empty slots included)
//     out   = LayerNorm(src + MultiheadAttention(query = src, key = x + pos_embedding, value = x, key_padding_mask = empty slots))
// PARITY UNPINNED (oracle/dense_ref.py stage_reduction_att restates the same published semantics; there is no reference output to compare with).
// Three plugins, composed by pipeline3d.py with four DsvtLinearPlugin launches (Q, K, V projections; out-proj + residual + LayerNorm epilogue):
//   DsvtVoxelPoolPlugin          integer: pooled coordinates in canonical (ascending pooled cell key) order, the [P2, pool_volume] child table,
//                                every voxel's pooled row, the counts P2 and P2 x pool_volume -- an occupancy bitmap of the pooled grid, one
//                                single-workgroup popcount scan, one ranking pass: no sort, no dense table of rows (219 KB of bitmap at 468 x 468 x 8)
//   DsvtPoolGatherPlugin         x [P, C], child table -> src [P2, C] (the max query) and the key input [P, C] = x + pos_embedding[slot of the voxel], per INPUT voxel:
//                                upstream materialises a [P2, pool_volume, C] tensor and projects all of it; the empty slots' keys and values are masked out of the
//                                softmax anyway, so K and V are projected for the P real voxels only (a third of the rows at stride (1, 1, 4)) -- the value input is x itself
//   DsvtPoolAttentionCorePlugin  q [P2, C], k, v [P, C], child table -> softmax(q k^T over the pool's children) v per head: one wavefront per pooled voxel
#include "plugin_base.h"
#include "device_utils.h"

namespace dsvt {

static bool vpF32(const DsvtPluginTensorDesc& t) { return t.type == DSVT_FLOAT && t.format == DSVT_FORMAT_LINEAR; }
static bool vpI32(const DsvtPluginTensorDesc& t) { return t.type == DSVT_INT32 && t.format == DSVT_FORMAT_LINEAR; }

struct VPParams {
    int max_voxels, max_pooled;
    int gx, gy, gz;            // sparse shape of the INPUT stage
    int sx, sy, sz;            // downsample stride
    int frames;                // coords.x = frame index of a multi-frame voxelizer (0 otherwise)
    __host__ __device__ int px() const { return (gx + sx - 1) / sx; }
    __host__ __device__ int py() const { return (gy + sy - 1) / sy; }
    __host__ __device__ int pz() const { return (gz + sz - 1) / sz; }
    __host__ __device__ int pv() const { return sx * sy * sz; }
    __host__ __device__ long ncell() const { return (long)px() * py() * pz() * frames; }
};

// coords rows are (b, z, y, x) (points2features.hip; plugins/src/points2Features.cu:755 writes (0, 0, y, x))
__device__ __forceinline__ bool vpCell(const uint4 c, const VPParams& p, uint32_t& cell, uint32_t& slot) {
    if (c.x >= (uint32_t)p.frames || c.y >= (uint32_t)p.gz || c.z >= (uint32_t)p.gy || c.w >= (uint32_t)p.gx) return false;
    const uint32_t z2 = c.y / p.sz, y2 = c.z / p.sy, x2 = c.w / p.sx;
    cell = ((c.x * (uint32_t)p.pz() + z2) * (uint32_t)p.py() + y2) * (uint32_t)p.px() + x2;          // canonical order: ascending (frame, z, y, x) of the pooled grid
    slot = ((c.w % p.sx) * p.sy + (c.z % p.sy)) * p.sz + (c.y % p.sz);                                // upstream's voxel_index_in_win (x-major)
    return true;
}

__global__ void __launch_bounds__(256)
vp_mark(const uint4* __restrict__ coords, const uint32_t* __restrict__ count, VPParams p, uint32_t* __restrict__ bitmap)
{
    const uint32_t n = min(*count, (uint32_t)p.max_voxels);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t cell, slot;
        if (vpCell(coords[i], p, cell, slot)) atomicOr(&bitmap[cell >> 5], 1u << (cell & 31));
    }
}

// exclusive prefix of the words' popcounts: one workgroup, rounds of 1024 x 8 words (lane-contiguous slabs)
__global__ void __launch_bounds__(1024)
vp_scan(const uint32_t* __restrict__ bitmap, int nwords, int max_pooled, int pv, uint32_t* __restrict__ base, uint32_t* __restrict__ pooled_num, uint32_t* __restrict__ row_num)
{
    __shared__ uint32_t smem[1024 / kWave + 1];
    constexpr int WPT = 8;
    uint32_t carry = 0;
    for (int w0 = 0; w0 < nwords; w0 += 1024 * WPT) {
        uint32_t cnt[WPT], sum = 0;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int w = w0 + (int)threadIdx.x * WPT + k;
            cnt[k] = w < nwords ? (uint32_t)__popc(bitmap[w]) : 0u;
            sum += cnt[k];
        }
        uint32_t total;
        uint32_t ex = blockExclusiveScan<1024>(sum, smem, &total) + carry;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int w = w0 + (int)threadIdx.x * WPT + k;
            if (w < nwords) base[w] = ex;
            ex += cnt[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        const uint32_t n = carry < (uint32_t)max_pooled ? carry : (uint32_t)max_pooled;
        *pooled_num = n; *row_num = n * (uint32_t)pv;
    }
}

__global__ void __launch_bounds__(256)
vp_fill(const uint4* __restrict__ coords, const uint32_t* __restrict__ count, VPParams p, const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ base,
        uint4* __restrict__ coords2, int32_t* __restrict__ pool_inds, int32_t* __restrict__ parent)
{
    const uint32_t n = min(*count, (uint32_t)p.max_voxels);
    const int pv = p.pv();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint4 c = coords[i];
        uint32_t cell, slot;
        int32_t row = -1;
        if (vpCell(c, p, cell, slot)) {
            const uint32_t r = base[cell >> 5] + (uint32_t)__popc(bitmap[cell >> 5] & ((1u << (cell & 31)) - 1u));
            if (r < (uint32_t)p.max_pooled) {
                row = (int32_t)r;
                pool_inds[(size_t)r * pv + slot] = (int32_t)i;                         // (voxels are unique: one writer per slot)
                coords2[r] = make_uint4(c.x, c.y / p.sz, c.z / p.sy, c.w / p.sx);      // (siblings write the same value)
            }
        }
        parent[i] = row;
    }
}

class DsvtVoxelPoolPlugin : public Plugin {
public:
    VPParams p_;
    explicit DsvtVoxelPoolPlugin(const VPParams& p) : p_(p) {}
    const char* type() const override { return "DsvtVoxelPoolPlugin"; }
    int nbOutputs() const override { return 5; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        const int b = in[0].d[0];
        switch (i) {
            case 0: *out = dims3(b, p_.max_pooled, 4); return 0;                      // pooled coordinates (b, z, y, x)
            case 1: *out = dims3(b, p_.max_pooled, p_.pv()); return 0;               // child row per slot, -1 = empty
            case 2: *out = dims3(b, p_.max_voxels, 1); return 0;                      // pooled row of every input voxel
            case 3: case 4: *out = dims1(b); return 0;                                // P2, P2 x pool_volume
        }
        return -1;
    }
    int outputType(int, const int32_t*, int) const override { return DSVT_INT32; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override { return pos >= 0 && pos <= 6 && vpI32(io[pos]); }
    int nwords() const { return (int)((p_.ncell() + 31) / 32); }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 2 * alignUp(sizeof(uint32_t) * (size_t)nwords()); }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void* workspace, hipStream_t stream) override {
        if (inDesc && inDesc[0].dims.d[0] != 1) return -2;
        WsCarver ws(workspace);
        uint32_t* bitmap = ws.take<uint32_t>(nwords());
        uint32_t* base = ws.take<uint32_t>(nwords());
        DSVT_CHECK(hipMemsetAsync(bitmap, 0, sizeof(uint32_t) * (size_t)nwords(), stream));
        DSVT_CHECK(hipMemsetAsync(out[1], 0xff, sizeof(int32_t) * (size_t)p_.max_pooled * p_.pv(), stream));      // -1: empty slots (the consumers' mask)
        if (zeroFill) {
            DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(uint32_t) * 4 * (size_t)p_.max_pooled, stream));
            DSVT_CHECK(hipMemsetAsync(out[2], 0xff, sizeof(int32_t) * (size_t)p_.max_voxels, stream));
        }
        const uint4* coords = static_cast<const uint4*>(in[0]);
        const uint32_t* cnt = static_cast<const uint32_t*>(in[1]);
        const int grid = cdiv(p_.max_voxels, 256) < 1024 ? cdiv(p_.max_voxels, 256) : 1024;
        hipLaunchKernelGGL(vp_mark, dim3(grid), dim3(256), 0, stream, coords, cnt, p_, bitmap);
        hipLaunchKernelGGL(vp_scan, dim3(1), dim3(1024), 0, stream, bitmap, nwords(), p_.max_pooled, p_.pv(), base, static_cast<uint32_t*>(out[3]), static_cast<uint32_t*>(out[4]));
        hipLaunchKernelGGL(vp_fill, dim3(grid), dim3(256), 0, stream, coords, cnt, p_, bitmap, base, static_cast<uint4*>(out[0]), static_cast<int32_t*>(out[1]),
                           static_cast<int32_t*>(out[2]));
        return lastError();
    }
    size_t serializationSize() const override { return 9 * sizeof(int); }
    void serialize(void* b) const override {
        char* d = static_cast<char*>(b);
        wr<int>(d, p_.max_voxels); wr<int>(d, p_.max_pooled); wr<int>(d, p_.gx); wr<int>(d, p_.gy); wr<int>(d, p_.gz); wr<int>(d, p_.sx); wr<int>(d, p_.sy); wr<int>(d, p_.sz); wr<int>(d, p_.frames);
    }
    Plugin* clone() const override { return new DsvtVoxelPoolPlugin(p_); }
};
static Plugin* vpNew(const VPParams& p) {
    auto no = [](const char* why) { createError() = std::string("DsvtVoxelPoolPlugin: ") + why; return static_cast<Plugin*>(nullptr); };
    if (p.max_voxels <= 0 || p.max_pooled <= 0 || p.gx <= 0 || p.gy <= 0 || p.gz <= 0 || p.frames < 1) return no("capacities, sparse_shape and frames must be positive");
    if (p.sx < 1 || p.sy < 1 || p.sz < 1 || p.pv() > 64) return no("stride must be positive with a pool volume of at most 64");
    if (p.ncell() >= (1l << 31)) return no("pooled grid cells x frames must stay below 2^31");
    return new DsvtVoxelPoolPlugin(p);
}
static Plugin* vpCreate(const DsvtPluginFieldCollection* fc) {
    VPParams p{};
    p.max_voxels = fieldInt(fc, "max_voxel_num"); p.max_pooled = fieldInt(fc, "max_pooled_num");
    int g[3], s[3]; fieldInts(fc, "sparse_shape", g, 3); fieldInts(fc, "stride", s, 3);
    p.gx = g[0]; p.gy = g[1]; p.gz = g[2]; p.sx = s[0]; p.sy = s[1]; p.sz = s[2]; p.frames = fieldInt(fc, "frames", 1);
    return vpNew(p);
}
static Plugin* vpDeser(const void* data, size_t len) {
    if (len != 9 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    VPParams p{};
    p.max_voxels = rd<int>(d); p.max_pooled = rd<int>(d); p.gx = rd<int>(d); p.gy = rd<int>(d); p.gz = rd<int>(d); p.sx = rd<int>(d); p.sy = rd<int>(d); p.sz = rd<int>(d); p.frames = rd<int>(d);
    return vpNew(p);
}
static Creator g_vpCreator{"DsvtVoxelPoolPlugin",
    {{"max_voxel_num", DSVT_FIELD_INT32}, {"max_pooled_num", DSVT_FIELD_INT32}, {"sparse_shape", DSVT_FIELD_INT32}, {"stride", DSVT_FIELD_INT32}},
    vpCreate, vpDeser, {}, {}};
static Registrar g_vpReg(&g_vpCreator);

// =====================================================================================================================
// DsvtPoolGatherPlugin: the prepool tensor of upstream's stage reduction, never materialised beyond what the three projections read
// =====================================================================================================================
__global__ void __launch_bounds__(256)
pool_gather_kernel(const float4* __restrict__ x, const int32_t* __restrict__ pool_inds, const uint32_t* __restrict__ pooled_num, const float4* __restrict__ pos,
                   int pv, int G, int max_pooled, float4* __restrict__ src, float4* __restrict__ kin)
{
    const uint32_t n = min(*pooled_num, (uint32_t)max_pooled);
    const size_t total = (size_t)n * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / G; const int c = (int)(i % G);
        float4 m = make_float4(-3.4028235e38f, -3.4028235e38f, -3.4028235e38f, -3.4028235e38f);
        for (int j = 0; j < pv; ++j) {
            const int32_t row = pool_inds[r * pv + j];
            const float4 v = row >= 0 ? x[(size_t)row * G + c] : make_float4(0.f, 0.f, 0.f, 0.f);      // empty slot = the zero row of upstream's preholder tensor
            m = make_float4(fmaxf(m.x, v.x), fmaxf(m.y, v.y), fmaxf(m.z, v.z), fmaxf(m.w, v.w));
            if (row >= 0) {                                                                // key input of child `row` (every voxel is the child of exactly one pool)
                const float4 pe = pos[(size_t)j * G + c];
                kin[(size_t)row * G + c] = make_float4(v.x + pe.x, v.y + pe.y, v.z + pe.z, v.w + pe.w);
            }
        }
        src[i] = m;
    }
}

class DsvtPoolGatherPlugin : public Plugin {
public:
    int max_pooled_, pv_, C_;      // (output 1 has the row capacity of input 0: one key-input row per input voxel)
    std::vector<float> pos_; float* pos_dev_ = nullptr; bool ok_ = true;
    DsvtPoolGatherPlugin(int mp, int pv, int c, const float* pos) : max_pooled_(mp), pv_(pv), C_(c), pos_(pos, pos + (size_t)pv * c) {
        ok_ = dsvtMalloc(&pos_dev_, sizeof(float) * pos_.size()) == hipSuccess && hipMemcpy(pos_dev_, pos_.data(), sizeof(float) * pos_.size(), hipMemcpyHostToDevice) == hipSuccess;
    }
    ~DsvtPoolGatherPlugin() override { if (pos_dev_) (void)dsvtFree(pos_dev_); }
    const char* type() const override { return "DsvtPoolGatherPlugin"; }
    int nbOutputs() const override { return 2; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i < 0 || i > 1) return -1;
        *out = dims3(in[0].d[0], i == 0 ? max_pooled_ : in[0].d[1], C_); return 0;
    }
    int outputType(int, const int32_t*, int) const override { return DSVT_FLOAT; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override { return (pos == 1 || pos == 2) ? vpI32(io[pos]) : pos >= 0 && pos <= 4 && vpF32(io[pos]); }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*, hipStream_t stream) override {
        if (!ok_) return static_cast<int>(hipErrorOutOfMemory);
        if (zeroFill) {
            DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(float) * (size_t)max_pooled_ * C_, stream));
            if (inDesc) DSVT_CHECK(hipMemsetAsync(out[1], 0, sizeof(float) * (size_t)inDesc[0].dims.d[1] * C_, stream));
        }
        hipLaunchKernelGGL(pool_gather_kernel, dim3(2048), dim3(256), 0, stream, static_cast<const float4*>(in[0]), static_cast<const int32_t*>(in[1]),
                           static_cast<const uint32_t*>(in[2]), reinterpret_cast<const float4*>(pos_dev_), pv_, C_ / 4, max_pooled_,
                           static_cast<float4*>(out[0]), static_cast<float4*>(out[1]));
        return lastError();
    }
    size_t serializationSize() const override { return 3 * sizeof(int) + sizeof(float) * pos_.size(); }
    void serialize(void* b) const override {
        char* d = static_cast<char*>(b); wr<int>(d, max_pooled_); wr<int>(d, pv_); wr<int>(d, C_);
        memcpy(d, pos_.data(), sizeof(float) * pos_.size());
    }
    Plugin* clone() const override { return new DsvtPoolGatherPlugin(max_pooled_, pv_, C_, pos_.data()); }
};
static Plugin* pgNew(int mp, int pv, int c, const float* pos) {
    return (mp > 0 && pv >= 1 && pv <= 64 && c > 0 && c % 4 == 0 && pos) ? new DsvtPoolGatherPlugin(mp, pv, c, pos) : nullptr;
}
static Plugin* pgCreate(const DsvtPluginFieldCollection* fc) {
    const int mp = fieldInt(fc, "max_pooled_num"), pv = fieldInt(fc, "pool_volume"), c = fieldInt(fc, "channel_num");
    if (mp <= 0 || pv < 1 || pv > 64 || c <= 0) return nullptr;
    return pgNew(mp, pv, c, fieldFloatArray(fc, "pos_embedding", (long)pv * c));
}
static Plugin* pgDeser(const void* data, size_t len) {
    if (len < 3 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    const int mp = rd<int>(d), pv = rd<int>(d), c = rd<int>(d);
    if (mp <= 0 || pv < 1 || pv > 64 || c <= 0 || len != 3 * sizeof(int) + sizeof(float) * (size_t)pv * c) return nullptr;
    return pgNew(mp, pv, c, reinterpret_cast<const float*>(d));
}
static Creator g_pgCreator{"DsvtPoolGatherPlugin",
    {{"max_pooled_num", DSVT_FIELD_INT32}, {"pool_volume", DSVT_FIELD_INT32}, {"channel_num", DSVT_FIELD_INT32}, {"pos_embedding", DSVT_FIELD_FLOAT32}},
    pgCreate, pgDeser, {}, {}};
static Registrar g_pgReg(&g_pgCreator);

// =====================================================================================================================
// DsvtPoolAttentionCorePlugin: one query, <= pool_volume keys (the children's rows of k / v, through the child table), H heads per pooled voxel -- one wavefront per voxel, lane l owns channels
// (C / 64) l .. : a head's dot product is a reduction over the 64 / H consecutive lanes that hold its channels
// =====================================================================================================================
template <int CPL>      // channels per lane (C = 64 CPL)
__global__ void __launch_bounds__(256)
pool_attention_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const int32_t* __restrict__ pool_inds,
                      const uint32_t* __restrict__ pooled_num, int pv, int lanes_per_head, int max_pooled, float* __restrict__ out)
{
    constexpr int C = 64 * CPL;
    const uint32_t n = min(*pooled_num, (uint32_t)max_pooled);
    const int lane = threadIdx.x & 63;
    for (uint32_t r = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); r < n; r += gridDim.x * (blockDim.x / 64)) {
        float qv[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) qv[c] = q[(size_t)r * C + lane * CPL + c];
        float mx = -3.4028235e38f, den = 0.f, acc[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
        // two passes over the (few) keys: the maximum of the unmasked scores, then the weighted sum (softmax over the keys, torch.nn.MultiheadAttention
        // with key_padding_mask: masked keys get -inf, i.e. weight 0)
        for (int pass = 0; pass < 2; ++pass)
            for (int j = 0; j < pv; ++j) {
                const int32_t row = pool_inds[(size_t)r * pv + j];
                if (row < 0) continue;                                                 // (wave-uniform) an empty slot: masked out of the softmax
                const float* kr = k + (size_t)row * C + lane * CPL;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < CPL; ++c) s += qv[c] * kr[c];
                for (int o = 1; o < lanes_per_head; o <<= 1) s += __shfl_xor(s, o, kWave);
                if (pass == 0) mx = fmaxf(mx, s);
                else {
                    const float e = expf(s - mx);
                    den += e;
                    const float* vr = v + (size_t)row * C + lane * CPL;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) acc[c] += e * vr[c];
                }
            }
#pragma unroll
        for (int c = 0; c < CPL; ++c) out[(size_t)r * C + lane * CPL + c] = den > 0.f ? acc[c] / den : 0.f;
    }
}

class DsvtPoolAttentionCorePlugin : public Plugin {
public:
    int max_pooled_, pv_, C_, H_;
    DsvtPoolAttentionCorePlugin(int mp, int pv, int c, int h) : max_pooled_(mp), pv_(pv), C_(c), H_(h) {}
    const char* type() const override { return "DsvtPoolAttentionCorePlugin"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override { if (i != 0) return -1; *out = dims3(in[0].d[0], max_pooled_, C_); return 0; }
    int outputType(int, const int32_t*, int) const override { return DSVT_FLOAT; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override { return (pos == 3 || pos == 4) ? vpI32(io[pos]) : pos >= 0 && pos <= 5 && vpF32(io[pos]); }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc*, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*, hipStream_t stream) override {
        if (zeroFill) DSVT_CHECK(hipMemsetAsync(out[0], 0, sizeof(float) * (size_t)max_pooled_ * C_, stream));
        const int lph = 64 / H_;
        auto launch = [&](auto cplTag) {
            constexpr int CPL = decltype(cplTag)::value;
            hipLaunchKernelGGL(pool_attention_kernel<CPL>, dim3(2048), dim3(256), 0, stream, static_cast<const float*>(in[0]), static_cast<const float*>(in[1]),
                               static_cast<const float*>(in[2]), static_cast<const int32_t*>(in[3]), static_cast<const uint32_t*>(in[4]), pv_, lph, max_pooled_, static_cast<float*>(out[0]));
        };
        if (C_ == 192) launch(std::integral_constant<int, 3>{}); else if (C_ == 128) launch(std::integral_constant<int, 2>{}); else if (C_ == 64) launch(std::integral_constant<int, 1>{});
        else return -3;
        return lastError();
    }
    size_t serializationSize() const override { return 4 * sizeof(int); }
    void serialize(void* b) const override { char* d = static_cast<char*>(b); wr<int>(d, max_pooled_); wr<int>(d, pv_); wr<int>(d, C_); wr<int>(d, H_); }
    Plugin* clone() const override { return new DsvtPoolAttentionCorePlugin(max_pooled_, pv_, C_, H_); }
};
static Plugin* paNew(int mp, int pv, int c, int h) {
    // a head's channels must be a whole number of lanes' worth: C / H a multiple of C / 64, H a power of two dividing 64
    const bool ok = mp > 0 && pv >= 1 && pv <= 64 && (c == 64 || c == 128 || c == 192) && h >= 1 && h <= 64 && (h & (h - 1)) == 0 && c % h == 0;
    return ok ? new DsvtPoolAttentionCorePlugin(mp, pv, c, h) : nullptr;
}
static Plugin* paCreate(const DsvtPluginFieldCollection* fc) { return paNew(fieldInt(fc, "max_pooled_num"), fieldInt(fc, "pool_volume"), fieldInt(fc, "channel_num"), fieldInt(fc, "num_heads")); }
static Plugin* paDeser(const void* data, size_t len) {
    if (len != 4 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    const int mp = rd<int>(d), pv = rd<int>(d), c = rd<int>(d), h = rd<int>(d);
    return paNew(mp, pv, c, h);
}
static Creator g_paCreator{"DsvtPoolAttentionCorePlugin",
    {{"max_pooled_num", DSVT_FIELD_INT32}, {"pool_volume", DSVT_FIELD_INT32}, {"channel_num", DSVT_FIELD_INT32}, {"num_heads", DSVT_FIELD_INT32}},
    paCreate, paDeser, {}, {}};
static Registrar g_paReg(&g_paCreator);

}  // namespace dsvt
