// decode.hip -- CenterHeadTopKPlugin: the CenterHead decode between the last head convolution and
// FilterBoxByScorePlugin, on the device in three launches.
//
// SURVEY.md section 8(f)-2.  The reference builds this stage from TensorRT layers
// (src/dsvt-ai-trt.cpp:1479-1669): sigmoid(heat map) -> TopK(500) per class over H*W -> TopK(500)
// over the 10 x 500 survivors -> class = idx / 500, cell = gathered per-class index, xs = cell % W,
// ys = cell / W -> gathers of center / center_z / dim / rot at the winners -> exp(dim),
// atan(rot[1] / rot[0]).  Its outputs are exactly the eight inputs of FilterBoxByScorePlugin.
//
// The two-stage TopK selects the 500 largest of ALL class x cell scores (every global winner is a
// per-class winner), and sigmoid is monotonic, so the selection runs once, on the raw logits:
//   1. topk_hist     4096-bin histogram of the order-preserving 32-bit key's top 12 bits (LDS
//                    histogram per workgroup, flushed with global atomics)
//   2. topk_collect  every workgroup finds the bin B holding the K-th largest key (suffix scan of
//                    the histogram) and appends the elements of bins >= B to a candidate list
//   3. topk_decode   one workgroup: (if the list is longer than 4096: a second 12-bit refinement
//                    inside bin B,) bitonic sort of <= 4096 (key, ~index) pairs in LDS, then the
//                    first K are decoded: sigmoid, cell -> (xs, ys), gathers, exp, atan.
// Order of the K rows: descending score, ties by ascending class * H*W + cell (the reference's TopK
// leaves ties unspecified).  The candidate lists are filled in atomic arrival order, which is harmless while
// everything fits (the sort key carries the index); a degenerate heat map -- a constant one, e.g. an empty frame
// whose head output is its bias -- can put more elements into the threshold bin than the lists hold.  topk_decode
// then falls back to an EXACT selection (exactSelect: radix select on the 32 key bits, then on the index among the
// elements equal to the threshold key, five passes of one workgroup over the heat map), so the K rows and their
// order are the rule above on every input, run to run.
#include "plugin_base.h"
#include "device_utils.h"

namespace dsvt {

struct TopKParams {
    int H, W, C;               // head tensor [1, H, W, C] fp32, channels-last
    int ncls, K;               // heat-map classes, rows to keep
    int off_center, off_z, off_dim, off_rot, off_hm;     // channel offsets inside C
};

static bool f32Linear(const DsvtPluginTensorDesc& t) { return t.type == DSVT_FLOAT && t.format == DSVT_FORMAT_LINEAR; }
static bool i32Linear(const DsvtPluginTensorDesc& t) { return t.type == DSVT_INT32 && t.format == DSVT_FORMAT_LINEAR; }

constexpr int TK_BINS = 4096, TK_SORT = 4096, TK_CAP = 65536;

__device__ __forceinline__ uint32_t floatKey(float f) {          // larger float <=> larger key
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float keyFloat(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Every heat-map element of one frame: fn(value, class, pixel).  The head tensor is [HW][C] with the ncls heat-map channels somewhere in C:
// the walk is a flat float4 stream over ALL channels (lane-contiguous 16-byte loads; a lane that reads its pixel's ncls values one by one
// makes lane-scattered 4-byte accesses 4 C bytes apart -- 36 us per pass over four 468 x 468 x 18 maps instead of the ~15 the bytes take),
// each lane sorting out which of its four values are heat-map values.  Frames whose size is not a multiple of four floats walk per pixel.
template <class F>
__device__ __forceinline__ void tkForHeat(const float* __restrict__ head, const TopKParams& p, F&& fn)
{
    const uint32_t HW = (uint32_t)(p.H * p.W), C = (uint32_t)p.C, total = HW * C;
    const uint32_t lo = (uint32_t)p.off_hm, hi = lo + (uint32_t)p.ncls;
    if ((total & 3u) == 0u) {
        const float4* h4 = reinterpret_cast<const float4*>(head);
        for (uint32_t v = blockIdx.x * 256u + threadIdx.x; v < total / 4u; v += gridDim.x * 256u) {
            const float4 q = h4[v];
            uint32_t px = (4u * v) / C, ch = 4u * v - px * C;
            const float vals[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ch >= lo && ch < hi) fn(vals[j], ch - lo, px);
                if (++ch == C) { ch = 0; ++px; }
            }
        }
    } else {
        for (uint32_t px = blockIdx.x * 256u + threadIdx.x; px < HW; px += gridDim.x * 256u) {
            const float* row = head + (size_t)px * C + lo;
            for (uint32_t c = 0; c < (uint32_t)p.ncls; ++c) fn(row[c], c, px);
        }
    }
}

__global__ void __launch_bounds__(256)
topk_hist(const float* __restrict__ head, TopKParams p, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t lh[TK_BINS];
    head += (size_t)blockIdx.y * p.H * p.W * p.C; hist += (size_t)blockIdx.y * (TK_BINS + 64);      // blockIdx.y = frame of a stack
    for (int i = threadIdx.x; i < TK_BINS; i += 256) lh[i] = 0;
    __syncthreads();
    tkForHeat(head, p, [&](float v, uint32_t, uint32_t) { atomicAdd(&lh[floatKey(v) >> 20], 1u); });
    __syncthreads();
    for (int i = threadIdx.x; i < TK_BINS; i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// bin of the K-th largest key: the largest B with count(bins >= B) >= K; *above = count(bins > B).  256 threads x 16 bins, thread t owning
// bins 16 (255 - t) ..: an EXCLUSIVE scan over t is then the count above a thread's bins (one workgroup scan on DPP adds; a one-thread walk
// over 256 partials in LDS was ~5 us at the head of every workgroup of topk_collect).
__device__ __forceinline__ int thresholdBin(const uint32_t* __restrict__ hist, int K, uint32_t* sh /* 258 */, uint32_t* above_out) {
    const int t = threadIdx.x, b0 = (255 - t) * 16;
    uint32_t local[16], sum = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { local[i] = hist[b0 + i]; sum += local[i]; }
    if (t == 0) { sh[256] = 0xffffffffu; sh[257] = 0; }
    uint32_t tot;
    uint32_t above = blockExclusiveScan<256>(sum, sh, &tot);       // (ends with a barrier: sh[256 ..] are published too)
    if (above < (uint32_t)K && above + sum >= (uint32_t)K) {
        for (int i = 15; i >= 0; --i) {
            if (above + local[i] >= (uint32_t)K) { sh[256] = (uint32_t)(b0 + i); sh[257] = above; break; }
            above += local[i];
        }
    }
    __syncthreads();
    const uint32_t b = sh[256];
    *above_out = sh[257];
    return b == 0xffffffffu ? 0 : (int)b;            // fewer than K elements in total: keep everything
}

__global__ void __launch_bounds__(256)
topk_collect(const float* __restrict__ head, TopKParams p, const uint32_t* __restrict__ hist, uint32_t* __restrict__ count,
             uint2* __restrict__ cand)
{
    __shared__ uint32_t sh[258];
    head += (size_t)blockIdx.y * p.H * p.W * p.C; hist += (size_t)blockIdx.y * (TK_BINS + 64); count += (size_t)blockIdx.y * (TK_BINS + 64);
    cand += (size_t)blockIdx.y * TK_CAP;
    uint32_t above;
    const uint32_t B = (uint32_t)thresholdBin(hist, p.K, sh, &above);
    if (blockIdx.x == 0 && threadIdx.x == 0) { count[2] = above; count[4] = B; }     // for topk_decode's second level / exact fallback
    const uint32_t HW = (uint32_t)(p.H * p.W);
    tkForHeat(head, p, [&](float v, uint32_t c, uint32_t px) {
        const uint32_t key = floatKey(v);
        const uint32_t bin = key >> 20;
        if (bin > B) {                               // fewer than K of these exist: cand[0 .. TK_SORT)
            const uint32_t slot = atomicAdd(count + 3, 1u);
            if (slot < TK_SORT) cand[slot] = make_uint2(key, c * HW + px);
        } else if (bin == B) {                       // the threshold bin: cand[TK_SORT ..)
            const uint32_t slot = atomicAdd(count, 1u);
            if (slot < TK_CAP - TK_SORT) cand[TK_SORT + slot] = make_uint2(key, c * HW + px);
        }
    });
}

// Exact top-K of a heat map whose threshold bin does not fit the candidate lists (one 1024-thread workgroup).  B1 = threshold bin of the
// key's top 12 bits, above1 = elements in higher bins (both from topk_collect).  Fills sk[0 .. K) with (key << 32) | ~index of exactly
// the K largest elements by (key descending, index ascending); returns nothing the caller does not already know (K).
__device__ void exactSelect(const float* __restrict__ head, const TopKParams& p, uint32_t B1, uint32_t above1, unsigned long long* sk,
                            uint32_t* sub /* TK_BINS */, uint32_t* sh /* 4 */, uint32_t* nsel)
{
    const int t = threadIdx.x, HW = p.H * p.W;
    auto clearSub = [&]() { for (int i = t; i < TK_BINS; i += 1024) sub[i] = 0; __syncthreads(); };
    // suffix walk: largest bin b with count(bins >= b) >= need; sh[0] = b, sh[1] = count(bins > b)
    auto suffixBin = [&](int nbins, uint32_t need) {
        __syncthreads();
        if (t == 0) {
            uint32_t run = 0; int b = 0; uint32_t ab = 0;
            for (int i = nbins - 1; i >= 0; --i) { if (run + sub[i] >= need) { b = i; ab = run; break; } run += sub[i]; }
            sh[0] = (uint32_t)b; sh[1] = ab;
        }
        __syncthreads();
    };
    // prefix walk: smallest bin b with count(bins <= b) >= need; sh[0] = b, sh[1] = count(bins < b)
    auto prefixBin = [&](int nbins, uint32_t need) {
        __syncthreads();
        if (t == 0) {
            uint32_t run = 0; int b = nbins - 1; uint32_t bl = 0;
            for (int i = 0; i < nbins; ++i) { if (run + sub[i] >= need) { b = i; bl = run; break; } run += sub[i]; }
            sh[0] = (uint32_t)b; sh[1] = bl;
        }
        __syncthreads();
    };
    auto forAll = [&](auto&& fn) {
        for (int px = t; px < HW; px += 1024) {
            const float* row = head + (size_t)px * p.C + p.off_hm;
            for (int c = 0; c < p.ncls; ++c) fn(floatKey(row[c]), (uint32_t)(c * HW + px));
        }
    };
    uint32_t need = (uint32_t)p.K - above1;                       // wanted from bin B1 (>= 1)
    clearSub();
    forAll([&](uint32_t key, uint32_t) { if ((key >> 20) == B1) atomicAdd(&sub[(key >> 8) & 0xfffu], 1u); });
    suffixBin(TK_BINS, need);
    const uint32_t B2 = sh[0]; need -= sh[1];
    clearSub();
    forAll([&](uint32_t key, uint32_t) { if ((key >> 8) == ((B1 << 12) | B2)) atomicAdd(&sub[key & 0xffu], 1u); });
    suffixBin(256, need);
    const uint32_t T = (B1 << 20) | (B2 << 8) | sh[0]; need -= sh[1];      // the threshold key; `need` elements equal to it are wanted
    clearSub();
    forAll([&](uint32_t key, uint32_t idx) { if (key == T) atomicAdd(&sub[idx >> 11], 1u); });       // index < 2^22 (tkNew bounds it)
    prefixBin(TK_BINS, need);
    const uint32_t I1 = sh[0]; need -= sh[1];
    clearSub();
    forAll([&](uint32_t key, uint32_t idx) { if (key == T && (idx >> 11) == I1) atomicAdd(&sub[idx & 0x7ffu], 1u); });
    prefixBin(2048, need);
    const uint32_t X = (I1 << 11) | sh[0];                         // elements equal to T are taken up to index X
    for (int i = t; i < TK_SORT; i += 1024) sk[i] = 0ull;
    if (t == 0) *nsel = 0;
    __syncthreads();
    forAll([&](uint32_t key, uint32_t idx) {
        if (key > T || (key == T && idx <= X)) {
            const uint32_t slot = atomicAdd(nsel, 1u);
            if (slot < TK_SORT) sk[slot] = ((unsigned long long)key << 32) | (uint32_t)~idx;
        }
    });
    __syncthreads();
}

__global__ void __launch_bounds__(1024)
topk_decode(const float* __restrict__ head, TopKParams p, const uint32_t* __restrict__ count,
            const uint2* __restrict__ cand, float* __restrict__ scores, int32_t* __restrict__ classes, int32_t* __restrict__ xs,
            int32_t* __restrict__ ys, float* __restrict__ center, float* __restrict__ center_z, float* __restrict__ angle,
            float* __restrict__ dim)
{
    __shared__ unsigned long long sk[TK_SORT];       // (key << 32) | ~index : descending order = score desc, index asc
    __shared__ uint32_t sub[TK_BINS];
    __shared__ uint32_t sh[4];
    __shared__ uint32_t nsel;
    const int t = threadIdx.x;
    {   // blockIdx.x = frame of a stack
        const size_t b = blockIdx.x, K = (size_t)p.K;
        head += b * p.H * p.W * p.C; count += b * (TK_BINS + 64); cand += b * TK_CAP;
        scores += b * K; classes += b * K; xs += b * K; ys += b * K; center += b * K * 2; center_z += b * K; angle += b * K; dim += b * K * 3;
    }
    uint32_t Ma = count[3], Me = count[0];           // elements above the threshold bin (< K) / inside it
    const bool listOverflow = Me > TK_CAP - TK_SORT || Ma > TK_SORT;      // the arrival-ordered lists dropped elements: exact path below
    if (Ma > TK_SORT) Ma = TK_SORT;
    if (Me > TK_CAP - TK_SORT) Me = TK_CAP - TK_SORT;
    for (int i = t; i < TK_SORT; i += 1024) sk[i] = 0ull;
    if (t == 0) nsel = Ma;
    __syncthreads();
    for (uint32_t i = t; i < Ma; i += 1024) { const uint2 c = cand[i]; sk[i] = ((unsigned long long)c.x << 32) | (uint32_t)~c.y; }
    const uint2* eq = cand + TK_SORT;
    if (Ma + Me <= TK_SORT) {
        for (uint32_t i = t; i < Me; i += 1024) { const uint2 c = eq[i]; sk[Ma + i] = ((unsigned long long)c.x << 32) | (uint32_t)~c.y; }
    } else {
        // second level inside the threshold bin: bits 19..8 of the key
        const uint32_t need = (uint32_t)p.K > count[2] ? (uint32_t)p.K - count[2] : 0u;     // how many of the bin's elements are wanted
        for (int i = t; i < TK_BINS; i += 1024) sub[i] = 0;
        __syncthreads();
        for (uint32_t i = t; i < Me; i += 1024) atomicAdd(&sub[(eq[i].x >> 8) & 0xfffu], 1u);
        __syncthreads();
        if (t == 0) {                                                               // serial suffix walk over 4096 sub-bins
            uint32_t run = 0; int b2 = 0;
            for (int i = TK_BINS - 1; i >= 0; --i) { run += sub[i]; if (run >= need) { b2 = i; break; } }
            sh[2] = (uint32_t)b2;
        }
        __syncthreads();
        const uint32_t B2 = sh[2];
        for (uint32_t i = t; i < Me; i += 1024) {
            const uint2 c = eq[i];
            if (((c.x >> 8) & 0xfffu) >= B2) {
                const uint32_t slot = atomicAdd(&nsel, 1u);
                if (slot < TK_SORT) sk[slot] = ((unsigned long long)c.x << 32) | (uint32_t)~c.y;
            }
        }
    }
    __syncthreads();
    // more candidates than the sort holds share the top 24 key bits, or the lists themselves overflowed: the degenerate case
    bool exact = listOverflow || (Ma + Me > TK_SORT && nsel > TK_SORT);
    if (exact) exactSelect(head, p, count[4], count[2], sk, sub, sh, &nsel);
    // bitonic sort, descending, of the smallest power of two >= the number of candidates (the rest of sk is zero = lowest)
    uint32_t tot = exact ? (uint32_t)p.K : (Ma + Me <= TK_SORT) ? Ma + Me : (nsel < TK_SORT ? nsel : TK_SORT);
    int NS = 512;
    while (NS < (int)tot) NS <<= 1;
    for (int k = 2; k <= NS; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = t; q < NS / 2; q += 1024) {
                const int lo = ((q & ~(j - 1)) << 1) | (q & (j - 1)), hi = lo | j;
                const bool desc = (lo & k) == 0;
                const unsigned long long a = sk[lo], b = sk[hi];
                if ((a < b) == desc) { sk[lo] = b; sk[hi] = a; }
            }
            __syncthreads();
        }
    // decode the first K
    const int HW = p.H * p.W;
    for (int i = t; i < p.K; i += 1024) {
        const unsigned long long e = sk[i];
        const uint32_t key = (uint32_t)(e >> 32), idx = ~(uint32_t)e;
        if (e == 0ull) {                                            // fewer than K elements exist (H*W*ncls < K)
            scores[i] = 0.f; classes[i] = 0; xs[i] = 0; ys[i] = 0; center[2 * i] = center[2 * i + 1] = 0.f; center_z[i] = 0.f;
            angle[i] = 0.f; dim[3 * i] = dim[3 * i + 1] = dim[3 * i + 2] = 0.f;
            continue;
        }
        const int cls = (int)(idx / (uint32_t)HW), cell = (int)(idx - (uint32_t)cls * HW);
        const float* row = head + (size_t)cell * p.C;
        scores[i] = 1.0f / (1.0f + expf(-keyFloat(key)));             // sigmoid (:1479-1483)
        classes[i] = cls; xs[i] = cell % p.W; ys[i] = cell / p.W;      // :1560-1600
        center[2 * i] = row[p.off_center]; center[2 * i + 1] = row[p.off_center + 1];
        center_z[i] = row[p.off_z];
        dim[3 * i] = expf(row[p.off_dim]); dim[3 * i + 1] = expf(row[p.off_dim + 1]); dim[3 * i + 2] = expf(row[p.off_dim + 2]);   // :1486-1491
        angle[i] = atanf(row[p.off_rot + 1] / row[p.off_rot]);        // rot[1] / rot[0] = sin / cos (:1494-1501, 1655-1669)
    }
}

class CenterHeadTopKPlugin : public Plugin {
public:
    TopKParams p_;
    explicit CenterHeadTopKPlugin(const TopKParams& p) : p_(p) {}
    const char* type() const override { return "CenterHeadTopKPlugin"; }
    bool handlesBatch() const override { return true; }              // blockIdx.y / x = frame of a stack of head tensors
    int nbOutputs() const override { return 8; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        const int b = in[0].d[0];
        switch (i) {
            case 0: case 1: case 2: case 3: *out = dims2(b, p_.K); return 0;            // scores, classes, xs, ys
            case 4: *out = dims4(b, 1, p_.K, 2); return 0;                              // center
            case 5: case 6: *out = dims4(b, 1, p_.K, 1); return 0;                      // center_z, angle
            case 7: *out = dims4(b, 1, p_.K, 3); return 0;                              // dim
        }
        return -1;
    }
    int outputType(int i, const int32_t*, int) const override { return (i >= 1 && i <= 3) ? DSVT_INT32 : DSVT_FLOAT; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int, int) const override {
        if (pos >= 2 && pos <= 4) return i32Linear(io[pos]);
        return pos >= 0 && pos <= 8 && f32Linear(io[pos]);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc* in, int nbIn, const DsvtPluginTensorDesc*, int) const override {
        const size_t nb = (in && nbIn > 0 && in[0].dims.nbDims >= 1 && in[0].dims.d[0] > 1) ? (size_t)in[0].dims.d[0] : 1;
        return alignUp(sizeof(uint32_t) * (TK_BINS + 64) * nb) + alignUp(sizeof(uint2) * TK_CAP * nb);
    }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void* ws,
                hipStream_t stream) override {
        const int nb = (inDesc && inDesc[0].dims.nbDims >= 1 && inDesc[0].dims.d[0] > 1) ? inDesc[0].dims.d[0] : 1;
        WsCarver c(ws);
        uint32_t* hist = c.take<uint32_t>((size_t)(TK_BINS + 64) * nb);          // per frame: histogram | counters
        uint32_t* count = hist + TK_BINS;
        uint2* cand = c.take<uint2>((size_t)TK_CAP * nb);
        if (hipMemsetAsync(hist, 0, sizeof(uint32_t) * (TK_BINS + 64) * nb, stream) != hipSuccess) return lastError();
        const float* head = static_cast<const float*>(in[0]);
        const int HW = p_.H * p_.W;
        int grid = cdiv(HW, 256); if (grid > 1024) grid = 1024;
        hipLaunchKernelGGL(topk_hist, dim3(grid, nb), dim3(256), 0, stream, head, p_, hist);
        hipLaunchKernelGGL(topk_collect, dim3(grid, nb), dim3(256), 0, stream, head, p_, hist, count, cand);
        hipLaunchKernelGGL(topk_decode, dim3(nb), dim3(1024), 0, stream, head, p_, count, cand, static_cast<float*>(out[0]),
                           static_cast<int32_t*>(out[1]), static_cast<int32_t*>(out[2]), static_cast<int32_t*>(out[3]),
                           static_cast<float*>(out[4]), static_cast<float*>(out[5]), static_cast<float*>(out[6]),
                           static_cast<float*>(out[7]));
        return lastError();
    }
    size_t serializationSize() const override { return 10 * sizeof(int); }
    void serialize(void* b) const override {
        char* d = static_cast<char*>(b);
        const int* v = reinterpret_cast<const int*>(&p_);
        for (int i = 0; i < 10; ++i) wr<int>(d, v[i]);
    }
    Plugin* clone() const override { return new CenterHeadTopKPlugin(p_); }
};
static Plugin* tkNew(const TopKParams& p) {
    if (p.H <= 0 || p.W <= 0 || p.C <= 0 || p.ncls <= 0 || p.K <= 0 || p.K > TK_SORT) return nullptr;
    if ((long)p.H * p.W * p.ncls >= (1l << 22)) return nullptr;        // the exact fallback selects on 22 index bits
    const int offs[5] = {p.off_center + 1, p.off_z, p.off_dim + 2, p.off_rot + 1, p.off_hm + p.ncls - 1};
    for (int o : offs) if (o < 0 || o >= p.C) return nullptr;
    return new CenterHeadTopKPlugin(p);
}
static Plugin* tkCreate(const DsvtPluginFieldCollection* fc) {
    TopKParams p{};
    p.H = fieldInt(fc, "feature_height"); p.W = fieldInt(fc, "feature_width"); p.C = fieldInt(fc, "channel_num");
    p.ncls = fieldInt(fc, "class_num"); p.K = fieldInt(fc, "max_top_k");
    p.off_center = fieldInt(fc, "center_offset", 0); p.off_z = fieldInt(fc, "center_z_offset", 2); p.off_dim = fieldInt(fc, "dim_offset", 3);
    p.off_rot = fieldInt(fc, "rot_offset", 6); p.off_hm = fieldInt(fc, "hm_offset", 8);
    return tkNew(p);
}
static Plugin* tkDeser(const void* data, size_t len) {
    if (len < 10 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    TopKParams p{}; int* v = reinterpret_cast<int*>(&p);
    for (int i = 0; i < 10; ++i) v[i] = rd<int>(d);
    return tkNew(p);
}
static Creator g_tkCreator{"CenterHeadTopKPlugin",
    {{"feature_height", DSVT_FIELD_INT32}, {"feature_width", DSVT_FIELD_INT32}, {"channel_num", DSVT_FIELD_INT32},
     {"class_num", DSVT_FIELD_INT32}, {"max_top_k", DSVT_FIELD_INT32}, {"center_offset", DSVT_FIELD_INT32},
     {"center_z_offset", DSVT_FIELD_INT32}, {"dim_offset", DSVT_FIELD_INT32}, {"rot_offset", DSVT_FIELD_INT32},
     {"hm_offset", DSVT_FIELD_INT32}},
    tkCreate, tkDeser, {}, {}};
static Registrar g_tkReg(&g_tkCreator);

}  // namespace dsvt
