// conv.hip -- DsvtConv2dPlugin: NHWC fp16 implicit-GEMM convolution on v_mfma_f32_16x16x32_f16.
//
// SURVEY.md section 8(f)-1, the first row "next" to the hot path: the reference's BEV ResNet,
// deblocks and CenterHead (src/dsvt-ai-trt.cpp:1144-1468: addConvolutionNd / addDeconvolutionNd +
// addScale(BN) + ReLU + ElementWise SUM, helpers :149-246).  One kernel covers all of them:
//   * 3x3 / 1x1 convolution, stride 1 or 2, zero padding                      (convBnLELU, convBn)
//   * folded BatchNorm = per-channel bias in the epilogue                      (addBatchNorm2d :149-180)
//   * residual add + ReLU in the epilogue                                      (:1165-1166 ...)
//   * ConvTranspose with kernel == stride as a 1x1 convolution whose output-channel chunk picks
//     the (dy, dx) sub-pixel it writes (pixel shuffle in the store)            (deconvBnLELU :217-246)
//   * channel offset / stride on the output => the 3-way concat (:1363) is free.
//
// GEMM view: rows = output pixels, columns = output channels, K = taps x Cin.  It is the fp16
// linear kernel (linear.hip) with a gathered A operand: for tap (ky,kx) the B-operand fragment of
// a pixel is 8 consecutive channels of the input pixel at (y*s+ky-p, x*s+kx-p) -- 16 contiguous
// bytes of an NHWC row, loaded straight from global memory (zero outside the image) -- while the
// weights stream through LDS in [128 cout][KC cin] slabs, double buffered, rows padded by 16 halfs
// (conflict-free ds_read_b128, enumerated).  Transposed product tile D[cout][pixel]: a lane owns one
// pixel and four consecutive output channels, so bias / residual / store are vector accesses.
// Workgroup = 4 waves = 128 pixels x 128 output channels; wave = 32 pixels.
#include "plugin_base.h"
#include "device_utils.h"
#include <cstdlib>

namespace dsvt {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int CNB = 128;          // output channels per workgroup
constexpr int CPX = 128;          // pixels per workgroup
constexpr int CNT = CNB / 16;     // n-tiles

struct ConvArgs {
    const _Float16* in; int H, W, Cin;           // NHWC input
    const _Float16* wt;                          // [CoutRows][KH*KW][Cin]
    const float* bias;                           // [Cout] (per real output channel) or nullptr
    const _Float16* res; int res_ld;             // residual NHWC on the OUTPUT grid, or nullptr
    void* out; int out_ld, out_coff, out_f32;    // NHWC output, channel stride / offset
    int Ho, Wo;                                  // GEMM pixel grid (= conv output grid before pixel shuffle)
    int CoutRows;                                // rows of wt = up*up*Cout
    int Cout;                                    // real output channels
    int KH, KW, stride, pad, up, relu;
};

// MT = 16-pixel MFMA tiles per wave, NW = waves per workgroup (128 pixels per workgroup either way)
template <int KC, int MT, int NW>
__global__ void __launch_bounds__(64 * NW, (MT == 1 ? 4 : 2))
conv_f16_kernel(ConvArgs a)
{
    constexpr int NTHR = 64 * NW;
    constexpr int KSTEPS = KC / 32, LDW = KC + 16, WPT = (CNB * KC / 8 + NTHR - 1) / NTHR;     // WPT: uint4 weight loads per thread per slab
    __shared__ __attribute__((aligned(16))) _Float16 sW[2][CNB * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int npix = a.Ho * a.Wo;
    const int n0 = blockIdx.y * CNB;
    const int nvalid = a.CoutRows - n0 < CNB ? a.CoutRows - n0 : CNB;
    const int ntiles = (nvalid + 15) / 16;
    const int Ktot = a.KH * a.KW * a.Cin;
    const int nck = a.Cin / KC, NS = a.KH * a.KW * nck;

    // XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8 (observed, speed only); giving every
    // XCD a contiguous run of pixel tiles keeps the three input rows a 3x3 tap window re-reads in that XCD's L2
    const int per_xcd = gridDim.x / 8;                    // grid.x is rounded up to a multiple of 8 by the host
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile * CPX >= npix) return;
    int py[MT], px[MT]; bool pv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int p = tile * CPX + wave * 16 * MT + mt * 16 + r;
        pv[mt] = p < npix;
        p = pv[mt] ? p : npix - 1;
        py[mt] = p / a.Wo; px[mt] = p % a.Wo;
    }

    floatx4 acc[MT][CNT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < CNT; ++t) acc[mt][t] = floatx4{0.f, 0.f, 0.f, 0.f};

    auto loadW = [&](int s, uint4 (&wr)[WPT]) {
        const int tap = s / nck, cc = s - tap * nck;
        const _Float16* base = a.wt + (size_t)n0 * Ktot + (size_t)tap * a.Cin + cc * KC;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int i = tid + j * NTHR, n = i / (KC / 8), c = i % (KC / 8);
            wr[j] = n < nvalid ? *reinterpret_cast<const uint4*>(base + (size_t)n * Ktot + c * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto storeW = [&](int buf, const uint4 (&wr)[WPT]) {
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int i = tid + j * NTHR, n = i / (KC / 8), c = i % (KC / 8);
            if (n < CNB) *reinterpret_cast<uint4*>(&sW[buf][n * LDW + c * 8]) = wr[j];
        }
    };
    auto loadA = [&](int s, half8 (&af)[MT][KSTEPS]) {
        const int tap = s / nck, cc = s - tap * nck;
        const int ky = tap / a.KW, kx = tap - ky * a.KW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int yi = py[mt] * a.stride + ky - a.pad, xi = px[mt] * a.stride + kx - a.pad;
            const bool inb = yi >= 0 && yi < a.H && xi >= 0 && xi < a.W;
            const _Float16* src = a.in + ((size_t)(inb ? yi : 0) * a.W + (inb ? xi : 0)) * a.Cin + cc * KC + g * 8;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                half8 v = *reinterpret_cast<const half8*>(src + ks * 32);
                if (!inb) v = half8{0, 0, 0, 0, 0, 0, 0, 0};
                af[mt][ks] = v;
            }
        }
    };

    uint4 wr[WPT];
    half8 cur[MT][KSTEPS], nxt[MT][KSTEPS];
    loadW(0, wr);
    loadA(0, cur);
    storeW(0, wr);
    __syncthreads();
    for (int s = 0; s < NS; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < NS;
        if (more) { loadW(s + 1, wr); loadA(s + 1, nxt); }          // next slab's traffic is in flight during the MFMAs
        const _Float16* pw = &sW[buf][r * LDW + g * 8];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int t = 0; t < CNT; ++t)
                if (t < ntiles) {
                    const half8 wf = *reinterpret_cast<const half8*>(pw + t * 16 * LDW + ks * 32);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, cur[mt][ks], acc[mt][t], 0, 0, 0);
                }
        if (more) {
            storeW(buf ^ 1, wr);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) cur[mt][ks] = nxt[mt][ks];
        }
        __syncthreads();
    }

    // ---- epilogue: lane = pixel (mt, r), output rows n0 + 16t + 4g + i of the weight matrix -------
    const int sub = n0 / a.Cout;                         // (dy, dx) chunk of a pixel-shuffle deconvolution; 0 otherwise
    const int dy = sub / a.up, dx = sub - dy * a.up;
    const int cbase = n0 - sub * a.Cout;
    const int Wout = a.Wo * a.up;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (!pv[mt]) continue;
        const size_t opix = (size_t)(py[mt] * a.up + dy) * Wout + (px[mt] * a.up + dx);
#pragma unroll
        for (int t = 0; t < CNT; ++t) {
            const int co = cbase + t * 16 + 4 * g;
            if (t >= ntiles || co >= a.Cout) continue;
            float v[4] = {acc[mt][t][0], acc[mt][t][1], acc[mt][t][2], acc[mt][t][3]};
            const bool full = co + 3 < a.Cout;
            if (a.bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (co + i < a.Cout) v[i] += a.bias[co + i];
            }
            if (a.res && full) {
                const half4 rv = *reinterpret_cast<const half4*>(a.res + opix * a.res_ld + co);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
            }
            if (a.relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            if (a.out_f32) {
                float* o = static_cast<float*>(a.out) + opix * a.out_ld + a.out_coff + co;
#pragma unroll
                for (int i = 0; i < 4; ++i) if (co + i < a.Cout) o[i] = v[i];
            } else {
                _Float16* o = static_cast<_Float16*>(a.out) + opix * a.out_ld + a.out_coff + co;
                if (full) {
                    half4 h; h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
                    *reinterpret_cast<half4*>(o) = h;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (co + i < a.Cout) o[i] = (_Float16)v[i];
                }
            }
        }
    }
}

static int launchConv(const ConvArgs& a, int KC, hipStream_t stream) {
    dim3 grid((unsigned)(cdiv(cdiv(a.Ho * a.Wo, CPX), 8) * 8), (unsigned)cdiv(a.CoutRows, CNB));
    static int variant = -1;       // 0: 4 waves x 32 pixels   1: 8 waves x 16 pixels (4 waves/SIMD)
    if (variant < 0) { const char* e = getenv("DSVT_CONV_VARIANT"); variant = e ? atoi(e) : 1; }
    if (variant == 0) {
        if (KC == 128) hipLaunchKernelGGL((conv_f16_kernel<128, 2, 4>), grid, dim3(256), 0, stream, a);
        else if (KC == 96) hipLaunchKernelGGL((conv_f16_kernel<96, 2, 4>), grid, dim3(256), 0, stream, a);
        else if (KC == 64) hipLaunchKernelGGL((conv_f16_kernel<64, 2, 4>), grid, dim3(256), 0, stream, a);
        else return -3;
    } else {
        if (KC == 128) hipLaunchKernelGGL((conv_f16_kernel<128, 1, 8>), grid, dim3(512), 0, stream, a);
        else if (KC == 96) hipLaunchKernelGGL((conv_f16_kernel<96, 1, 8>), grid, dim3(512), 0, stream, a);
        else if (KC == 64) hipLaunchKernelGGL((conv_f16_kernel<64, 1, 8>), grid, dim3(512), 0, stream, a);
        else return -3;
    }
    return lastError();
}

// -------------------------------------------------------------------------------------
struct ConvCfg {
    int H, W, Cin, Cout, KH, KW, stride, pad, up, relu, has_res, out_ld, out_coff, out_f32;
};

class DsvtConv2dPlugin : public Plugin {
public:
    ConvCfg c_;
    std::vector<float> w_, b_;           // w_: [up*up*Cout][KH*KW][Cin]
    _Float16* w_dev_ = nullptr; float* b_dev_ = nullptr;
    bool ok_ = false;
    int Ho() const { return (c_.H + 2 * c_.pad - c_.KH) / c_.stride + 1; }
    int Wo() const { return (c_.W + 2 * c_.pad - c_.KW) / c_.stride + 1; }
    int rows() const { return c_.up * c_.up * c_.Cout; }
    int KC() const {
        if (const char* e = getenv("DSVT_CONV_KC")) { int k = atoi(e); if (k > 0 && c_.Cin % k == 0) return k; }   // tuning knob
        // large images: 64-channel slabs (150 VGPRs, 40 KB LDS => 3 waves/SIMD) hide the per-slab barrier better than
        // 128-channel ones (measured +10 % on the 468x468 layers); small images prefer fewer, fatter slabs
        if (Ho() * Wo() >= 100000 && c_.Cin % 64 == 0) return 64;
        return c_.Cin % 128 == 0 ? 128 : c_.Cin % 96 == 0 ? 96 : 64;
    }
    DsvtConv2dPlugin(const ConvCfg& c, const float* w, const float* b) : c_(c) {
        const size_t nw = (size_t)rows() * c.KH * c.KW * c.Cin;
        w_.assign(w, w + nw);
        if (b) b_.assign(b, b + c.Cout);
        std::vector<_Float16> wh(nw);
        for (size_t i = 0; i < nw; ++i) wh[i] = (_Float16)w_[i];
        ok_ = hipMalloc(&w_dev_, sizeof(_Float16) * nw) == hipSuccess &&
              hipMemcpy(w_dev_, wh.data(), sizeof(_Float16) * nw, hipMemcpyHostToDevice) == hipSuccess;
        if (ok_ && b) ok_ = hipMalloc(&b_dev_, sizeof(float) * c.Cout) == hipSuccess &&
                            hipMemcpy(b_dev_, b_.data(), sizeof(float) * c.Cout, hipMemcpyHostToDevice) == hipSuccess;
    }
    ~DsvtConv2dPlugin() override { if (w_dev_) (void)hipFree(w_dev_); if (b_dev_) (void)hipFree(b_dev_); }
    const char* type() const override { return "DsvtConv2dPlugin"; }
    int nbOutputs() const override { return 1; }
    int outputDims(int i, const DsvtDims* in, int, DsvtDims* out) const override {
        if (i != 0) return -1;
        *out = dims4(in[0].d[0], Ho() * c_.up, Wo() * c_.up, c_.out_ld); return 0;
    }
    int outputType(int, const int32_t*, int) const override { return c_.out_f32 ? DSVT_FLOAT : DSVT_HALF; }
    bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int nbIn, int) const override {
        if (io[pos].format != DSVT_FORMAT_LINEAR) return false;
        return pos < nbIn ? io[pos].type == DSVT_HALF : io[pos].type == (c_.out_f32 ? DSVT_FLOAT : DSVT_HALF);
    }
    size_t workspaceSize(const DsvtPluginTensorDesc*, int, const DsvtPluginTensorDesc*, int) const override { return 0; }
    int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc*, const void* const* in, void* const* out, void*,
                hipStream_t stream) override {
        if (!ok_) return static_cast<int>(hipErrorOutOfMemory);
        ConvArgs a{};
        a.in = static_cast<const _Float16*>(in[0]); a.H = c_.H; a.W = c_.W; a.Cin = c_.Cin;
        a.wt = w_dev_; a.bias = b_dev_;
        a.res = c_.has_res ? static_cast<const _Float16*>(in[1]) : nullptr;
        a.res_ld = (c_.has_res && inDesc) ? inDesc[1].dims.d[inDesc[1].dims.nbDims - 1] : c_.Cout;
        a.out = out[0]; a.out_ld = c_.out_ld; a.out_coff = c_.out_coff; a.out_f32 = c_.out_f32;
        a.Ho = Ho(); a.Wo = Wo(); a.CoutRows = rows(); a.Cout = c_.Cout;
        a.KH = c_.KH; a.KW = c_.KW; a.stride = c_.stride; a.pad = c_.pad; a.up = c_.up; a.relu = c_.relu;
        return launchConv(a, KC(), stream);
    }
    size_t serializationSize() const override { return 14 * sizeof(int) + sizeof(int) + sizeof(float) * (w_.size() + b_.size()); }
    void serialize(void* buf) const override {
        char* d = static_cast<char*>(buf);
        const int* ci = reinterpret_cast<const int*>(&c_);
        for (int i = 0; i < 14; ++i) wr<int>(d, ci[i]);
        wr<int>(d, b_.empty() ? 0 : 1);
        memcpy(d, w_.data(), sizeof(float) * w_.size()); d += sizeof(float) * w_.size();
        memcpy(d, b_.data(), sizeof(float) * b_.size());
    }
    Plugin* clone() const override { return new DsvtConv2dPlugin(c_, w_.data(), b_.empty() ? nullptr : b_.data()); }
};

static Plugin* convNew(const ConvCfg& c, const float* w, const float* b) {
    if (c.H <= 0 || c.W <= 0 || c.Cin <= 0 || c.Cin % 32 != 0 || c.Cout <= 0 || c.KH <= 0 || c.KW <= 0 || !w) return nullptr;
    if (c.stride < 1 || c.pad < 0 || c.up < 1 || c.out_ld < c.out_coff + c.Cout) return nullptr;
    if (c.Cin % 64 != 0) return nullptr;                               // K slabs are 64 / 96 / 128 channels wide
    if (c.up > 1 && (c.KH != 1 || c.KW != 1 || c.stride != 1 || c.Cout % CNB != 0)) return nullptr;   // pixel-shuffle chunks are whole workgroup columns
    if (!c.out_f32 && (c.out_ld % 4 != 0 || c.out_coff % 4 != 0)) return nullptr;
    return new DsvtConv2dPlugin(c, w, b);
}
static Plugin* convCreate(const DsvtPluginFieldCollection* fc) {
    ConvCfg c{};
    c.H = fieldInt(fc, "in_height"); c.W = fieldInt(fc, "in_width"); c.Cin = fieldInt(fc, "in_channels"); c.Cout = fieldInt(fc, "out_channels");
    c.KH = c.KW = fieldInt(fc, "kernel_size", 1); c.stride = fieldInt(fc, "stride", 1); c.pad = fieldInt(fc, "padding", 0);
    c.up = fieldInt(fc, "pixel_shuffle", 1); c.relu = fieldInt(fc, "relu", 0); c.has_res = fieldInt(fc, "has_residual", 0);
    c.out_ld = fieldInt(fc, "out_channel_stride", c.Cout); c.out_coff = fieldInt(fc, "out_channel_offset", 0); c.out_f32 = fieldInt(fc, "out_f32", 0);
    const DsvtPluginField* w = findField(fc, "weight"); const DsvtPluginField* b = findField(fc, "bias");
    if (!w || !w->data || c.Cin <= 0 || c.Cout <= 0 || c.up < 1) return nullptr;
    if ((long)w->length != (long)c.up * c.up * c.Cout * c.KH * c.KW * c.Cin) return nullptr;
    if (b && b->data && b->length != c.Cout) return nullptr;
    return convNew(c, static_cast<const float*>(w->data), (b && b->data) ? static_cast<const float*>(b->data) : nullptr);
}
static Plugin* convDeser(const void* data, size_t len) {
    if (len < 15 * sizeof(int)) return nullptr;
    const char* d = static_cast<const char*>(data);
    ConvCfg c{}; int* ci = reinterpret_cast<int*>(&c);
    for (int i = 0; i < 14; ++i) ci[i] = rd<int>(d);
    int has_b = rd<int>(d);
    if (c.Cin <= 0 || c.Cout <= 0 || c.up < 1 || c.KH <= 0 || c.KW <= 0) return nullptr;
    size_t nw = (size_t)c.up * c.up * c.Cout * c.KH * c.KW * c.Cin;
    if (len < 15 * sizeof(int) + sizeof(float) * (nw + (has_b ? c.Cout : 0))) return nullptr;
    std::vector<float> w(nw), b(has_b ? c.Cout : 0);
    memcpy(w.data(), d, sizeof(float) * nw); if (has_b) memcpy(b.data(), d + sizeof(float) * nw, sizeof(float) * c.Cout);
    return convNew(c, w.data(), has_b ? b.data() : nullptr);
}
static Creator g_convCreator{"DsvtConv2dPlugin",
    {{"in_height", DSVT_FIELD_INT32}, {"in_width", DSVT_FIELD_INT32}, {"in_channels", DSVT_FIELD_INT32}, {"out_channels", DSVT_FIELD_INT32},
     {"kernel_size", DSVT_FIELD_INT32}, {"stride", DSVT_FIELD_INT32}, {"padding", DSVT_FIELD_INT32}, {"pixel_shuffle", DSVT_FIELD_INT32},
     {"relu", DSVT_FIELD_INT32}, {"has_residual", DSVT_FIELD_INT32}, {"out_channel_stride", DSVT_FIELD_INT32},
     {"out_channel_offset", DSVT_FIELD_INT32}, {"out_f32", DSVT_FIELD_INT32}, {"weight", DSVT_FIELD_FLOAT32}, {"bias", DSVT_FIELD_FLOAT32}},
    convCreate, convDeser, {}, {}};
static Registrar g_convReg(&g_convCreator);

}  // namespace dsvt
