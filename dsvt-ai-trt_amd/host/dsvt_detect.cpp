// dsvt_detect -- the reference's host executable (src/dsvt-ai-trt.cpp: `dsvt-ai-trt -d`) on libdsvt_hip.so.
//
// C++ host side above the C ABI of include/dsvt_plugin.h: nothing but that header and the HIP runtime is used.
//   * loadWeights            include/helper.h:328-366  (.wts text file: count, then `name n hex hex ...`)
//   * createEngine           src/dsvt-ai-trt.cpp:532-1762: the network wiring, replayed on plugins created through the creator
//                            protocol (getFieldNames -> createPlugin, like include/plugin_helper.h:15-678): fused pillar feature net,
//                            per-layer QKV linear with the position-embedding table, set attention, fused encoder MLP, BEV ResNet +
//                            CenterHead on the HIP convolution, device decode / NMS.  Three precisions:
//                              --fp32 (default)  the reference's own arithmetic is fp32 (include/params.h:332 leaves USE_FP16 commented out):
//                                                every GEMM / convolution operand a (hi, lo) fp16 pair, three fp16 MFMAs per product,
//                                                fp32 accumulate, fp32 tensors in the DSVT stage -- boxes within 1e-3 of the fp32 oracle on
//                                                all nine columns (the Python host's COMPUTE_SPLIT, bench.py's headline mode)
//                              --fp8-head        the same with the head convolutions' two correction products on the fp8 scaled MFMA
//                                                (DsvtPipeline(head_mx=True): faster, yaw of ill-conditioned boxes above 1e-3)
//                              --fp16            fp16 operands (BASELINE configs[2] "fp16"; z / size 2e-3 .. 4e-3 from the oracle)
//                            --frames N: N frames per enqueue, their pillar rows concatenated (the Python host's frames=N)
//   * the -d loop            src/dsvt-ai-trt.cpp:1876-1960: every .bin of a directory -> enqueue (one HIP-graph launch per frame) ->
//                            <name>.txt in save_txt's layout (include/helper.h:441-481)
// BatchNorm folding, weight re-layout and the Q / sqrt(head_dim) scaling are done here in fp32 exactly as the Python pipeline
// (dsvt-ai-trt_amd/pipeline.py) does them: compiled with -ffp-contract=off, the two hosts hand bit-identical fields to the plugins and
// produce bit-identical boxes (tests/test_host_executor_gpu.py).
//
//   * --gpus N               frame-batch data parallelism over the N devices of a node inside this process: file j of a batch of N x frames on device j mod N, one
//                            RCCL gather of the result rows to device 0 per batch (runMultiGpu; --rccl-gather: the same path on one device, communicator of size 1)
//   dsvt_detect --wts dsvt.wts --data DIR --out DIR [--fp32 | --fp8-head | --fp16] [--frames N] [--in-flight K] [--gpus N] [--rccl-gather] [--ref-caps] [--no-graph] [--dump-raw] [--repeat N]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "dsvt_plugin.h"

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); exit(2); } } while (0)

[[noreturn]] static void die(const std::string& m) { fprintf(stderr, "dsvt_detect: %s\n", m.c_str()); exit(1); }

// ---- network constants (include/params.h) -------------------------------------------------------------------------
static const float X_MIN = -74.88f, X_MAX = 74.88f, Y_MIN = -74.88f, Y_MAX = 74.88f, Z_MIN = -5.0f, Z_MAX = 3.0f;
static const float VX = 0.32f, VY = 0.32f, VZ = 8.0f;
static const int GX = 468, GY = 468, GZ = 1;
static const int WINS[2][2][3] = {{{12, 12, 1}, {0, 0, 0}}, {{24, 24, 1}, {6, 6, 0}}};       // params.h:47-66 (shape, shift)
static const int L_SET = 36, C = 192, H = 8, C_FFN = 384, TOP_K = 500;
static const float SCORE_THR = 0.3f, NMS_THRESH = 0.01f;

struct Caps { int N = 196608, Nk = 196608, P = 65536, W = 2048, Vw = 576, S = 4096; };
// capacities of an engine that takes `frames` frames per enqueue (pipeline.py Caps.for_frames): per-frame point capacity, TOTAL kept-point /
// pillar / window / set capacities (every frame can touch every window cell of the grid: 40 x 40 cells for the 12 x 12 windows)
static Caps capsForFrames(int frames) {
    Caps c; if (frames <= 1) return c;
    const int nw = 1600, ppf = 65536;
    c.Nk = c.N * frames; c.P = ppf * frames; c.W = (nw * frames + 1023) / 1024 * 1024;
    c.S = ((ppf * frames + L_SET - 1) / L_SET + nw * frames + 1023) / 1024 * 1024;
    return c;
}
enum Mode { MODE_F16 = 0, MODE_SPLIT = 1, MODE_SPLIT_MX = 2 };

// ---- weights ----------------------------------------------------------------------------------------------------------
typedef std::vector<float> Vec;
typedef std::map<std::string, Vec> WeightMap;

static WeightMap loadWeights(const std::string& path) {      // include/helper.h:328-366
    std::ifstream in(path);
    if (!in) die("cannot open " + path);
    int count = 0;
    in >> count;
    if (count <= 0) die("bad weight file header");
    WeightMap w;
    std::string name, hex;
    for (int i = 0; i < count; ++i) {
        size_t n = 0;
        in >> name >> std::dec >> n;
        Vec v(n);
        for (size_t k = 0; k < n; ++k) {
            in >> hex;
            uint32_t bits = (uint32_t)strtoul(hex.c_str(), nullptr, 16);
            memcpy(&v[k], &bits, 4);
        }
        if (!in) die("truncated weight file at " + name);
        w[name] = std::move(v);
    }
    return w;
}
static const Vec& W_(const WeightMap& w, const std::string& k, size_t n) {
    auto it = w.find(k);
    if (it == w.end()) die("weight " + k + " missing");
    if (it->second.size() != n) die("weight " + k + ": " + std::to_string(it->second.size()) + " elements, expected " + std::to_string(n));
    return it->second;
}

// src/dsvt-ai-trt.cpp:99-122: scale = gamma / sqrt(var + eps), shift = beta - mean * gamma / sqrt(var + eps)
static void bnFold(const WeightMap& w, const std::string& p, int n, float eps, Vec& scale, Vec& shift) {
    const Vec &g = W_(w, p + ".weight", n), &b = W_(w, p + ".bias", n), &m = W_(w, p + ".running_mean", n), &v = W_(w, p + ".running_var", n);
    scale.resize(n); shift.resize(n);
    for (int i = 0; i < n; ++i) {
        const float sd = sqrtf(v[i] + eps);
        scale[i] = g[i] / sd;
        shift[i] = b[i] - m[i] * g[i] / sd;
    }
}
// FC [N,K] followed by a BatchNorm == FC with scaled rows and a bias
static void foldLinearBn(const WeightMap& w, const std::string& lin, const std::string& bn, int N, int K, float eps, bool bias, Vec& Wo, Vec& bo) {
    Vec s, sh; bnFold(w, bn, N, eps, s, sh);
    const Vec& W = W_(w, lin + ".weight", (size_t)N * K);
    Wo.resize((size_t)N * K); bo.resize(N);
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) Wo[(size_t)n * K + k] = W[(size_t)n * K + k] * s[n];
    if (bias) { const Vec& b = W_(w, lin + ".bias", N); for (int n = 0; n < N; ++n) bo[n] = sh[n] + b[n] * s[n]; }
    else bo = sh;
}
// torch Conv2d weight [Cout,Cin,k,k] (x per-Cout scale) -> DsvtConv2dPlugin rows [Cout][k*k][Cin]
static Vec convRows(const Vec& W, const Vec* s, int Cout, int Cin, int k) {
    Vec r((size_t)Cout * k * k * Cin);
    for (int co = 0; co < Cout; ++co) for (int ci = 0; ci < Cin; ++ci) for (int t = 0; t < k * k; ++t) {
        float v = W[((size_t)co * Cin + ci) * k * k + t];
        if (s) v = v * (*s)[co];
        r[((size_t)co * k * k + t) * Cin + ci] = v;
    }
    return r;
}
// torch ConvTranspose2d weight [Cin,Cout,k,k], stride == k (x per-Cout scale) -> pixel-shuffle rows [(dy*k+dx)*Cout + co][Cin]
static Vec deconvRows(const Vec& W, const Vec& s, int Cin, int Cout, int k) {
    Vec r((size_t)k * k * Cout * Cin);
    for (int ci = 0; ci < Cin; ++ci) for (int co = 0; co < Cout; ++co) for (int dy = 0; dy < k; ++dy) for (int dx = 0; dx < k; ++dx)
        r[(((size_t)dy * k + dx) * Cout + co) * Cin + ci] = W[(((size_t)ci * Cout + co) * k + dy) * k + dx] * s[co];
    return r;
}

// rows [R][taps][Cin] fp32 -> [R][taps][3 Cin] holding [w_hi | w_hi | w_lo] per tap (plugin.py split_weight_rows): w_hi = fp16(w), w_lo = fp16(w - w_hi)
static Vec splitRows(const Vec& rows, int taps, int cin) {
    const size_t R = rows.size() / ((size_t)taps * cin);
    Vec r(rows.size() * 3);
    for (size_t q = 0; q < R * taps; ++q)
        for (int ci = 0; ci < cin; ++ci) {
            const float w = rows[q * cin + ci];
            const float hi = (float)(_Float16)w, lo = (float)(_Float16)(w - hi);
            r[q * 3 * cin + ci] = hi; r[q * 3 * cin + cin + ci] = hi; r[q * 3 * cin + 2 * cin + ci] = lo;
        }
    return r;
}

// ---- plugins through the creator protocol -------------------------------------------------------------------------------
struct Fields {       // a PluginFieldCollection under construction (include/plugin_helper.h builds them the same way)
    std::vector<std::unique_ptr<std::vector<int>>> ints;
    std::vector<std::unique_ptr<Vec>> floats;
    std::vector<std::string> names;
    std::vector<DsvtPluginField> f;
    Fields& i(const char* n, std::vector<int> v) {
        ints.emplace_back(new std::vector<int>(std::move(v))); names.emplace_back(n);
        f.push_back({nullptr, ints.back()->data(), DSVT_FIELD_INT32, (int32_t)ints.back()->size()}); return *this;
    }
    Fields& i(const char* n, int v) { return i(n, std::vector<int>{v}); }
    Fields& fl(const char* n, Vec v) {
        floats.emplace_back(new Vec(std::move(v))); names.emplace_back(n);
        f.push_back({nullptr, floats.back()->data(), DSVT_FIELD_FLOAT32, (int32_t)floats.back()->size()}); return *this;
    }
    Fields& fl(const char* n, float v) { return fl(n, Vec{v}); }
};

struct Tensor {
    void* ptr = nullptr; DsvtDims dims{}; int32_t type = DSVT_FLOAT;
    size_t count() const { size_t n = 1; for (int k = 0; k < dims.nbDims; ++k) n *= (size_t)dims.d[k]; return n; }
    size_t bytes() const { return count() * (type == DSVT_HALF ? 2 : 4); }
};
static Tensor devTensor(std::vector<int> shape, int32_t type) {
    Tensor t; t.type = type; t.dims.nbDims = (int)shape.size();
    for (size_t k = 0; k < shape.size(); ++k) t.dims.d[k] = shape[k];
    HIP_OK(hipMalloc(&t.ptr, std::max<size_t>(t.bytes(), 256)));
    HIP_OK(hipMemset(t.ptr, 0, std::max<size_t>(t.bytes(), 256)));
    return t;
}

struct Op {
    DsvtPlugin* h = nullptr; std::string type; std::vector<Tensor> outs; void* ws = nullptr; bool built = false;
    Op() {}
    Op(const char* pluginType, Fields& fld, const char* layer, bool zeroFill = true) : type(pluginType) {
        // walk the advertised names like the reference factories do; fields the creator does not advertise (optional ones) follow
        for (size_t k = 0; k < fld.f.size(); ++k) fld.f[k].name = fld.names[k].c_str();
        if (!dsvtGetFieldNames(pluginType, DSVT_PLUGIN_VERSION)) die(std::string("no creator for ") + pluginType);
        DsvtPluginFieldCollection fc{(int32_t)fld.f.size(), fld.f.data()};
        h = dsvtCreatePlugin(pluginType, DSVT_PLUGIN_VERSION, layer, &fc);
        if (!h) die(std::string("createPlugin(") + pluginType + ") rejected its fields: " + dsvtGetLastCreateError());
        if (!zeroFill) dsvtPluginSetZeroFill(h, 0);
    }
    // first call: size outputs / workspace from the plugin (getOutputDimensions / getOutputDataType / getWorkspaceSize)
    void build(const std::vector<Tensor>& in, const std::vector<Tensor>* preset) {
        std::vector<DsvtDims> id; std::vector<int32_t> it; std::vector<DsvtPluginTensorDesc> ind, outd;
        for (const Tensor& t : in) { id.push_back(t.dims); it.push_back(t.type); ind.push_back({t.dims, t.type, DSVT_FORMAT_LINEAR, 1.0f}); }
        const int no = dsvtPluginGetNbOutputs(h);
        for (int o = 0; o < no; ++o) {
            if (preset) { outs.push_back((*preset)[o]); }
            else {
                DsvtDims d{};
                if (dsvtPluginGetOutputDimensions(h, o, id.data(), (int)id.size(), &d) != 0) die(type + ": getOutputDimensions");
                Tensor t; t.dims = d; t.type = dsvtPluginGetOutputDataType(h, o, it.data(), (int)it.size());
                HIP_OK(hipMalloc(&t.ptr, std::max<size_t>(t.bytes(), 256)));
                HIP_OK(hipMemset(t.ptr, 0, std::max<size_t>(t.bytes(), 256)));
                outs.push_back(t);
            }
            outd.push_back({outs.back().dims, outs.back().type, DSVT_FORMAT_LINEAR, 1.0f});
        }
        // configurePlugin(in, nbInputs, out, nbOutputs): the reference's engine build calls it once the shapes are known (points2Features.cu:257-260)
        if (dsvtPluginConfigurePlugin(h, ind.data(), (int)ind.size(), outd.data(), (int)outd.size()) != 0) die(type + ": configurePlugin");
        const size_t wsz = dsvtPluginGetWorkspaceSize(h, ind.data(), (int)ind.size(), outd.data(), (int)outd.size());
        HIP_OK(hipMalloc(&ws, std::max<size_t>(wsz, 256)));
        built = true;
    }
    const std::vector<Tensor>& operator()(const std::vector<Tensor>& in, hipStream_t s, const std::vector<Tensor>* preset = nullptr) {
        if (!built) build(in, preset);
        std::vector<DsvtPluginTensorDesc> ind, outd; std::vector<const void*> ip; std::vector<void*> op;
        for (const Tensor& t : in) { ind.push_back({t.dims, t.type, DSVT_FORMAT_LINEAR, 1.0f}); ip.push_back(t.ptr); }
        for (const Tensor& t : outs) { outd.push_back({t.dims, t.type, DSVT_FORMAT_LINEAR, 1.0f}); op.push_back(t.ptr); }
        const int rc = dsvtPluginEnqueue(h, ind.data(), outd.data(), ip.data(), op.data(), ws, (dsvtStream_t)s);
        if (rc != 0) die(type + ".enqueue returned " + std::to_string(rc));
        return outs;
    }
};

// ---- createEngine (src/dsvt-ai-trt.cpp:532-1762) ---------------------------------------------------------------------------
struct Layer { Op qkv, attn, mlp; Tensor table; };
struct Engine {
    Caps c; Mode mode; int frames;
    bool split() const { return mode != MODE_F16; }
    bool mx() const { return mode == MODE_SPLIT_MX; }
    Op voxelizer, pfn, part, map2bev, shared, heads0, heads1, topk, filter, nms;
    Layer layers[4][2];
    std::vector<Op> tableOps;                   // (split mode: the exact-fp32 linears whose outputs are the position tables)
    std::map<std::string, Op> conv;
    Tensor cat_bev;
    Tensor points, count;                       // static inputs: points [1, frames * N, 4], count [frames]
    std::vector<Tensor> result;                 // rows [frames,500,9], idx, count [frames]

    Engine(const WeightMap& w, const Caps& caps, hipStream_t s, Mode m, int nframes) : c(caps), mode(m), frames(nframes) {
        points = devTensor({1, frames * c.N, 4}, DSVT_FLOAT); count = devTensor({frames}, DSVT_INT32);
        {   // Points2Features (plugin_helper.h:15-123); range order (xmin,ymin,zmin,xmax,ymax,zmax) :33-38
            Fields f; f.i("max_points_num", c.N).i("max_points_num_voxel_filter", c.Nk).i("max_pillars_num", c.P).i("point_feature_num", 4)
                       .i("feature_num", 10).i("max_num_points_per_voxel", 48).fl("point_cloud_range", Vec{X_MIN, Y_MIN, Z_MIN, X_MAX, Y_MAX, Z_MAX})
                       .fl("voxel_size", Vec{VX, VY, VZ}).i("grid_size", {GX, GY, GZ});
            if (frames != 1) f.i("frames", frames);
            f.i("point_id_slots", 1);                                // (the fused pillar feature net reads slot 0 of the [P, 48] table: pipeline.py)
            voxelizer = Op("Points2FeaturesPlugin", f, "voxelGeneratorlayer", false);
        }
        {   // PFN: FC (no bias) + BN1d(1e-5) + ReLU twice, TorchScatterMax twice (:565-589), BN folded
            Vec W0, b0, W1, b1;
            foldLinearBn(w, "module.vfe.pfn_layers.0.linear", "module.vfe.pfn_layers.0.norm", 96, 10, 1e-5f, false, W0, b0);
            foldLinearBn(w, "module.vfe.pfn_layers.1.linear", "module.vfe.pfn_layers.1.norm", 192, 192, 1e-5f, false, W1, b1);
            Fields f; f.i("max_pillars_num", c.P).fl("weight0", W0).fl("bias0", b0).fl("weight1", W1).fl("bias1", b1).i("pack_small_pillars", 1);
            if (split()) f.i("split_precision", 1);
            pfn = Op("DsvtPillarFeatureNetPlugin", f, "pillar_feature_net_layer", false);
        }
        {   // WindowPartition + GetSet of both window configurations (:592-601) in one fused op: in-window coordinates, set indices / masks / counts
            Fields f; f.i("max_win_num", c.W).i("max_voxel_num_per_win", c.Vw).i("voxel_num_set", L_SET).i("max_set_num", c.S).i("max_pillars_num", c.P)
                       .i("sparse_shape", {GX, GY, GZ}).i("num_configs", 2)
                       .i("win_shapes", {WINS[0][0][0], WINS[0][0][1], WINS[0][0][2], WINS[1][0][0], WINS[1][0][1], WINS[1][0][2]})
                       .i("shift_lists", {WINS[0][1][0], WINS[0][1][1], WINS[0][1][2], WINS[1][1][0], WINS[1][1][1], WINS[1][1][2]}).i("frames", frames);
            part = Op("DsvtSetPartitionPlugin", f, "set_partition_layer", false);
        }
        if (split()) buildPosTablesF32(w, s); else buildPosTables(w, s);
        const float scale = (float)std::sqrt((double)C / H);        // np.float32(math.sqrt(C / H))
        for (int b = 0; b < 4; ++b) for (int l = 0; l < 2; ++l) {
            const std::string lp = "module.backbone_3d.stage_0." + std::to_string(b) + ".encoder_list." + std::to_string(l);
            Vec wi = W_(w, lp + ".win_attn.self_attn.in_proj_weight", (size_t)3 * C * C), bi = W_(w, lp + ".win_attn.self_attn.in_proj_bias", 3 * C);
            for (size_t k = 0; k < (size_t)C * C; ++k) wi[k] = wi[k] / scale;        // Q / sqrt(head_dim) after the bias (:386-405)
            for (int k = 0; k < C; ++k) bi[k] = bi[k] / scale;
            Layer& L = layers[b][l];
            {
                Fields f; f.i("max_rows", c.P).i("in_features", C).i("out_features", 3 * C).i("row_mult", 1).i("activation", 0).i("add_cols", 2 * C)
                           .i("num_layer_norms", 0).fl("ln_eps", 0.f).i("compute_type", split() ? 2 : 1).i("input_half", split() ? 0 : 1).i("output_mode", split() ? 0 : 1)
                           .i("add_gather_width", WINS[l][0][0]).fl("weight", wi).fl("bias", bi);
                L.qkv = Op("DsvtLinearPlugin", f, "linear_layer", false);
            }
            {
                Fields f; f.i("max_win_num", c.S).i("voxel_num_set", L_SET).i("channel_num", C).i("num_heads", H).i("axis_id", l)
                           .i("max_pillars_num", c.P).i("io_half", split() ? 2 : 1);
                L.attn = Op("DsvtSetAttentionPlugin", f, "set_attention_layer", false);
            }
            {
                Vec lg, lb;
                std::vector<std::string> norms = {lp + ".win_attn.norm1", lp + ".win_attn.norm2", lp + ".norm"};
                if (l == 1) norms.push_back("module.backbone_3d.residual_norm_stage_0." + std::to_string(b));       // block LayerNorm (:750-756)
                for (const std::string& n : norms) {
                    const Vec &g = W_(w, n + ".weight", C), &be = W_(w, n + ".bias", C);
                    lg.insert(lg.end(), g.begin(), g.end()); lb.insert(lb.end(), be.begin(), be.end());
                }
                Fields f; f.i("max_rows", c.P).i("has_block_norm", l == 1).fl("ln_eps", 0.f)
                           .fl("out_proj_weight", W_(w, lp + ".win_attn.self_attn.out_proj.weight", (size_t)C * C))
                           .fl("out_proj_bias", W_(w, lp + ".win_attn.self_attn.out_proj.bias", C))
                           .fl("linear1_weight", W_(w, lp + ".win_attn.linear1.weight", (size_t)C_FFN * C)).fl("linear1_bias", W_(w, lp + ".win_attn.linear1.bias", C_FFN))
                           .fl("linear2_weight", W_(w, lp + ".win_attn.linear2.weight", (size_t)C * C_FFN)).fl("linear2_bias", W_(w, lp + ".win_attn.linear2.bias", C))
                           .fl("ln_weights", lg).fl("ln_bias", lb).i("frames", frames);
                if (split()) f.i("split_precision", 1);
                L.mlp = Op("DsvtEncoderMlpPlugin", f, "encoder_mlp_layer", false);
            }
        }
        if (split()) buildHeadSplit(w); else buildHead(w);
        buildPost();
    }

    static Vec cellGrid(int k, int ncell) {      // (x - wx / 2, y - wy / 2) of every cell of window shape k, zero padded to ncell rows
        const int wx = WINS[k][0][0], wy = WINS[k][0][1];
        Vec g((size_t)ncell * 2, 0.f);
        for (int y = 0; y < wy; ++y) for (int x = 0; x < wx; ++x) { g[(size_t)(y * wx + x) * 2] = (float)x - wx / 2.0f; g[(size_t)(y * wx + x) * 2 + 1] = (float)y - wy / 2.0f; }
        return g;
    }

    // the eight position-embedding MLPs (:461-492, 603-637) evaluated once on the cell grids of the two window shapes
    void buildPosTables(const WeightMap& w, hipStream_t s) {
        const int ncell = 24 * 24;
        Vec pw, pb, ww, bb; std::vector<int> src;
        for (int b = 0; b < 4; ++b) for (int l = 0; l < 2; ++l) {
            const std::string pre = "module.backbone_3d.input_layer.posembed_layers.0." + std::to_string(b) + "." + std::to_string(l) + ".position_embedding_head";
            Vec Wa, ba; foldLinearBn(w, pre + ".0", pre + ".1", C, 2, 1e-5f, true, Wa, ba);
            const Vec &W3 = W_(w, pre + ".3.weight", (size_t)C * C), &b3 = W_(w, pre + ".3.bias", C);
            pw.insert(pw.end(), Wa.begin(), Wa.end()); pb.insert(pb.end(), ba.begin(), ba.end());
            ww.insert(ww.end(), W3.begin(), W3.end()); bb.insert(bb.end(), b3.begin(), b3.end());
            src.push_back(l);
        }
        Fields f; f.i("max_rows", ncell).i("num_layers", 8).i("layer_input", src).fl("pe_weight", pw).fl("pe_bias", pb).fl("weight", ww).fl("bias", bb);
        Op tab("DsvtPosEmbedPlugin", f, "pos_embed_layer");
        Tensor cnt = devTensor({1}, DSVT_INT32), grid[2];
        HIP_OK(hipMemcpy(cnt.ptr, &ncell, 4, hipMemcpyHostToDevice));
        for (int k = 0; k < 2; ++k) {
            const Vec g = cellGrid(k, ncell);
            grid[k] = devTensor({1, ncell, 2}, DSVT_FLOAT);
            HIP_OK(hipMemcpy(grid[k].ptr, g.data(), g.size() * 4, hipMemcpyHostToDevice));
        }
        const std::vector<Tensor>& o = tab({cnt, grid[0], grid[1]}, s);
        HIP_OK(hipStreamSynchronize(s));
        for (int b = 0; b < 4; ++b) for (int l = 0; l < 2; ++l) layers[b][l].table = o[2 * b + l];      // (the tables live on in the op's output buffers)
    }
    // fp32-grade frame: the same per-layer cell tables in fp32, from two exact-fp32 linears per layer (FC(2 -> 192) + BN + ReLU, FC(192 -> 192)),
    // construction time only -- what pipeline.py does for COMPUTE_SPLIT
    void buildPosTablesF32(const WeightMap& w, hipStream_t s) {
        const int ncell = 24 * 24;
        Tensor cnt = devTensor({1}, DSVT_INT32);
        HIP_OK(hipMemcpy(cnt.ptr, &ncell, 4, hipMemcpyHostToDevice));
        tableOps.reserve(16);
        for (int b = 0; b < 4; ++b) for (int l = 0; l < 2; ++l) {
            const std::string pre = "module.backbone_3d.input_layer.posembed_layers.0." + std::to_string(b) + "." + std::to_string(l) + ".position_embedding_head";
            Vec Wa, ba; foldLinearBn(w, pre + ".0", pre + ".1", C, 2, 1e-5f, true, Wa, ba);
            const Vec g = cellGrid(l, ncell);
            Tensor grid = devTensor({1, ncell, 2}, DSVT_FLOAT);
            HIP_OK(hipMemcpy(grid.ptr, g.data(), g.size() * 4, hipMemcpyHostToDevice));
            auto linear = [&](const Vec& Wt, const Vec& bs, int K, int act) {
                Fields f; f.i("max_rows", ncell).i("in_features", K).i("out_features", C).i("row_mult", 1).i("activation", act).i("add_cols", 0)
                           .i("num_layer_norms", 0).fl("ln_eps", 0.f).i("compute_type", 0).i("input_half", 0).i("output_mode", 0).i("add_gather_width", 0)
                           .fl("weight", Wt).fl("bias", bs);
                tableOps.emplace_back("DsvtLinearPlugin", f, "linear_layer");
                return &tableOps.back();
            };
            Op* a = linear(Wa, ba, 2, 1);
            const Tensor h1 = (*a)({grid, cnt}, s)[0];
            Op* fc = linear(W_(w, pre + ".3.weight", (size_t)C * C), W_(w, pre + ".3.bias", C), C, 0);
            layers[b][l].table = (*fc)({h1, cnt}, s)[0];                             // (lives on in the op's output buffer)
        }
        HIP_OK(hipStreamSynchronize(s));
    }

    Op convOp(const Vec& rows, const Vec& bias, int Hh, int cin, int cout, int k, int stride, int pad, int shuffle, bool relu, bool res,
              int ostride, int ooff, bool f32out, int splitOut = 0, int splitRes = 0, int splitIn = 0) {
        Fields f; f.i("in_height", Hh).i("in_width", Hh).i("in_channels", cin).i("out_channels", cout).i("kernel_size", k).i("stride", stride)
                   .i("padding", pad).i("pixel_shuffle", shuffle).i("relu", relu).i("has_residual", res).i("out_channel_stride", ostride)
                   .i("out_channel_offset", ooff).i("out_f32", f32out).fl("weight", rows);
        if (splitOut) f.i("split_output", splitOut);
        if (splitRes) f.i("split_residual", splitRes);
        if (splitIn) f.i("split_input", splitIn);
        f.fl("bias", bias);
        return Op("DsvtConv2dPlugin", f, "conv2d_layer");
    }
    Op convBn(const WeightMap& w, const std::string& cv, const std::string& bn, int Hh, int cin, int cout, int k, int stride, bool relu, bool res) {
        Vec s, sh; bnFold(w, bn, cout, 1e-3f, s, sh);                                                   // :191,208
        return convOp(convRows(W_(w, cv + ".weight", (size_t)cout * cin * k * k), &s, cout, cin, k), sh, Hh, cin, cout, k, stride, k / 2, 1, relu, res, cout, 0, false);
    }

    // the five CenterHead stems as one 64 -> 320 layer, the five output convolutions as one block-diagonal 320 -> 18 layer (:1378-1468)
    void headRows(const WeightMap& w, Vec& W0r, Vec& b0, Vec& W1r, Vec& b1) {
        const char* names[5] = {"center", "center_z", "dim", "rot", "hm"}; const int outs[5] = {2, 1, 3, 2, 10};     // the iou head is dead (:1440-1452)
        Vec W0, W1((size_t)18 * 320 * 9, 0.f); b1.assign(18, 0.f); b0.clear();
        int o = 0;
        for (int k = 0; k < 5; ++k) {
            const std::string p = std::string("module.dense_head.heads_list.0.") + names[k];
            Vec s, sh; bnFold(w, p + ".0.1", 64, 1e-3f, s, sh);
            const Vec& Wk = W_(w, p + ".0.0.weight", (size_t)64 * 64 * 9);
            for (int co = 0; co < 64; ++co) for (size_t e = 0; e < (size_t)64 * 9; ++e) W0.push_back(Wk[(size_t)co * 64 * 9 + e] * s[co]);
            b0.insert(b0.end(), sh.begin(), sh.end());
            const Vec &Wl = W_(w, p + ".1.weight", (size_t)outs[k] * 64 * 9), &bl = W_(w, p + ".1.bias", outs[k]);
            for (int r = 0; r < outs[k]; ++r) {                                    // block-diagonal second convolutions
                for (int ci = 0; ci < 64; ++ci) for (int t = 0; t < 9; ++t) W1[((size_t)(o + r) * 320 + 64 * k + ci) * 9 + t] = Wl[((size_t)r * 64 + ci) * 9 + t];
                b1[o + r] = bl[r];
            }
            o += outs[k];
        }
        W0r = convRows(W0, nullptr, 320, 64, 3); W1r = convRows(W1, nullptr, 18, 320, 3);
    }

    // Map2BEV + BEV ResNet + CenterHead (:1128-1468), fp16 operands
    void buildHead(const WeightMap& w) {
        { Fields f; f.i("max_pillars_num", c.P).i("channel_num", C).i("grid_size_x", GX).i("grid_size_y", GY);
          if (frames != 1) f.i("frames", frames);
          f.i("persistent_output", 1); map2bev = Op("Map2BevPlugin", f, "map2bev_layer"); }
        const int blk[3][4] = {{192, 128, 1, 2}, {128, 128, 2, 3}, {128, 256, 2, 3}};
        int Hh = GY;
        for (int i = 0; i < 3; ++i) {
            const int cin = blk[i][0], cout = blk[i][1], stride = blk[i][2], nb = blk[i][3];
            for (int j = 0; j < nb; ++j) {
                const std::string p = "module.backbone_2d.blocks." + std::to_string(i) + "." + std::to_string(j);
                const int st = j == 0 ? stride : 1, ci = j == 0 ? cin : cout;
                conv[p + ".1"] = convBn(w, p + ".conv1", p + ".bn1", Hh, ci, cout, 3, st, true, false);
                const int Ho = (Hh + 2 - 3) / st + 1;
                if (j == 0) conv[p + ".d"] = convBn(w, p + ".downsample_layer.0", p + ".downsample_layer.1", Hh, ci, cout, 1, st, false, false);
                conv[p + ".2"] = convBn(w, p + ".conv2", p + ".bn2", Ho, cout, cout, 3, 1, true, true);     // + identity, ReLU (:1165-1166)
                Hh = Ho;
            }
            const int k = 1 << i;
            const std::string p = "module.backbone_2d.deblocks." + std::to_string(i);
            Vec s, sh; bnFold(w, p + ".1", 128, 1e-3f, s, sh);
            conv[p] = convOp(deconvRows(W_(w, p + ".0.weight", (size_t)cout * 128 * k * k), s, cout, 128, k), sh, Hh, cout, 128, 1, 1, 0, k, true, false, 384, 128 * i, false);
        }
        shared = convBn(w, "module.dense_head.shared_conv.0", "module.dense_head.shared_conv.1", GY, 384, 64, 3, 1, true, false);
        Vec W0r, b0, W1r, b1; headRows(w, W0r, b0, W1r, b1);
        heads0 = convOp(W0r, b0, GY, 64, 320, 3, 1, 1, 1, true, false, 320, 0, false);
        heads1 = convOp(W1r, b1, GY, 320, 18, 3, 1, 1, 1, false, false, 18, 0, true);
        cat_bev = devTensor({frames, GY, GX, 384}, DSVT_HALF);
    }

    // the same stage at fp32 grade (pipeline.py _build_hip_head_split): every convolution over 3 Cin operand channels [hi | lo | hi] x rows
    // [w_hi | w_hi | w_lo] (three fp16 MFMAs per product), the activations travelling as fp16 triples written by the producing layer's epilogue, the
    // residual read as hi + lo, the last layer fp32.  --fp8-head: the third plane holds the fp8 operands (x8), the 3 x 3 stride-1 layers with more than
    // 32 output channels and the 1 x 1 stride-1 layers take the REAL fp32 rows (split_input = 2: the plugin packs the fp16 + fp8 weights itself)
    Op convSplit(const Vec& rows, const Vec& bias, int Hh, int cin, int cout, int k, int stride, bool relu, bool res, bool f32out,
                 int plane = 0, bool lo = true, bool resOnly = false, int shuffle = 1, int ooff = 0) {
        if (!plane) plane = cout;
        const int up2 = shuffle * shuffle;
        const bool wide = mx() && stride == 1 && ((k == 3 && cout > 32 && up2 == 1) || (k == 1 && (up2 * cout) % 128 == 0));
        // three fp16 products (default): [hi | lo | -] out (4), the third plane's phases of the input alias plane 0 (split_input 1) -- pipeline.py conv()
        const int splitOut = f32out ? 0 : (mx() ? (resOnly ? 4 : lo ? 2 : 3) : 4);
        return convOp(wide ? rows : splitRows(rows, k * k, cin), bias, Hh, 3 * cin, cout, k, stride, k / 2, shuffle, relu, res, f32out ? plane : 3 * plane, ooff, f32out,
                      splitOut, res ? 1 : 0, mx() ? (wide ? 2 : 1) : 1);
    }
    Op convBnSplit(const WeightMap& w, const std::string& cv, const std::string& bn, int Hh, int cin, int cout, int k, int stride, bool relu, bool res,
                   bool lo = true, bool resOnly = false) {
        Vec s, sh; bnFold(w, bn, cout, 1e-3f, s, sh);
        return convSplit(convRows(W_(w, cv + ".weight", (size_t)cout * cin * k * k), &s, cout, cin, k), sh, Hh, cin, cout, k, stride, relu, res, false, 0, lo, resOnly);
    }
    void buildHeadSplit(const WeightMap& w) {
        { Fields f; f.i("max_pillars_num", c.P).i("channel_num", C).i("grid_size_x", GX).i("grid_size_y", GY);
          if (frames != 1) f.i("frames", frames);
          f.i("persistent_output", 1).i("split_output", mx() ? 2 : 3); map2bev = Op("Map2BevPlugin", f, "map2bev_layer"); }
        const int blk[3][4] = {{192, 128, 1, 2}, {128, 128, 2, 3}, {128, 256, 2, 3}};
        int Hh = GY;
        for (int i = 0; i < 3; ++i) {
            const int cin = blk[i][0], cout = blk[i][1], stride = blk[i][2], nb = blk[i][3];
            for (int j = 0; j < nb; ++j) {
                const std::string p = "module.backbone_2d.blocks." + std::to_string(i) + "." + std::to_string(j);
                const int st = j == 0 ? stride : 1, ci = j == 0 ? cin : cout;
                conv[p + ".1"] = convBnSplit(w, p + ".conv1", p + ".bn1", Hh, ci, cout, 3, st, true, false, /*lo=*/false);        // (read by conv2 only)
                const int Ho = (Hh + 2 - 3) / st + 1;
                if (j == 0) conv[p + ".d"] = convBnSplit(w, p + ".downsample_layer.0", p + ".downsample_layer.1", Hh, ci, cout, 1, st, false, false, true, /*resOnly=*/true);
                conv[p + ".2"] = convBnSplit(w, p + ".conv2", p + ".bn2", Ho, cout, cout, 3, 1, true, true);                      // + identity, ReLU (:1165-1166)
                Hh = Ho;
            }
            const int k = 1 << i;
            const std::string p = "module.backbone_2d.deblocks." + std::to_string(i);
            Vec s, sh; bnFold(w, p + ".1", 128, 1e-3f, s, sh);
            conv[p] = convSplit(deconvRows(W_(w, p + ".0.weight", (size_t)cout * 128 * k * k), s, cout, 128, k), sh, Hh, cout, 128, 1, 1, true, false, false,
                                /*plane=*/384, /*lo=*/false, false, /*shuffle=*/k, /*ooff=*/128 * i);                              // (the concat buffer: read by the shared conv only)
        }
        shared = convBnSplit(w, "module.dense_head.shared_conv.0", "module.dense_head.shared_conv.1", GY, 384, 64, 3, 1, true, false, /*lo=*/false);
        Vec W0r, b0, W1r, b1; headRows(w, W0r, b0, W1r, b1);
        heads0 = convSplit(W0r, b0, GY, 64, 320, 3, 1, true, false, false, 0, true, /*resOnly=*/true);       // (read by heads1 only: three fp16 products over hi and lo)
        heads1 = convSplit(W1r, b1, GY, 320, 18, 3, 1, false, false, /*f32out=*/true);
        cat_bev = devTensor({frames, GY, GX, 3 * 384}, DSVT_HALF);
    }
    // decode (:1479-1669) + FilterBoxByScore (:1684-1736) + nms_cpu's device twin
    void buildPost() {
        { Fields f; f.i("feature_height", GY).i("feature_width", GX).i("channel_num", 18).i("class_num", 10).i("max_top_k", TOP_K).i("center_offset", 0)
                     .i("center_z_offset", 2).i("dim_offset", 3).i("rot_offset", 6).i("hm_offset", 8); topk = Op("CenterHeadTopKPlugin", f, "center_head_topk_layer"); }
        { Fields f; f.i("max_top_k", TOP_K).fl("point_cloud_range", Vec{X_MIN, X_MAX, Y_MIN, Y_MAX, Z_MIN, Z_MAX}).fl("voxel_size", Vec{VX, VY, VZ})
                     .fl("score_threshold", SCORE_THR); filter = Op("FilterBoxByScorePlugin", f, "filter_box_by_score_layer"); }
        { Fields f; f.i("max_boxes", TOP_K).fl("nms_thresh", NMS_THRESH); nms = Op("RotatedNmsPlugin", f, "rotated_nms_layer"); }
    }

    // one forward (`frames` frames): every op enqueued on `s` (context->enqueueV2, src/dsvt-ai-trt.cpp:1928)
    void enqueue(hipStream_t s) {
        const std::vector<Tensor>& v = voxelizer({points, count}, s);             // feat, pidx, coords, pcnt, P, Nk
        const Tensor coords = v[2], Pn = v[4];
        const std::vector<Tensor>& pf = pfn({v[0], v[1], v[3], Pn}, s);           // pillar features fp32 (, fp16)
        const std::vector<Tensor>& po = part({coords, Pn}, s);                    // per configuration k: c2d, inds, mask, S at 4k .. 4k+3
        Tensor x = pf[0], xh = split() ? pf[0] : pf[1];                           // (fp32 grade: the GEMM operand IS the fp32 residual stream)
        for (int b = 0; b < 4; ++b) {
            const Tensor xb = x;
            const Tensor* g = &po[4 * (b % 2) + 1];                               // inds, mask, S of window configuration b % 2
            for (int l = 0; l < 2; ++l) {
                Layer& L = layers[b][l];
                const Tensor qkv = L.qkv({xh, Pn, L.table, po[4 * l]}, s)[0];     // position table of window configuration l (:603-637)
                const Tensor att = L.attn({qkv, g[0], g[1], g[2]}, s)[0];
                const std::vector<Tensor>& o = l == 1 ? L.mlp({att, Pn, x, xb}, s) : L.mlp({att, Pn, x}, s);
                x = o[0]; xh = split() ? o[0] : o[1];
            }
        }
        Tensor t = map2bev({xh, coords, Pn}, s)[0];                               // [frames,468,468,192] fp16 NHWC, or the [hi | lo | hi / x8] triple map
        const int nb[3] = {2, 3, 3};
        std::vector<Tensor> cat = {cat_bev};
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < nb[i]; ++j) {
                const std::string p = "module.backbone_2d.blocks." + std::to_string(i) + "." + std::to_string(j);
                const Tensor y = conv[p + ".1"]({t}, s)[0];
                const Tensor idn = j == 0 ? conv[p + ".d"]({t}, s)[0] : t;
                t = conv[p + ".2"]({y, idn}, s)[0];
            }
            conv["module.backbone_2d.deblocks." + std::to_string(i)]({t}, s, &cat);      // deblock + concat (:1363)
        }
        const Tensor head = heads1({heads0({shared({cat_bev}, s)[0]}, s)[0]}, s)[0];
        const std::vector<Tensor>& fb = filter(topk({head}, s), s);
        result = nms({fb[0], fb[1]}, s);
    }
};

// ---- host I/O (include/helper.h:28-72, 441-481) -----------------------------------------------------------------------
static std::vector<float> loadBin(const std::string& path, int maxPoints, int& n) {
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) die("cannot open " + path);
    const std::streamsize sz = in.tellg();
    if (sz % 16) die(path + ": size is not a multiple of 4 floats");
    n = (int)(sz / 16);
    if (n > maxPoints) die(path + ": " + std::to_string(n) + " points exceed the cap " + std::to_string(maxPoints));      // helper.h:47-53
    std::vector<float> p((size_t)n * 4);
    in.seekg(0); in.read(reinterpret_cast<char*>(p.data()), sz);
    return p;
}
static void saveTxt(const std::string& path, const float* rows, int k, double ms) {
    FILE* fh = fopen(path.c_str(), "w");
    if (!fh) die("cannot write " + path);
    fprintf(fh, "%.6f\n", ms);
    for (int r = 0; r < k; ++r) {
        const float* b = rows + (size_t)r * 9;
        fprintf(fh, "%.6f,  %.6f,  %.6f,  %.6f,  %.6f,  %.6f,  %.6f,  %d,  %.6f\n", b[0], b[1], b[2], b[3], b[4], b[5], b[6], (int)b[7], b[8]);
    }
    fclose(fh);
}


// --in-flight K (K > 1): K engines on K streams, the groups of `frames` frames go to them round-robin -- the upload of one group and the download of another's boxes
// run under a third's forward.  The reference's loop (src/dsvt-ai-trt.cpp:1884-1970) is synchronous; this is the same forward per group (same kernels, same boxes: every
// engine is the engine of the synchronous loop), only the waiting is taken out.  Timed: the LAST pass over the directory, wall clock from the first host copy to the last
// download (frames cached in host memory after the first pass; the .txt files are written after the clock stops).
static int runPipelined(const WeightMap& w, const Caps& caps, Mode mode, int frames, int inflight, bool graph, bool raw, int repeat,
                        const std::string& data, const std::string& out, const std::vector<std::string>& files) {
    struct Slot {
        hipStream_t s = nullptr; std::unique_ptr<Engine> eng; hipGraphExec_t exec = nullptr;
        float* hpts = nullptr; int* hcnt = nullptr; float* hrows = nullptr; int* hkept = nullptr;
        long group = -1;                                                          // the group in flight on this slot, -1 = none
    };
    std::vector<Slot> slots(inflight);
    for (Slot& sl : slots) {
        HIP_OK(hipStreamCreate(&sl.s));
        sl.eng.reset(new Engine(w, caps, sl.s, mode, frames));
        for (int k = 0; k < 2; ++k) sl.eng->enqueue(sl.s);                        // warm-up on empty frames (sizes every buffer)
        HIP_OK(hipStreamSynchronize(sl.s));
        if (graph) {
            hipGraph_t g;
            HIP_OK(hipStreamBeginCapture(sl.s, hipStreamCaptureModeThreadLocal));
            sl.eng->enqueue(sl.s);
            HIP_OK(hipStreamEndCapture(sl.s, &g));
            HIP_OK(hipGraphInstantiate(&sl.exec, g, nullptr, nullptr, 0));
        }
        HIP_OK(hipHostMalloc(&sl.hpts, (size_t)frames * caps.N * 16)); HIP_OK(hipHostMalloc(&sl.hcnt, 4 * frames));
        HIP_OK(hipHostMalloc(&sl.hrows, (size_t)frames * TOP_K * 9 * 4)); HIP_OK(hipHostMalloc(&sl.hkept, 4 * frames));
    }
    const size_t nfile = files.size(), ngroup = (nfile + frames - 1) / frames;
    std::vector<std::vector<float>> cache(nfile); std::vector<int> npts(nfile, 0);
    std::vector<std::vector<float>> rows(nfile); std::vector<int> kept(nfile, 0);
    auto collect = [&](Slot& sl) {                                                // wait for the slot's group and keep its boxes
        if (sl.group < 0) return;
        HIP_OK(hipStreamSynchronize(sl.s));
        const size_t f0 = (size_t)sl.group * frames;
        for (int f = 0; f < frames && f0 + f < nfile; ++f) {
            kept[f0 + f] = sl.hkept[f];
            rows[f0 + f].assign(sl.hrows + (size_t)f * TOP_K * 9, sl.hrows + (size_t)f * TOP_K * 9 + (size_t)sl.hkept[f] * 9);
        }
        sl.group = -1;
    };
    double lastMs = 0;
    for (int r = 0; r < repeat; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t g = 0; g < ngroup; ++g) {
            Slot& sl = slots[g % inflight];
            collect(sl);
            const size_t f0 = g * frames;
            const int nf = (int)std::min<size_t>(frames, nfile - f0);
            for (int f = 0; f < frames; ++f) sl.hcnt[f] = 0;
            for (int f = 0; f < nf; ++f) {
                if (cache[f0 + f].empty()) cache[f0 + f] = loadBin(data + "/" + files[f0 + f], caps.N, npts[f0 + f]);
                memcpy(sl.hpts + (size_t)f * caps.N * 4, cache[f0 + f].data(), (size_t)npts[f0 + f] * 16); sl.hcnt[f] = npts[f0 + f];
                HIP_OK(hipMemcpyAsync((char*)sl.eng->points.ptr + (size_t)f * caps.N * 16, sl.hpts + (size_t)f * caps.N * 4, (size_t)sl.hcnt[f] * 16, hipMemcpyHostToDevice, sl.s));
            }
            HIP_OK(hipMemcpyAsync(sl.eng->count.ptr, sl.hcnt, 4 * frames, hipMemcpyHostToDevice, sl.s));
            if (graph) HIP_OK(hipGraphLaunch(sl.exec, sl.s)); else sl.eng->enqueue(sl.s);
            HIP_OK(hipMemcpyAsync(sl.hkept, sl.eng->result[2].ptr, 4 * frames, hipMemcpyDeviceToHost, sl.s));
            HIP_OK(hipMemcpyAsync(sl.hrows, sl.eng->result[0].ptr, (size_t)frames * TOP_K * 9 * 4, hipMemcpyDeviceToHost, sl.s));
            sl.group = (long)g;
        }
        for (Slot& sl : slots) collect(sl);
        lastMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    for (size_t i = 0; i < nfile; ++i) {
        const std::string stem = files[i].substr(0, files[i].size() - 4);
        saveTxt(out + "/" + stem + ".txt", rows[i].data(), kept[i], lastMs / nfile);
        if (raw) {
            FILE* fh = fopen((out + "/" + stem + ".rows").c_str(), "wb");
            if (!fh) die("cannot write raw rows");
            fwrite(&kept[i], 4, 1, fh); fwrite(rows[i].data(), 4, (size_t)kept[i] * 9, fh); fclose(fh);
        }
        printf("%s: %d points -> %d boxes, %.3f ms (its share of the pipelined pass)\n", stem.c_str(), npts[i], kept[i], lastMs / nfile);
    }
    printf("dsvt_detect: %zu frames, %s, %d frame(s) per forward, %s, %d forwards in flight: %.3f ms per frame, %.1f frames/s (host copy + upload + forward + download of the final boxes, "
           "wall clock of the last pass%s)\n",
           nfile, mode == MODE_F16 ? "fp16" : mode == MODE_SPLIT ? "fp32 grade (f16x3)" : "fp32 grade with fp8 head corrections", frames, graph ? "HIP graph" : "host launches", inflight,
           lastMs / std::max<size_t>(nfile, 1), 1e3 * nfile / std::max(lastMs, 1e-9), repeat > 1 ? ", frames cached in host memory" : ", disk reads inside");
    return 0;
}

// --gpus N (round 6; BASELINE configs[3] from the C++ host): frame-batch data parallelism inside ONE process -- an engine, a stream and a HIP graph per device, a batch =
// N x `frames` consecutive files, file j of the batch on device j mod N (slot j / N of that device's forward), and ONE collective per batch: the result rows and kept
// counts of every device gathered to device 0 with ncclGather (RCCL over xGMI; grouped, one thread drives all devices), from where they go to the host and into the
// .txt files.  The reference binds one device (cudaSetDevice(DEVICE), src/dsvt-ai-trt.cpp:1783; include/params.h:333) and has nothing to gather; the loop shape -- a
// result per frame, every frame -- is its -d loop (:1884-1970).  --rccl-gather forces this path with N = 1 (a communicator of size 1: how a one-GPU box runs it).
#define NCCL_OK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { \
    fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #expr, ncclGetErrorString(r_)); exit(2); } } while (0)
static int runMultiGpu(const WeightMap& w, const Caps& caps, Mode mode, int frames, int ngpu, bool graph, bool raw, int repeat,
                       const std::string& data, const std::string& out, const std::vector<std::string>& files) {
    struct Dev {
        hipStream_t s = nullptr; std::unique_ptr<Engine> eng; hipGraphExec_t exec = nullptr;
        float* hpts = nullptr; int* hcnt = nullptr;
    };
    std::vector<Dev> devs(ngpu);
    std::vector<int> ids(ngpu); for (int d = 0; d < ngpu; ++d) ids[d] = d;
    std::vector<ncclComm_t> comms(ngpu);
    NCCL_OK(ncclCommInitAll(comms.data(), ngpu, ids.data()));
    const size_t rowFloats = (size_t)frames * TOP_K * 9;
    float* gRows = nullptr; int* gKept = nullptr; float* hRows = nullptr; int* hKept = nullptr;
    for (int d = 0; d < ngpu; ++d) {
        Dev& dv = devs[d];
        HIP_OK(hipSetDevice(d));
        HIP_OK(hipStreamCreate(&dv.s));
        dv.eng.reset(new Engine(w, caps, dv.s, mode, frames));
        for (int k = 0; k < 2; ++k) dv.eng->enqueue(dv.s);                        // warm-up on empty frames (sizes every buffer)
        HIP_OK(hipStreamSynchronize(dv.s));
        if (graph) {
            hipGraph_t g;
            HIP_OK(hipStreamBeginCapture(dv.s, hipStreamCaptureModeThreadLocal));
            dv.eng->enqueue(dv.s);
            HIP_OK(hipStreamEndCapture(dv.s, &g));
            HIP_OK(hipGraphInstantiate(&dv.exec, g, nullptr, nullptr, 0));
        }
        HIP_OK(hipHostMalloc(&dv.hpts, (size_t)frames * caps.N * 16)); HIP_OK(hipHostMalloc(&dv.hcnt, 4 * frames));
        if (d == 0) {                                                             // the gather's receive buffers, allocated once
            HIP_OK(hipMalloc(&gRows, (size_t)ngpu * rowFloats * 4)); HIP_OK(hipMalloc(&gKept, (size_t)ngpu * frames * 4));
            HIP_OK(hipHostMalloc(&hRows, (size_t)ngpu * rowFloats * 4)); HIP_OK(hipHostMalloc(&hKept, (size_t)ngpu * frames * 4));
        }
    }
    const size_t nfile = files.size(), per = (size_t)frames * ngpu, nbatch = (nfile + per - 1) / per;
    std::vector<std::vector<float>> cache(nfile); std::vector<int> npts(nfile, 0);
    std::vector<std::vector<float>> rows(nfile); std::vector<int> kept(nfile, 0);
    double lastMs = 0;
    for (int r = 0; r < repeat; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t b = 0; b < nbatch; ++b) {
            const size_t f0 = b * per;
            for (int d = 0; d < ngpu; ++d) {
                Dev& dv = devs[d];
                HIP_OK(hipSetDevice(d));
                for (int sl = 0; sl < frames; ++sl) {
                    const size_t f = f0 + (size_t)sl * ngpu + d;                  // file j = sl * N + d of the batch: device j mod N, slot j / N
                    dv.hcnt[sl] = 0;
                    if (f >= nfile) continue;
                    if (cache[f].empty()) cache[f] = loadBin(data + "/" + files[f], caps.N, npts[f]);
                    memcpy(dv.hpts + (size_t)sl * caps.N * 4, cache[f].data(), (size_t)npts[f] * 16); dv.hcnt[sl] = npts[f];
                    HIP_OK(hipMemcpyAsync((char*)dv.eng->points.ptr + (size_t)sl * caps.N * 16, dv.hpts + (size_t)sl * caps.N * 4, (size_t)npts[f] * 16, hipMemcpyHostToDevice, dv.s));
                }
                HIP_OK(hipMemcpyAsync(dv.eng->count.ptr, dv.hcnt, 4 * frames, hipMemcpyHostToDevice, dv.s));
                if (graph) HIP_OK(hipGraphLaunch(dv.exec, dv.s)); else dv.eng->enqueue(dv.s);
            }
            // the one collective of the batch: rows [frames, 500, 9] and kept [frames] of every device -> device 0, rank-major
            NCCL_OK(ncclGroupStart());
            for (int d = 0; d < ngpu; ++d) {
                NCCL_OK(ncclGather(devs[d].eng->result[0].ptr, gRows, rowFloats, ncclFloat, 0, comms[d], devs[d].s));
                NCCL_OK(ncclGather(devs[d].eng->result[2].ptr, gKept, (size_t)frames, ncclInt32, 0, comms[d], devs[d].s));
            }
            NCCL_OK(ncclGroupEnd());
            HIP_OK(hipSetDevice(0));
            HIP_OK(hipMemcpyAsync(hKept, gKept, (size_t)ngpu * frames * 4, hipMemcpyDeviceToHost, devs[0].s));
            HIP_OK(hipMemcpyAsync(hRows, gRows, (size_t)ngpu * rowFloats * 4, hipMemcpyDeviceToHost, devs[0].s));
            for (int d = ngpu - 1; d >= 0; --d) { HIP_OK(hipSetDevice(d)); HIP_OK(hipStreamSynchronize(devs[d].s)); }      // (device 0 last: the downloads)
            for (int d = 0; d < ngpu; ++d)
                for (int sl = 0; sl < frames; ++sl) {
                    const size_t f = f0 + (size_t)sl * ngpu + d;
                    if (f >= nfile) continue;
                    kept[f] = hKept[(size_t)d * frames + sl];
                    const float* src = hRows + ((size_t)d * frames + sl) * TOP_K * 9;
                    rows[f].assign(src, src + (size_t)kept[f] * 9);
                }
        }
        lastMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    for (size_t i = 0; i < nfile; ++i) {
        const std::string stem = files[i].substr(0, files[i].size() - 4);
        saveTxt(out + "/" + stem + ".txt", rows[i].data(), kept[i], lastMs / nfile);
        if (raw) {
            FILE* fh = fopen((out + "/" + stem + ".rows").c_str(), "wb");
            if (!fh) die("cannot write raw rows");
            fwrite(&kept[i], 4, 1, fh); fwrite(rows[i].data(), 4, (size_t)kept[i] * 9, fh); fclose(fh);
        }
        printf("%s: %d points -> %d boxes, %.3f ms (its share of the pass), device %d\n", stem.c_str(), npts[i], kept[i], lastMs / nfile, (int)((i % per) % ngpu));
    }
    printf("dsvt_detect: %zu frames on %d GPU(s), %s, %d frame(s) per forward, %s, one RCCL gather per batch of %zu frames: %.3f ms per frame, %.1f frames/s (host copy + upload + forward + "
           "gather to device 0 + download of the final boxes, wall clock of the last pass%s)\n",
           nfile, ngpu, mode == MODE_F16 ? "fp16" : mode == MODE_SPLIT ? "fp32 grade (f16x3)" : "fp32 grade with fp8 head corrections", frames, graph ? "HIP graph" : "host launches", per,
           lastMs / std::max<size_t>(nfile, 1), 1e3 * nfile / std::max(lastMs, 1e-9), repeat > 1 ? ", frames cached in host memory" : ", disk reads inside");
    for (int d = 0; d < ngpu; ++d) ncclCommDestroy(comms[d]);
    return 0;
}

int main(int argc, char** argv) {
    std::string wts, data, out; bool refCaps = false, graph = true, raw = false, rcclGather = false; int repeat = 1, frames = 1, inflight = 1, gpus = 1; Mode mode = MODE_SPLIT;
    for (int a = 1; a < argc; ++a) {
        const std::string s = argv[a];
        if (s == "--wts" && a + 1 < argc) wts = argv[++a];
        else if (s == "--data" && a + 1 < argc) data = argv[++a];
        else if (s == "--out" && a + 1 < argc) out = argv[++a];
        else if (s == "--ref-caps") refCaps = true;
        else if (s == "--no-graph") graph = false;
        else if (s == "--dump-raw") raw = true;
        else if (s == "--fp32") mode = MODE_SPLIT;
        else if (s == "--fp8-head") mode = MODE_SPLIT_MX;
        else if (s == "--fp16") mode = MODE_F16;
        else if (s == "--frames" && a + 1 < argc) frames = atoi(argv[++a]);
        else if (s == "--repeat" && a + 1 < argc) repeat = atoi(argv[++a]);
        else if (s == "--in-flight" && a + 1 < argc) inflight = atoi(argv[++a]);
        else if (s == "--gpus" && a + 1 < argc) gpus = atoi(argv[++a]);
        else if (s == "--rccl-gather") rcclGather = true;
        else die("usage: dsvt_detect --wts F --data DIR --out DIR [--fp32 | --fp8-head | --fp16] [--frames N] [--in-flight K] [--gpus N] [--rccl-gather] [--ref-caps] [--no-graph] [--dump-raw] [--repeat N]");
    }
    if (wts.empty() || data.empty() || out.empty()) die("--wts, --data and --out are required (the reference's dsvt.wts is not shipped)");
    if (frames < 1 || frames > 16) die("--frames must be 1 .. 16");
    if (inflight < 1 || inflight > 4) die("--in-flight must be 1 .. 4");
    if (refCaps && frames != 1) die("--ref-caps is the reference's one-frame configuration (its kernels only ever read frame 0's counts)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) die("no GPU visible (there is no CPU path)");
    if (gpus < 1 || gpus > 64) die("--gpus must be 1 .. 64");
    if (gpus > ndev) die("--gpus " + std::to_string(gpus) + ": only " + std::to_string(ndev) + " GPU(s) visible (one engine per DEVICE: devices are never shared)");
    if ((gpus > 1 || rcclGather) && inflight > 1) die("--gpus / --rccl-gather and --in-flight are separate loops");
    HIP_OK(hipSetDevice(0));                                                       // cudaSetDevice(DEVICE) :1783
    Caps caps = capsForFrames(frames);
    if (refCaps) { caps.N = 50000; caps.Nk = 30000; caps.P = 10000; caps.W = 800; caps.Vw = 576; caps.S = 800; }      // params.h:24-27,68-69
    std::vector<std::string> files;
    if (DIR* d = opendir(data.c_str())) {
        while (dirent* e = readdir(d)) { std::string n = e->d_name; if (n.size() > 4 && n.substr(n.size() - 4) == ".bin") files.push_back(n); }
        closedir(d);
    } else die("cannot list " + data);
    if (files.empty()) die("no .bin frames under " + data);
    std::sort(files.begin(), files.end());

    const WeightMap w = loadWeights(wts);
    if (gpus > 1 || rcclGather) return runMultiGpu(w, caps, mode, frames, gpus, graph, raw, repeat, data, out, files);
    if (inflight > 1) return runPipelined(w, caps, mode, frames, inflight, graph, raw, repeat, data, out, files);
    hipStream_t s; HIP_OK(hipStreamCreate(&s));
    Engine eng(w, caps, s, mode, frames);
    // warm-up on empty frames (sizes every buffer), then record the forward into a HIP graph
    for (int k = 0; k < 2; ++k) eng.enqueue(s);
    HIP_OK(hipStreamSynchronize(s));
    hipGraphExec_t exec = nullptr;
    if (graph) {
        hipGraph_t g;
        HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        eng.enqueue(s);
        HIP_OK(hipStreamEndCapture(s, &g));
        HIP_OK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    }
    float* hpts; int* hcnt; float* hrows; int* hkept;
    HIP_OK(hipHostMalloc(&hpts, (size_t)frames * caps.N * 16)); HIP_OK(hipHostMalloc(&hcnt, 4 * frames));
    HIP_OK(hipHostMalloc(&hrows, (size_t)frames * TOP_K * 9 * 4)); HIP_OK(hipHostMalloc(&hkept, 4 * frames));
    double totalMs = 0; size_t totalFrames = 0;
    for (size_t f0 = 0; f0 < files.size(); f0 += frames) {
        const int nf = (int)std::min<size_t>(frames, files.size() - f0);          // (a last, partial group leaves its other slots empty: count 0)
        for (int f = 0; f < frames; ++f) hcnt[f] = 0;
        for (int f = 0; f < nf; ++f) {
            int n = 0;
            const std::vector<float> pts = loadBin(data + "/" + files[f0 + f], caps.N, n);
            memcpy(hpts + (size_t)f * caps.N * 4, pts.data(), (size_t)n * 16); hcnt[f] = n;
        }
        double ms = 0;
        for (int r = 0; r < repeat; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int f = 0; f < nf; ++f)      // n x 16 bytes per frame, not the zero-padded cap (:1925)
                HIP_OK(hipMemcpyAsync((char*)eng.points.ptr + (size_t)f * caps.N * 16, hpts + (size_t)f * caps.N * 4, (size_t)hcnt[f] * 16, hipMemcpyHostToDevice, s));
            HIP_OK(hipMemcpyAsync(eng.count.ptr, hcnt, 4 * frames, hipMemcpyHostToDevice, s));
            if (graph) HIP_OK(hipGraphLaunch(exec, s)); else eng.enqueue(s);
            HIP_OK(hipMemcpyAsync(hkept, eng.result[2].ptr, 4 * frames, hipMemcpyDeviceToHost, s));
            HIP_OK(hipMemcpyAsync(hrows, eng.result[0].ptr, (size_t)frames * TOP_K * 9 * 4, hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
            ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (r == repeat - 1) { totalMs += ms; totalFrames += nf; }
        }
        for (int f = 0; f < nf; ++f) {
            const std::string& name = files[f0 + f];
            const std::string stem = name.substr(0, name.size() - 4);
            const float* rows = hrows + (size_t)f * TOP_K * 9;
            saveTxt(out + "/" + stem + ".txt", rows, hkept[f], ms / nf);
            if (raw) {
                FILE* fh = fopen((out + "/" + stem + ".rows").c_str(), "wb");
                if (!fh) die("cannot write raw rows");
                fwrite(&hkept[f], 4, 1, fh); fwrite(rows, 4, (size_t)hkept[f] * 9, fh); fclose(fh);
            }
            printf("%s: %d points -> %d boxes, %.3f ms%s\n", stem.c_str(), hcnt[f], hkept[f], ms / nf, frames > 1 ? " (its share of a multi-frame forward)" : "");
        }
    }
    printf("dsvt_detect: %zu frames, %s, %d frame(s) per forward, %s: %.3f ms per frame, %.1f frames/s (upload + forward + download of the final boxes, last repeat of each group)\n",
           totalFrames, mode == MODE_F16 ? "fp16" : mode == MODE_SPLIT ? "fp32 grade (f16x3)" : "fp32 grade with fp8 head corrections", frames,
           graph ? "HIP graph" : "host launches", totalMs / std::max<size_t>(totalFrames, 1), 1e3 * totalFrames / std::max(totalMs, 1e-9));
    return 0;
}
