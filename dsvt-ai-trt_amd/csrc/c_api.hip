// c_api.hip -- extern "C" surface of libdsvt_hip.so (declared in include/dsvt_plugin.h).
#include "plugin_base.h"
#include <mutex>
#include <unordered_map>

namespace dsvt {

// ---- the plugins' own device memory (dsvtSetGpuAllocator) ----------------------------------------------------------------
namespace {
struct GpuAllocator { DsvtGpuAllocFn alloc = nullptr; DsvtGpuFreeFn free_ = nullptr; void* user = nullptr; };
std::mutex& allocMu() { static std::mutex m; return m; }
GpuAllocator& gpuAllocator() { static GpuAllocator a; return a; }
std::unordered_map<void*, GpuAllocator>& hostedBlocks() { static std::unordered_map<void*, GpuAllocator> m; return m; }     // blocks that came from a host allocator
}  // namespace
void setGpuAllocator(DsvtGpuAllocFn alloc, DsvtGpuFreeFn free_, void* user) {
    std::lock_guard<std::mutex> lk(allocMu());
    GpuAllocator& a = gpuAllocator();
    if (alloc && free_) { a.alloc = alloc; a.free_ = free_; a.user = user; } else a = GpuAllocator{};
}
hipError_t deviceMallocBytes(void** p, size_t bytes) {
    GpuAllocator a;
    { std::lock_guard<std::mutex> lk(allocMu()); a = gpuAllocator(); }
    if (!a.alloc) return hipMalloc(p, bytes);
    void* q = a.alloc(bytes ? bytes : 1, a.user);
    if (!q) { *p = nullptr; return hipErrorOutOfMemory; }
    { std::lock_guard<std::mutex> lk(allocMu()); hostedBlocks()[q] = a; }
    *p = q;
    return hipSuccess;
}
hipError_t deviceFreeBytes(void* p) {
    if (!p) return hipSuccess;
    GpuAllocator a;
    {
        std::lock_guard<std::mutex> lk(allocMu());
        auto it = hostedBlocks().find(p);
        if (it == hostedBlocks().end()) return hipFree(p);
        a = it->second; hostedBlocks().erase(it);
    }
    a.free_(p, a.user);
    return hipSuccess;
}
std::vector<Creator*>& registry() {
    static std::vector<Creator*> r;
    return r;
}
static Creator* findCreator(const char* type, const char* version) {
    if (!type) return nullptr;
    if (version && strcmp(version, DSVT_PLUGIN_VERSION) != 0) return nullptr;
    for (Creator* c : registry())
        if (!strcmp(c->name, type)) return c;
    return nullptr;
}
}  // namespace dsvt

using namespace dsvt;

// Recorded by dsvtPluginConfigurePlugin: the number of inputs and, for a batch of B > 1 frames, WHICH tensors are per-frame stacks.
// enqueue never guesses that from the shapes it is handed (a shared table or count tensor whose leading dimension
// happens to equal B must not be sliced, and a caller whose descriptors carry no leading batch dimension must get one plain enqueue).
struct DsvtPlugin {
    Plugin* impl; int nbInputs = -1;
    int batch = 1;                               // B of the configured shapes (in[0].dims.d[0] when every output is a [B, ...] stack too)
    std::vector<char> inBatched, outBatched;     // per tensor: leading dimension == B at configure time and the plugin does not declare it shared
};

static DsvtPlugin* wrap(Plugin* p, const char* layerName) {
    if (!p) return nullptr;
    if (layerName) p->layerName = layerName;
    DsvtPlugin* w = new DsvtPlugin; w->impl = p; return w;
}

// Nothing may unwind through the C boundary: every entry point that allocates or runs plugin code is wrapped, and a C++ exception
// (std::bad_alloc from a weight copy, ...) becomes the entry point's failure value.
#define DSVT_GUARD(fail, ...) try { __VA_ARGS__ } catch (...) { return fail; }
constexpr int32_t kErrNullArg = -1;      // enqueue & co.: a required pointer was NULL
constexpr int32_t kErrException = -3;    // a C++ exception was caught at the boundary

extern "C" {

int32_t dsvtGetNbPluginTypes(void) { return static_cast<int32_t>(registry().size()); }

const char* dsvtGetPluginTypeName(int32_t i) {
    return (i >= 0 && i < static_cast<int32_t>(registry().size())) ? registry()[i]->name : nullptr;
}

const DsvtPluginFieldCollection* dsvtGetFieldNames(const char* type, const char* version) {
    DSVT_GUARD(nullptr,
        Creator* c = findCreator(type, version);
        if (!c) return nullptr;
        if (c->fieldStore.empty()) {
            for (const FieldDef& f : c->fields) c->fieldStore.push_back(DsvtPluginField{f.name, nullptr, f.type, 1});
            c->fc.nbFields = static_cast<int32_t>(c->fieldStore.size());
            c->fc.fields = c->fieldStore.data();
        }
        return &c->fc;)
}

DsvtPlugin* dsvtCreatePlugin(const char* type, const char* version, const char* layerName,
                             const DsvtPluginFieldCollection* fc) {
    DSVT_GUARD(nullptr,
        createError().clear();
        Creator* c = findCreator(type, version);
        if (!c) { createError() = "no plugin of this type and version is registered"; return nullptr; }
        if (!fc || fc->nbFields < 0 || (fc->nbFields > 0 && !fc->fields)) { createError() = "null or malformed field collection"; return nullptr; }
        fieldError() = false;
        Plugin* p = c->create(fc);
        if (p && fieldError()) { delete p; p = nullptr; createError() = "an array field is shorter than the number of elements the creator reads"; }
        if (!p && createError().empty()) createError() = "the creator rejected the field values (missing field, value out of range, or a capacity this build does not support)";
        return wrap(p, layerName);)
}

DsvtPlugin* dsvtDeserializePlugin(const char* type, const char* version, const char* layerName,
                                  const void* data, size_t len) {
    DSVT_GUARD(nullptr,
        createError().clear();
        Creator* c = findCreator(type, version);
        if (!c || !data) { createError() = !c ? "no plugin of this type and version is registered" : "null serial data"; return nullptr; }
        Plugin* p = c->deserialize(data, len);
        if (!p && createError().empty()) createError() = "the serialized blob is too short or holds values the plugin rejects";
        return wrap(p, layerName);)
}

const char* dsvtGetLastCreateError(void) { return createError().c_str(); }

const char* dsvtPluginGetType(const DsvtPlugin* p) { return p ? p->impl->type() : nullptr; }
const char* dsvtPluginGetVersion(const DsvtPlugin*) { return DSVT_PLUGIN_VERSION; }
int32_t dsvtPluginGetNbOutputs(const DsvtPlugin* p) { return p ? p->impl->nbOutputs() : kErrNullArg; }

int32_t dsvtPluginGetOutputDimensions(const DsvtPlugin* p, int32_t idx, const DsvtDims* in, int32_t nbIn, DsvtDims* out) {
    if (!p || !out || (nbIn > 0 && !in)) return kErrNullArg;
    DSVT_GUARD(kErrException, return p->impl->outputDims(idx, in, nbIn, out);)
}
int32_t dsvtPluginGetOutputDataType(const DsvtPlugin* p, int32_t idx, const int32_t* inTypes, int32_t nbIn) {
    if (!p || (nbIn > 0 && !inTypes)) return kErrNullArg;
    DSVT_GUARD(kErrException, return p->impl->outputType(idx, inTypes, nbIn);)
}
int32_t dsvtPluginSupportsFormatCombination(const DsvtPlugin* p, int32_t pos, const DsvtPluginTensorDesc* io,
                                            int32_t nbIn, int32_t nbOut) {
    if (!p || !io || pos < 0 || pos >= nbIn + nbOut) return 0;
    DSVT_GUARD(0, return p->impl->supportsFormat(pos, io, nbIn, nbOut) ? 1 : 0;)
}
size_t dsvtPluginGetWorkspaceSize(const DsvtPlugin* p, const DsvtPluginTensorDesc* in, int32_t nbIn,
                                  const DsvtPluginTensorDesc* out, int32_t nbOut) {
    if (!p) return 0;
    DSVT_GUARD(0, return p->impl->workspaceSize(in, nbIn, out, nbOut);)
}
// The batch dimension.  The reference carries it in every tensor shape but its kernels index with the scalar counts of frame 0
// (points2Features.cu:678,900,919), so only batch 1 works there.  Here a batch of B frames is B consecutive batch-1
// enqueues on the same stream.  Which tensors are stacks of per-frame slabs is decided ONCE, by dsvtPluginConfigurePlugin (the
// reference's configurePlugin sees the same descriptors): B = in[0].dims.d[0], valid as a batch only if in[0] has a row dimension behind it
// and EVERY output is a [B, ...] tensor (the plugins' own getOutputDimensions produce exactly that); a tensor is a stack
// when its leading dimension is B and the plugin does not declare that input shared (Plugin::sharedInput: tables such as the QKV
// linear's position table); everything else is passed to every frame as is.  A plugin that was not configured, or whose enqueue
// descriptors disagree with the configured batch, is refused (-2) rather than sliced by guesswork; descriptors without a leading batch
// dimension (rank-2 [P, C] tensors) only look like a batch when the caller configured EVERY output as a [P, ...] tensor as well.  The workspace is reused (same stream => serialised).
static size_t slabBytes(const DsvtPluginTensorDesc& d) {
    size_t n = d.type == DSVT_HALF ? 2 : (d.type == DSVT_INT8 || d.type == DSVT_BOOL) ? 1 : 4;
    for (int k = 1; k < d.dims.nbDims; ++k) n *= (size_t)d.dims.d[k];
    return n;
}
static int32_t enqueueBatched(const DsvtPlugin* p, const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc* outDesc,
                              const void* const* inputs, void* const* outputs, void* ws, hipStream_t stream) {
    Plugin* impl = p->impl;
    const int B = p->batch, nIn = p->nbInputs, nOut = impl->nbOutputs();
    std::vector<DsvtPluginTensorDesc> id(inDesc, inDesc + nIn), od(outDesc, outDesc + nOut);
    std::vector<const void*> ip(nIn); std::vector<void*> op(nOut);
    for (int k = 0; k < nIn; ++k) if (p->inBatched[k]) { if (id[k].dims.nbDims < 1 || id[k].dims.d[0] != B) return -2; id[k].dims.d[0] = 1; }
    for (int k = 0; k < nOut; ++k) if (p->outBatched[k]) { if (od[k].dims.nbDims < 1 || od[k].dims.d[0] != B) return -2; od[k].dims.d[0] = 1; }
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < nIn; ++k)
            ip[k] = p->inBatched[k] ? static_cast<const char*>(inputs[k]) + (size_t)b * slabBytes(inDesc[k]) : inputs[k];
        for (int k = 0; k < nOut; ++k)
            op[k] = p->outBatched[k] ? static_cast<char*>(outputs[k]) + (size_t)b * slabBytes(outDesc[k]) : outputs[k];
        const int32_t rc = impl->enqueue(id.data(), od.data(), ip.data(), op.data(), ws, stream);
        if (rc != 0) return rc;
    }
    return 0;
}

int32_t dsvtPluginEnqueue(DsvtPlugin* p, const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc* outDesc,
                          const void* const* inputs, void* const* outputs, void* ws, dsvtStream_t stream) {
    if (!p || !inputs || !outputs) return kErrNullArg;
    DSVT_GUARD(kErrException,
        if (!p->impl->handlesBatch()) {
            if (p->batch > 1) {                  // configured as a batch: the enqueue descriptors must describe that batch
                if (!inDesc || !outDesc || inDesc[0].dims.nbDims < 2 || inDesc[0].dims.d[0] != p->batch) return -2;
                return enqueueBatched(p, inDesc, outDesc, inputs, outputs, ws, reinterpret_cast<hipStream_t>(stream));
            }
            // not configured (enqueue's signature, like TensorRT's, does not carry the number of inputs): a rank >= 3 first input with a
            // leading dimension > 1 can only be a batch, which cannot be served -- refused, not mis-sliced.  Rank-2 [rows, C] descriptors
            // and everything configured with B = 1 pass through as one enqueue.
            // Configured with a batch the loop above cannot serve (batch == -1: B > 1 frames in, but not every output a [B, ...] stack): refused too --
            // a plugin that ignores the batch dimension would process frame 0 and report success.
            if ((p->nbInputs < 1 || p->batch < 0) && inDesc && outDesc && inDesc[0].dims.nbDims >= 3 && inDesc[0].dims.d[0] > 1) return -2;
        }
        return p->impl->enqueue(inDesc, outDesc, inputs, outputs, ws, reinterpret_cast<hipStream_t>(stream));)
}
int32_t dsvtPluginConfigurePlugin(DsvtPlugin* p, const DsvtPluginTensorDesc* in, int32_t nbIn, const DsvtPluginTensorDesc* out, int32_t nbOut) {
    if (!p || nbIn < 1 || !in || (nbOut > 0 && !out)) return kErrNullArg;
    if (nbOut != p->impl->nbOutputs()) return -2;
    p->nbInputs = nbIn;
    p->batch = 1; p->inBatched.assign(nbIn, 0); p->outBatched.assign(nbOut, 0);
    const int B = in[0].dims.nbDims >= 2 ? in[0].dims.d[0] : 1;
    if (B > 1 && !p->impl->handlesBatch()) {
        bool stacks = nbOut > 0;
        for (int k = 0; k < nbOut; ++k) stacks = stacks && out[k].dims.nbDims >= 1 && out[k].dims.d[0] == B;      // ([B] count outputs are stacks too)
        if (stacks) {
            p->batch = B;
            for (int k = 0; k < nbIn; ++k) p->inBatched[k] = in[k].dims.nbDims >= 1 && in[k].dims.d[0] == B && !p->impl->sharedInput(k);
            for (int k = 0; k < nbOut; ++k) p->outBatched[k] = 1;
        } else if (in[0].dims.nbDims >= 3) p->batch = -1;        // a batch that cannot be sliced: enqueue refuses it
    }
    return 0;
}
size_t dsvtPluginGetSerializationSize(const DsvtPlugin* p) {
    if (!p) return 0;
    DSVT_GUARD(0, return p->impl->serializationSize();)
}
void dsvtPluginSerialize(const DsvtPlugin* p, void* buf) {
    if (!p || !buf) return;
    try { p->impl->serialize(buf); } catch (...) {}
}
DsvtPlugin* dsvtPluginClone(const DsvtPlugin* p) {
    if (!p) return nullptr;
    DSVT_GUARD(nullptr,
        Plugin* c = p->impl->clone();
        if (!c) return nullptr;
        c->zeroFill = p->impl->zeroFill;
        DsvtPlugin* w = wrap(c, p->impl->layerName.c_str());
        w->nbInputs = p->nbInputs; w->batch = p->batch; w->inBatched = p->inBatched; w->outBatched = p->outBatched;
        return w;)
}
void dsvtPluginDestroy(DsvtPlugin* p) {
    if (!p) return;
    try { delete p->impl; } catch (...) {}
    delete p;
}
void dsvtPluginSetZeroFill(DsvtPlugin* p, int32_t enable) { if (p) p->impl->zeroFill = enable != 0; }
void dsvtSetGpuAllocator(DsvtGpuAllocFn alloc, DsvtGpuFreeFn free_, void* user) { dsvt::setGpuAllocator(alloc, free_, user); }

const char* dsvtGetBuildInfo(void) { return "libdsvt_hip gfx950 (CDNA4) hand-written HIP, built " __DATE__; }

}  // extern "C"
