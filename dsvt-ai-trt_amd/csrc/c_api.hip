// c_api.hip -- extern "C" surface of libdsvt_hip.so (declared in include/dsvt_plugin.h).
#include "plugin_base.h"

namespace dsvt {
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

struct DsvtPlugin { Plugin* impl; int nbInputs = -1; };      // nbInputs: recorded by dsvtPluginConfigurePlugin

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
        Creator* c = findCreator(type, version);
        if (!c || !fc || fc->nbFields < 0 || (fc->nbFields > 0 && !fc->fields)) return nullptr;
        fieldError() = false;
        Plugin* p = c->create(fc);
        if (p && fieldError()) { delete p; p = nullptr; }      // an array field shorter than what the creator reads
        return wrap(p, layerName);)
}

DsvtPlugin* dsvtDeserializePlugin(const char* type, const char* version, const char* layerName,
                                  const void* data, size_t len) {
    DSVT_GUARD(nullptr,
        Creator* c = findCreator(type, version);
        return (c && data) ? wrap(c->deserialize(data, len), layerName) : nullptr;)
}

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
// enqueues on the same stream: every tensor whose leading dimension is B is a stack of per-frame slabs (the layout the reference's
// own output shapes describe), tensors with another leading dimension (shared tables) are passed to every frame, the workspace is
// reused (same stream => serialised).  Needs the tensor descriptors, like TensorRT always supplies them.
static size_t slabBytes(const DsvtPluginTensorDesc& d) {
    size_t n = d.type == DSVT_HALF ? 2 : (d.type == DSVT_INT8 || d.type == DSVT_BOOL) ? 1 : 4;
    for (int k = 1; k < d.dims.nbDims; ++k) n *= (size_t)d.dims.d[k];
    return n;
}
static int32_t enqueueBatched(Plugin* impl, int B, int nIn, const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc* outDesc,
                              const void* const* inputs, void* const* outputs, void* ws, hipStream_t stream) {
    const int nOut = impl->nbOutputs();
    std::vector<DsvtPluginTensorDesc> id(inDesc, inDesc + nIn), od(outDesc, outDesc + nOut);
    std::vector<const void*> ip(nIn); std::vector<void*> op(nOut);
    for (int k = 0; k < nIn; ++k) if (id[k].dims.d[0] == B) id[k].dims.d[0] = 1;
    for (int k = 0; k < nOut; ++k) if (od[k].dims.d[0] == B) od[k].dims.d[0] = 1;
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < nIn; ++k)
            ip[k] = inDesc[k].dims.d[0] == B ? static_cast<const char*>(inputs[k]) + (size_t)b * slabBytes(inDesc[k]) : inputs[k];
        for (int k = 0; k < nOut; ++k)
            op[k] = outDesc[k].dims.d[0] == B ? static_cast<char*>(outputs[k]) + (size_t)b * slabBytes(outDesc[k]) : outputs[k];
        const int32_t rc = impl->enqueue(id.data(), od.data(), ip.data(), op.data(), ws, stream);
        if (rc != 0) return rc;
    }
    return 0;
}

int32_t dsvtPluginEnqueue(DsvtPlugin* p, const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc* outDesc,
                          const void* const* inputs, void* const* outputs, void* ws, dsvtStream_t stream) {
    if (!p || !inputs || !outputs) return kErrNullArg;
    DSVT_GUARD(kErrException,
        const int B = (inDesc && outDesc && inDesc[0].dims.nbDims >= 1) ? inDesc[0].dims.d[0] : 1;
        if (B > 1 && !p->impl->handlesBatch()) {
            // enqueue's signature (like TensorRT's) does not carry the number of inputs: configurePlugin delivers it beforehand
            if (p->nbInputs < 1) return -2;
            return enqueueBatched(p->impl, B, p->nbInputs, inDesc, outDesc, inputs, outputs, ws, reinterpret_cast<hipStream_t>(stream));
        }
        return p->impl->enqueue(inDesc, outDesc, inputs, outputs, ws, reinterpret_cast<hipStream_t>(stream));)
}
int32_t dsvtPluginConfigurePlugin(DsvtPlugin* p, const DsvtPluginTensorDesc* in, int32_t nbIn, const DsvtPluginTensorDesc* out, int32_t nbOut) {
    if (!p || nbIn < 1 || !in || (nbOut > 0 && !out)) return kErrNullArg;
    if (nbOut != p->impl->nbOutputs()) return -2;
    p->nbInputs = nbIn;
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
        w->nbInputs = p->nbInputs;
        return w;)
}
void dsvtPluginDestroy(DsvtPlugin* p) {
    if (!p) return;
    try { delete p->impl; } catch (...) {}
    delete p;
}
void dsvtPluginSetZeroFill(DsvtPlugin* p, int32_t enable) { if (p) p->impl->zeroFill = enable != 0; }
const char* dsvtGetBuildInfo(void) { return "libdsvt_hip gfx950 (CDNA4) hand-written HIP, built " __DATE__; }

}  // extern "C"
