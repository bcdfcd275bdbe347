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

struct DsvtPlugin { Plugin* impl; };

static DsvtPlugin* wrap(Plugin* p, const char* layerName) {
    if (!p) return nullptr;
    if (layerName) p->layerName = layerName;
    return new DsvtPlugin{p};
}

extern "C" {

int32_t dsvtGetNbPluginTypes(void) { return static_cast<int32_t>(registry().size()); }

const char* dsvtGetPluginTypeName(int32_t i) {
    return (i >= 0 && i < static_cast<int32_t>(registry().size())) ? registry()[i]->name : nullptr;
}

const DsvtPluginFieldCollection* dsvtGetFieldNames(const char* type, const char* version) {
    Creator* c = findCreator(type, version);
    if (!c) return nullptr;
    if (c->fieldStore.empty()) {
        for (const FieldDef& f : c->fields) c->fieldStore.push_back(DsvtPluginField{f.name, nullptr, f.type, 1});
        c->fc.nbFields = static_cast<int32_t>(c->fieldStore.size());
        c->fc.fields = c->fieldStore.data();
    }
    return &c->fc;
}

DsvtPlugin* dsvtCreatePlugin(const char* type, const char* version, const char* layerName,
                             const DsvtPluginFieldCollection* fc) {
    Creator* c = findCreator(type, version);
    return c ? wrap(c->create(fc), layerName) : nullptr;
}

DsvtPlugin* dsvtDeserializePlugin(const char* type, const char* version, const char* layerName,
                                  const void* data, size_t len) {
    Creator* c = findCreator(type, version);
    return (c && data) ? wrap(c->deserialize(data, len), layerName) : nullptr;
}

const char* dsvtPluginGetType(const DsvtPlugin* p) { return p->impl->type(); }
const char* dsvtPluginGetVersion(const DsvtPlugin*) { return DSVT_PLUGIN_VERSION; }
int32_t dsvtPluginGetNbOutputs(const DsvtPlugin* p) { return p->impl->nbOutputs(); }

int32_t dsvtPluginGetOutputDimensions(const DsvtPlugin* p, int32_t idx, const DsvtDims* in, int32_t nbIn, DsvtDims* out) {
    return p->impl->outputDims(idx, in, nbIn, out);
}
int32_t dsvtPluginGetOutputDataType(const DsvtPlugin* p, int32_t idx, const int32_t* inTypes, int32_t nbIn) {
    return p->impl->outputType(idx, inTypes, nbIn);
}
int32_t dsvtPluginSupportsFormatCombination(const DsvtPlugin* p, int32_t pos, const DsvtPluginTensorDesc* io,
                                            int32_t nbIn, int32_t nbOut) {
    return p->impl->supportsFormat(pos, io, nbIn, nbOut) ? 1 : 0;
}
size_t dsvtPluginGetWorkspaceSize(const DsvtPlugin* p, const DsvtPluginTensorDesc* in, int32_t nbIn,
                                  const DsvtPluginTensorDesc* out, int32_t nbOut) {
    return p->impl->workspaceSize(in, nbIn, out, nbOut);
}
int32_t dsvtPluginEnqueue(DsvtPlugin* p, const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc* outDesc,
                          const void* const* inputs, void* const* outputs, void* ws, dsvtStream_t stream) {
    return p->impl->enqueue(inDesc, outDesc, inputs, outputs, ws, reinterpret_cast<hipStream_t>(stream));
}
size_t dsvtPluginGetSerializationSize(const DsvtPlugin* p) { return p->impl->serializationSize(); }
void dsvtPluginSerialize(const DsvtPlugin* p, void* buf) { p->impl->serialize(buf); }
DsvtPlugin* dsvtPluginClone(const DsvtPlugin* p) {
    Plugin* c = p->impl->clone();
    c->zeroFill = p->impl->zeroFill;
    return wrap(c, p->impl->layerName.c_str());
}
void dsvtPluginDestroy(DsvtPlugin* p) {
    if (!p) return;
    delete p->impl;
    delete p;
}
void dsvtPluginSetZeroFill(DsvtPlugin* p, int32_t enable) { p->impl->zeroFill = enable != 0; }
const char* dsvtGetBuildInfo(void) { return "libdsvt_hip gfx950 (CDNA4) hand-written HIP, built " __DATE__; }

}  // extern "C"
