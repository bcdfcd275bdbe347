// plugin_base.h -- host-side plugin object model of libdsvt_hip.so.
//
// Mirrors the subset of nvinfer1::IPluginV2DynamicExt / IPluginCreator the reference
// plugins implement (plugins/include/*.h), minus everything TensorRT-specific.  One
// subclass per plugin type; include/dsvt_plugin.h exports the objects as a C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/dsvt_plugin.h"

namespace dsvt {

// 256-byte workspace alignment, as CUDA_MEM_ALIGN in e.g. plugins/src/getSet.cu:34,71-102
constexpr size_t kWsAlign = 256;
inline size_t alignUp(size_t v) { return (v + kWsAlign - 1) / kWsAlign * kWsAlign; }

// carves consecutive aligned regions out of the caller's workspace
struct WsCarver {
    char* base; size_t off = 0;
    explicit WsCarver(void* p) : base(static_cast<char*>(p)) {}
    template <class T> T* take(size_t count) {
        T* r = reinterpret_cast<T*>(base + off);
        off += alignUp(count * sizeof(T));
        return r;
    }
};

// device memory of the plugins' own state: hipMalloc / hipFree unless the host installed an allocator (include/dsvt_plugin.h dsvtSetGpuAllocator)
void setGpuAllocator(DsvtGpuAllocFn alloc, DsvtGpuFreeFn free_, void* user);      // c_api.hip
hipError_t deviceMallocBytes(void** p, size_t bytes);
hipError_t deviceFreeBytes(void* p);
template <class T> inline hipError_t dsvtMalloc(T** p, size_t bytes) { return deviceMallocBytes(reinterpret_cast<void**>(p), bytes); }
inline hipError_t dsvtFree(void* p) { return deviceFreeBytes(p); }

struct FieldDef { const char* name; DsvtPluginFieldType type; };

class Plugin {
public:
    virtual ~Plugin() {}
    virtual const char* type() const = 0;
    virtual int nbOutputs() const = 0;
    virtual int outputDims(int index, const DsvtDims* in, int nbIn, DsvtDims* out) const = 0;
    virtual int outputType(int index, const int32_t* inTypes, int nbIn) const = 0;
    virtual bool supportsFormat(int pos, const DsvtPluginTensorDesc* io, int nbIn, int nbOut) const = 0;
    virtual size_t workspaceSize(const DsvtPluginTensorDesc* in, int nbIn, const DsvtPluginTensorDesc* out, int nbOut) const = 0;
    virtual int enqueue(const DsvtPluginTensorDesc* inDesc, const DsvtPluginTensorDesc* outDesc,
                        const void* const* inputs, void* const* outputs, void* workspace, hipStream_t stream) = 0;
    virtual size_t serializationSize() const = 0;
    virtual void serialize(void* buffer) const = 0;
    virtual Plugin* clone() const = 0;
    // true: enqueue() itself processes a leading batch dimension > 1 (one launch for all images); false (default): the C ABI layer runs
    // the per-frame slabs one after the other (c_api.hip)
    virtual bool handlesBatch() const { return false; }
    // input `index` is shared by all frames of a batch even if its leading dimension equals the batch size (a table, not a per-frame stack)
    virtual bool sharedInput(int /*index*/) const { return false; }
    bool zeroFill = true;
    std::string layerName;
};

struct Creator {
    const char* name;
    std::vector<FieldDef> fields;
    Plugin* (*create)(const DsvtPluginFieldCollection* fc);
    Plugin* (*deserialize)(const void* data, size_t len);
    // materialised lazily for dsvtGetFieldNames
    std::vector<DsvtPluginField> fieldStore;
    DsvtPluginFieldCollection fc;
};

std::vector<Creator*>& registry();
struct Registrar { explicit Registrar(Creator* c) { registry().push_back(c); } };

// ---- field / buffer helpers -----------------------------------------------------------
inline const DsvtPluginField* findField(const DsvtPluginFieldCollection* fc, const char* name) {
    if (!fc) return nullptr;
    for (int i = 0; i < fc->nbFields; ++i)
        if (fc->fields[i].name && !strcmp(fc->fields[i].name, name)) return &fc->fields[i];
    return nullptr;
}
inline int fieldInt(const DsvtPluginFieldCollection* fc, const char* name, int def = 0) {
    const DsvtPluginField* f = findField(fc, name);
    return (f && f->data) ? static_cast<const int*>(f->data)[0] : def;
}
inline float fieldFloat(const DsvtPluginFieldCollection* fc, const char* name, float def = 0.f) {
    const DsvtPluginField* f = findField(fc, name);
    return (f && f->data) ? static_cast<const float*>(f->data)[0] : def;
}
// Array fields.  The reference's factories pass length = 1 even for arrays (include/plugin_helper.h:92-104) and its creators read
// a fixed number of elements, so length <= 1 means "unspecified" here too; a caller that DOES state a length states the truth, and a
// field shorter than what the creator reads is an error (fieldError() makes dsvtCreatePlugin return NULL instead of reading past it).
inline bool& fieldError() { static thread_local bool e = false; return e; }
// why the last dsvtCreatePlugin / dsvtDeserializePlugin of this thread returned NULL (dsvtGetLastCreateError); creators that can say set it
inline std::string& createError() { static thread_local std::string m; return m; }
inline bool fieldHolds(const DsvtPluginField* f, long n) {
    if (f && f->data && f->length > 1 && f->length < n) { fieldError() = true; return false; }
    return f && f->data;
}
inline void fieldInts(const DsvtPluginFieldCollection* fc, const char* name, int* out, int n) {
    const DsvtPluginField* f = findField(fc, name);
    const bool ok = fieldHolds(f, n);
    for (int i = 0; i < n; ++i) out[i] = ok ? static_cast<const int*>(f->data)[i] : 0;
}
inline void fieldFloats(const DsvtPluginFieldCollection* fc, const char* name, float* out, int n) {
    const DsvtPluginField* f = findField(fc, name);
    const bool ok = fieldHolds(f, n);
    for (int i = 0; i < n; ++i) out[i] = ok ? static_cast<const float*>(f->data)[i] : 0.f;
}
// pointer to an n-element float / int array field, or nullptr (absent, or shorter than n)
inline const float* fieldFloatArray(const DsvtPluginFieldCollection* fc, const char* name, long n) {
    const DsvtPluginField* f = findField(fc, name);
    return fieldHolds(f, n) ? static_cast<const float*>(f->data) : nullptr;
}
inline const int* fieldIntArray(const DsvtPluginFieldCollection* fc, const char* name, long n) {
    const DsvtPluginField* f = findField(fc, name);
    return fieldHolds(f, n) ? static_cast<const int*>(f->data) : nullptr;
}

// writeToBuffer / readFromBuffer, as in e.g. plugins/src/getSet.cu:45-59
template <class T> inline void wr(char*& d, const T& v) { memcpy(d, &v, sizeof(T)); d += sizeof(T); }
template <class T> inline T rd(const char*& d) { T v; memcpy(&v, d, sizeof(T)); d += sizeof(T); return v; }

inline DsvtDims dims1(int a) { DsvtDims d{}; d.nbDims = 1; d.d[0] = a; return d; }
inline DsvtDims dims2(int a, int b) { DsvtDims d{}; d.nbDims = 2; d.d[0] = a; d.d[1] = b; return d; }
inline DsvtDims dims3(int a, int b, int c) { DsvtDims d{}; d.nbDims = 3; d.d[0] = a; d.d[1] = b; d.d[2] = c; return d; }
inline DsvtDims dims4(int a, int b, int c, int e) { DsvtDims d{}; d.nbDims = 4; d.d[0] = a; d.d[1] = b; d.d[2] = c; d.d[3] = e; return d; }

inline int lastError() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : static_cast<int>(e); }
#define DSVT_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return static_cast<int>(_e); } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Optional trailing int fields of a serialized plugin (appended by later rounds, older blobs stay valid): how many of them a blob of `len`
// bytes carries after its `base` bytes, or -1 when the length is none of base, base + 4, ... base + 4 maxInts -- a padded or truncated
// buffer is refused instead of having stray bytes read as flags.
inline int trailingInts(size_t len, size_t base, int maxInts) {
    if (len < base || (len - base) % sizeof(int) != 0 || (len - base) / sizeof(int) > (size_t)maxInts) return -1;
    return (int)((len - base) / sizeof(int));
}

// Ablation / trace / A-B switches.  The product library (libdsvt_hip.so) reads NO environment variable: ablateEnv() is the constant
// default there, so no load, store or MFMA of a timed kernel can be skipped from outside.  `python dsvt-ai-trt_amd/build.py --ablate`
// builds libdsvt_hip_ablate.so with -DDSVT_ABLATE for tools/ (trace_*.py, ablate_conv.sh, one_attn.py ...), where the switch is read.
#ifdef DSVT_ABLATE
inline int ablateEnv(const char* name, int def) { const char* e = getenv(name); return e ? atoi(e) : def; }
constexpr bool kAblate = true;           // the kernel instantiations only an ablation switch can select are compiled
#else
inline int ablateEnv(const char*, int def) { return def; }
constexpr bool kAblate = false;          // ... and are NOT part of the product library (`if constexpr (kAblate)` at their launch sites)
#endif

}  // namespace dsvt
