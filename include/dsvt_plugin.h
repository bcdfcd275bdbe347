/*
 * dsvt_plugin.h -- C ABI of libdsvt_hip.so, the MI355X-native replacement for the
 * DSVT-AI-TRT plugin libraries.
 *
 * The reference registers ten TensorRT plugins (nvinfer1::IPluginV2DynamicExt +
 * nvinfer1::IPluginCreator, version "1") and its host code talks to them through
 *   getPluginRegistry()->getPluginCreator(name,"1") -> getFieldNames() ->
 *   createPlugin(name, fc) -> addPluginV2(...)            (include/plugin_helper.h:15-678)
 * and, at run time, through
 *   int enqueue(const PluginTensorDesc* in, const PluginTensorDesc* out,
 *               const void* const* inputs, void* const* outputs,
 *               void* workspace, cudaStream_t stream)     (e.g. plugins/include/getSet.h:41-44)
 * TensorRT cannot exist on ROCm, so the same protocol is exported here as plain C:
 * same plugin type names, same field names / types / order, same tensor slots and
 * dtypes, same serialisation byte layout, same "caller owns every tensor and the
 * workspace" convention, hipStream_t where the reference has cudaStream_t.
 * Every entry point below names the reference interface it replaces.
 *
 * Plugin types exported (dsvtGetPluginTypeName enumerates them):
 *   Points2FeaturesPlugin     plugins/src/points2Features.cu   (PLUGIN_NAME :41)
 *   TorchScatterMaxPlugin     plugins/src/torchScatterMax.cu
 *   WindowPartitionPlugin     plugins/src/windowPartition.cu
 *   GetSetPlugin              plugins/src/getSet.cu
 *   GetValueByIndexPlugin     plugins/src/getValueByIndex.cu
 *   MapSetFeature2VoxelPlugin plugins/src/mapSetFeature2voxel.cu
 *   LayerNormPlugin           plugins/src/layerNorm.cu
 *   GeluPlugin                plugins/src/gelu.cu
 *   Map2BevPlugin             plugins/src/map2bev.cu
 *   FilterBoxByScorePlugin    plugins/src/filterBoxByScore.cu
 *   MultiHeadAttentionPlugin  NEW: replaces the ~35 TensorRT layers built by
 *                             multHeadAttention(), src/dsvt-ai-trt.cpp:288-458
 *   (fused pipeline ops, no reference counterpart, see DESIGN.md:)
 *   DsvtLinearPlugin          FC (+prologue/epilogue) used where the reference calls
 *                             addFullyConnected (src/dsvt-ai-trt.cpp:283,476,490,506,525)
 *   DsvtSetAttentionPlugin    GetValueByIndex + MHA core + MapSetFeature2Voxel in one
 *   DsvtEncoderMlpPlugin      out-proj + LayerNorm -> FC1 + GELU -> FC2 + LayerNorms of one encoder
 *                             layer in one launch (src/dsvt-ai-trt.cpp:669-756)
 *   DsvtPosEmbedPlugin        all position-embedding MLPs of a frame in one launch (src/dsvt-ai-trt.cpp:461-492)
 *   DsvtPillarFeatureNetPlugin  both PFN layers + both TorchScatterMax reductions in one launch
 *                             (src/dsvt-ai-trt.cpp:565-589)
 *   DsvtConv2dPlugin          convBnLELU / convBn / deconvBnLELU / conv_with_bias of the BEV
 *                             backbone and CenterHead (src/dsvt-ai-trt.cpp:149-246, 1144-1468)
 *   CenterHeadTopKPlugin      the decode the reference builds from TensorRT layers: sigmoid,
 *                             two-stage TopK, gathers, exp, atan (src/dsvt-ai-trt.cpp:1479-1669);
 *                             its outputs are FilterBoxByScorePlugin's inputs
 *   RotatedNmsPlugin          nms_cpu / box_overlap of the host post-processing
 *                             (include/helper.h:166-283) on the device
 *
 * Differences from the reference that a caller can observe are listed in
 * INTEGRATION.md (deterministic canonical ordering instead of atomic arrival
 * order; runtime caps instead of params.h macros; capacity guards; non-zero
 * return from enqueue instead of abort()).
 */
#ifndef DSVT_PLUGIN_H_
#define DSVT_PLUGIN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSVT_PLUGIN_VERSION "1"   /* PLUGIN_VERSION, e.g. plugins/src/getSet.cu:40 */
#define DSVT_MAX_DIMS 8           /* nvinfer1::Dims::MAX_DIMS */

/* hipStream_t (opaque); replaces cudaStream_t in enqueue() */
typedef struct ihipStream_t* dsvtStream_t;

/* nvinfer1::PluginFieldType (values as in NvInferRuntimeCommon.h) */
typedef enum {
    DSVT_FIELD_FLOAT16 = 0, DSVT_FIELD_INT8 = 1, DSVT_FIELD_INT16 = 2, DSVT_FIELD_INT32 = 3,
    DSVT_FIELD_CHAR = 4, DSVT_FIELD_FLOAT32 = 5, DSVT_FIELD_FLOAT64 = 6, DSVT_FIELD_DIMS = 7,
    DSVT_FIELD_UNKNOWN = 8
} DsvtPluginFieldType;

/* nvinfer1::PluginField {name, data, type, length} */
typedef struct {
    const char* name;
    const void* data;
    int32_t type;     /* DsvtPluginFieldType */
    int32_t length;   /* number of elements behind data (the reference passes 1 even for
                         arrays, plugin_helper.h:92-104; array lengths are fixed per field) */
} DsvtPluginField;

/* nvinfer1::PluginFieldCollection */
typedef struct {
    int32_t nbFields;
    const DsvtPluginField* fields;
} DsvtPluginFieldCollection;

/* nvinfer1::Dims */
typedef struct {
    int32_t nbDims;
    int32_t d[DSVT_MAX_DIMS];
} DsvtDims;

/* nvinfer1::DataType */
typedef enum { DSVT_FLOAT = 0, DSVT_HALF = 1, DSVT_INT8 = 2, DSVT_INT32 = 3, DSVT_BOOL = 4 } DsvtDataType;
/* nvinfer1::TensorFormat::kLINEAR is the only format any reference plugin accepts */
#define DSVT_FORMAT_LINEAR 0

/* nvinfer1::PluginTensorDesc */
typedef struct {
    DsvtDims dims;
    int32_t type;     /* DsvtDataType */
    int32_t format;   /* DSVT_FORMAT_LINEAR */
    float scale;
} DsvtPluginTensorDesc;

/* opaque plugin instance = one nvinfer1::IPluginV2DynamicExt object */
typedef struct DsvtPlugin DsvtPlugin;

/* ---- registry / creator side (nvinfer1::IPluginCreator) ------------------------- */

/* number of registered plugin types / their names; replaces getPluginRegistry()
 * enumeration (REGISTER_TENSORRT_PLUGIN, e.g. plugins/include/points2Features.h:104) */
int32_t dsvtGetNbPluginTypes(void);
const char* dsvtGetPluginTypeName(int32_t index);

/* IPluginCreator::getFieldNames(), e.g. Points2FeaturesPluginCreator::getFieldNames
 * plugins/src/points2Features.cu:1109-1112.  Field data pointers are NULL; name, type
 * and order are exactly the reference creator's.  NULL if the type is unknown. */
const DsvtPluginFieldCollection* dsvtGetFieldNames(const char* pluginType, const char* pluginVersion);

/* IPluginCreator::createPlugin(name, fc), e.g. plugins/src/points2Features.cu:1113-1195.
 * Returns NULL on unknown type/version, NULL fc, or invalid fields -- including an array field
 * whose stated length (> 1) is shorter than the number of elements the creator reads; length
 * <= 1 is taken as "unspecified" because the reference's factories pass 1 for every field. */
DsvtPlugin* dsvtCreatePlugin(const char* pluginType, const char* pluginVersion, const char* layerName,
                             const DsvtPluginFieldCollection* fc);

/* Why the calling thread's last dsvtCreatePlugin / dsvtDeserializePlugin returned NULL ("" after a success).  The reference reports such
 * failures through the TensorRT logger (e.g. the printf / assert lines of plugins/src/points2Features.cu:1113-1195); there is no logger
 * object in this ABI, so the text is kept per thread.  The pointer is valid until the thread's next create / deserialize call. */
const char* dsvtGetLastCreateError(void);

/* IPluginCreator::deserializePlugin(name, data, length), e.g. points2Features.cu:1196-1200 */
DsvtPlugin* dsvtDeserializePlugin(const char* pluginType, const char* pluginVersion, const char* layerName,
                                  const void* serialData, size_t serialLength);

/* ---- plugin side (nvinfer1::IPluginV2DynamicExt) -------------------------------- */

const char* dsvtPluginGetType(const DsvtPlugin* p);        /* getPluginType()    points2Features.cu:1008 */
const char* dsvtPluginGetVersion(const DsvtPlugin* p);     /* getPluginVersion() points2Features.cu:1013 */
int32_t dsvtPluginGetNbOutputs(const DsvtPlugin* p);       /* getNbOutputs()     points2Features.cu:1018 */

/* getOutputDimensions(outputIndex, inputs, nbInputs, exprBuilder): the reference builds
 * DimsExprs from its constructor parameters; here the result is concrete.  Returns 0 on
 * success.  e.g. plugins/src/points2Features.cu:133-189 */
int32_t dsvtPluginGetOutputDimensions(const DsvtPlugin* p, int32_t outputIndex, const DsvtDims* inputs,
                                      int32_t nbInputs, DsvtDims* out);

/* getOutputDataType(index, inputTypes, nbInputs), e.g. points2Features.cu:1000-1006 */
int32_t dsvtPluginGetOutputDataType(const DsvtPlugin* p, int32_t index, const int32_t* inputTypes, int32_t nbInputs);

/* supportsFormatCombination(pos, inOut, nbInputs, nbOutputs), e.g. points2Features.cu:203-255 */
int32_t dsvtPluginSupportsFormatCombination(const DsvtPlugin* p, int32_t pos, const DsvtPluginTensorDesc* inOut,
                                            int32_t nbInputs, int32_t nbOutputs);

/* getWorkspaceSize(inputs, nbInputs, outputs, nbOutputs), e.g. points2Features.cu:262-277.
 * The caller allocates that many bytes of device memory (256-byte aligned) and passes it
 * to enqueue. */
size_t dsvtPluginGetWorkspaceSize(const DsvtPlugin* p, const DsvtPluginTensorDesc* inputs, int32_t nbInputs,
                                  const DsvtPluginTensorDesc* outputs, int32_t nbOutputs);

/* configurePlugin(in, nbInputs, out, nbOutputs), e.g. Points2FeaturesPlugin::configurePlugin plugins/src/points2Features.cu:257-260
 * (an empty body there).  Here it records nbInputs (which enqueue's signature does not carry) and, for a batch of B = inputs[0].dims.d[0]
 * > 1 frames, WHICH tensors are stacks of per-frame slabs: B counts as a batch only when inputs[0] has rank >= 2 and every output is a
 * [B, ...] tensor; an input is a stack when its leading dimension is B and the plugin does not declare it shared (tables).
 * enqueue uses this record and never infers a batch from the shapes it is handed.
 * Returns 0; -1 on NULL arguments; -2 if nbOutputs is not the plugin's. */
int32_t dsvtPluginConfigurePlugin(DsvtPlugin* p, const DsvtPluginTensorDesc* inputs, int32_t nbInputs,
                                  const DsvtPluginTensorDesc* outputs, int32_t nbOutputs);

/* enqueue(inputDesc, outputDesc, inputs, outputs, workspace, stream), e.g.
 * Points2FeaturesPlugin::enqueue plugins/src/points2Features.cu:896-990.
 * All pointers are device pointers owned by the caller; work is issued asynchronously on
 * `stream`; no host synchronisation happens inside.  Returns 0 on success; a HIP launch
 * error returns its hipError_t value (> 0; the reference abort()s instead); -1 = a required
 * pointer (plugin, inputs, outputs) is NULL; -2 = unsupported tensor shape (batch != 1, like
 * the reference, whose kernels ignore the batch dimension: points2Features.cu:678,900); -3 = a
 * C++ exception was caught at the boundary (nothing ever unwinds into the caller).
 * Batch: a plugin configured (dsvtPluginConfigurePlugin) with B > 1 runs B batch-1 enqueues on `stream` over the tensors recorded as
 * per-frame stacks -- the layout the reference's output shapes describe, although its kernels only ever process frame 0
 * (points2Features.cu:678,900,919) -- the other tensors are shared by all frames; descriptors that disagree with the configured batch, or
 * a rank >= 3 first input with a leading dimension > 1 on a plugin that was never configured, return -2.  Descriptors without a leading
 * batch dimension ([rows, C]) are one plain enqueue. */
int32_t dsvtPluginEnqueue(DsvtPlugin* p, const DsvtPluginTensorDesc* inputDesc, const DsvtPluginTensorDesc* outputDesc,
                          const void* const* inputs, void* const* outputs, void* workspace, dsvtStream_t stream);

/* getSerializationSize()/serialize(buffer): byte layout identical to the reference's
 * (SURVEY.md section 8b column 3), e.g. points2Features.cu:1033-1060 */
size_t dsvtPluginGetSerializationSize(const DsvtPlugin* p);
void dsvtPluginSerialize(const DsvtPlugin* p, void* buffer);

/* clone() / destroy(), e.g. points2Features.cu:123-131, 1062-1065 */
DsvtPlugin* dsvtPluginClone(const DsvtPlugin* p);
void dsvtPluginDestroy(DsvtPlugin* p);

/* ---- extras (no reference counterpart) ------------------------------------------ */

/* Non-reference knob: when 0 the plugin stops zero-filling the padded tail of its outputs
 * (rows >= the device-side valid count).  Default 1 = reference behaviour ("every enqueue
 * zero-fills its whole outputs", e.g. points2Features.cu:919-937).  The fused pipeline sets
 * 0 because every consumer honours the device-side counts. */
void dsvtPluginSetZeroFill(DsvtPlugin* p, int32_t enable);

/* Device memory of the plugins' OWN state (packed weights, tables, LayerNorm parameters): the reference allocates it with cudaMalloc in the plugin
 * constructor (plugins/src/layerNorm.cu:150-155) and TensorRT lets a host replace the allocator of an engine's memory through nvinfer1::IGpuAllocator
 * (IBuilder::setGpuAllocator / IRuntime::setGpuAllocator).  Same idea: after dsvtSetGpuAllocator(alloc, free, user) every plugin created or deserialized
 * takes its device memory from alloc(bytes, user) (NULL = failure; 256-byte alignment expected) and returns it through free(ptr, user); NULL, NULL
 * restores hipMalloc / hipFree.  Memory is always released by the allocator it came from.  Tensors and workspaces stay the caller's, as in the
 * reference.  (tests/test_guard_pages_gpu.py puts unmapped guard pages around these buffers too.) */
typedef void* (*DsvtGpuAllocFn)(size_t bytes, void* user);
typedef void (*DsvtGpuFreeFn)(void* ptr, void* user);
void dsvtSetGpuAllocator(DsvtGpuAllocFn alloc, DsvtGpuFreeFn free_, void* user);

/* Library build info ("gfx950 ...") */
const char* dsvtGetBuildInfo(void);

#ifdef __cplusplus
}
#endif
#endif /* DSVT_PLUGIN_H_ */
