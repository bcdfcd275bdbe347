"""Host-side mirror of the reference's plugin interface, on top of the C ABI of
libdsvt_hip.so (include/dsvt_plugin.h).

`Plugin` plays the role of nvinfer1::IPluginV2DynamicExt: created from a
PluginFieldCollection through the creator protocol (getFieldNames -> createPlugin), asked
for output dims / workspace size, and run with enqueue(inputs, outputs, workspace, stream).
The `add_*_op` functions have the names, argument order and argument meaning of the
reference factories in include/plugin_helper.h:15-678 (minus the TensorRT `network`
handle: they return the plugin object, and calling it enqueues on the current stream).

PyTorch is used for device memory and streams only.  There is no CPU fallback: if the
HIP library is missing, import fails.
"""
import ctypes as C
import os
import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("DSVT_HIP_LIB") or os.path.join(_HERE, "libdsvt_hip.so")      # (DSVT_HIP_LIB: A/B runs against another build)

FIELD_INT32, FIELD_FLOAT32 = 3, 5
DT_FLOAT, DT_HALF, DT_INT32 = 0, 1, 3
MAX_DIMS = 8


class PluginField(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("type", C.c_int32), ("length", C.c_int32)]


class PluginFieldCollection(C.Structure):
    _fields_ = [("nbFields", C.c_int32), ("fields", C.POINTER(PluginField))]


class Dims(C.Structure):
    _fields_ = [("nbDims", C.c_int32), ("d", C.c_int32 * MAX_DIMS)]


class PluginTensorDesc(C.Structure):
    _fields_ = [("dims", Dims), ("type", C.c_int32), ("format", C.c_int32), ("scale", C.c_float)]


def _load():
    if not os.path.exists(_SO):
        raise ImportError(
            f"{_SO} is missing: the HIP extension must be built first "
            "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    lib = C.CDLL(_SO)
    lib.dsvtGetNbPluginTypes.restype = C.c_int32
    lib.dsvtGetPluginTypeName.restype = C.c_char_p
    lib.dsvtGetPluginTypeName.argtypes = [C.c_int32]
    lib.dsvtGetFieldNames.restype = C.POINTER(PluginFieldCollection)
    lib.dsvtGetFieldNames.argtypes = [C.c_char_p, C.c_char_p]
    lib.dsvtCreatePlugin.restype = C.c_void_p
    lib.dsvtCreatePlugin.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(PluginFieldCollection)]
    lib.dsvtDeserializePlugin.restype = C.c_void_p
    lib.dsvtDeserializePlugin.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t]
    lib.dsvtPluginGetType.restype = C.c_char_p
    lib.dsvtPluginGetType.argtypes = [C.c_void_p]
    lib.dsvtPluginGetVersion.restype = C.c_char_p
    lib.dsvtPluginGetVersion.argtypes = [C.c_void_p]
    lib.dsvtPluginGetNbOutputs.restype = C.c_int32
    lib.dsvtPluginGetNbOutputs.argtypes = [C.c_void_p]
    lib.dsvtPluginGetOutputDimensions.restype = C.c_int32
    lib.dsvtPluginGetOutputDimensions.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Dims), C.c_int32, C.POINTER(Dims)]
    lib.dsvtPluginGetOutputDataType.restype = C.c_int32
    lib.dsvtPluginGetOutputDataType.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    lib.dsvtPluginSupportsFormatCombination.restype = C.c_int32
    lib.dsvtPluginSupportsFormatCombination.argtypes = [C.c_void_p, C.c_int32, C.POINTER(PluginTensorDesc), C.c_int32, C.c_int32]
    lib.dsvtPluginGetWorkspaceSize.restype = C.c_size_t
    lib.dsvtPluginGetWorkspaceSize.argtypes = [C.c_void_p, C.POINTER(PluginTensorDesc), C.c_int32, C.POINTER(PluginTensorDesc), C.c_int32]
    lib.dsvtPluginEnqueue.restype = C.c_int32
    lib.dsvtPluginEnqueue.argtypes = [C.c_void_p, C.POINTER(PluginTensorDesc), C.POINTER(PluginTensorDesc),
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]
    lib.dsvtPluginConfigurePlugin.restype = C.c_int32
    lib.dsvtPluginConfigurePlugin.argtypes = [C.c_void_p, C.POINTER(PluginTensorDesc), C.c_int32, C.POINTER(PluginTensorDesc), C.c_int32]
    lib.dsvtPluginGetSerializationSize.restype = C.c_size_t
    lib.dsvtPluginGetSerializationSize.argtypes = [C.c_void_p]
    lib.dsvtPluginSerialize.argtypes = [C.c_void_p, C.c_void_p]
    lib.dsvtPluginClone.restype = C.c_void_p
    lib.dsvtPluginClone.argtypes = [C.c_void_p]
    lib.dsvtPluginDestroy.argtypes = [C.c_void_p]
    lib.dsvtPluginSetZeroFill.argtypes = [C.c_void_p, C.c_int32]
    lib.dsvtGetBuildInfo.restype = C.c_char_p
    lib.dsvtGetLastCreateError.restype = C.c_char_p
    lib.dsvtGetLastCreateError.argtypes = []
    lib.dsvtSetGpuAllocator.restype = None
    lib.dsvtSetGpuAllocator.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


LIB = _load()

# Optional per-launch timing hook used by bench.py's roofline measurement: when PROFILE is a dict
# {plugin_type: [(start_event, end_event, plugin), ...]} every enqueue of a listed plugin type is
# bracketed by HIP events on the launching stream.
PROFILE = None
# Test seam (tests/test_pipeline_gpu.py::test_no_plugin_writes_outside_its_buffers): with GUARD_BYTES > 0 every output and workspace the
# convenience call path allocates sits between two bands of GUARD_BYTES 0xA5 bytes, listed in GUARDED as (whole uint8 buffer, payload bytes).
GUARD_BYTES = 0
GUARDED = []


def _alloc(shape, dtype, device, zero):
    if not GUARD_BYTES:
        return torch.zeros(shape, dtype=dtype, device=device) if zero else torch.empty(shape, dtype=dtype, device=device)
    n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    whole = torch.full((n + 2 * GUARD_BYTES,), 0xA5, dtype=torch.uint8, device=device)
    payload = whole[GUARD_BYTES:GUARD_BYTES + n]
    if zero:
        payload.zero_()
    GUARDED.append((whole, n))
    return payload.view(dtype).view(shape)

EXPORTED_SYMBOLS = [
    "dsvtGetNbPluginTypes", "dsvtGetPluginTypeName", "dsvtGetFieldNames", "dsvtCreatePlugin",
    "dsvtDeserializePlugin", "dsvtPluginGetType", "dsvtPluginGetVersion", "dsvtPluginGetNbOutputs",
    "dsvtPluginGetOutputDimensions", "dsvtPluginGetOutputDataType", "dsvtPluginSupportsFormatCombination",
    "dsvtPluginGetWorkspaceSize", "dsvtPluginConfigurePlugin", "dsvtPluginEnqueue", "dsvtPluginGetSerializationSize",
    "dsvtPluginSerialize", "dsvtPluginClone", "dsvtPluginDestroy", "dsvtPluginSetZeroFill", "dsvtGetBuildInfo", "dsvtGetLastCreateError",
    "dsvtSetGpuAllocator",
]


GPU_ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)
GPU_FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_gpu_allocator = None          # the pair installed now
_gpu_allocator_thunks = []     # EVERY pair ever installed, never cleared: the library keeps the raw function pointers per outstanding block (c_api.hip deviceFreeBytes:
                               # "memory is always released by the allocator it came from"), so a thunk may be called long after it was replaced or reset


def set_gpu_allocator(alloc=None, free=None):
    """nvinfer1::IGpuAllocator for the plugins' OWN device memory (packed weights, tables): alloc(bytes) -> address (0 = failure), free(address).
    None, None restores hipMalloc / hipFree.  Tensors and workspaces stay the caller's."""
    global _gpu_allocator
    if alloc is None or free is None:
        LIB.dsvtSetGpuAllocator(None, None, None); _gpu_allocator = None
        return
    a = GPU_ALLOC_FN(lambda n, _u: alloc(n)); f = GPU_FREE_FN(lambda p_, _u: free(p_))
    _gpu_allocator = (a, f); _gpu_allocator_thunks.append((a, f))
    LIB.dsvtSetGpuAllocator(a, f, None)


def plugin_types():
    return [LIB.dsvtGetPluginTypeName(i).decode() for i in range(LIB.dsvtGetNbPluginTypes())]


def get_field_names(plugin_type, version="1"):
    """IPluginCreator::getFieldNames(): [(name, type)] in the creator's advertised order."""
    fc = LIB.dsvtGetFieldNames(plugin_type.encode(), version.encode())
    if not fc:
        return None
    return [(fc.contents.fields[i].name.decode(), fc.contents.fields[i].type) for i in range(fc.contents.nbFields)]


def _torch_dtype(t):
    return {DT_FLOAT: torch.float32, DT_HALF: torch.float16, DT_INT32: torch.int32}[t]


def _dt_code(t):
    if t.dtype == torch.float32:
        return DT_FLOAT
    if t.dtype == torch.float16:
        return DT_HALF
    if t.dtype in (torch.int32, torch.uint32):
        return DT_INT32
    raise TypeError(f"unsupported tensor dtype {t.dtype}")


def _desc(shape, code):
    d = PluginTensorDesc()
    d.dims.nbDims = len(shape)
    for i, s in enumerate(shape):
        d.dims.d[i] = int(s)
    d.type, d.format, d.scale = code, 0, 1.0
    return d


class Plugin:
    """One plugin instance (nvinfer1::IPluginV2DynamicExt)."""

    rows_kind = "P"      # which device-side count bounds this op's rows (bench.py flop accounting)

    def __init__(self, plugin_type, fields=None, layer_name="", version="1", _handle=None):
        self.plugin_type = plugin_type
        self._keep = []
        if _handle is None:
            advertised = get_field_names(plugin_type, version)
            if advertised is None:
                raise ValueError(f"no plugin creator registered for ({plugin_type!r}, {version!r})")
            # like plugin_helper.h: walk the creator's advertised names and supply those we have
            fields = dict(fields or {})
            flist = []
            names = [n for n, _ in advertised] + [n for n in fields if n not in dict(advertised)]
            for name in names:
                if name not in fields:
                    continue
                val = fields[name]
                if isinstance(val, torch.Tensor):
                    val = val.detach().cpu().numpy()
                if isinstance(val, np.ndarray):
                    arr = np.ascontiguousarray(val.reshape(-1), np.int32 if val.dtype.kind in "iu" else np.float32)
                elif isinstance(val, (list, tuple)):
                    isint = all(isinstance(v, (int, np.integer)) for v in val)
                    arr = np.asarray(val, np.int32 if isint else np.float32)
                elif isinstance(val, (int, np.integer)):
                    arr = np.asarray([val], np.int32)
                else:
                    arr = np.asarray([val], np.float32)
                self._keep.append(arr)
                ftype = FIELD_INT32 if arr.dtype == np.int32 else FIELD_FLOAT32
                flist.append(PluginField(name.encode(), arr.ctypes.data, ftype, arr.size))
            arr_t = (PluginField * max(len(flist), 1))(*flist)
            fc = PluginFieldCollection(len(flist), arr_t)
            _handle = LIB.dsvtCreatePlugin(plugin_type.encode(), version.encode(), layer_name.encode(), C.byref(fc))
            if not _handle:
                why = LIB.dsvtGetLastCreateError()
                raise ValueError(f"createPlugin({plugin_type}) rejected fields {fields}: {why.decode() if why else 'no reason given'}")
        self._h = C.c_void_p(_handle)
        self.fields = {k: (v if not isinstance(v, np.ndarray) or v.size <= 8 else v.shape) for k, v in (fields or {}).items()}
        self.nb_outputs = LIB.dsvtPluginGetNbOutputs(self._h)
        self._cache = {}
        self._packs = {}
        self._keepalive = []

    # ---- IPluginV2DynamicExt surface --------------------------------------------------
    def get_plugin_type(self):
        return LIB.dsvtPluginGetType(self._h).decode()

    def get_output_dimensions(self, index, input_shapes):
        ins = (Dims * len(input_shapes))()
        for i, s in enumerate(input_shapes):
            ins[i].nbDims = len(s)
            for j, v in enumerate(s):
                ins[i].d[j] = int(v)
        out = Dims()
        rc = LIB.dsvtPluginGetOutputDimensions(self._h, index, ins, len(input_shapes), C.byref(out))
        if rc != 0:
            raise IndexError(f"{self.plugin_type}: no output {index}")
        return tuple(out.d[i] for i in range(out.nbDims))

    def get_output_data_type(self, index, input_types):
        arr = (C.c_int32 * len(input_types))(*input_types)
        return LIB.dsvtPluginGetOutputDataType(self._h, index, arr, len(input_types))

    def supports_format_combination(self, pos, descs, nb_in, nb_out):
        arr = (PluginTensorDesc * len(descs))(*descs)
        return bool(LIB.dsvtPluginSupportsFormatCombination(self._h, pos, arr, nb_in, nb_out))

    def get_workspace_size(self, in_descs, out_descs):
        a = (PluginTensorDesc * len(in_descs))(*in_descs)
        b = (PluginTensorDesc * len(out_descs))(*out_descs)
        return LIB.dsvtPluginGetWorkspaceSize(self._h, a, len(in_descs), b, len(out_descs))

    def serialize(self):
        n = LIB.dsvtPluginGetSerializationSize(self._h)
        buf = (C.c_char * n)()
        LIB.dsvtPluginSerialize(self._h, buf)
        return bytes(buf)

    @classmethod
    def deserialize(cls, plugin_type, data, layer_name="", version="1"):
        h = LIB.dsvtDeserializePlugin(plugin_type.encode(), version.encode(), layer_name.encode(), data, len(data))
        if not h:
            raise ValueError(f"deserializePlugin({plugin_type}) failed")
        return cls(plugin_type, _handle=h)

    def clone(self):
        return Plugin(self.plugin_type, _handle=LIB.dsvtPluginClone(self._h))

    def set_zero_fill(self, enable):
        LIB.dsvtPluginSetZeroFill(self._h, 1 if enable else 0)
        return self

    def enqueue(self, inputs, outputs, workspace=None, stream=None):
        """enqueue(inputDesc, outputDesc, inputs, outputs, workspace, stream); tensors are device
        tensors owned by the caller.  Raises on a non-zero return."""
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        ind = (PluginTensorDesc * len(inputs))(*[_desc(t.shape, _dt_code(t)) for t in inputs])
        outd = (PluginTensorDesc * len(outputs))(*[_desc(t.shape, _dt_code(t)) for t in outputs])
        inp = (C.c_void_p * len(inputs))(*[t.data_ptr() for t in inputs])
        outp = (C.c_void_p * len(outputs))(*[t.data_ptr() for t in outputs])
        ws = workspace.data_ptr() if workspace is not None else None
        rc = LIB.dsvtPluginEnqueue(self._h, ind, outd, inp, outp, ws, C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(f"{self.plugin_type}.enqueue returned {rc}")

    # ---- convenience: allocate outputs/workspace once per input signature, then enqueue ----
    def __call__(self, *inputs, out=None):
        # Hot path: every buffer of the pipeline is static, so the marshalled argument pack
        # (descriptors + pointer arrays) is built once per distinct set of input pointers and a
        # call is then a single ctypes call into dsvtPluginEnqueue.
        # out: optional list of caller-owned output tensors (e.g. a shared concat buffer).
        key = tuple(t.data_ptr() for t in inputs)
        if out is not None:
            key = key + tuple(t.data_ptr() for t in out)
        pack = self._packs.get(key)
        if pack is None:
            pack = self._make_pack(inputs, key, out)
        outs, ind, outd, inp, outp, ws = pack
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        prof = PROFILE.get(self.plugin_type) if PROFILE is not None else None
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = LIB.dsvtPluginEnqueue(self._h, ind, outd, inp, outp, ws, stream)
            e1.record()
            prof.append((e0, e1, self))
        else:
            rc = LIB.dsvtPluginEnqueue(self._h, ind, outd, inp, outp, ws, stream)
        if rc != 0:
            raise RuntimeError(f"{self.plugin_type}.enqueue returned {rc}")
        return outs

    def _make_pack(self, inputs, key, out=None):
        for t in inputs:
            if not t.is_cuda:
                raise RuntimeError(f"{self.plugin_type}: inputs must be device tensors (no CPU path exists)")
            if not t.is_contiguous():
                raise RuntimeError(f"{self.plugin_type}: inputs must be contiguous (kLINEAR)")
        sig = tuple((tuple(t.shape), t.dtype) for t in inputs)
        ent = self._cache.get(sig)
        if ent is None:
            shapes = [tuple(t.shape) for t in inputs]
            codes = [_dt_code(t) for t in inputs]
            outs = []
            for i in range(self.nb_outputs):
                shp = self.get_output_dimensions(i, shapes)
                dt = _torch_dtype(self.get_output_data_type(i, codes))
                outs.append(_alloc(shp, dt, inputs[0].device, True))
            wsz = self.get_workspace_size([_desc(s, c) for s, c in zip(shapes, codes)],
                                          [_desc(o.shape, _dt_code(o)) for o in outs])
            ws = _alloc((max(wsz, 256),), torch.uint8, inputs[0].device, False)
            ent = (outs, ws)
            self._cache[sig] = ent
        outs, ws = ent
        if out is not None:
            outs = list(out)
        ind = (PluginTensorDesc * len(inputs))(*[_desc(t.shape, _dt_code(t)) for t in inputs])
        outd = (PluginTensorDesc * len(outs))(*[_desc(t.shape, _dt_code(t)) for t in outs])
        inp = (C.c_void_p * len(inputs))(*[t.data_ptr() for t in inputs])
        outp = (C.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
        if LIB.dsvtPluginConfigurePlugin(self._h, ind, len(inputs), outd, len(outs)) != 0:          # configurePlugin(in, nbInputs, out, nbOutputs)
            raise RuntimeError(f"{self.plugin_type}.configurePlugin failed")
        pack = (outs, ind, outd, inp, outp, C.c_void_p(ws.data_ptr()))
        self._keepalive.append((tuple(inputs), tuple(outs)))      # the cached raw pointers must stay valid
        self._packs[key] = pack
        return pack

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                LIB.dsvtPluginDestroy(self._h)
                self._h = None
        except Exception:
            pass


# --------------------------------------------------------------------------------------
# the reference's factory functions (include/plugin_helper.h), minus `network`
# --------------------------------------------------------------------------------------
def add_voxel_generator(max_points_num, max_points_num_voxel_filter, max_pillars_num, point_feature_num,
                        feature_num, max_num_points_per_voxel, x_min, x_max, y_min, y_max, z_min, z_max,
                        voxel_size_x, voxel_size_y, voxel_size_z, grid_size_x, grid_size_y, grid_size_z, frames=1, point_id_slots=None):
    """plugin_helper.h:15-123.  Inputs at call time: points[1,N,4] f32, points_size[1] i32.
    frames > 1 (not in the reference): points [1, frames * N, 4] (frame f = rows f * N ...), points_size [frames]; the frames' pillars are
    concatenated (ascending (frame, cell)), coords = (frame, z, y, x), max_pillars_num / max_points_num_voxel_filter bound the totals.
    point_id_slots (not in the reference): how many slots of each row of the [P, 48] point-id table are written (default: all); a pillar's rows are
    consecutive, so the fused pillar feature net needs slot 0 only and the frame pipeline passes 1."""
    extra = dict(frames=int(frames)) if frames != 1 else {}
    if point_id_slots is not None and int(point_id_slots) != max_num_points_per_voxel:
        extra["point_id_slots"] = int(point_id_slots)
    return Plugin("Points2FeaturesPlugin", dict(extra, 
        max_points_num=max_points_num, max_points_num_voxel_filter=max_points_num_voxel_filter,
        max_pillars_num=max_pillars_num, point_feature_num=point_feature_num, feature_num=feature_num,
        max_num_points_per_voxel=max_num_points_per_voxel,
        point_cloud_range=[float(x_min), float(y_min), float(z_min), float(x_max), float(y_max), float(z_max)],   # :33-38
        voxel_size=[float(voxel_size_x), float(voxel_size_y), float(voxel_size_z)],
        grid_size=[grid_size_x, grid_size_y, grid_size_z]), "voxelGeneratorlayer")


def add_torch_scatter_max(max_points_num, max_pillars_num, feature_num):
    """plugin_helper.h:125-172.  Inputs: feat, pidx, pcnt, pillar_num."""
    return Plugin("TorchScatterMaxPlugin", dict(max_points_num=max_points_num, max_pillars_num=max_pillars_num,
                                                feature_num=feature_num), "torch_scatter_max_layer")


def add_window_partition(max_win_num, max_voxel_num_per_win, sparse_shape_x, sparse_shape_y, sparse_shape_z,
                         win_shape_x, win_shape_y, win_shape_z, shift_x, shift_y, shift_z):
    """plugin_helper.h:174-251.  Inputs: coords, pillar_num."""
    return Plugin("WindowPartitionPlugin", dict(
        max_win_num=max_win_num, max_voxel_num_per_win=max_voxel_num_per_win,
        sparse_shape=[sparse_shape_x, sparse_shape_y, sparse_shape_z],
        win_shape=[win_shape_x, win_shape_y, win_shape_z], shift_list=[shift_x, shift_y, shift_z]),
        "window_partition_layer")


def add_get_set_op(max_win_num, max_voxel_num_per_win, voxel_num_set, win_shape_x, win_shape_y, win_shape_z, max_set_num=None):
    """plugin_helper.h:253-314.  Inputs: gidx, cinw, vcnt, win_num.
    max_set_num (not in the reference, whose set dimension is MAX_WIN_NUM too): capacity of the set dimension of the outputs."""
    f = dict(max_win_num=max_win_num, max_voxel_num_per_win=max_voxel_num_per_win, voxel_num_set=voxel_num_set,
             win_shape=[win_shape_x, win_shape_y, win_shape_z])
    if max_set_num is not None and max_set_num != max_win_num:
        f["max_set_num"] = int(max_set_num)
    return Plugin("GetSetPlugin", f, "get_set_layer")


def add_set_partition_op(max_win_num, max_voxel_num_per_win, voxel_num_set, max_set_num, max_pillars_num, sparse_shape, wins, frames=1):
    """WindowPartition + GetSet of several window configurations in four launches (csrc/partition_ops.hip DsvtSetPartitionPlugin).
    wins: [(win_shape xyz, shift xyz), ...].  Inputs: coords [1,P,4], pillar_num [1].  Outputs per configuration k: c2d_k [1,P,3]
    (= WindowPartition output 4), inds_k [1,2,S,36], mask_k [1,2,S,36], set_num_k [1] (= GetSet outputs 0..2)."""
    return Plugin("DsvtSetPartitionPlugin", dict(
        max_win_num=max_win_num, max_voxel_num_per_win=max_voxel_num_per_win, voxel_num_set=voxel_num_set, max_set_num=max_set_num,
        max_pillars_num=max_pillars_num, sparse_shape=[int(v) for v in sparse_shape], num_configs=len(wins),
        win_shapes=[int(v) for w_, _s in wins for v in w_], shift_lists=[int(v) for _w, s_ in wins for v in s_], frames=int(frames)),
        "set_partition_layer")


def add_get_value_by_index_op(max_win_num, voxel_num_set, channel_num, axis_id):
    """plugin_helper.h:316-369.  Inputs: voxel_features, pose_features, voxel_inds, valid_set_num."""
    return Plugin("GetValueByIndexPlugin", dict(max_win_num=max_win_num, voxel_num_set=voxel_num_set,
                                                channel_num=channel_num, axis_id=axis_id), "get_value_by_index_layer")


def add_map_2_bev_op(max_pillars_num, channel_num, grid_size_x, grid_size_y, frames=1, split_output=False, persistent_output=False):
    """plugin_helper.h:371-425.  Inputs: voxel_features, coors, valid_voxel_num.  frames > 1: coords.x selects one of `frames` stacked maps.
    split_output: fp32 rows in, the fp16 triple [hi | lo | hi] (3 C channels per cell) out -- the operand of the fp32-grade convolutions.
    persistent_output: the caller passes the SAME output buffer on every call and nobody else writes it: a call then zeroes only the cells the call
    before it wrote instead of the whole map (another address, and the first call, take the full fill)."""
    extra = dict(frames=int(frames)) if frames != 1 else {}
    if persistent_output:
        extra["persistent_output"] = 1
    if split_output:
        extra["split_output"] = int(split_output)           # 2: the third plane holds the fp8 operands (x8) instead of repeating hi; 3: it is not written at all
    return Plugin("Map2BevPlugin", dict(extra, max_pillars_num=max_pillars_num, channel_num=channel_num,
                                        grid_size_x=grid_size_x, grid_size_y=grid_size_y), "map2bev_layer")


def add_map_set_feature2voxel_op(max_win_num, voxel_num_set, channel_num, axis_id, max_pillars_num):
    """plugin_helper.h:427-487.  Inputs: set features, voxel_inds, valid_set_num."""
    return Plugin("MapSetFeature2VoxelPlugin", dict(max_win_num=max_win_num, voxel_num_set=voxel_num_set,
                                                    channel_num=channel_num, axis_id=axis_id,
                                                    max_pillars_num=max_pillars_num), "map_set_feature2voxel_layer")


def add_layer_norm_op(weights, bias, max_pillars_num, channel_num, weights_size, eps):
    """plugin_helper.h:489-555.  Inputs: voxel_features, valid_voxel_num.
    NOTE the reference quirk kept on purpose: the creator advertises the field "pes", the
    factory offers "eps" only for names the creator advertises (plugin_helper.h:527), so `eps`
    is never delivered and the plugin runs with eps = 0 (layerNorm.cu:497 vs :558)."""
    offered = dict(max_pillars_num=max_pillars_num, channel_num=channel_num, weights_size=weights_size,
                   eps=float(eps), weights=np.asarray(weights, np.float32), bias=np.asarray(bias, np.float32))
    advertised = {n for n, _ in get_field_names("LayerNormPlugin")}
    return Plugin("LayerNormPlugin", {k: v for k, v in offered.items() if k in advertised}, "layer_norm_layer")


def add_gelu_op(max_pillars_num, channel_num):
    """plugin_helper.h:557-605.  Inputs: voxel_features, valid_voxel_num."""
    return Plugin("GeluPlugin", dict(max_pillars_num=max_pillars_num, channel_num=channel_num), "gelu_layer")


def add_center_head_topk_op(feature_height, feature_width, channel_num=18, class_num=10, max_top_k=500,
                            center_offset=0, center_z_offset=2, dim_offset=3, rot_offset=6, hm_offset=8):
    """CenterHead decode (src/dsvt-ai-trt.cpp:1479-1669: sigmoid, two-stage TopK, gathers, exp, atan) on the device.
    Input: head tensor [1,H,W,C] fp32 channels-last.  Outputs: the eight inputs of FilterBoxByScorePlugin."""
    return Plugin("CenterHeadTopKPlugin", dict(feature_height=feature_height, feature_width=feature_width, channel_num=channel_num,
                                               class_num=class_num, max_top_k=max_top_k, center_offset=center_offset,
                                               center_z_offset=center_z_offset, dim_offset=dim_offset, rot_offset=rot_offset,
                                               hm_offset=hm_offset), "center_head_topk_layer")


def add_pos_embed_op(max_rows, layer_input, pe_weights, pe_biases, weights, biases):
    """All position-embedding MLPs of a frame (FC(2->192)+BN+ReLU -> FC(192->192), src/dsvt-ai-trt.cpp:461-492, one per encoder layer)
    in one launch.  layer_input[l] selects which xy input layer l reads.  Inputs: count [1], xy tensors [1,rows,2] f32.
    Outputs: one [1,rows,192] fp16 tensor per layer.  BatchNorm folded into pe_weights / pe_biases by the caller."""
    L = len(layer_input)
    f32 = lambda xs: np.ascontiguousarray(np.stack([np.asarray(x, np.float32) for x in xs])).reshape(-1)
    return Plugin("DsvtPosEmbedPlugin", dict(max_rows=int(max_rows), num_layers=L, layer_input=[int(v) for v in layer_input],
                                             pe_weight=f32(pe_weights), pe_bias=f32(pe_biases), weight=f32(weights), bias=f32(biases)),
                  "pos_embed_layer")


def add_pillar_feature_net_op(max_pillars_num, weight0, bias0, weight1, bias1, pack_small_pillars=True, split_precision=False):
    """Both PFN layers + both scatter-max reductions in one launch, no per-point activation in memory (csrc/pfn.hip).
    BatchNorm folded by the caller: weight0 [96,10], weight1 [192,192] (columns 0..95 act on x0, 96..191 on its pillar max).
    Inputs: feat [1,Nk,10], pidx [1,P,T], pcnt [1,P,1], pillar_num [1].  Outputs: pillar features [1,P,192] fp32 + fp16.
    split_precision: layer 1 on (hi, lo) fp16 operand pairs (fp32 grade); one output, fp32."""
    extra = dict(split_precision=1) if split_precision else {}
    return Plugin("DsvtPillarFeatureNetPlugin", dict(extra, 
        max_pillars_num=int(max_pillars_num), weight0=np.asarray(weight0, np.float32).reshape(-1),
        bias0=np.asarray(bias0, np.float32).reshape(-1), weight1=np.ascontiguousarray(np.asarray(weight1, np.float32)).reshape(-1),
        bias1=np.asarray(bias1, np.float32).reshape(-1), pack_small_pillars=int(bool(pack_small_pillars))), "pillar_feature_net_layer")


def add_voxel_pool_op(max_voxel_num, max_pooled_num, sparse_shape, stride, frames=1):
    """Stage reduction of a 3-D voxel DSVT, integer half (csrc/voxel_pool.hip; no reference counterpart: SURVEY 8f-4).  sparse_shape / stride: (x, y, z).
    Inputs: coords [1,P,4] (b, z, y, x), count [1].  Outputs: pooled coords [1,P2,4] in ascending pooled-cell order, the child table [1,P2,pool_volume]
    (input row per slot (x % sx) sy sz + (y % sy) sz + z % sz, -1 = empty), every voxel's pooled row [1,P,1], P2 [1], P2 x pool_volume [1]."""
    f = dict(max_voxel_num=int(max_voxel_num), max_pooled_num=int(max_pooled_num), sparse_shape=[int(v) for v in sparse_shape], stride=[int(v) for v in stride])
    if frames != 1:
        f["frames"] = int(frames)
    return Plugin("DsvtVoxelPoolPlugin", f, "voxel_pool_layer")


def add_pool_gather_op(max_pooled_num, pool_volume, channel_num, pos_embedding):
    """x [1,P,C], child table, P2 -> src [1,P2,C] = max over the pool_volume slots (empty slots count as zero rows) and the key input [1,P,C] =
    x + pos_embedding[slot of the voxel in its pool] per INPUT voxel (the value input is x itself; csrc/voxel_pool.hip)"""
    return Plugin("DsvtPoolGatherPlugin", dict(max_pooled_num=int(max_pooled_num), pool_volume=int(pool_volume), channel_num=int(channel_num),
                                               pos_embedding=np.asarray(pos_embedding, np.float32).reshape(-1)), "pool_gather_layer")


def add_pool_attention_core_op(max_pooled_num, pool_volume, channel_num, num_heads):
    """q [1,P2,C] (already scaled by 1 / sqrt(head_dim)), k, v [1,P,C] (per input voxel), child table, P2 -> softmax over the pool's children, per head: [1,P2,C]"""
    return Plugin("DsvtPoolAttentionCorePlugin", dict(max_pooled_num=int(max_pooled_num), pool_volume=int(pool_volume), channel_num=int(channel_num),
                                                      num_heads=int(num_heads)), "pool_attention_core_layer")


def add_rotated_nms_op(max_boxes=500, nms_thresh=0.01):
    """nms_cpu (include/helper.h:257-283) on the device.  Inputs: FilterBoxByScorePlugin's rows [1,K,9] and count [1].
    Outputs: kept rows [1,K,9] in score order, their input row numbers [1,K], count [1]."""
    return Plugin("RotatedNmsPlugin", dict(max_boxes=int(max_boxes), nms_thresh=float(nms_thresh)), "rotated_nms_layer")


def add_filter_box_by_score_op(max_top_k, min_x_range, max_x_range, min_y_range, max_y_range, min_z_range,
                               max_z_range, voxel_x_size, voxel_y_size, voxel_z_size, score_threshold):
    """plugin_helper.h:607-678.  Inputs: scores, classes, xs, ys, center, center_z, angle, dim."""
    return Plugin("FilterBoxByScorePlugin", dict(
        max_top_k=max_top_k,
        point_cloud_range=[float(min_x_range), float(max_x_range), float(min_y_range), float(max_y_range),
                           float(min_z_range), float(max_z_range)],                       # :627-632
        voxel_size=[float(voxel_x_size), float(voxel_y_size), float(voxel_z_size)],
        score_threshold=float(score_threshold)), "filter_box_by_score_layer")


# --------------------------------------------------------------------------------------
# ops with no plugin in the reference (it builds them out of TensorRT layers)
# --------------------------------------------------------------------------------------
def add_multi_head_attention_op(in_proj_weight, in_proj_bias, out_proj_weight, out_proj_bias, max_win_num,
                                voxel_num_set, channel_num, num_heads):
    """Drop-in for multHeadAttention() (src/dsvt-ai-trt.cpp:288-458).
    Inputs: q, k, v [1,S,36,C], attn_mask [1,S,H,36] (GetSet output 3), valid_set_num [1]."""
    f = lambda a: np.asarray(a, np.float32).reshape(-1)
    return Plugin("MultiHeadAttentionPlugin", dict(
        max_win_num=max_win_num, voxel_num_set=voxel_num_set, channel_num=channel_num, num_heads=num_heads,
        in_proj_weight=f(in_proj_weight), in_proj_bias=f(in_proj_bias),
        out_proj_weight=f(out_proj_weight), out_proj_bias=f(out_proj_bias)), "multi_head_attention_layer")


ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


# precision of the matrix products: exact fp32 MFMA | fp16 operands (BASELINE configs[2]) | split precision: every operand a (hi, lo) fp16
# pair, three fp16 MFMAs per product, fp32 accumulate -- the reference's fp32 arithmetic (include/params.h:332) at fp16-matrix-core speed
COMPUTE_F32, COMPUTE_F16, COMPUTE_SPLIT = 0, 1, 2


OUT_F32, OUT_F16, OUT_BOTH = 0, 1, 2


def add_linear_op(weight, bias, max_rows, row_mult=1, activation=ACT_NONE, add_cols=0, layer_norms=(), ln_eps=0.0,
                  compute_type=COMPUTE_F32, input_half=False, output_mode=OUT_F32, pe_weight=None, pe_bias=None, add_gather_width=0,
                  add_gather_height=0):
    """FC with fused prologue/epilogue (csrc/linear.hip), used where the reference calls
    addFullyConnected (src/dsvt-ai-trt.cpp:283,476,490,506,525) + ElementWise/LayerNorm/GELU.
    Inputs: A [1,rows,K], count [1], (A2 if add_cols), then one residual per LayerNorm stage.
    layer_norms: sequence of (gamma, beta).
    add_gather_width = wx > 0: A2 is a [1, cells, K] table and one more input follows it, the [1, rows, 3] window coordinates
    (z, y, x) of WindowPartition (output 4); row m adds table row y * wx + x."""
    weight = np.asarray(weight, np.float32)
    N, K = weight.shape
    fields = dict(max_rows=max_rows, in_features=K, out_features=N, row_mult=row_mult, activation=activation,
                  add_cols=add_cols, num_layer_norms=len(layer_norms), ln_eps=float(ln_eps), compute_type=compute_type,
                  input_half=int(bool(input_half)), output_mode=output_mode, add_gather_width=int(add_gather_width), weight=weight.reshape(-1))
    if add_gather_height:      # 3-D windows (BASELINE configs[4]): table row (z * wy + y) * wx + x
        fields["add_gather_height"] = int(add_gather_height)
    if bias is not None:
        fields["bias"] = np.asarray(bias, np.float32).reshape(-1)
    if pe_weight is not None:      # fused K_in = 2 first FC (+BN, ReLU) of the position-embedding MLP: input 0 becomes xy [1,rows,2]
        fields["pe_weight"] = np.asarray(pe_weight, np.float32).reshape(-1)
        fields["pe_bias"] = np.asarray(pe_bias, np.float32).reshape(-1)
    if layer_norms:
        fields["ln_weights"] = np.concatenate([np.asarray(g, np.float32).reshape(-1) for g, _ in layer_norms])
        fields["ln_bias"] = np.concatenate([np.asarray(b, np.float32).reshape(-1) for _, b in layer_norms])
    return Plugin("DsvtLinearPlugin", fields, "linear_layer")


def add_set_attention_op(max_win_num, voxel_num_set, channel_num, num_heads, axis_id, max_pillars_num, io_half=False, split_precision=False):
    """GetValueByIndex + attention core + MapSetFeature2Voxel fused (csrc/attention.hip).
    Inputs: qkv [1,P,3C] (per-voxel projections), inds [1,2,S,36], mask [1,2,S,36], valid_set_num [1].
    io_half: fp16 rows in / out on fp16 MFMA.  split_precision: fp32 rows in / out, both products on (hi, lo) fp16 operand pairs
    (fp32 grade, set_attention_split_kernel) instead of v_mfma_f32_16x16x4_f32."""
    assert not (io_half and split_precision)
    return Plugin("DsvtSetAttentionPlugin", dict(max_win_num=max_win_num, voxel_num_set=voxel_num_set,
                                                 channel_num=channel_num, num_heads=num_heads, axis_id=axis_id,
                                                 max_pillars_num=max_pillars_num, io_half=2 if split_precision else int(bool(io_half))),
                  "set_attention_layer")


def conv_weight_rows(w):
    """torch Conv2d weight [Cout, Cin, KH, KW] -> DsvtConv2dPlugin rows [Cout][KH*KW][Cin]."""
    w = np.asarray(w, np.float32)
    return np.ascontiguousarray(w.transpose(0, 2, 3, 1)).reshape(w.shape[0], -1)


def deconv_weight_rows(w):
    """torch ConvTranspose2d weight [Cin, Cout, k, k] with stride == k -> pixel-shuffle rows [(dy*k+dx)*Cout + co][Cin]."""
    w = np.asarray(w, np.float32)
    return np.ascontiguousarray(w.transpose(2, 3, 1, 0)).reshape(-1, w.shape[0])


def split_weight_rows(weight_rows, taps, in_channels):
    """rows [R][taps][Cin] fp32 -> [R][taps][3 Cin] holding [w_hi | w_hi | w_lo] per tap (w_hi = fp16(w), w_lo = fp16(w - w_hi)): with an input
    [hi | lo | hi] (DsvtSplitHalfPlugin, or a producer with split_output) one fp16-MFMA convolution computes hi w_hi + lo w_hi + hi w_lo: the
    fp32 product up to the dropped lo w_lo term (2^-22 relative) for operands whose lo part is a NORMAL fp16 number, i.e. |v| >= 2^-3 or so;
    below that lo falls into the fp16 subnormals (absolute step 2^-24), so a 0.01-sized BN-folded weight keeps ~2^-18 relative precision
    -- still 50x finer than the 1e-3 box bar needs (tests/test_split_kernels_gpu.py::test_split_operand_range; the MFMA does not flush
    fp16 subnormal inputs)."""
    w = np.asarray(weight_rows, np.float32).reshape(-1, taps, in_channels)
    hi = w.astype(np.float16).astype(np.float32)
    lo = (w - hi).astype(np.float16).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([hi, hi, lo], axis=2)).reshape(w.shape[0], -1)


def add_split_half_op(channel_num, relu=False, has_residual=False):
    """y = relu?(x (+ residual)) in fp32 and its [hi | lo | hi] fp16 split (csrc/conv.hip DsvtSplitHalfPlugin).
    Inputs: x [1,H,W,C] f32 (, residual f32).  Outputs: y f32, y3 [1,H,W,3C] fp16."""
    return Plugin("DsvtSplitHalfPlugin", dict(channel_num=int(channel_num), relu=int(bool(relu)), has_residual=int(bool(has_residual))),
                  "split_half_layer")


def add_conv2d_op(weight_rows, bias, in_height, in_width, in_channels, out_channels, kernel_size=1, stride=1, padding=0,
                  pixel_shuffle=1, relu=False, has_residual=False, out_channel_stride=None, out_channel_offset=0, out_f32=False,
                  split_output=0, split_residual=False, split_input=0, kernel_variant=0):
    """NHWC fp16 implicit-GEMM convolution with fused bias / residual / ReLU / pixel-shuffle / concat offset
    (csrc/conv.hip): replaces the reference's addConvolutionNd / addDeconvolutionNd + addScale + ReLU + SUM
    groups (src/dsvt-ai-trt.cpp:149-246, 1144-1468).  Inputs: x [1,H,W,Cin] fp16 (, residual [1,Ho,Wo,C] fp16).
    split_output: 1 = the result leaves as the fp16 triple [hi | lo | hi] (out_channel_stride = 3 x the plane width), 2 = as [hi | lo | x8]
    (third plane: the OCP-fp8 operands of the correction terms, csrc/conv.hip ConvArgs::x8_out), 3 = [hi | - | x8] (no lo plane: read by [hi | x8] layers only),
    4 = [hi | lo | -] (no third plane: a tensor that is only ever a residual); split_residual: the residual input is such
    a triple (value hi + lo).  split_input (the input is a [hi | lo | x8] triple of in_channels / 3 real channels): 1 = weight_rows are
    split_weight_rows(...) and the third plane's phases read plane 0; 2 = weight_rows are the REAL fp32 rows [R][9][in_channels / 3] and the
    layer runs on the fp16 + fp8 K loop (3 x 3, stride 1, more than 32 output channels).  kernel_variant (a test knob, not serialized): 1 = the layer
    never takes conv_rows_kernel (csrc/conv_rows.hip), i.e. runs on round 5's conv_wide_kernel -- the two must agree bit for bit."""
    fields = dict(in_height=in_height, in_width=in_width, in_channels=in_channels, out_channels=out_channels,
                  kernel_size=kernel_size, stride=stride, padding=padding, pixel_shuffle=pixel_shuffle, relu=int(bool(relu)),
                  has_residual=int(bool(has_residual)),
                  out_channel_stride=out_channels if out_channel_stride is None else out_channel_stride,
                  out_channel_offset=out_channel_offset, out_f32=int(bool(out_f32)),
                  weight=np.asarray(weight_rows, np.float32).reshape(-1))
    if split_output:
        fields["split_output"] = int(split_output)
    if split_residual:
        fields["split_residual"] = int(split_residual)     # 2: the residual triple has no lo plane, its lo part is read from the x8 plane (to 2^-15)
    if split_input:
        fields["split_input"] = int(split_input)
    if kernel_variant:
        fields["kernel_variant"] = int(kernel_variant)
    if bias is not None:
        fields["bias"] = np.asarray(bias, np.float32).reshape(-1)
    return Plugin("DsvtConv2dPlugin", fields, "conv2d_layer")


def add_encoder_mlp_op(out_proj_weight, out_proj_bias, linear1_weight, linear1_bias, linear2_weight, linear2_bias,
                       layer_norms, max_rows, ln_eps=0.0, frames=0, split_precision=False):
    """Everything after the attention core of one DSVT encoder layer in one launch (csrc/mlp.hip):
    s1 = LN1(att Wo^T + bo + x); h = GELU(s1 W1^T + b1); x' = LN3(LN2(s1 + h W2^T + b2) + x) [; x' = LN4(x' + xb)].
    layer_norms: [(gamma, beta)] x 3 (norm1, norm2, encoder norm) or x 4 (+ block residual norm).
    Inputs: att [1,P,192] fp16, count [1], x [1,P,192] fp32 (, xb [1,P,192] fp32).  Outputs: x' fp32, x' fp16.
    frames: how many frames' rows a launch carries (picks the kernel: >= 3 the two-workgroups-per-CU one); 0 = decided on the device.
    split_precision: att arrives as fp32, all three GEMMs on (hi, lo) fp16 operand pairs (fp32 grade); one output, x' fp32."""
    f = lambda a: np.asarray(a, np.float32).reshape(-1)
    assert len(layer_norms) in (3, 4)
    return Plugin("DsvtEncoderMlpPlugin", dict(
        max_rows=max_rows, has_block_norm=int(len(layer_norms) == 4), ln_eps=float(ln_eps),
        out_proj_weight=f(out_proj_weight), out_proj_bias=f(out_proj_bias), linear1_weight=f(linear1_weight),
        linear1_bias=f(linear1_bias), linear2_weight=f(linear2_weight), linear2_bias=f(linear2_bias),
        ln_weights=np.concatenate([f(g) for g, _ in layer_norms]), ln_bias=np.concatenate([f(b) for _, b in layer_norms]),
        **({"frames": int(frames)} if frames else {}), **({"split_precision": 1} if split_precision else {})),
        "encoder_mlp_layer")
