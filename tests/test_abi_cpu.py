"""The C-ABI library loads without a GPU and exports every symbol include/dsvt_plugin.h declares;
creators advertise the reference's field names in the reference's order; serialisation has the
reference's byte layout.  No compute calls here."""
import ctypes
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (plugin type, advertised fields in creator order, reference file:line of the creator)
REFERENCE_FIELDS = {
    "Points2FeaturesPlugin": ["max_points_num", "max_points_num_voxel_filter", "max_pillars_num", "point_feature_num",
                              "feature_num", "max_num_points_per_voxel", "point_cloud_range", "voxel_size",
                              "grid_size"],                                     # points2Features.cu:1084-1092
    "TorchScatterMaxPlugin": ["max_points_num", "max_pillars_num", "feature_num"],            # torchScatterMax.cu:376-378
    "WindowPartitionPlugin": ["max_win_num", "max_voxel_num_per_win", "sparse_shape", "win_shape", "shift_list"],  # windowPartition.cu:549-553
    "GetSetPlugin": ["max_win_num", "max_voxel_num_per_win", "voxel_num_set", "win_shape"],   # getSet.cu:782-785
    "GetValueByIndexPlugin": ["max_win_num", "voxel_num_set", "channel_num", "axis_id"],      # getValueByIndex.cu:427-430
    "MapSetFeature2VoxelPlugin": ["max_win_num", "voxel_num_set", "channel_num", "axis_id", "max_pillars_num"],  # mapSetFeature2voxel.cu:393-397
    "LayerNormPlugin": ["max_pillars_num", "channel_num", "weights_size", "pes", "weights", "bias"],              # layerNorm.cu:494-500 ("pes" sic)
    "GeluPlugin": ["max_pillars_num", "channel_num"],                                         # gelu.cu:325-326
    "Map2BevPlugin": ["max_pillars_num", "channel_num", "grid_size_x", "grid_size_y"],        # map2bev.cu:383-386
    "FilterBoxByScorePlugin": ["max_top_k", "point_cloud_range", "voxel_size", "score_threshold"],  # filterBoxByScore.cu:459-462
}


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "dsvt_plugin.h")).read()
    declared = sorted(set(re.findall(r"\b(dsvt[A-Z]\w+)\s*\(", hdr)))
    assert len(declared) >= 19
    lib = ctypes.CDLL(os.path.join(ROOT, "dsvt-ai-trt_amd", "libdsvt_hip.so"))
    for name in declared:
        assert getattr(lib, name) is not None, name
    assert sorted(pkg.plugin.EXPORTED_SYMBOLS) == declared
    lib.dsvtGetBuildInfo.restype = ctypes.c_char_p
    assert b"gfx950" in lib.dsvtGetBuildInfo()


def test_registry_and_field_names(pkg):
    P = pkg.plugin
    types = P.plugin_types()
    for t, fields in REFERENCE_FIELDS.items():
        assert t in types
        assert [n for n, _ in P.get_field_names(t)] == fields
    for t in ("MultiHeadAttentionPlugin", "DsvtLinearPlugin", "DsvtSetAttentionPlugin"):
        assert t in types
    assert P.get_field_names("Points2FeaturesPlugin", "2") is None        # only version "1" is registered
    assert P.get_field_names("NoSuchPlugin") is None
    with pytest.raises(ValueError):
        P.Plugin("NoSuchPlugin", {})


def test_serialisation_layouts_match_reference(pkg):
    P = pkg.plugin
    f32 = lambda *v: struct.pack("<%df" % len(v), *[np.float32(x) for x in v])
    i32 = lambda *v: struct.pack("<%di" % len(v), *v)
    vg = P.add_voxel_generator(50000, 30000, 10000, 4, 10, 48, -74.88, 74.88, -74.88, 74.88, -5.0, 3.0, 0.32, 0.32, 8.0, 468, 468, 1)
    # points2Features.cu:1038-1060: 6 ints, (xmin,xmax,ymin,ymax,zmin,zmax,vx,vy,vz), 3 ints
    assert vg.serialize() == i32(50000, 30000, 10000, 4, 10, 48) + f32(-74.88, 74.88, -74.88, 74.88, -5.0, 3.0, 0.32, 0.32, 8.0) + i32(468, 468, 1)
    assert P.add_torch_scatter_max(30000, 10000, 96).serialize() == i32(30000, 10000, 96)
    wp = P.add_window_partition(800, 576, 468, 468, 1, 24, 24, 1, 6, 6, 0)
    assert wp.serialize() == i32(468, 468, 1, 24, 24, 1, 6, 6, 0, 800, 576)              # windowPartition.cu:511-525
    assert P.add_get_set_op(800, 576, 36, 12, 12, 1).serialize() == i32(36, 800, 576, 12, 12, 1)     # getSet.cu:749-758
    assert P.add_get_value_by_index_op(800, 36, 192, 1).serialize() == i32(36, 800, 192, 1)
    assert P.add_map_set_feature2voxel_op(800, 36, 192, 1, 10000).serialize() == i32(36, 800, 192, 10000, 1)
    assert P.add_gelu_op(10000, 384).serialize() == i32(10000, 384)
    assert P.add_map_2_bev_op(10000, 192, 468, 468).serialize() == i32(10000, 192, 468, 468)
    fb = P.add_filter_box_by_score_op(500, -74.88, 74.88, -74.88, 74.88, -5.0, 3.0, 0.32, 0.32, 8.0, 0.3)
    assert fb.serialize() == i32(500) + f32(-74.88, 74.88, -74.88, 74.88, -5.0, 3.0, 0.32, 0.32, 8.0, 0.3)   # filterBoxByScore.cu:420-435
    g = np.linspace(0.8, 1.2, 192).astype(np.float32); b = np.linspace(-0.1, 0.1, 192).astype(np.float32)
    ln = P.add_layer_norm_op(g, b, 10000, 192, 192, 1e-5)
    assert ln.serialize() == i32(10000, 192, 192) + f32(0.0) + g.tobytes() + b.tobytes()  # eps = 0: the "pes" quirk
    # deserialize(serialize(x)) serialises to the same bytes; clone too
    for op in (vg, wp, fb, ln):
        blob = op.serialize()
        assert P.Plugin.deserialize(op.plugin_type, blob).serialize() == blob
        assert op.clone().serialize() == blob
    with pytest.raises(ValueError):
        P.Plugin.deserialize("GetSetPlugin", b"\x00" * 8)          # truncated blob
    # optional trailing fields are recognised by the blob's exact length: a padded buffer is refused, not read as flags
    gs7 = P.add_get_set_op(800, 576, 36, 12, 12, 1, max_set_num=1200).serialize()
    assert len(gs7) == 28 and P.Plugin.deserialize("GetSetPlugin", gs7).serialize() == gs7
    mb6 = P.add_map_2_bev_op(10000, 192, 468, 468, frames=2, split_output=2).serialize()
    assert mb6 == i32(10000, 192, 468, 468, 2, 2) and P.Plugin.deserialize("Map2BevPlugin", mb6).serialize() == mb6
    for ptype, blob in (("GetSetPlugin", gs7 + b"\0" * 4), ("GetSetPlugin", gs7[:-2]), ("Map2BevPlugin", mb6 + b"\0" * 4),
                        ("Points2FeaturesPlugin", vg.serialize() + b"\0" * 8)):
        with pytest.raises(ValueError):
            P.Plugin.deserialize(ptype, blob)


def test_output_dimensions_and_types(pkg):
    P = pkg.plugin
    vg = P.add_voxel_generator(50000, 30000, 10000, 4, 10, 48, -74.88, 74.88, -74.88, 74.88, -5.0, 3.0, 0.32, 0.32, 8.0, 468, 468, 1)
    ins = [(1, 50000, 4), (1,)]
    assert [vg.get_output_dimensions(i, ins) for i in range(6)] == [(1, 30000, 10), (1, 10000, 48), (1, 10000, 4), (1, 10000, 1), (1,), (1,)]
    assert vg.nb_outputs == 6 and vg.get_output_data_type(0, [0, 3]) == 0 and vg.get_output_data_type(2, [0, 3]) == 3
    with pytest.raises(IndexError):
        vg.get_output_dimensions(6, ins)
    gs = P.add_get_set_op(800, 576, 36, 12, 12, 1)
    ins = [(1, 800, 576), (1, 800, 576, 3), (1, 800), (1,)]
    assert [gs.get_output_dimensions(i, ins) for i in range(5)] == [(1, 2, 800, 36), (1, 2, 800, 36), (1,), (1, 800, 8, 36), (1, 800, 8, 36)]
    wp = P.add_window_partition(800, 576, 468, 468, 1, 12, 12, 1, 0, 0, 0)
    ins = [(1, 10000, 4), (1,)]
    assert [wp.get_output_dimensions(i, ins) for i in range(6)] == [(1, 800, 576), (1, 800, 576, 3), (1, 800), (1,), (1, 10000, 3), (1, 10000, 2)]
    fb = P.add_filter_box_by_score_op(500, -74.88, 74.88, -74.88, 74.88, -5.0, 3.0, 0.32, 0.32, 8.0, 0.3)
    assert fb.get_output_dimensions(0, [(1, 500)]) == (1, 500, 9) and fb.get_output_dimensions(1, [(1, 500)]) == (1,)
    # invalid construction parameters are rejected by createPlugin (returns NULL)
    with pytest.raises(ValueError):
        P.add_get_set_op(0, 576, 36, 12, 12, 1)
    with pytest.raises(ValueError):
        P.add_voxel_generator(50000, 30000, 10000, 4, 10, 100, -1, 1, -1, 1, -1, 1, 0.1, 0.1, 1, 20, 20, 1)   # > 64 points per pillar unsupported


def test_null_arguments_and_short_fields_are_rejected(pkg):
    """C boundary hygiene: NULL handles / pointer arrays give an error code instead of a crash; an array field that states a length
    shorter than what the creator reads makes createPlugin return NULL (length 1 = "unspecified", the reference's own habit,
    include/plugin_helper.h:92-104, is still accepted)."""
    P = pkg.plugin
    L = P.LIB
    assert L.dsvtPluginEnqueue(None, None, None, None, None, None, None) == -1
    assert L.dsvtPluginGetNbOutputs(None) == -1
    assert L.dsvtPluginGetSerializationSize(None) == 0
    assert not L.dsvtPluginClone(None)
    assert not L.dsvtCreatePlugin(b"GeluPlugin", b"1", b"x", None)
    L.dsvtPluginDestroy(None); L.dsvtPluginSerialize(None, None); L.dsvtPluginSetZeroFill(None, 1)
    op = P.add_gelu_op(16, 8)
    assert L.dsvtPluginEnqueue(op._h, None, None, None, None, None, None) == -1
    out = P.Dims()
    assert L.dsvtPluginGetOutputDimensions(op._h, 0, None, 1, ctypes.byref(out)) == -1

    def create(ptype, fields):
        keep, fl = [], []
        for name, arr, length in fields:
            arr = np.ascontiguousarray(arr); keep.append(arr)
            fl.append(P.PluginField(name.encode(), arr.ctypes.data, P.FIELD_INT32 if arr.dtype == np.int32 else P.FIELD_FLOAT32, length))
        fc = P.PluginFieldCollection(len(fl), (P.PluginField * len(fl))(*fl))
        h = L.dsvtCreatePlugin(ptype.encode(), b"1", b"t", ctypes.byref(fc))
        if h:
            L.dsvtPluginDestroy(ctypes.c_void_p(h))
        return bool(h)

    i = lambda *v: np.asarray(v, np.int32)
    f = lambda *v: np.asarray(v, np.float32)
    fb = lambda rng_len: [("max_top_k", i(500), 1), ("point_cloud_range", f(-74.88, 74.88, -74.88, 74.88, -5.0, 3.0), rng_len),
                          ("voxel_size", f(0.32, 0.32, 8.0), 3), ("score_threshold", f(0.3), 1)]
    assert create("FilterBoxByScorePlugin", fb(6))
    assert create("FilterBoxByScorePlugin", fb(1))            # the reference's factories say 1 for every field
    assert not create("FilterBoxByScorePlugin", fb(3))        # states 3 floats, the creator reads 6
    wp = lambda n: [("max_win_num", i(800), 1), ("max_voxel_num_per_win", i(576), 1), ("sparse_shape", i(468, 468, 1), 3),
                    ("win_shape", i(12, 12, 1), n), ("shift_list", i(0, 0, 0), 3)]
    assert create("WindowPartitionPlugin", wp(3)) and not create("WindowPartitionPlugin", wp(2))
    ln = lambda n: [("max_pillars_num", i(100), 1), ("channel_num", i(192), 1), ("weights_size", i(192), 1),
                    ("weights", np.ones(192, np.float32), n), ("bias", np.zeros(192, np.float32), 192)]
    assert create("LayerNormPlugin", ln(192)) and not create("LayerNormPlugin", ln(64))


def test_configure_plugin_protocol(pkg):
    """configurePlugin (points2Features.cu:257-260) records nbInputs for batched enqueues; wrong output count is refused"""
    P = pkg.plugin
    L = P.LIB
    op = P.add_gelu_op(16, 8)
    ind = (P.PluginTensorDesc * 2)(P._desc((2, 16, 8), P.DT_FLOAT), P._desc((2,), P.DT_INT32))
    outd = (P.PluginTensorDesc * 1)(P._desc((2, 16, 8), P.DT_FLOAT))
    assert L.dsvtPluginConfigurePlugin(op._h, ind, 2, outd, 1) == 0
    assert L.dsvtPluginConfigurePlugin(op._h, ind, 2, outd, 3) == -2
    assert L.dsvtPluginConfigurePlugin(None, ind, 2, outd, 1) == -1
    assert L.dsvtPluginConfigurePlugin(op._h, None, 2, outd, 1) == -1
    # a batch of 2 frames whose output is NOT a [2, ...] stack cannot be sliced: enqueue refuses it (-2) instead of serving frame 0 only
    import ctypes as C
    bad = (P.PluginTensorDesc * 1)(P._desc((16, 8), P.DT_FLOAT))
    assert L.dsvtPluginConfigurePlugin(op._h, ind, 2, bad, 1) == 0
    ptrs_in, ptrs_out = (C.c_void_p * 2)(1, 1), (C.c_void_p * 1)(1)          # (never dereferenced: the shape check comes first)
    assert L.dsvtPluginEnqueue(op._h, ind, bad, ptrs_in, ptrs_out, None, None) == -2


def test_no_cpu_path(pkg):
    """The product path refuses host tensors instead of silently computing somewhere else."""
    import torch
    op = pkg.plugin.add_gelu_op(16, 8)
    with pytest.raises(RuntimeError):
        op(torch.zeros((1, 16, 8)), torch.zeros((1,), dtype=torch.int32))


def test_wts_roundtrip(pkg, tmp_path):
    w = {k: v for k, v in list(pkg.synth.make_weights(with_bev=False).items())[:6]}
    path = str(tmp_path / "t.wts")
    pkg.synth.write_wts(path, w)
    r = pkg.synth.read_wts(path)
    assert list(r) == list(w)
    for k in w:
        assert np.array_equal(r[k], w[k].reshape(-1))
    s = pkg.synth.split_in_proj(pkg.synth.make_weights(with_bev=False))
    k = "module.backbone_3d.stage_0.0.encoder_list.0.win_attn.self_attn.in_proj_weight"
    assert s[k + ".query"].shape == (192, 192) and np.array_equal(s[k + ".value"], s[k][384:])


def test_plain_c_client_compiles_links_and_runs(pkg, tmp_path):
    """include/dsvt_plugin.h is valid C (gcc -std=c99 -Wall -Werror) and a C program can drive the creator side of the ABI
    exactly like include/plugin_helper.h:253-310 drives TensorRT's registry (tests/c_client/abi_client.c)."""
    import subprocess
    libdir = os.path.join(ROOT, "dsvt-ai-trt_amd")
    exe = str(tmp_path / "abi_client")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_client", "abi_client.c"), "-o", exe,
                           "-L", libdir, "-l:libdsvt_hip.so", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_client ok" in out.stdout and "gfx950" in out.stdout


def test_create_failure_says_why(pkg):
    """dsvtGetLastCreateError: a rejected createPlugin leaves its reason for the calling thread (the reference logs through TensorRT)"""
    P = pkg.plugin
    with pytest.raises(ValueError) as e:
        P.add_voxel_generator(1 << 29, 1 << 29, 65536, 4, 10, 48, -74.88, 74.88, -74.88, 74.88, -5.0, 3.0, 0.32, 0.32, 8.0, 468, 468, 1, frames=8)
    assert "2^31" in str(e.value)
    with pytest.raises(ValueError) as e:
        P.add_voxel_generator(4096, 4096, 64, 4, 10, 200, -74.88, 74.88, -74.88, 74.88, -5.0, 3.0, 0.32, 0.32, 8.0, 468, 468, 1)
    assert "max_num_points_per_voxel" in str(e.value)
    L = P.LIB
    assert not L.dsvtCreatePlugin(b"NoSuchPlugin", b"1", b"x", None)
    assert b"registered" in L.dsvtGetLastCreateError()
    ok = P.add_voxel_generator(4096, 4096, 64, 4, 10, 48, -74.88, 74.88, -74.88, 74.88, -5.0, 3.0, 0.32, 0.32, 8.0, 468, 468, 1)
    assert ok is not None and L.dsvtGetLastCreateError() == b""
