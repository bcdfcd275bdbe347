"""Synthetic inputs for the DSVT hot path: Waymo-shaped point clouds and seeded
weights, plus the reference's `.wts` text format (reader/writer).

The point-cloud generator and the weight distributions are the test-data spec of
SURVEY.md section 8(d); the known-answer counts in tests/ depend on the exact
call order of `lidar_like`.  Weight names/shapes follow the reference's
`weightMap[...]` keys (src/dsvt-ai-trt.cpp:577-1468); the
`.wts` format is tools/gen_wts.py:86-99 / include/helper.h:328-439.
"""
import struct
import numpy as np


def lidar_like(N, seed):
    """[N,4] float32 (x,y,z,intensity), 64-beam spinning-lidar-like cloud."""
    rng = np.random.default_rng(seed)
    th = rng.uniform(0, 2 * np.pi, N)
    beams = np.deg2rad(np.linspace(-17.6, 2.4, 64))
    phi = beams[rng.integers(0, 64, N)]
    with np.errstate(divide="ignore"):
        rg = np.where(phi < 0, 1.8 / np.tan(-np.minimum(phi, -1e-4)), np.inf)
    hit = rng.random(N) < 0.35
    ro = 2.0 + rng.exponential(18.0, N)
    r = np.where(hit, np.minimum(ro, rg), rg)
    r = np.where(np.isfinite(r), r, rng.uniform(5, 75, N))
    r = np.minimum(r, 105.0)
    x = r * np.cos(phi) * np.cos(th)
    y = r * np.cos(phi) * np.sin(th)
    z = r * np.sin(phi)
    p = np.stack([x, y, z, rng.random(N)], 1).astype(np.float32)
    p[:, :3] += rng.normal(0, 0.02, (N, 3)).astype(np.float32)
    return p


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
C_MODEL, C_FFN, N_HEADS = 192, 384, 8

# (name, out_ch, in_ch, k, stride, pad) of every 2-D conv in the BEV backbone
# (include/params.h:89-211; src/dsvt-ai-trt.cpp:1144-1351)
BEV_BLOCKS = [
    # block index, in, out, stride of first conv, number of basic blocks
    (0, 192, 128, 1, 2),
    (1, 128, 128, 2, 3),
    (2, 128, 256, 2, 3),
]
# deblocks: (index, in, out, kernel=stride)   params.h:214-233
BEV_DEBLOCKS = [(0, 128, 128, 1), (1, 128, 128, 2), (2, 256, 128, 4)]
HEADS = [("center", 2), ("center_z", 1), ("dim", 3), ("rot", 2), ("iou", 1), ("hm", 10)]


def _lin(rng, out_f, in_f):
    return (rng.standard_normal((out_f, in_f)) / np.sqrt(in_f)).astype(np.float32)


def _conv(rng, o, i, k):
    return (rng.standard_normal((o, i, k, k)) / np.sqrt(i * k * k)).astype(np.float32)


def _bias(rng, n):
    return (rng.standard_normal(n) * 0.02).astype(np.float32)


def _bn(rng, w, prefix, n):
    w[prefix + ".weight"] = rng.uniform(0.8, 1.2, n).astype(np.float32)
    w[prefix + ".bias"] = (rng.standard_normal(n) * 0.05).astype(np.float32)
    w[prefix + ".running_mean"] = (rng.standard_normal(n) * 0.05).astype(np.float32)
    w[prefix + ".running_var"] = rng.uniform(0.5, 1.5, n).astype(np.float32)


def _ln(rng, w, prefix, n):
    w[prefix + ".weight"] = rng.uniform(0.8, 1.2, n).astype(np.float32)
    w[prefix + ".bias"] = (rng.standard_normal(n) * 0.05).astype(np.float32)


def make_weights(seed=1234, blocks=4, with_bev=True):
    """Seeded random weights with the reference's tensor names (dict name -> float32 ndarray)."""
    rng = np.random.default_rng(seed)
    w = {}
    C = C_MODEL
    # PFN (src/dsvt-ai-trt.cpp:577, :587) -- linears have no bias
    w["module.vfe.pfn_layers.0.linear.weight"] = _lin(rng, 96, 10)
    _bn(rng, w, "module.vfe.pfn_layers.0.norm", 96)
    w["module.vfe.pfn_layers.1.linear.weight"] = _lin(rng, C, C)
    _bn(rng, w, "module.vfe.pfn_layers.1.norm", C)
    # position-embedding MLPs (:603-637)
    for b in range(blocks):
        for l in range(2):
            p = f"module.backbone_3d.input_layer.posembed_layers.0.{b}.{l}.position_embedding_head"
            w[p + ".0.weight"] = _lin(rng, C, 2)
            w[p + ".0.bias"] = _bias(rng, C)
            _bn(rng, w, p + ".1", C)
            w[p + ".3.weight"] = _lin(rng, C, C)
            w[p + ".3.bias"] = _bias(rng, C)
    # DSVT blocks (:653-756)
    for b in range(blocks):
        for l in range(2):
            p = f"module.backbone_3d.stage_0.{b}.encoder_list.{l}"
            w[p + ".win_attn.self_attn.in_proj_weight"] = _lin(rng, 3 * C, C)
            w[p + ".win_attn.self_attn.in_proj_bias"] = _bias(rng, 3 * C)
            w[p + ".win_attn.self_attn.out_proj.weight"] = _lin(rng, C, C)
            w[p + ".win_attn.self_attn.out_proj.bias"] = _bias(rng, C)
            w[p + ".win_attn.linear1.weight"] = _lin(rng, C_FFN, C)
            w[p + ".win_attn.linear1.bias"] = _bias(rng, C_FFN)
            w[p + ".win_attn.linear2.weight"] = _lin(rng, C, C_FFN)
            w[p + ".win_attn.linear2.bias"] = _bias(rng, C)
            _ln(rng, w, p + ".win_attn.norm1", C)
            _ln(rng, w, p + ".win_attn.norm2", C)
            _ln(rng, w, p + ".norm", C)
        _ln(rng, w, f"module.backbone_3d.residual_norm_stage_0.{b}", C)
    if not with_bev:
        return w
    # BEV ResNet (:1144-1364)
    for (i, cin, cout, stride, nb) in BEV_BLOCKS:
        for j in range(nb):
            p = f"module.backbone_2d.blocks.{i}.{j}"
            w[p + ".conv1.weight"] = _conv(rng, cout, cin if j == 0 else cout, 3)
            _bn(rng, w, p + ".bn1", cout)
            w[p + ".conv2.weight"] = _conv(rng, cout, cout, 3)
            _bn(rng, w, p + ".bn2", cout)
            if j == 0:
                w[p + ".downsample_layer.0.weight"] = _conv(rng, cout, cin, 1)
                _bn(rng, w, p + ".downsample_layer.1", cout)
    for (i, cin, cout, k) in BEV_DEBLOCKS:
        p = f"module.backbone_2d.deblocks.{i}"
        # ConvTranspose2d layout [in, out, k, k]
        w[p + ".0.weight"] = (rng.standard_normal((cin, cout, k, k)) / np.sqrt(cin)).astype(np.float32)
        _bn(rng, w, p + ".1", cout)
    # CenterHead (:1369-1468)
    w["module.dense_head.shared_conv.0.weight"] = _conv(rng, 64, 384, 3)
    _bn(rng, w, "module.dense_head.shared_conv.1", 64)
    for (name, k) in HEADS:
        p = f"module.dense_head.heads_list.0.{name}"
        w[p + ".0.0.weight"] = _conv(rng, 64, 64, 3)
        _bn(rng, w, p + ".0.1", 64)
        w[p + ".1.weight"] = _conv(rng, k, 64, 3)
        w[p + ".1.bias"] = _bias(rng, k)
    # heat-map head: shift so that a useful fraction of the top-500 exceed the 0.3
    # score threshold; per-class offsets keep class maxima distinct.
    w["module.dense_head.heads_list.0.hm.1.bias"] = (
        np.float32(-1.8) + np.linspace(-0.3, 0.3, 10).astype(np.float32))
    return w


def split_in_proj(w):
    """include/helper.h:348-433: `*.in_proj_weight/bias` are cut into three row blocks
    `.query/.key/.value`.  Returns a new dict with those extra keys."""
    out = dict(w)
    for k, v in w.items():
        if k.endswith("in_proj_weight") or k.endswith("in_proj_bias"):
            n = v.shape[0] // 3
            out[k + ".query"], out[k + ".key"], out[k + ".value"] = v[:n], v[n:2 * n], v[2 * n:]
    return out


def write_wts(path, weights):
    """tools/gen_wts.py:86-99: first line = tensor count; then `name count hex...`,
    hex = big-endian IEEE-754 float32."""
    with open(path, "w") as f:
        f.write(f"{len(weights)}\n")
        for k, v in weights.items():
            flat = np.asarray(v, np.float32).reshape(-1)
            hx = flat.astype(">f4").tobytes().hex()
            f.write(f"{k} {flat.size} " + " ".join(hx[i:i + 8] for i in range(0, len(hx), 8)) + "\n")


def weight_shapes(blocks=4):
    """name -> shape of every tensor the pipeline consumes (the wiring of src/dsvt-ai-trt.cpp:577-1468)"""
    return {k: v.shape for k, v in make_weights(0, blocks).items()}


def shape_weights(flat, blocks=4):
    """`.wts` tensors are flat (the file carries element counts only, tools/gen_wts.py:93-99; the reference hands them to TensorRT
    layers whose dimensions come from include/params.h).  Gives every tensor the pipeline consumes its shape, drops the
    state_dict entries it does not use (num_batches_tracked, the dead iou head's statistics, ...) and refuses a file whose
    element counts do not fit the architecture."""
    out, missing = {}, []
    for name, shape in weight_shapes(blocks).items():
        if name not in flat:
            missing.append(name)
            continue
        a = np.asarray(flat[name], np.float32)
        if a.size != int(np.prod(shape)):
            raise ValueError(f"{name}: {a.size} elements in the file, the architecture needs {shape}")
        out[name] = a.reshape(shape)
    if missing:
        raise KeyError(f"{len(missing)} tensors missing from the weight file, e.g. {missing[:3]}")
    return out


def read_wts(path):
    """include/helper.h:328-366 (`loadWeights_new`): returns dict name -> flat float32 array."""
    out = {}
    with open(path) as f:
        count = int(f.readline())
        for _ in range(count):
            parts = f.readline().split()
            name, n = parts[0], int(parts[1])
            raw = bytes.fromhex("".join(parts[2:2 + n]))
            out[name] = np.frombuffer(raw, ">f4").astype(np.float32)
    return out


def make_weights_3d(seed=4321, stages=2, pool_volumes=(4,)):
    """Seeded random weights of a multi-stage 3-D voxel DSVT backbone (upstream DSVT's tensor names where the reference has none):
    the pillar feature net, per stage one DSVT block (two encoder layers) with position-embedding MLPs over (x, y, z) in-window coordinates, and between
    stages the attention-style stage reduction (in-proj / out-proj of an 8-head attention, a [pool_volume, C] position embedding, a LayerNorm)."""
    rng = np.random.default_rng(seed)
    w = {}
    C = C_MODEL
    w["module.vfe.pfn_layers.0.linear.weight"] = _lin(rng, 96, 10); _bn(rng, w, "module.vfe.pfn_layers.0.norm", 96)
    w["module.vfe.pfn_layers.1.linear.weight"] = _lin(rng, C, C); _bn(rng, w, "module.vfe.pfn_layers.1.norm", C)
    for s_ in range(stages):
        for l in range(2):
            p = f"module.backbone_3d.input_layer.posembed_layers.{s_}.0.{l}.position_embedding_head"
            w[p + ".0.weight"] = _lin(rng, C, 3); w[p + ".0.bias"] = _bias(rng, C); _bn(rng, w, p + ".1", C)
            w[p + ".3.weight"] = _lin(rng, C, C); w[p + ".3.bias"] = _bias(rng, C)
            p = f"module.backbone_3d.stage_{s_}.0.encoder_list.{l}"
            w[p + ".win_attn.self_attn.in_proj_weight"] = _lin(rng, 3 * C, C); w[p + ".win_attn.self_attn.in_proj_bias"] = _bias(rng, 3 * C)
            w[p + ".win_attn.self_attn.out_proj.weight"] = _lin(rng, C, C); w[p + ".win_attn.self_attn.out_proj.bias"] = _bias(rng, C)
            w[p + ".win_attn.linear1.weight"] = _lin(rng, C_FFN, C); w[p + ".win_attn.linear1.bias"] = _bias(rng, C_FFN)
            w[p + ".win_attn.linear2.weight"] = _lin(rng, C, C_FFN); w[p + ".win_attn.linear2.bias"] = _bias(rng, C)
            _ln(rng, w, p + ".win_attn.norm1", C); _ln(rng, w, p + ".win_attn.norm2", C); _ln(rng, w, p + ".norm", C)
        _ln(rng, w, f"module.backbone_3d.residual_norm_stage_{s_}.0", C)
        if s_ + 1 < stages:
            p = f"module.backbone_3d.stage_{s_}_reduction"
            w[p + ".self_attn.in_proj_weight"] = _lin(rng, 3 * C, C); w[p + ".self_attn.in_proj_bias"] = _bias(rng, 3 * C)
            w[p + ".self_attn.out_proj.weight"] = _lin(rng, C, C); w[p + ".self_attn.out_proj.bias"] = _bias(rng, C)
            w[p + ".pos_embedding"] = (rng.standard_normal((pool_volumes[s_], C)) * 0.5).astype(np.float32)     # (upstream initialises at std 0.01; 0.5 makes a wrong slot order visible)
            _ln(rng, w, p + ".norm", C)
    return w
