"""End-to-end parity of the HIP pipeline against the oracle's restatement of the reference
network (seeded synthetic weights: the reference's dsvt.wts is not shipped)."""
import numpy as np
import pytest
import torch

from tests import cases
from tests.parity import match_boxes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(pkg, pipe, pts, n):
    return pipe.forward(torch.from_numpy(pts[None]).to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV))


@pytest.fixture(scope="module")
def weights(pkg):
    return pkg.synth.make_weights()


def _frame_and_caps(pkg, frame):
    """a reference frame under the reference caps, or `lidar_like(N, 0)` (frame = "lidar<N>") under the Waymo-sized caps"""
    if frame.startswith("lidar"):                 # "lidar<N>" (seed 0) or "lidar<N>s<seed>"
        caps = pkg.pipeline.Caps()
        npts, _, seed = frame[5:].partition("s")
        pts, n = cases.pad_points(pkg.synth.lidar_like(int(npts), int(seed or 0)), caps.N)
    else:
        caps = pkg.pipeline.Caps.reference()
        pts, n = cases.load_frame(frame, caps.N)
    return caps, pts, n


def _oracle_cfg(caps):
    from oracle import dense_ref as D
    return D.OracleCfg(max_points=caps.N, max_points_filter=caps.Nk, max_pillars=caps.P, max_win=caps.W, max_vox_per_win=caps.Vw,
                       max_sets=caps.S)


def test_backbone_features_frame000000(pkg, oracle, weights):
    """config 2 shape of test (voxelize + partition + DSVT blocks, fp32) on the reference frame:
    per-layer voxel features against the oracle."""
    from oracle import dense_ref as D
    caps = pkg.pipeline.Caps.reference()
    pipe = pkg.pipeline.DsvtPipeline(weights, caps=caps, with_head=False, device=DEV)
    pts, n = cases.load_frame("000000", caps.N)
    st = pipe.voxel_stage(torch.from_numpy(pts[None]).to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV))
    tr = {}
    x = pipe.backbone(st, trace=tr)
    torch.cuda.synchronize()
    cfg = D.OracleCfg()
    ost = D.voxel_stage(pts, n, weights, cfg)
    otr = {}
    ox = D.dsvt_blocks(ost, weights, cfg, trace=otr)
    Pn = ost["P"]
    vf = st["vfeat"][0, :Pn].cpu().numpy()
    assert np.abs(vf - ost["vfeat"][:Pn]).max() < 1e-5 * max(1.0, np.abs(ost["vfeat"]).max())     # PFN + scatter-max
    for b in range(4):
        ref = otr[(b, "res")][:Pn]
        got = tr[(b, 1)][0, :Pn].cpu().numpy()
        err = np.abs(got - ref).max()
        assert err < 2e-4, (b, err)          # LayerNorm-ed O(1) activations; fp32 summation-order noise grows with depth
    assert np.abs(x[0, :Pn].cpu().numpy() - ox[:Pn]).max() < 2e-4


def test_config1_60k_cloud_one_block_fp32(pkg, oracle, weights):
    """BASELINE configs[1]: lidar_like(60000, 0), caps 65536 / 32768 / 2048, voxelize + WindowPartition_0 + GetSet_0 + ONE DSVT block
    (two encoder layers: gather / MHA / scatter / LN / FFN, then the block LayerNorm) in fp32 on one MI355X.  Indices and set
    assignment bit-exact against the oracle, features per layer within 2e-4 (fp32 summation order through two layers)."""
    from oracle import dense_ref as D
    c = cases.caps("mid")
    caps = pkg.pipeline.Caps(c["N"], c["Nk"], c["P"], c["W"], c["Vw"])
    assert caps.overflow_free()
    pts, n = cases.pad_points(pkg.synth.lidar_like(60000, 0), caps.N)
    pipe = pkg.pipeline.DsvtPipeline(weights, caps=caps, blocks=1, with_head=False, device=DEV, zero_fill=True)
    st = pipe.voxel_stage(torch.from_numpy(pts[None]).to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV))
    tr = {}
    x = pipe.backbone(st, trace=tr)
    torch.cuda.synchronize()
    cfg = D.OracleCfg(max_points=caps.N, max_points_filter=caps.Nk, max_pillars=caps.P, max_win=caps.W, max_vox_per_win=caps.Vw,
                      max_sets=caps.S, blocks=1)
    ost = D.voxel_stage(pts, n, weights, cfg)
    otr = {}
    ox = D.dsvt_blocks(ost, weights, cfg, trace=otr)
    Pn, v = ost["P"], ost["vox"]
    assert (int(st["P"][0]), int(st["Nk"][0])) == (Pn, ost["Nk"]) == (19445, 57577)              # SURVEY 8c known answers
    u32 = lambda t: t[0].cpu().numpy().view(np.uint32)
    assert np.array_equal(u32(st["coords"]), v["coords"]) and np.array_equal(u32(st["pidx"]), v["pidx"])
    assert np.array_equal(u32(st["pcnt"]), v["pcnt"])
    wp, gs, owp, ogs = st["wps"][0], st["gss"][0], ost["wps"][0], ost["gss"][0]
    assert int(wp[3][0]) == owp["W"] == 1250 and int(gs[2][0]) == ogs["S"] == 1471
    assert np.array_equal(u32(wp[0]), owp["gidx"]) and np.array_equal(u32(wp[1]), owp["cinw"]) and np.array_equal(u32(wp[2]), owp["vcnt"])
    assert np.array_equal(u32(gs[0]), ogs["inds"]) and np.array_equal(gs[1][0].cpu().numpy(), ogs["mask"])
    assert np.abs(st["vfeat"][0, :Pn].cpu().numpy() - ost["vfeat"][:Pn]).max() < 1e-5 * np.abs(ost["vfeat"]).max()
    for l in range(2):
        assert np.abs(tr[(0, l)][0, :Pn].cpu().numpy() - (otr[(0, l)][:Pn] if l == 0 else otr[(0, "res")][:Pn])).max() < 2e-4, l
    assert np.abs(x[0, :Pn].cpu().numpy() - ox[:Pn]).max() < 2e-4
    assert not x[0, Pn:].any()


@pytest.mark.parametrize("frame", ["000000", "000004", "lidar180000", "lidar60000s3"])
def test_boxes_reference_frames(pkg, oracle, weights, frame):
    """fp32 mode of the HIP path against the oracle at the north-star tolerance, on reference frames and on the bench frame
    (BASELINE configs[2] size: 180k points, 34.5k pillars, 1714 / 1158 sets)."""
    from tests import golden_oracle as GO
    caps, pts, n = _frame_and_caps(pkg, frame)
    pipe = pkg.pipeline.DsvtPipeline(weights, caps=caps, device=DEV)
    boxes, cnt = _run(pkg, pipe, pts, n)
    torch.cuda.synchronize()
    eb, ec = GO.forward(frame, pts, n, weights, caps)          # dense_ref.forward's rows for this cloud (tests/golden/oracle_boxes.npz, tools/make_golden.py)
    assert 0 < ec <= 500 and (ec < 500 or frame.startswith("lidar"))      # the score threshold actually filters the reference frames
    worst, unmatched = match_boxes(boxes[0].cpu().numpy(), int(cnt[0]), eb, ec)
    assert unmatched == 0 and worst < 1e-3, (worst, unmatched)     # north-star tolerance: 1e-3 fp32


def test_pipeline_is_deterministic(pkg, weights):
    """Every kernel of the frame is hand-written and race-free (no vendor-library stage since round 2): features AND boxes are
    bit-reproducible run to run."""
    caps = pkg.pipeline.Caps.reference()
    pipe = pkg.pipeline.DsvtPipeline(weights, caps=caps, device=DEV)
    pts, n = cases.load_frame("000003", caps.N)
    p_d, n_d = torch.from_numpy(pts[None]).to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV)
    x0 = pipe.backbone(pipe.voxel_stage(p_d, n_d)).clone()
    a, ca = pipe.forward(p_d, n_d); a, ca = a.clone(), ca.clone()
    for _ in range(2):
        x1 = pipe.backbone(pipe.voxel_stage(p_d, n_d))
        assert torch.equal(x0, x1), float((x0 - x1).abs().max())
        b, cb = pipe.forward(p_d, n_d)
        assert torch.equal(a, b) and torch.equal(ca, cb)


@pytest.mark.parametrize("mode,persistent_bev", [("f16", False), ("split", False), ("split", True)])
def test_boxes_do_not_depend_on_stale_buffer_contents(pkg, weights, mode, persistent_bev):
    """Uninitialised-read sanitiser: every plugin's outputs and workspace are overwritten with a byte pattern (zeros, 0xff = NaN / -1,
    random) before a forward; the boxes are the same bits each time.  (TensorRT hands plugins uninitialised workspace and outputs.)
    persistent_bev (the pipeline's default): Map2Bev's output is BY CONTRACT a buffer the plugin alone writes, call after call (it zeroes only the cells
    its previous call wrote) -- that one buffer is left alone, every other one is overwritten as before."""
    P = pkg.plugin
    made, init = [], P.Plugin.__init__
    def tracking(self, *a, **k):
        init(self, *a, **k); made.append(self)
    P.Plugin.__init__ = tracking
    try:
        kw = dict(linear_compute=P.COMPUTE_SPLIT) if mode == "split" else dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)
        caps = pkg.pipeline.Caps()
        pipe = pkg.pipeline.DsvtPipeline(weights, caps=caps, device=DEV, device_nms=True, persistent_bev=persistent_bev, **kw)
    finally:
        P.Plugin.__init__ = init
    if persistent_bev:
        made = [pl for pl in made if pl.plugin_type != "Map2BevPlugin"]
    pts, n = cases.pad_points(pkg.synth.lidar_like(120000, 4), caps.N)
    p_d, n_d = torch.from_numpy(pts[None]).to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV)
    b0, c0 = [t.clone() for t in pipe.forward(p_d, n_d)]
    assert int(c0[0]) > 50 and len(made) > 40
    g = torch.Generator(device=DEV); g.manual_seed(99)
    for kind in ("ff", "rand", "zero"):
        for pl in made:
            for outs, ws in pl._cache.values():
                for t in list(outs) + [ws]:
                    v = t.view(-1).view(torch.uint8)
                    if kind == "ff":
                        v.fill_(255)
                    elif kind == "rand":
                        v.copy_(torch.randint(0, 256, v.shape, dtype=torch.uint8, device=DEV, generator=g))
                    else:
                        v.zero_()
        b, c = pipe.forward(p_d, n_d)
        assert torch.equal(c, c0) and torch.equal(b, b0), (kind, c.tolist(), c0.tolist())


@pytest.mark.parametrize("mode,frames", [("split", 1), ("f16", 2)])
def test_consecutive_forwards_do_not_see_each_others_bev_cells(pkg, weights, mode, frames):
    """Map2Bev's persistent_output (the pipeline's default: a forward zeroes only the cells the forward before it wrote): a dense cloud, a sparse one,
    an EMPTY frame and the dense one again through ONE captured graph (inputs refilled in place) give the bits of a stateless pipeline
    (persistent_bev=False: whole-map fill per call) that sees each cloud alone."""
    P = pkg.plugin
    kw = dict(linear_compute=P.COMPUTE_SPLIT) if mode == "split" else dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)
    caps = pkg.pipeline.Caps() if frames == 1 else pkg.pipeline.Caps.for_frames(frames)
    pipe = pkg.pipeline.DsvtPipeline(weights, caps=caps, device=DEV, device_nms=True, frames=frames, **kw)
    ref = pkg.pipeline.DsvtPipeline(weights, caps=caps, device=DEV, device_nms=True, frames=frames, persistent_bev=False, **kw)
    seqs = [[180000, 150000], [30000, 7], [0, 0], [180000, 60000], [1, 120000]]
    pts_d = torch.zeros((1, frames * caps.N, 4), device=DEV); n_d = torch.zeros(frames, dtype=torch.int32, device=DEV)
    def fill(step, counts):
        buf = np.zeros((1, frames * caps.N, 4), np.float32)
        for f in range(frames):
            if counts[f]:
                p = pkg.synth.lidar_like(counts[f], 20 + 3 * step + f); buf[0, f * caps.N:f * caps.N + len(p)] = p
        pts_d.copy_(torch.from_numpy(buf)); n_d.copy_(torch.tensor(counts[:frames], dtype=torch.int32))
    fill(0, seqs[0])
    pipe.capture(pts_d, n_d)
    total = 0
    for step, counts in enumerate(seqs):
        fill(step, counts)
        rows, cnt = pipe.replay()
        torch.cuda.synchronize()
        rows, cnt = rows.clone(), cnt.clone()
        r2, c2 = ref.forward(pts_d, n_d)
        torch.cuda.synchronize()
        assert torch.equal(cnt, c2) and torch.equal(rows, r2), (step, cnt.tolist(), c2.tolist())
        total += int(cnt.sum())
    assert total > 100


@pytest.mark.parametrize("mode,frames", [("f16", 4), ("split", 1), ("f16", 1)])
def test_no_plugin_writes_outside_its_buffers(pkg, weights, mode, frames):
    """Guard-band sanitiser: every output and workspace of every plugin of the frame is allocated between two 4 KB bands of 0xA5 bytes
    (plugin.GUARD_BYTES); after forwards over dense, sparse and empty frames every band is intact."""
    P = pkg.plugin
    P.GUARD_BYTES, P.GUARDED[:] = 4096, []
    try:
        kw = dict(linear_compute=P.COMPUTE_SPLIT) if mode == "split" else dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)
        caps = pkg.pipeline.Caps() if frames == 1 else pkg.pipeline.Caps.for_frames(frames)
        pipe = pkg.pipeline.DsvtPipeline(weights, caps=caps, device=DEV, device_nms=True, frames=frames, **kw)
        counts = []
        for trial, npts in enumerate([[180000, 60000, 0, 196608][:frames], [1, 37, 120000, 5][:frames]]):
            buf = np.zeros((1, frames * caps.N, 4), np.float32)
            for f, m in enumerate(npts):
                if m:
                    p = pkg.synth.lidar_like(m, 20 + 4 * trial + f); buf[0, f * caps.N:f * caps.N + len(p)] = p
            b, c = pipe.forward(torch.from_numpy(buf).to(DEV), torch.tensor(npts, dtype=torch.int32, device=DEV))
            torch.cuda.synchronize()
            counts.append(c.tolist())
        assert len(P.GUARDED) > 100 and max(counts[0]) > 50
        G = P.GUARD_BYTES
        for whole, n in P.GUARDED:
            assert bool((whole[:G] == 0xA5).all()) and bool((whole[G + n:] == 0xA5).all()), (n, int((whole[:G] != 0xA5).sum()), int((whole[G + n:] != 0xA5).sum()))
    finally:
        P.GUARD_BYTES, P.GUARDED[:] = 0, []


def _box_errors(got, n_got, exp, n_exp):
    """per-field max abs error over rows matched by (class, nearest centre within 0.2 m); fraction matched"""
    got, exp = got[:n_got], exp[:n_exp]
    used = np.zeros(n_got, bool)
    errs, matched = [], 0
    for e in exp:
        d = np.abs(got[:, :2] - e[:2]).max(1) + (got[:, 7] != e[7]) * 1e3 + used * 1e3
        j = int(np.argmin(d))
        if d[j] > 0.2:
            continue
        used[j] = True; matched += 1
        d = np.abs(got[j] - e)
        d[6] = min(d[6], abs(np.pi - d[6]))      # yaw = atan(sin/cos) lives in (-pi/2, pi/2): +-pi/2 are the same heading (cos ~ 0 flips the sign)
        errs.append(d)
    errs = np.array(errs)
    return errs.max(0), matched / max(n_exp, 1)


# Bounds of the fp16 frame against the fp32 oracle: centre x,y / z / size / score, yaw.  The north-star 1e-3 is met by the fp32 mode
# (test_boxes_reference_frames, 1e-5 measured).  For fp16 OPERANDS it is out of reach, and not because of one tensor:
# profiles/r02_f16_error_attribution.txt swaps stages between the two modes and shows that ONE fp16 rounding site -- the weights of the
# last convolution alone, or its input alone, or the pillar feature net alone -- already moves z / size by 3.5e-4 .. 5.5e-4 (2^-12 mean
# relative operand error x O(1) outputs, max over 500 boxes; size = exp(d) multiplies it by the box size), and the ~60 sites of the frame
# add in quadrature to what is measured here: 180k-point frame xy 7.3e-4, z 2.0e-3, size 4.0e-3, score 5.8e-4; reference frames
# xy 5.7e-4, z 1.7e-3, size 1.9e-3, score 4e-4.  The bounds are those figures x 1.5.
F16_TOL = dict(xy=1.2e-3, z=3e-3, size=6e-3, score=1e-3, yaw=1e-1)


@pytest.mark.parametrize("frame", ["000000", "000003", "000004", "lidar180000", "lidar60000s3", "lidar196000s5"])
def test_boxes_f16_mode(pkg, oracle, weights, frame):
    """BASELINE configs[2] precision ("fp16"): fp16 MFMA operands / fp16 BEV activations, fp32 accumulate,
    fp32 LayerNorm / softmax / decode -- the mode bench.py times by default, on the three distinct reference frames and on the
    bench frame itself (lidar_like(180000, 0)).  The reference's own fp16 build (TensorRT kFP16, include/params.h:332) is not
    reproducible, so this mode is judged against the fp32 oracle with the fp16-sized bounds F16_TOL derived above.  The yaw is atan(sin/cos) of two
    raw head outputs (src/dsvt-ai-trt.cpp:1668-1669): ill-conditioned when cos ~ 0, so its bound is loose."""
    from tests import golden_oracle as GO
    caps, pts, n = _frame_and_caps(pkg, frame)
    pipe = pkg.pipeline.DsvtPipeline(weights, caps=caps, device=DEV, linear_compute=pkg.plugin.COMPUTE_F16,
                                     head_dtype=torch.float16)
    assert pipe.hip_head
    boxes, cnt = _run(pkg, pipe, pts, n)
    torch.cuda.synchronize()
    eb, ec = GO.forward(frame, pts, n, weights, caps)
    err, frac = _box_errors(boxes[0].cpu().numpy(), int(cnt[0]), eb, ec)
    print("f16 box errors per field", frame, err, "matched", frac, "counts", int(cnt[0]), ec)
    assert frac >= 0.99
    t = F16_TOL
    assert err[:2].max() < t["xy"] and err[2] < t["z"] and err[3:6].max() < t["size"] and err[8] < t["score"], err
    assert err[6] < t["yaw"]
    assert abs(int(cnt[0]) - ec) <= 2


def test_frame_uploader_feeds_the_pipeline(pkg):
    """hostio.FrameUploader moves n x 16 bytes (not the zero-padded cap) from pinned memory; the plugins bound every loop by
    the device-side count, so stale rows beyond n from an earlier, larger frame must not matter."""
    P = pkg.plugin
    c = cases.caps("ref")
    up = pkg.hostio.FrameUploader(c["N"], depth=1)
    vox = P.add_voxel_generator(c["N"], c["Nk"], c["P"], 4, 10, 48, -74.88, 74.88, -74.88, 74.88, -5.0, 3.0, 0.32, 0.32, 8.0, 468, 468, 1)
    big, nb = pkg.hostio.load_bin(f"{cases.GOLDEN}/000000.bin", c["N"])
    small, ns = pkg.hostio.load_bin(f"{cases.GOLDEN}/000004.bin", c["N"])
    assert nb != ns
    outs = {}
    for name, (pts, n) in (("big", (big, nb)), ("small", (small, ns)), ("big2", (big, nb))):
        d, dn = up.upload(pts, n)
        outs[name] = [t.clone() for t in vox(d, dn)]
    torch.cuda.synchronize()
    for a, b in zip(outs["big"], outs["big2"]):
        assert torch.equal(a, b)
    # the same frame through a freshly zero-padded buffer
    pad, n = cases.load_frame("000004", c["N"])
    ref = vox(torch.from_numpy(pad[None]).to("cuda:0"), torch.tensor([n], dtype=torch.int32, device="cuda:0"))
    torch.cuda.synchronize()
    for a, b in zip(outs["small"], ref):
        assert torch.equal(a, b)


def test_full_size_frame_properties(pkg):
    """BASELINE configs[2] size (180k-point Waymo-shaped cloud, fp16 mode, frame ends at the final boxes): properties that do
    not need the CPU oracle (10 s per frame) -- reproducibility, ordering / range of the boxes, NMS idempotence, empty input."""
    P = pkg.plugin
    caps = pkg.pipeline.Caps()
    w = pkg.synth.make_weights()
    pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, linear_compute=P.COMPUTE_F16, head_dtype=torch.float16, device_nms=True)
    p = pkg.synth.lidar_like(180000, seed=3)
    buf = np.zeros((1, caps.N, 4), np.float32); buf[0, :p.shape[0]] = p
    pts = torch.from_numpy(buf).to("cuda:0"); n = torch.tensor([p.shape[0]], dtype=torch.int32, device="cuda:0")
    rows, cnt = [t.clone() for t in pipe.forward(pts, n)]
    rows2, cnt2 = [t.clone() for t in pipe.forward(pts, n)]
    torch.cuda.synchronize()
    assert torch.equal(rows, rows2) and torch.equal(cnt, cnt2)                  # bit-reproducible, atomics included
    k = int(cnt.cpu()[0])
    r = rows[0, :k].cpu().numpy()
    assert 0 < k <= 500 and not rows[0, k:].any()
    assert np.all(r[:-1, 8] >= r[1:, 8]) and r[:, 8].min() >= 0.3 and r[:, 8].max() <= 1.0     # score order, threshold (params.h:328)
    assert np.all((r[:, 0] >= -74.88) & (r[:, 0] < 74.88) & (r[:, 1] >= -74.88) & (r[:, 1] < 74.88))
    assert np.all((r[:, 7] >= 0) & (r[:, 7] < 10) & (r[:, 7] == np.round(r[:, 7])))
    assert np.all(r[:, 3:6] > 0)                                                                # exp(dim)
    # NMS of an NMS result keeps everything, in the same order
    again, idx, c2 = P.add_rotated_nms_op(500, 0.01)(rows, cnt)
    torch.cuda.synchronize()
    assert int(c2.cpu()[0]) == k and torch.equal(again[0, :k], rows[0, :k]) and idx[0, :k].cpu().tolist() == list(range(k))
    # empty cloud
    rows0, cnt0 = pipe.forward(pts, torch.zeros_like(n))
    torch.cuda.synchronize()
    assert int(cnt0.cpu()[0]) == 0 and not rows0.any()


def test_detect_directory_writes_reference_txt(pkg, weights, tmp_path):
    """detect.run_directory = the reference's `-d` loop (src/dsvt-ai-trt.cpp:1876-1960): every .bin of the directory -> one .txt
    in save_txt's layout, holding the boxes the pipeline (device top-K + FilterBoxByScore + rotated NMS) returns for that frame."""
    import subprocess, sys, os
    out = tmp_path / "outputs"
    done = pkg.detect.run_directory(cases.GOLDEN, str(out), weights, caps=pkg.pipeline.Caps.reference(), fp16=False, log=lambda *_: None)
    assert [d[0] for d in done] == ["000000", "000003", "000004"]
    c = pkg.pipeline.Caps.reference()
    pipe = pkg.pipeline.DsvtPipeline(weights, caps=c, device_nms=True, linear_compute=pkg.plugin.COMPUTE_SPLIT)      # detect's fp32-grade mode
    for name, kept, ms in done:
        sec, rows = pkg.detect.read_txt(str(out / f"{name}.txt"))
        assert rows.shape == (kept, 9) and abs(sec - ms) < 1e-3
        pts, n = cases.load_frame(name, c.N)
        r, cnt = pipe.forward(torch.from_numpy(pts[None]).to("cuda:0"), torch.tensor([n], dtype=torch.int32, device="cuda:0"))
        k = int(cnt[0])
        assert k == kept
        exp = r.reshape(-1, 9)[:k].cpu().numpy()
        assert np.abs(rows - exp).max() < 2e-6 * max(1.0, np.abs(exp).max())        # text carries 6 decimals
        assert np.array_equal(rows[:, 7], exp[:, 7])
    # the command-line wrapper: same files
    out2 = tmp_path / "cli"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(root, "tools", "detect.py"), "--data", cases.GOLDEN, "--out", str(out2), "--fp32", "--ref-caps"],
                   check=True, timeout=600, capture_output=True)
    for name, _, _ in done:
        _, a = pkg.detect.read_txt(str(out / f"{name}.txt"))
        _, b = pkg.detect.read_txt(str(out2 / f"{name}.txt"))
        assert a.shape == b.shape and np.array_equal(a, b)      # (another process, the same deterministic kernels: the same text)


@pytest.mark.parametrize("n_pts", [0, 1, 37])
def test_degenerate_frames_run_through_the_fp16_pipeline(pkg, weights, n_pts):
    """Empty / one-point / few-point frames: every kernel of the fp16 frame (elastic row tiles, set attention with zero or
    one set, convolutions over an empty BEV map, top-K / NMS on bias-only logits) returns, twice the same, with finite rows."""
    P = pkg.plugin
    c = pkg.pipeline.Caps.reference()
    pipe = pkg.pipeline.DsvtPipeline(weights, caps=c, linear_compute=P.COMPUTE_F16, head_dtype=torch.float16, device_nms=True)
    pts = np.zeros((1, c.N, 4), np.float32)
    if n_pts:
        pts[0, :n_pts] = pkg.synth.lidar_like(max(n_pts, 8), 3)[:n_pts]
    outs = []
    for _ in range(2):
        rows, cnt = pipe.forward(torch.from_numpy(pts).to(DEV), torch.tensor([n_pts], dtype=torch.int32, device=DEV))
        torch.cuda.synchronize()
        outs.append((rows.clone(), int(cnt[0])))
    assert outs[0][1] == outs[1][1] and 0 <= outs[0][1] <= 500
    k = outs[0][1]
    assert torch.equal(outs[0][0].reshape(-1, 9)[:k], outs[1][0].reshape(-1, 9)[:k])
    assert torch.isfinite(outs[0][0].reshape(-1, 9)[:k]).all()


def test_weights_from_a_wts_file_give_the_same_boxes(pkg, weights, tmp_path):
    """tools/detect.py --wts: the .wts reader returns flat tensors (like the reference's loadWeights, include/helper.h:328-366);
    detect.load_weights shapes them.  Same boxes, bit for bit, as the in-memory weights."""
    path = str(tmp_path / "dsvt.wts")
    pkg.synth.write_wts(path, weights)
    w2 = pkg.detect.load_weights(path, log=lambda *_: None)
    caps = pkg.pipeline.Caps.reference()
    pts, n = cases.load_frame("000003", caps.N)
    kw = dict(caps=caps, device=DEV, linear_compute=pkg.plugin.COMPUTE_F16, head_dtype=torch.float16, device_nms=True)
    a = [t.clone() for t in _run(pkg, pkg.pipeline.DsvtPipeline(weights, **kw), pts, n)]
    b = [t.clone() for t in _run(pkg, pkg.pipeline.DsvtPipeline(w2, **kw), pts, n)]
    torch.cuda.synchronize()
    assert int(a[1][0]) > 0 and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_two_frames_per_forward_equal_two_forwards(pkg, weights):
    """DsvtPipeline(frames=2): the rows of both frames concatenated through ONE launch per backbone layer (Points2Features / set partition /
    Map2Bev with the frame index as an extra grid dimension, rows = sum of the frames' pillars), the dense stage / decode / NMS per frame
    through the C ABI's batched enqueue.  Every row, set and window is computed exactly as in its own single-frame run, so the final
    boxes of each frame are the same BITS as a frames=1 pipeline gives for it -- on two different clouds, in both slot orders."""
    P = pkg.plugin
    kw = dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16, device_nms=True, device=DEV)
    one = pkg.pipeline.DsvtPipeline(weights, caps=pkg.pipeline.Caps(), **kw)
    caps2 = pkg.pipeline.Caps.for_frames(2)
    two = pkg.pipeline.DsvtPipeline(weights, caps=caps2, frames=2, **kw)
    clouds = [pkg.synth.lidar_like(180000, 5), pkg.synth.lidar_like(120000, 6)]
    singles = []
    for p in clouds:
        pts, n = cases.pad_points(p, one.caps.N)
        r, c = _run(pkg, one, pts, n)
        torch.cuda.synchronize()
        singles.append((r[0].clone(), int(c[0])))
    assert singles[0][1] > 0 and singles[1][1] > 0 and singles[0][1] != singles[1][1]
    for order in ((0, 1), (1, 0)):
        buf = np.zeros((1, 2 * caps2.N, 4), np.float32)
        for slot, k in enumerate(order):
            buf[0, slot * caps2.N:slot * caps2.N + clouds[k].shape[0]] = clouds[k]
        n = torch.tensor([clouds[k].shape[0] for k in order], dtype=torch.int32, device=DEV)
        rows, cnt = two.forward(torch.from_numpy(buf).to(DEV), n)
        torch.cuda.synchronize()
        assert rows.shape == (2, 500, 9) and cnt.shape == (2,)
        for slot, k in enumerate(order):
            assert int(cnt[slot]) == singles[k][1]
            assert torch.equal(rows[slot], singles[k][0]), float((rows[slot] - singles[k][0]).abs().max())
    # graph replay of the two-frame forward
    pts_d = torch.from_numpy(buf).to(DEV)
    two.capture(pts_d, n)
    g_rows, g_cnt = two.replay()
    torch.cuda.synchronize()
    assert torch.equal(g_rows, rows) and torch.equal(g_cnt, cnt)


def test_four_frames_per_forward_equal_single_frame_forwards(pkg, weights):
    """DsvtPipeline(frames=4), bench.py's default (BASELINE configs[3]: four frames per GPU and batch).  At this row count the encoder MLP
    runs its four-wave x 32-row kernel (two workgroups per CU) instead of the elastic one; both accumulate every row's dot products in
    the same k order, so the boxes are still the bits of the single-frame runs -- four different clouds, one of them small."""
    P = pkg.plugin
    kw = dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16, device_nms=True, device=DEV)
    one = pkg.pipeline.DsvtPipeline(weights, caps=pkg.pipeline.Caps(), **kw)
    caps4 = pkg.pipeline.Caps.for_frames(4)
    four = pkg.pipeline.DsvtPipeline(weights, caps=caps4, frames=4, **kw)
    clouds = [pkg.synth.lidar_like(180000, 11), pkg.synth.lidar_like(150000, 12), pkg.synth.lidar_like(180000, 13), pkg.synth.lidar_like(30000, 14)]
    singles = []
    for p in clouds:
        pts, n = cases.pad_points(p, one.caps.N)
        r, c = _run(pkg, one, pts, n)
        torch.cuda.synchronize()
        singles.append((r[0].clone(), int(c[0])))
    buf = np.zeros((1, 4 * caps4.N, 4), np.float32)
    for slot, p in enumerate(clouds):
        buf[0, slot * caps4.N:slot * caps4.N + p.shape[0]] = p
    n = torch.tensor([p.shape[0] for p in clouds], dtype=torch.int32, device=DEV)
    pts_d = torch.from_numpy(buf).to(DEV)
    rows, cnt = four.forward(pts_d, n)
    torch.cuda.synchronize()
    assert rows.shape == (4, 500, 9) and cnt.shape == (4,)
    for slot in range(4):
        assert int(cnt[slot]) == singles[slot][1] and singles[slot][1] > 0
        assert torch.equal(rows[slot], singles[slot][0]), (slot, float((rows[slot] - singles[slot][0]).abs().max()))
    four.capture(pts_d, n)
    g_rows, g_cnt = four.replay()
    torch.cuda.synchronize()
    assert torch.equal(g_rows, rows) and torch.equal(g_cnt, cnt)


def test_four_frames_per_forward_with_an_empty_and_a_one_point_frame(pkg, weights):
    """ragged stack: [180k points, NO points, ONE point, 120k points] in one forward() -- the empty frames contribute no pillar rows,
    no sets and an all-bias BEV map (whose top-K takes the exact fallback), the others must come out as from their own single-frame
    runs, bit for bit, whatever sits in the neighbouring slots."""
    P = pkg.plugin
    kw = dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16, device_nms=True, device=DEV)
    one = pkg.pipeline.DsvtPipeline(weights, caps=pkg.pipeline.Caps(), **kw)
    caps4 = pkg.pipeline.Caps.for_frames(4)
    four = pkg.pipeline.DsvtPipeline(weights, caps=caps4, frames=4, **kw)
    single_pt = np.array([[3.1, -7.9, -1.0, 0.5]], np.float32)
    clouds = [pkg.synth.lidar_like(180000, 21), np.zeros((0, 4), np.float32), single_pt, pkg.synth.lidar_like(120000, 22)]
    singles = []
    for p in clouds:
        pts, n = cases.pad_points(p, one.caps.N)
        r, c = _run(pkg, one, pts, n)
        torch.cuda.synchronize()
        singles.append((r[0].clone(), int(c[0])))
    buf = np.zeros((1, 4 * caps4.N, 4), np.float32)
    for slot, p in enumerate(clouds):
        buf[0, slot * caps4.N:slot * caps4.N + p.shape[0]] = p
    n = torch.tensor([p.shape[0] for p in clouds], dtype=torch.int32, device=DEV)
    rows, cnt = four.forward(torch.from_numpy(buf).to(DEV), n)
    torch.cuda.synchronize()
    assert singles[0][1] > 0 and singles[3][1] > 0
    for slot in range(4):
        assert int(cnt[slot]) == singles[slot][1], (slot, int(cnt[slot]), singles[slot][1])
        k = singles[slot][1]
        assert torch.equal(rows[slot][:k], singles[slot][0][:k]), (slot, float((rows[slot][:k] - singles[slot][0][:k]).abs().max()))
