"""The reference's detection loop (src/dsvt-ai-trt.cpp:1876-1960, the `-d` mode of its executable) around the hot path:
read `<data>/*.bin`, run the frame, write `<out>/<name>.txt`.

What the reference does per frame, and what happens here instead:
  * loadData + a zero-padded cap-sized cudaMemcpy (:1907-1925)  ->  hostio.load_bin + FrameUploader (n x 16 bytes, pinned, async)
  * context->enqueueV2 (:1928)                                 ->  one HIP-graph replay of the C-ABI plugin chain
  * D2H of the 500 x 9 rows, save_result, nms_cpu (:1931-1946) ->  RotatedNmsPlugin on the device; only the kept rows + their
                                                                   count come back
  * save_txt(nms_pred, "<name>.txt", seconds) (:1950-1958)      ->  hostio.save_txt, same text layout
The reference names ten frames 000000..000009 under ../data/bin; here every *.bin of the directory is processed in sorted order.
Weights come from a `.wts` file in the reference's format (tools/gen_wts.py:86-99); the reference's dsvt.wts is not shipped with
it, so without --wts the seeded synthetic weights of synth.make_weights are used (and the caller is told so)."""
import glob
import os
import time

import numpy as np
import torch

from . import hostio, synth
from . import plugin as P
from .pipeline import Caps, DsvtPipeline


class Detector:
    """One graph-captured frame pipeline + an uploader; detect(points) -> (rows [k, 9] float32 numpy, milliseconds)."""

    def __init__(self, weights, caps=None, fp16=False, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("dsvt detect: no GPU visible (the pipeline has no CPU path)")
        self.caps = caps or Caps()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(self.device)
        # default: the reference's own precision -- fp32 arithmetic (include/params.h:332) -- on split-precision fp16 MFMA (boxes within 1e-3 of the
        # fp32 oracle); fp16 = BASELINE configs[2]'s faster operands (z / size 2e-3 .. 4e-3 off)
        kw = dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16) if fp16 else dict(linear_compute=P.COMPUTE_SPLIT)
        with torch.cuda.stream(self.stream):
            self.pipe = DsvtPipeline(weights, caps=self.caps, device=device, device_nms=True, **kw)
            self.up = hostio.FrameUploader(self.caps.N, device=device, depth=1)
            self.points, self.count = self.up.dev[0], self.up.dcnt[0]
            self.count.zero_()
            self.rows, self.kept = self.pipe.capture(self.points, self.count)
        torch.cuda.synchronize(self.device)

    def detect(self, points, n=None):
        t0 = time.perf_counter()
        with torch.cuda.stream(self.stream):
            self.up.upload(points, n)
            self.pipe.replay()
            k = int(self.kept.cpu()[0])                      # synchronises the frame
            rows = self.rows.reshape(-1, 9)[:k].cpu().numpy()
        return rows, (time.perf_counter() - t0) * 1e3


def run_directory(data_dir, out_dir, weights, caps=None, fp16=False, device="cuda:0", log=print):
    """-> [(frame name, boxes kept, milliseconds)]; writes <out_dir>/<frame>.txt in the reference's save_txt layout."""
    files = sorted(glob.glob(os.path.join(data_dir, "*.bin")))
    if not files:
        raise FileNotFoundError(f"no .bin frames under {data_dir}")
    os.makedirs(out_dir, exist_ok=True)
    det = Detector(weights, caps=caps, fp16=fp16, device=device)
    done = []
    for path in files:
        pts, n = hostio.load_bin(path, det.caps.N)
        rows, ms = det.detect(pts, n)
        name = os.path.splitext(os.path.basename(path))[0]
        hostio.save_txt(os.path.join(out_dir, name + ".txt"), rows, ms)     # the reference's "seconds" is in ms too (:1948)
        log(f"{name}: {n} points -> {rows.shape[0]} boxes, {ms:.3f} ms")
        done.append((name, int(rows.shape[0]), ms))
    return done


def load_weights(wts=None, seed=1234, log=print):
    if wts:
        return synth.shape_weights(synth.read_wts(wts))      # the file holds flat tensors; the pipeline needs shapes
    log(f"no --wts given: seeded synthetic weights (seed {seed}); the boxes are meaningless but the path is the real one")
    return synth.make_weights(seed)


def read_txt(path):
    """inverse of hostio.save_txt -> (seconds, rows [k, 9] float32)"""
    with open(path) as fh:
        lines = [l.strip() for l in fh if l.strip()]
    rows = np.array([[float(v) for v in l.split(",")] for l in lines[1:]], dtype=np.float32).reshape(-1, 9)
    return float(lines[0]), rows
