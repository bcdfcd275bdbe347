"""Host I/O around the hot path: the reference's .bin loader, an upload path that moves only the
points that exist, and its result writer.

Reference behaviour mirrored here:
  * loadData            include/helper.h:28-72     read the whole file; the caller zero-pads to MAX_POINTS_NUM and copies
                                                   the full cap to the device every frame (src/dsvt-ai-trt.cpp:1925)
  * save_result/save_txt include/helper.h:441-481  "seconds" line, then one line per box
                                                   x,  y,  z,  l,  w,  h,  rt,  id,  score   (fixed, 6 decimals; id is an int)
The plugins bound every loop by the device-side point count, so the zero padding never has to travel: FrameUploader keeps
pinned staging buffers and copies n x 16 bytes + the count asynchronously on the frame's stream."""
import numpy as np
import torch


def load_bin(path, max_points=None):
    """-> (points [n, 4] float32, n).  Like loadData + the reference's size check (helper.h:47-53): a frame with more than
    max_points points is an error (the reference prints and exit(-1)s)."""
    raw = np.fromfile(path, dtype=np.float32)
    if raw.size % 4:
        raise ValueError(f"{path}: size is not a multiple of 4 floats")
    pts = raw.reshape(-1, 4)
    if max_points is not None and pts.shape[0] > max_points:
        raise ValueError(f"{path}: {pts.shape[0]} points exceed the cap {max_points}")
    return pts, pts.shape[0]


class FrameUploader:
    """depth pinned host buffers + device buffers [1, max_points, 4] / [1]; upload() is asynchronous on the current stream."""

    def __init__(self, max_points, device="cuda:0", depth=2):
        self.max_points, self.depth, self.i = max_points, depth, 0
        self.host = [torch.zeros((max_points, 4), dtype=torch.float32).pin_memory() for _ in range(depth)]
        self.hcnt = [torch.zeros((1,), dtype=torch.int32).pin_memory() for _ in range(depth)]
        self.dev = [torch.zeros((1, max_points, 4), dtype=torch.float32, device=device) for _ in range(depth)]
        self.dcnt = [torch.zeros((1,), dtype=torch.int32, device=device) for _ in range(depth)]
        self.done = [None] * depth

    def upload(self, points, n=None):
        """points: [n, 4] float32 numpy array or CPU tensor.  Returns (device points, device count) of the slot used."""
        k = self.i % self.depth
        self.i += 1
        n = points.shape[0] if n is None else int(n)
        if n > self.max_points:
            raise ValueError(f"{n} points exceed the cap {self.max_points}")
        if self.done[k] is not None:
            self.done[k].synchronize()                  # the previous copy out of this staging buffer has finished
        src = torch.from_numpy(points) if isinstance(points, np.ndarray) else points
        self.host[k][:n].copy_(src[:n])
        self.hcnt[k][0] = n
        self.dev[k][0, :n].copy_(self.host[k][:n], non_blocking=True)
        self.dcnt[k].copy_(self.hcnt[k], non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
        self.done[k] = ev
        return self.dev[k], self.dcnt[k]


def format_results(rows, seconds):
    """rows [k, 9] float: x, y, z, l, w, h, rt, id, score (RotatedNmsPlugin / save_result order) -> the text save_txt writes"""
    out = [f"{float(seconds):.6f}"]
    for r in np.asarray(rows, dtype=np.float32).reshape(-1, 9):
        f = [f"{float(v):.6f}" for v in r]
        f[7] = str(int(r[7]))
        out.append(",  ".join(f))
    return "\n".join(out) + "\n"


def save_txt(path, rows, seconds):
    with open(path, "w") as fh:
        fh.write(format_results(rows, seconds))
