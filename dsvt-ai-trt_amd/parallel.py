"""Frame-batch data parallelism: one process per GPU, frames sharded across ranks, and ONE
collective per batch -- the gather of the per-frame results to rank 0 (RCCL over xGMI when the
backend is "nccl"; the same code runs on "gloo" for the CPU tests).

The reference is single-GPU, batch 1 (`cudaSetDevice(DEVICE)`, src/dsvt-ai-trt.cpp:1783;
include/params.h:333), frames are independent units of work (one enqueueV2 per frame,
src/dsvt-ai-trt.cpp:1884-1970), so the path shards by frame with no data-path exchange at all;
the only thing that has to meet on one rank is the tiny result: [frames, 500*9 + 1] float32
(18 kB per frame, latency-bound -- far below the ~153 GB/s of one xGMI link).
"""
import os

import torch
import torch.distributed as dist

ROW = 500 * 9 + 1        # boxes[500,9] flattened + the valid count (as float)


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend=None, single_rank_group=False, device_index=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  No-op for a
    single process unless single_rank_group: then a communicator of size 1 is created so that the
    result gather really goes through the collective library (RCCL on a GPU box) -- how the
    collective is exercised when only one GPU is visible.  device_index: the GPU of this
    rank when it is not LOCAL_RANK (bench.py --share-gpu: several ranks on one device, a dry run)."""
    rank, local_rank, world = env_world()
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank if device_index is None else device_index)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_frames(n_frames, rank, world):
    """Frame f belongs to rank f mod world.  Returns this rank's global frame ids."""
    return list(range(rank, n_frames, world))


def pack_result(boxes, count, out_row):
    """boxes [500,9] f32, count [1] i32 (device) -> one row of the result buffer, no host sync."""
    out_row[:ROW - 1].copy_(boxes.reshape(-1))
    out_row[ROW - 1:].copy_(count.reshape(-1).to(torch.float32))


def unpack_result(row):
    return row[:ROW - 1].reshape(500, 9), int(round(float(row[ROW - 1])))


class GatherBuffers:
    """The tensors one result gather needs, allocated ONCE (before any HIP graph is captured) and reused by every call: the padded send rows, on `dst` the
    per-rank receive buffers and the output in global frame order.  gather_results() without them allocates all three through the caching allocator on
    every call -- while captured graphs are alive that is the one ingredient of the steady-state loop (a gather after every batch, the loop shape of
    src/dsvt-ai-trt.cpp:1884-1970) that differs from the pattern known to run, so the product loop does not do it."""

    def __init__(self, n_frames, rank, world, device, dtype=torch.float32, dst=0, staged=False):
        self.n_frames, self.rank, self.world, self.dst = n_frames, rank, world, dst
        per = (n_frames + world - 1) // world
        gdev = torch.device("cpu") if staged else torch.device(device)
        self.pad = torch.zeros((per, ROW), dtype=dtype, device=gdev)
        self.bufs = [torch.empty_like(self.pad) for _ in range(world)] if rank == dst else None
        self.flat = torch.empty((world * per, ROW), dtype=dtype, device=gdev)          # (all-gather form: every rank receives every rank's rows)
        self.out = torch.empty((n_frames, ROW), dtype=dtype, device=device) if rank == dst else None
        self.ids = [torch.tensor(shard_frames(n_frames, r, world), dtype=torch.long, device=device) for r in range(world)] if rank == dst else None


def gather_results(local, n_frames, rank, world, dst=0, force_collective=False, buffers=None, collective="gather"):
    """local: [frames_of_this_rank, ROW].  Returns on `dst` a [n_frames, ROW] tensor in global frame
    order (None elsewhere).  One gather for the whole batch; ranks may own a different number of
    frames, so rows are padded to the maximum.  A single process skips the collective unless
    force_collective (needs init(single_rank_group=True)).  buffers: a GatherBuffers made for the same (n_frames, rank, world): no allocation in the call
    (the returned tensor is buffers.out, overwritten by the next call)."""
    if world == 1 and not (force_collective and dist.is_initialized()):
        return local
    per = (n_frames + world - 1) // world
    # gloo gathers host tensors only: device rows are staged through the host (the CPU tests, and bench.py --share-gpu, where RCCL
    # refuses two ranks on one device: "Duplicate GPU detected"); with nccl (= RCCL) the rows never leave the devices
    stage = local.is_cuda and dist.get_backend() == "gloo"
    if buffers is not None:
        assert (buffers.n_frames, buffers.rank, buffers.world, buffers.dst) == (n_frames, rank, world, dst)
        assert buffers.pad.is_cuda == (local.is_cuda and not stage), "GatherBuffers(staged=...) does not match the backend"
        buffers.pad[:local.shape[0]].copy_(local)
        if collective == "all_gather":
            dist.all_gather_into_tensor(buffers.flat, buffers.pad)
            recv = [buffers.flat[r * per:(r + 1) * per] for r in range(world)]
        else:
            dist.gather(buffers.pad, buffers.bufs, dst=dst)
            recv = buffers.bufs
        if rank != dst:
            return None
        for r in range(world):
            k = buffers.ids[r].shape[0]
            if k:
                buffers.out.index_copy_(0, buffers.ids[r], recv[r][:k].to(buffers.out.device, non_blocking=False))
        return buffers.out
    gdev = torch.device("cpu") if stage else local.device
    pad = torch.zeros((per, ROW), dtype=local.dtype, device=gdev)
    pad[:local.shape[0]].copy_(local)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = torch.empty((n_frames, ROW), dtype=local.dtype, device=local.device)
    for r in range(world):
        ids = shard_frames(n_frames, r, world)
        out[ids] = bufs[r][:len(ids)].to(local.device)
    return out


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def max_over_ranks_vec(values, device):
    """element-wise maximum over the ranks of a list of floats: ONE all-reduce (bench.py: the per-repeat times of a run)"""
    if not dist.is_initialized():
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def all_ranks_vec(values, device):
    """every rank's list of floats on every rank: ONE all-gather (bench.py: the per-repeat times of a run -- the maximum over the ranks is the job's
    time of a repeat, the per-rank values say which rank was slow).  Returns [world][len(values)]."""
    if not dist.is_initialized():
        return [list(values)]
    dev = "cpu" if dist.get_backend() == "gloo" else device
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(v) for v in o] for o in out]


def device_identity(device):
    """16 bytes that name the physical GPU of this rank: its UUID where the runtime reports one, else PCI bus / device / domain, else name + index"""
    import hashlib
    import socket
    p = torch.cuda.get_device_properties(device)
    ident = None
    for attr in ("uuid", "pci_bus_id"):
        v = getattr(p, attr, None)
        if v is not None and str(v) not in ("", "None"):
            ident = f"{attr}:{v}"
            break
    if ident is None:
        ident = f"name:{p.name}:index:{torch.device(device).index}"
    if not ident.startswith("uuid"):
        ident = socket.gethostname() + "/" + ident + f"/{getattr(p, 'pci_device_id', '')}/{getattr(p, 'pci_domain_id', '')}"
    return hashlib.md5(ident.encode()).digest(), ident


def gather_device_identities(device):
    """[(md5 hex, text)] of every rank's device, on every rank: one all-gather of 16 bytes, to be called BEFORE any HIP graph exists (bench.py:
    n_gpus must be the number of DISTINCT devices that took part, not the number of processes)"""
    digest, text = device_identity(device)
    if not dist.is_initialized():
        return [(digest.hex(), text)]
    dev = "cpu" if dist.get_backend() == "gloo" else device
    t = torch.tensor(list(digest), dtype=torch.uint8, device=dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [(bytes(o.cpu().tolist()).hex(), text if i == dist.get_rank() else "") for i, o in enumerate(out)]


def own_rows_match(gathered, local, n_frames, rank, world):
    """rank `dst`'s check of the gather: the rows of its own shard in the gathered tensor are its local rows, bit for bit"""
    ids = shard_frames(n_frames, rank, world)
    return bool(torch.equal(gathered[ids].view(torch.int32), local[:len(ids)].view(torch.int32)))
