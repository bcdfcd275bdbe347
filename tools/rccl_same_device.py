"""probe: can two ranks of one RCCL communicator share ONE GPU?  torchrun --nproc-per-node 2 tools/rccl_same_device.py"""
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl")
    r = dist.get_rank()
    t = torch.full((4,), float(r + 1), device="cuda:0")
    bufs = [torch.empty_like(t) for _ in range(2)] if r == 0 else None
    dist.gather(t, bufs, dst=0)
    torch.cuda.synchronize()
    if r == 0:
        print("rccl gather on one shared device OK:", [b.tolist() for b in bufs])
    dist.destroy_process_group()
except Exception as e:
    print("rccl on a shared device FAILED:", type(e).__name__, str(e)[:300])
    sys.exit(0)
