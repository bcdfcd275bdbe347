import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
net = pkg.pipeline3d.Dsvt3dBackbone(pkg.synth.make_weights_3d(), device=dev)
for s in range(4):
    p = pkg.synth.lidar_like(300000, s)
    buf = np.zeros((1, net.N, 4), np.float32); buf[0, :len(p)] = p
    tr = {}
    x, coords, Pn = net.forward(torch.from_numpy(buf).to(dev), torch.tensor([len(p)], dtype=torch.int32, device=dev), trace=tr)
    torch.cuda.synchronize()
    print("seed", s, "P0", int(tr[("in", 0)][2][0]), "W/S stage0", int(tr[("block", 0)][1]["W"][0]), int(tr[("block", 0)][1]["S"][0]), "P1", int(Pn[0]), "W/S stage1", int(tr[("block", 1)][1]["W"][0]), int(tr[("block", 1)][1]["S"][0]), flush=True)
print("eager ok", flush=True)
