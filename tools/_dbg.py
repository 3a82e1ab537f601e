import sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from fedmlp_amd.engine import Engine
from tests.test_effnet_bf16_gpu import _load, _pw_convs, _engine_weight, C_
e = Engine("Efficient_b0", C_, 96, 96, 8, precision="bf16")
e.stochastic = False
_load(e)
L, S = 144, 48
ci, info = next((c, i) for c, i in _pw_convs(e) if i["cin_p"] == L and i["cout_p"] == S and i["cin"] > i["cout"])
h, w = info["hout"], info["wout"]; HWo = h*w
imgs, groups = 2, 1
npix = imgs*HWo
g = torch.Generator().manual_seed(1)
dyp = (torch.randn((npix, S), generator=g)*0.5).to(torch.bfloat16); dyp[:, info["cout"]:] = 0
yd = torch.zeros((npix, L)).to(torch.bfloat16)
bn = torch.zeros((7, groups, L)); bn[0] = 1; bn[3] = 1; bn[4] = 1   # v = y = 0: sg = 0.5, ca = 1
gate = torch.ones((imgs, L)); ds = torch.zeros((imgs, L))
W = torch.from_numpy(_engine_weight(e, ci, info)).to(torch.bfloat16).float()
dev = e.device
d = (dyp.float() @ W).to(torch.bfloat16).float()
dy = torch.empty((npix, L), dtype=torch.bfloat16, device=dev)
e.debug_proj_bwd(ci, 1, dyp.to(dev), yd.to(dev), bn.to(dev), gate.to(dev), ds.to(dev), imgs, groups, dy)
torch.cuda.synchronize()
got = dy.float().cpu()*2          # dy = d*0.5
err = (got - d).abs()
print("max err", err.max().item(), "max d", d.abs().max().item())
bad = (err > 1e-2*d.abs().max()).nonzero()
print("bad count", len(bad), "of", err.numel())
rows = torch.unique(bad[:,0]); cols = torch.unique(bad[:,1])
print("bad rows (pixel idx)", rows[:40].tolist(), "... n", len(rows))
print("bad cols", cols[:60].tolist(), "n", len(cols))
# which s contribute: test with single-s dyp
for s0 in (0, 8, 31, 32, 36, 39):
    dyp1 = torch.zeros((npix, S)).to(torch.bfloat16); dyp1[:, s0] = 1
    e.debug_proj_bwd(ci, 1, dyp1.to(dev), yd.to(dev), bn.to(dev), gate.to(dev), ds.to(dev), imgs, groups, dy)
    torch.cuda.synchronize()
    got = dy.float().cpu()*2
    want = W[s0][None,:].expand(npix, L).to(torch.bfloat16).float()
    print("s", s0, "err", (got-want).abs().max().item(), "max", want.abs().max().item())
