import os, sys, torch
sys.path.insert(0, os.getcwd())
from transfer_em_amd.cgan import EM2EM
m = EM2EM(132, "sl", checkpoint_root="/tmp/sl_ck", precision=(sys.argv[1] if len(sys.argv) > 1 else "fp32"))
x = torch.randn(1, 132, 132, 132, 1, device="cuda")
m.train_step(x, x); torch.cuda.synchronize()
st = m._compiled(1)
for name, ws in zip(("G", "F", "DX", "DY"), st._keep[:4]):
    tot = 0
    for layer, t in ws.buf.items():
        mb = t.numel() * 4 / 1e6; tot += mb
        print(f"{name} {layer:6s} slabs {t.shape[0]:5d} x {t.shape[1]:7d} floats = {mb:7.2f} MB   per call: {[n for _, n, _ in ws.requests[layer]]}")
    print(name, "total MB", round(tot, 1))
