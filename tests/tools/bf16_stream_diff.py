"""Which layer's gradient differs between the multi-stream and the single-stream bf16 step (debugging aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from transfer_em_amd.cgan import EM2EM
n = int(os.environ.get("N", "132"))
rx, ry = torch.rand(1, n, n, n, 1, device="cuda"), torch.rand(1, n, n, n, 1, device="cuda")
grads = {}
for tag, streams in (("multi", True), ("single", False), ("multi2", True), ("multi3", True), ("multi4", True)):
    m = EM2EM(n, tag, seed=42, checkpoint_root="/tmp/tem_sd", precision="bf16", two_streams=streams)
    m.train_step(rx, ry)
    torch.cuda.synchronize()
    grads[tag] = {}
    for name, net in zip(("G", "F", "DX", "DY"), m._nets):
        for layer in net.params.shapes:
            grads[tag][name + "." + layer] = net.params.g(layer).clone()
    del m
    torch.cuda.empty_cache()
for k in grads["multi"]:
    a = grads["single"][k]
    d = [(a - grads[t][k]).abs().max().item() for t in ("multi", "multi2", "multi3", "multi4")]
    if any(d):
        print(k, tuple(a.shape), "single vs multi*: max|d|", ["%.2e" % v for v in d], "max|g| %.3e" % a.abs().max().item())
print("done")
