"""Debug helper: per generator call, compare every gated input-gradient of the HIP backward with the oracle's."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from test_gpu_step import _inputs, _state, _load
from oracle import graph, ops
from transfer_em_amd.cgan import EM2EM, _CompiledStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
shape = (B, 74, 74, 74, 1)
rx, ry = _inputs(shape, 11), _inputs(shape, 12)
st = _state(graph, True, True)
model = EM2EM(74, "b2", checkpoint_root="/tmp/b2ck")
if len(sys.argv) > 2 and sys.argv[2] == "direct":
    model._steps[B] = _CompiledStep(model, B, direct=True)
_load(model, st)
model.train_step(torch.from_numpy(rx), torch.from_numpy(ry))
torch.cuda.synchronize()
cs = model._steps[B]
rec, cur = [], None
orig_gate, orig_gb = ops.leaky_relu_grad_from_out, graph.generator_backward
def gate(g, out):
    r = orig_gate(g, out)
    if cur is not None: cur.append(r)
    return r
def gb(P, sv, dy, need_dx=False):
    global cur
    cur = []
    r = orig_gb(P, sv, dy, need_dx)
    rec.append(cur); cur = None
    return r
ops.leaky_relu_grad_from_out = gate; graph.generator_backward = gb
graph.train_step_grads(st["g"], st["f"], st["dx"], st["dy"], rx, ry, True, 2.0, 42, 0)
names = ("f1", "u1b", "u1a", "mid", "u2b", "u2a", "d2b", "d2a", "d1b", "d1a", "c0")
for call, gates in zip(("g3", "f3", "f2", "g2", "g1", "f1"), rec):
    bw = cs.bwd[call]; R = bw.fwd.regions
    for nm, ref in zip(names, gates):
        lo, hi = R[nm]
        ref = ref[:, lo:hi, lo:hi, lo:hi, :]
        got = bw.grads[nm].cpu().numpy()
        if nm in ("u1b", "u2b"):
            keep = got != 0
            ref = np.where(keep, ref * 2, 0)
        d = np.abs(got - ref)
        i = np.unravel_index(d.argmax(), d.shape)
        print(call, nm, got.shape, "max-rel %.2e l2-rel %.2e worst at %s got %.4e ref %.4e" % (d.max() / np.abs(ref).max(), np.linalg.norm(d) / np.linalg.norm(ref), i, got[i], ref[i]))
