"""Debug helper (not a test): per-layer gradient errors of one train step vs the oracle."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import rel_err, scaled_params
from oracle import graph
from transfer_em_amd.cgan import EM2EM
from test_gpu_step import _inputs, _load, _state
is3d = len(sys.argv) > 1 and sys.argv[1] == "3d"
batch = 1 if is3d else 2
n = 74
shape = (batch, n if is3d else 1, n, n, 1)
rx, ry = _inputs(shape, 1234), _inputs(shape, 5678)
st = _state(graph, is3d, True)
model = EM2EM(n, "dbg", is3d=is3d, seed=42, checkpoint_root="/tmp/dbg_ck")
_load(model, st)
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for _ in range(nsteps - 1):
    model.train_step(torch.from_numpy(rx), torch.from_numpy(ry))
    graph.train_step(st, rx, ry, is3d, 2.0, 42)
    _load(model, st)
got = model.train_step(torch.from_numpy(rx), torch.from_numpy(ry)).cpu().numpy()
grads_hip = {k: net.params.to_dict("grad") for k, net in zip(("g", "f", "dx", "dy"), model._nets)}
print("step counters", int(model.step_dev.item()), st["step"])
losses, grads, aux = graph.train_step(st, rx, ry, is3d, 2.0, 42)
print("losses", got, losses)
cs = model._steps[batch]
for key, plan in (("fake_y", "g1"), ("cyc_x", "f2"), ("fake_x", "f1"), ("cyc_y", "g2"), ("same_x", "f3"), ("same_y", "g3")):
    b = model.buffer
    ref = aux[key]
    if key.startswith("cyc"):
        ref = ref[:, b:-b, b:-b, b:-b, :] if is3d else ref[:, :, b:-b, b:-b, :]
    print(key, rel_err(cs.fwd[plan].y.cpu().numpy(), ref))
print("d_fake_y", rel_err(cs.bwd["f2"].dx.cpu().numpy(), aux["d_fake_y"]))
for net in ("g", "f", "dx", "dy"):
    for name, ref in grads[net].items():
        print(net, name, "rel", rel_err(grads_hip[net][name], ref), "max", np.abs(ref).max())
