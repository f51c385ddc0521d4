"""Debug helper: isolate what changes generator kernel-gradient parity on the second step."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import rel_err
from oracle import graph
from transfer_em_amd.cgan import EM2EM
from transfer_em_amd import hip_ops as H
from test_gpu_step import _inputs, _load, _state
is3d = False; batch = 2; n = 74
shape = (batch, 1, n, n, 1)
rx, ry = _inputs(shape, 1234), _inputs(shape, 5678)
st = _state(graph, is3d, True)
model = EM2EM(n, "dbg", is3d=is3d, seed=42, checkpoint_root="/tmp/dbg_ck2")
_load(model, st)
cs = model._compiled(batch)
cs.real_x.copy_(torch.from_numpy(rx)); cs.real_y.copy_(torch.from_numpy(ry))

def grads_model():
    cs.losses.zero_(); H.run(cs.compute); torch.cuda.synchronize()
    return {k: net.params.to_dict("grad") for k, net in zip(("g", "f", "dx", "dy"), model._nets)}

def cmp(tag, gm, go):
    for net in ("g", "f", "dy"):
        worst = max((rel_err(gm[net][k], go[net][k]), k) for k in go[net])
        print(tag, net, "worst rel err %.2e at %s" % worst)

def oracle(step):
    return graph.train_step_grads(st["g"], st["f"], st["dx"], st["dy"], rx, ry, is3d, 2.0, 42, step)[1]

g0a = grads_model(); g0b = grads_model()
print("run-to-run identical:", all(np.array_equal(g0a[n_][k], g0b[n_][k]) for n_ in g0a for k in g0a[n_]))
cmp("step_dev=0 vs oracle(step 0)", g0b, oracle(0))
model.step_dev.fill_(1)
g1 = grads_model()
cmp("step_dev=1 vs oracle(step 1)", g1, oracle(1))
cmp("step_dev=1 vs oracle(step 0)", g1, oracle(0))
model.step_dev.fill_(7)
cmp("step_dev=7 vs oracle(step 7)", grads_model(), oracle(7))

print("---- full flow")
model.step_dev.fill_(0)
_load(model, st)
model.train_step(torch.from_numpy(rx), torch.from_numpy(ry))
graph.train_step(st, rx, ry, is3d, 2.0, 42)
for key, net in zip(("g", "f", "dx", "dy"), model._nets):
    th = net.params.to_dict("theta")
    print(key, "theta drift model-vs-oracle after step 0: %.3e" % max(np.abs(th[k] - st[key][k]).max() for k in th))
_load(model, st)
for key, net in zip(("g", "f", "dx", "dy"), model._nets):
    th = net.params.to_dict("theta")
    print(key, "theta equal after _load:", all(np.array_equal(th[k], st[key][k]) for k in th))
print("step_dev", int(model.step_dev.item()), "oracle step", st["step"])
g1 = grads_model()
cmp("after Adam, step 1", g1, oracle(1))
