import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from util import scaled_params, rel_err
from test_gpu_step import _inputs
from oracle import graph
from transfer_em_amd import hip_ops as H
from transfer_em_amd.models.generator import unet_generator, GenForward
n = int(sys.argv[1]) if len(sys.argv) > 1 else 260
net, _ = unet_generator(n, seed=5)
P = scaled_params(graph.generator_param_shapes(True), 3)
net.params.load_dict(P)
x = torch.from_numpy(_inputs((1, n, n, n, 1), 7)).cuda()
a = GenForward(net, x); H.run(a.launches)
b = GenForward(net, x, direct=True); H.run(b.launches)
torch.cuda.synchronize()
for k in a.act:
    ta, tb = a.act[k], b.act[k]
    d = (ta - tb).abs()
    print(k, tuple(ta.shape), "tiled-vs-direct max rel %.2e" % (d.max().item() / tb.abs().max().item()), a.launches[list(a.act).index(k)].meta["kernel"] if list(a.act).index(k) < len(a.launches) else "")
