import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from util import gate_flips, prior_layers, scaled_params
from test_prior import _chain
from test_gpu_step import _inputs as std_inputs, _load
from oracle import graph
from transfer_em_amd.cgan import EM2EM
from transfer_em_amd.models.prior import PriorNet
use_prior = len(sys.argv) < 2 or sys.argv[1] != "noprior"
layers, cut = prior_layers(False)
chain = _chain(layers, cut) if use_prior else None
pc = 32 if use_prior else 0
model = EM2EM(74, "prior", is3d=False, disc_prior=PriorNet(layers, cut) if use_prior else None, checkpoint_root="/tmp/pck")
st = graph.new_state(False, prior_channels=pc)
gs = graph.generator_param_shapes(False)
st["g"], st["f"] = scaled_params(gs, 10), scaled_params(gs, 11)
st["dx"] = scaled_params(graph.discriminator_param_shapes(False), 12)
st["dy"] = scaled_params(graph.discriminator_param_shapes(False, prior_channels=pc), 13)
_load(model, st)
rx, ry = std_inputs((2, 1, 74, 74, 1), 1234), std_inputs((2, 1, 74, 74, 1), 5678)
got = model.train_step(torch.from_numpy(rx), torch.from_numpy(ry)).cpu().numpy()
gh = {k: net.params.to_dict("grad") for k, net in zip(("g", "f", "dx", "dy"), model._nets)}
losses, grads, aux = graph.train_step(st, rx, ry, False, 2.0, 42, prior_y=chain)
print("flips", gate_flips(model._steps[2], aux["saved"], False))
for net in ("g", "f", "dx", "dy"):
    for name, ref in grads[net].items():
        d = gh[net][name] - ref
        print(net, name, "max-rel %.2e  l2-rel %.2e  refmax %.2e" % (np.abs(d).max() / np.abs(ref).max(), np.linalg.norm(d) / np.linalg.norm(ref), np.abs(ref).max()))
