import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from transfer_em_amd import hip_ops as H
from oracle import ops
H.require_gpu()
from test_gpu_ops import _bww, dev, rnd
rng=np.random.default_rng(0)
for n in (14, 45):
    x=rnd(rng,1,n,n,n,1); o=n-2; g=rnd(rng,1,o,o,o,8)
    ref=ops.conv_bwd_weight(x,g,(3,3,3))
    got,kern=_bww(H,dev(x),dev(g),ref.shape,3)
    r=got.ravel()/ref.ravel()
    print(n, kern, 'ratio stats', np.median(r), r.min(), r.max(), 'first rows', r[:16].round(3))
