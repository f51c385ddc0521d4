import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dist.init_process_group(os.environ.get("TEM_DIST_BACKEND", "gloo"))
rank = dist.get_rank()
torch.cuda.set_device(0)
from transfer_em_amd.cgan import EM2EM
from transfer_em_amd import hip_ops as H, distributed as D
m = EM2EM(132, "dp", checkpoint_root=f"/tmp/dp{rank}")
x = torch.randn(1, 132, 132, 132, 1, device="cuda"); y = torch.randn_like(x)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = m._compiled(1); st.real_x.copy_(x); st.real_y.copy_(y)
    m._compute(st); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    D.allreduce_sum_(m.grad_all, m.pg); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    H.run(st.update, H.current_stream()); torch.cuda.synchronize(); t5 = time.perf_counter()
    if rank == 0:
        print("enqueue %.1f ms, compute done %.1f, allreduce call %.1f, allreduce done %.1f, update %.1f" % tuple(1e3 * v for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)), flush=True)
