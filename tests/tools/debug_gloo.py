import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
g = torch.ones(621698, device="cuda")
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dist.all_reduce(g)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if rank == 0: print("gloo all_reduce of 2.5 MB cuda tensor: %.1f ms" % ((t1 - t0) * 1e3), flush=True)
c = g.cpu()
for i in range(3):
    t0 = time.perf_counter(); dist.all_reduce(c); t1 = time.perf_counter()
    if rank == 0: print("gloo all_reduce of 2.5 MB cpu tensor: %.1f ms" % ((t1 - t0) * 1e3), flush=True)
