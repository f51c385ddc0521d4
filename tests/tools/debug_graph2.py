import sys, faulthandler; faulthandler.enable()
import torch
a = torch.zeros(1 << 20, device="cuda"); b = torch.zeros(1 << 20, device="cuda"); c = torch.zeros(1 << 20, device="cuda")
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
ev = {k: torch.cuda.Event() for k in "abcdefg"}
def body(variant):
    cur = torch.cuda.current_stream()
    a.add_(1)
    ev["a"].record(cur)
    s1.wait_event(ev["a"]); s2.wait_event(ev["a"])
    with torch.cuda.stream(s1):
        b.add_(1); ev["b"].record(s1)
        b.add_(1)
    with torch.cuda.stream(s2):
        c.add_(1)
        if variant >= 1:
            s2.wait_event(ev["b"])          # cross edge between side streams
        c.add_(1); ev["c"].record(s2)
    if variant == 2:
        with torch.cuda.stream(s1):
            s1.wait_event(ev["c"]); b.add_(1)
    if variant == 3:      # same DAG, the tail of s1 hops to a fresh stream
        with torch.cuda.stream(s1):
            ev["e"].record(s1)
        with torch.cuda.stream(s3):
            s3.wait_event(ev["e"]); s3.wait_event(ev["c"]); b.add_(1); ev["d"].record(s3)
    else:
      with torch.cuda.stream(s1):
        ev["d"].record(s1)
    cur.wait_event(ev["c"]); cur.wait_event(ev["d"])
    a.add_(1)
for variant in (0, 1, 3, 2):
    body(variant); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(variant)
    g.replay(); torch.cuda.synchronize()
    print("variant", variant, "ok", a[0].item(), b[0].item(), c[0].item(), flush=True)
