"""Progress of the step's stream lists WITHOUT an event pair around every launch (tests/tools/timeline.py's event pairs stretch
a 4 ms bf16 step to 5.3 ms): a timing event after every `every`-th launch of each list, times relative to the step's start.
   python tests/tools/milestones.py [fp32|bf16] [every=8]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from transfer_em_amd.cgan import EM2EM
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
every = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = EM2EM(132, "ms", checkpoint_root="/tmp/ms_ck", precision=prec)
x = torch.randn(1, 132, 132, 132, 1, device="cuda"); y = torch.randn(1, 132, 132, 132, 1, device="cuda")
for _ in range(5):
    m.train_step(x, y)
torch.cuda.synchronize()
st = m._compiled(1)
lists, marks = [], []
for i, lst in enumerate(st.lists_fused):
    out, k, last = [], 0, None
    for it in lst:
        out.append(it)
        if not isinstance(it, tuple):
            k += 1; last = it.name
            if k % every == 0:
                nm = f"ms{i}_{k}"
                st.events[nm] = torch.cuda.Event(enable_timing=True)
                out.append(("record", nm)); marks.append((i, k, nm, last))
    nm = f"ms{i}_end"
    st.events[nm] = torch.cuda.Event(enable_timing=True)
    # (the main list ends with its waits for the other lists: mark its last launch, not the joins)
    idx = max(j for j, it in enumerate(out) if not isinstance(it, tuple)) + 1
    out.insert(idx, ("record", nm)); marks.append((i, k, nm, "END " + str(last)))
    lists.append(out)
NT = 5
acc = {}
tot = 0.0
for it in range(NT):
    for _ in range(3):
        m.train_step(x, y)
    st.losses.zero_()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    m._run_streams(st, lists=lists)
    t1.record(); torch.cuda.synchronize()
    tot += t0.elapsed_time(t1) / NT
    for i, k, nm, last in marks:
        acc[nm] = acc.get(nm, 0.0) + t0.elapsed_time(st.events[nm]) / NT
print(f"step {tot:.3f} ms ({prec}); list: launches done -> ms since step start (last launch)")
for i in range(len(lists)):
    print(f"list {i}: " + "  ".join(f"{k}:{acc[nm]:.2f}" for ii, k, nm, last in marks if ii == i and not last.startswith("END")))
    for ii, k, nm, last in marks:
        if ii == i and last.startswith("END"):
            print(f"   ends {acc[nm]:.3f} ms after {k} launches ({last})")
