import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import reagent_amd._lib as L
from reagent_amd.engine import make_stack
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
dims=[128,512,512,512,16]; B=65536
g = torch.Generator().manual_seed(0)
ws = [torch.nn.Parameter((torch.randn(o, i, generator=g) * (1.4 / i ** 0.5)).to(dev)) for i, o in zip(dims, dims[1:])]
bs = [torch.nn.Parameter(torch.zeros(o).to(dev)) for o in dims[1:]]
st = make_stack(ws, bs, [1,1,1,0], L.PREC_BF16); st.stage_weights(True)
x = torch.randn(B, dims[0], device=dev); out = torch.zeros(B, dims[-1], device=dev)
for probe, name in [(0,"full"),(1,"B from fixed address (no L2 streaming)"),(2,"A from fixed LDS address"),(3,"both fixed"),(4,"no loads in the main loop (MFMA only)")]:
    os.environ["RG_FUSED_PROBE"]=str(probe)
    print(f"{name:45s} {timeit(lambda: st.forward(x, out, save=False)):7.1f} us")
for dims2 in ([128,512,16],[128,512,512,16],[128,512,512,512,16]):
    ws2 = [torch.nn.Parameter((torch.randn(o, i, generator=g) * (1.4 / i ** 0.5)).to(dev)) for i, o in zip(dims2, dims2[1:])]
    bs2 = [torch.nn.Parameter(torch.zeros(o).to(dev)) for o in dims2[1:]]
    st2 = make_stack(ws2, bs2, [1]*(len(dims2)-2)+[0], L.PREC_BF16); st2.stage_weights(True)
    out2 = torch.zeros(B, 16, device=dev)
    for probe in (0, 4):
        os.environ["RG_FUSED_PROBE"]=str(probe)
        print(f"dims {dims2} probe {probe}: {timeit(lambda: st2.forward(x, out2, save=False)):7.1f} us")
