"""Does the fragment round trip (saving forward -> backward -> weight gradient) get cheaper when a step's rows are
processed in chunks whose fragments fit the 256 MB Infinity Cache?  Times the three launches at B = 65536 / 32768 /
16384 rows (C2 network) and prints us per launch and per 65536 rows.
run: python profiles/microbench/mall_chunks.py [bf16|bf16x3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import reagent_amd._lib as L
from reagent_amd import ops, synthetic
from reagent_amd.engine import FusedMLP

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda")
dims = [128, 512, 512, 512, 16]
acts = [L.ACT["relu"]] * 3 + [L.ACT["linear"]]
w = synthetic.fc_init(dims, ["relu"] * 3 + ["linear"], seed=1)
ws = [p.to(dev) for p in w[0::2]]
bs = [p.to(dev) for p in w[1::2]]
for B in (65536, 32768, 16384):
    st = FusedMLP(ws, bs, acts, x3=(mode == "bf16x3"))
    st.stage_weights(need_transposed=True)
    x = torch.randn(B, dims[0], device=dev).to(torch.bfloat16 if mode == "bf16" else torch.float32)
    out = torch.empty(B, dims[-1], device=dev)
    dq = torch.randn(B, dims[-1], device=dev) * 1e-3
    dw = [torch.empty_like(p) for p in ws]
    db = [torch.empty_like(p) for p in bs]
    chunks = 65536 // B
    def seq():
        for _ in range(chunks):
            st.forward(x, out, save=True)
            st.backward(dq, None, dw, db)
    for _ in range(3):
        seq()
    torch.cuda.synchronize()
    n = 10
    with ops.profile() as prof:
        for _ in range(n):
            seq()
    rows = prof.summary()
    tot = 0.0
    line = []
    for r in rows:
        us = r["ms"] * 1e3 / r["calls"]
        tot += r["ms"] * 1e3 / n
        line.append(f"{r['name'].replace('rg_mlp_', '')} {us:7.1f} us/launch")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        seq()
    e1.record()
    torch.cuda.synchronize()
    print(f"{mode} B={B:6d} x{chunks}: " + " | ".join(line) + f" | sum per 65536 rows {tot:7.1f} us | back-to-back {e0.elapsed_time(e1) * 1e3 / n:7.1f} us")
