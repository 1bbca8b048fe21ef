"""TEST INFRASTRUCTURE — on the MI355X:  python tests/fuzz/fuzz_compact_head_gpu.py [seed]

rg_qr_compact_head (the quantile-Huber loss on grouped rows: a 256-wide bitonic network on v_med3 / DPP / ds_swizzle, fp64 prefix
sums by DPP row moves, twelve lock-step bisections per lane) against the N x N pair loop of qrdqn_trainer.py:143-160 in float64, for
EVERY kind of quantile count — 1, 2, 3, around the lane count (63 / 64 / 65), around the half and full network (127 .. 129, 255, 256),
BASELINE's 200 — with ties and differences of exactly 0 and +-1 planted.  The cross-lane instructions are the part an interpreter
models and only the hardware decides (tests/test_qrdqn_trainer.py::test_compact_head_against_the_pair_loop is the N = 37 instance)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from reagent_amd import ops
from reagent_amd.qr_engine import GroupedSpace

assert torch.cuda.is_available()
dev = torch.device("cuda", 0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
bad = 0
for case, N in enumerate([1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 191, 192, 193, 199, 200, 201, 254, 255, 256]):
    g = torch.Generator().manual_seed(seed * 1000 + case)
    B, G = int(torch.randint(1, 400, (1,), generator=g)), int(torch.randint(1, 7, (1,), generator=g))
    ld = (N + 3) // 4 * 4 + 4 * int(torch.randint(0, 3, (1,), generator=g))
    key = torch.randint(0, G, (B,), generator=g).to(torch.int32)
    sp = GroupedSpace(B, G, dev).build(key.to(dev))
    R = sp.rows
    z = torch.randn(R, ld, generator=g) * 1.5
    zt = torch.randn(B, ld, generator=g) * 1.5
    k5, k7 = max(1, N // 7), max(1, N // 5)
    z[:, :k5] = torch.round(z[:, :k5])
    zt[:, :k7] = torch.round(zt[:, :k7])
    if N > 8:
        zt[:, N - 3:N] = zt[:, :3]  # repeated targets
    reward = torch.where(torch.arange(B) % 2 == 0, torch.zeros(B), torch.randn(B, generator=g))
    nt = torch.where(torch.arange(B) % 5 == 0, torch.zeros(B), torch.ones(B))
    boosts = torch.randn(G, generator=g) * 0.25
    gamma = 1.0 if case % 2 == 0 else 0.97
    tau = (0.5 + torch.arange(N).float()) / N
    dz = torch.full((R, ld), 7.0)
    D = lambda t: t.to(dev)  # noqa: E731
    dzd, lpd, tld = D(dz), torch.zeros(R, device=dev), torch.zeros(sp.n_tiles, device=dev)
    ops.qr_compact_head(D(z), D(zt), sp.rowmap, D(key), D(reward), D(boosts), D(nt), gamma, None, D(tau), B, N, dzd, lpd, tld)
    torch.cuda.synchronize()
    rm = sp.rowmap.cpu()
    ref_dz = torch.zeros(R, ld, dtype=torch.float64)
    ref_l = torch.zeros(R, dtype=torch.float64)
    inv = 1.0 / (N * B * N)
    for r in range(R):
        b = int(rm[r])
        if b < 0:
            continue
        T = (reward[b] + boosts[int(key[b])] + (torch.tensor(gamma).float() * nt[b]) * zt[b, :N]).double()  # fp32 targets, as the kernel forms them
        C = z[r, :N].double()
        td = T[:, None] - C[None, :]
        ad = td.abs()
        hub = torch.where(ad < 1, 0.5 * td * td, ad - 0.5)
        dh = torch.where(ad < 1, td, torch.sign(td))
        w = (tau.double()[None, :] - (td < 0).double()).abs()
        ref_l[r] = (hub * w).sum() * inv
        ref_dz[r, :N] = -(dh * w).sum(0) * inv
    e_dz = (dzd.cpu().double() - ref_dz).abs().max().item() / (ref_dz.abs().max().item() + 1e-30)
    e_l = (lpd.cpu().double() - ref_l).abs().max().item() / (ref_l.abs().max().item() + 1e-30)
    e_t = abs(tld.cpu().double().sum().item() - ref_l.sum().item()) / (abs(ref_l.sum().item()) + 1e-30)
    pad_ok = bool((dzd.cpu()[:, N:] == 0).all())  # the padding columns of every row are written as zeros
    ok = e_dz <= 2e-6 and e_l <= 4e-6 and e_t <= 2e-6 and pad_ok
    bad += 0 if ok else 1
    print("OK " if ok else "BAD", f"N {N:3d} B {B:3d} G {G} ld {ld:3d} rows {R:4d}: dz {e_dz:.1e} loss {e_l:.1e} tile sums {e_t:.1e} padding zero {pad_ok}", flush=True)
print("bad cases:", bad)
sys.exit(1 if bad else 0)
