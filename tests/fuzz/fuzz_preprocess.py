import os, sys, random
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
import gpu_ops  # FUZZ_ON_GPU=1: the real library on cuda instead of the interpreter
DEV = gpu_ops.device()
from oracle import restated as R
from reagent_amd.core.parameters import NormalizationParameters as NP
from reagent_amd.preprocessing import Preprocessor

# Preprocessor.forward (rg_normalize_dense) on random feature tables — every feature type in random mixtures and id orders,
# random parameters, inputs that sit ON the special points (0, quantile knots, enum values, range ends, negative and huge
# values, missing features) — against oracle/restated.py::preprocess, the restatement pinned to the reference by
# tests/golden/preprocessor_all_types.npz (preprocessing/preprocessor.py:115-170, :197-525)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
random.seed(seed)
TYPES = list(R.FEATURE_TYPES)
bad = 0
for case in range(cases):
    g = torch.Generator().manual_seed(seed * 1000 + case)
    nf = random.choice([1, 2, 5, 17, 64, 130])
    ids = random.sample(range(1, 5000), nf)
    norm, special = {}, {}
    for f in ids:
        t = random.choice(TYPES)
        kw, pts = dict(feature_type=t), [0.0, -1.0, 1.0, 1e9, -1e9]
        if t in ("CONTINUOUS", "BOXCOX"):
            kw.update(mean=random.uniform(-3, 3), stddev=random.uniform(0.1, 5))
        if t == "BOXCOX":
            kw.update(boxcox_lambda=random.choice([-1.5, -0.5, 0.3, 1.0, 2.0]), boxcox_shift=random.uniform(0, 3))
            pts += [-kw["boxcox_shift"]]
        if t == "ENUM":
            kw.update(possible_values=sorted(random.sample(range(-5, 40), random.choice([1, 2, 5, 9]))))
            pts += [float(v) for v in kw["possible_values"]]
        if t == "QUANTILE":
            qs = sorted({round(random.uniform(-10, 10), 2) for _ in range(random.choice([2, 3, 7, 20]))})
            if len(qs) < 2:
                qs = [-1.0, 1.0]
            kw.update(quantiles=qs)
            pts += qs + [qs[0] - 1, qs[-1] + 1]
        if t == "CONTINUOUS_ACTION":
            lo = random.uniform(-5, 5)
            kw.update(min_value=lo, max_value=lo + random.uniform(0.1, 10))
            pts += [kw["min_value"], kw["max_value"]]
        if t == "PROBABILITY":
            pts += [1e-7, 0.5, 1 - 1e-7]
        norm[f], special[f] = NP(**kw), pts
    order = R.sort_features({f: SimpleNamespace(**vars(p)) if not hasattr(p, "feature_type") else p for f, p in norm.items()})
    B = random.choice([1, 3, 64, 257])
    x = torch.randn(B, nf, generator=g) * random.choice([0.5, 3.0, 30.0])
    for j, f in enumerate(order):  # plant the special points
        for r in range(B):
            if random.random() < 0.3:
                x[r, j] = random.choice(special[f])
    presence = (torch.rand(B, nf, generator=g) < 0.85).to(torch.uint8)
    pre = Preprocessor(norm, device=DEV)
    assert pre.sorted_features == order
    got = pre(x.to(DEV), presence.to(DEV)).cpu()
    want = R.preprocess(norm, x, presence)
    both_nan = torch.isnan(got) & torch.isnan(want)
    diff = torch.where(both_nan, torch.zeros(()), (got - want).abs())
    tol = 2e-6 * torch.clamp(want.abs(), min=1.0)
    ok = got.shape == want.shape and bool((diff <= torch.where(torch.isnan(tol), torch.zeros(()), tol)).all()) \
        and bool((torch.isnan(got) == torch.isnan(want)).all())
    if not ok:
        j = int(torch.nonzero(~(diff <= tol) | (torch.isnan(got) != torch.isnan(want)))[0, 1])
        cols, k = [], 0
        for f in order:  # output column -> feature (ENUM expands)
            w = len(norm[f].possible_values) if norm[f].feature_type == "ENUM" else 1
            cols += [f] * w
        print("   first bad column:", j, norm[cols[j]], "got", got[:, j][~(diff[:, j] <= tol[:, j])][:3], "want", want[:, j][~(diff[:, j] <= tol[:, j])][:3])
    print("OK " if ok else "BAD", dict(features=nf, B=B, types=sorted({p.feature_type for p in norm.values()})), "max diff %.2e" % diff.max().item())
    bad += 0 if ok else 1
print("bad cases:", bad)
sys.exit(1 if bad else 0)
