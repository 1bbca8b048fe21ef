"""rg_table_dqn_batch at the C2 shape on the GPU box: python profiles/microbench/table_batch.py
(table of N rows x 128 CONTINUOUS features, 16 actions, batch 65536 random rows; bf16 and fp32 rows)."""
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
from reagent_amd.data import OfflineTable  # noqa: E402
from reagent_amd.preprocessing import DiscreteDqnBatchPreprocessor, Preprocessor  # noqa: E402

N, F, A, B = 1 << 20, 128, 16, 65536
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
norm = {i: SimpleNamespace(feature_type="CONTINUOUS", mean=0.1 * (i % 7), stddev=1.0 + 0.01 * i, boxcox_lambda=None,
                           boxcox_shift=None, possible_values=None, quantiles=None, min_value=None, max_value=None)
        for i in range(F)}
pre = Preprocessor(norm, device=dev)
cols = dict(
    state_features=torch.randn(N, F, device=dev, generator=g), next_state_features=torch.randn(N, F, device=dev, generator=g),
    state_features_presence=torch.rand(N, F, device=dev, generator=g) > 0.05,
    next_state_features_presence=torch.rand(N, F, device=dev, generator=g) > 0.05,
    action=torch.randint(A, (N,), device=dev, generator=g), next_action=torch.randint(A + 1, (N,), device=dev, generator=g),
    reward=torch.randn(N, device=dev, generator=g), action_probability=torch.rand(N, device=dev, generator=g),
    time_diff=torch.ones(N, dtype=torch.int64, device=dev), step=torch.ones(N, dtype=torch.int64, device=dev),
    mdp_id=torch.arange(N, device=dev), sequence_number=torch.arange(N, device=dev),
    possible_actions_mask=torch.ones(N, A, dtype=torch.uint8, device=dev),
    possible_next_actions_mask=(torch.rand(N, A, device=dev, generator=g) > 0.2).to(torch.uint8),
)
table = OfflineTable(cols, A, device=dev)
print(f"table: {len(table)} rows, {table.nbytes / 1e9:.2f} GB in HBM")
for dt in (torch.bfloat16, torch.float32):
    bp = DiscreteDqnBatchPreprocessor(A, pre, state_dtype=dt)
    idx = torch.randint(N, (B,), device=dev, generator=g)
    for _ in range(3):
        bp.from_table(table, idx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        out = bp.from_table(table, idx)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    # the launch alone, outputs preallocated
    from reagent_amd import ops
    f32, i64 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int64, device=dev)
    bufs = dict(state=torch.empty(B, F, dtype=dt, device=dev), next_state=torch.empty(B, F, dtype=dt, device=dev),
                action=torch.empty(B, A, **f32), next_action=torch.empty(B, A, **f32), reward=torch.empty(B, 1, **f32),
                time_diff=torch.empty(B, 1, **f32), step=torch.empty(B, 1, **f32), not_terminal=torch.empty(B, 1, **f32),
                possible_actions_mask=torch.empty(B, A, **f32), possible_next_actions_mask=torch.empty(B, A, **f32),
                action_probability=torch.empty(B, 1, **f32), mdp_id=torch.empty(B, 1, **i64),
                sequence_number=torch.empty(B, 1, **i64))
    for _ in range(3):
        ops.table_dqn_batch(table, idx, pre._col_table, F, pre._quantiles, bufs)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.table_dqn_batch(table, idx, pre._col_table, F, pre._quantiles, bufs)
    e1.record()
    torch.cuda.synchronize()
    us_k = e0.elapsed_time(e1) * 1e3 / reps
    es = 2 if dt == torch.bfloat16 else 4
    bytes_per_row = 2 * F * 5 + 2 * F * es + 2 * A + 16 * A + 8 + 60
    print(f"{dt}: launch {us_k:.1f} us, from_table (13 output allocations) {us:.1f} us per batch of {B}; "
          f"{bytes_per_row} B/row algorithmic -> {B * bytes_per_row / us_k / 1e6:.2f} TB/s")
    ref = pre(table.columns["state_features"][idx], table.columns["state_features_presence"][idx])
    print("   max |state - Preprocessor(rows)| =", float((out.state.float_features.float() - ref).abs().max()))
