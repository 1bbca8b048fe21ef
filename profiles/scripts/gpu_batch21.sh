#!/bin/bash
# round 4, batch 21: batch splits per action group in the grouped head's weight gradient (16 groups x 2 k-groups x splits workgroups
# next to the trunk's weight gradient on the other stream): 8 (rounds 2-4) / 4 / 2 / 16, C3 bf16 and split-bf16, same box
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
for rep in 1 2; do
for prec in bf16 bf16x3; do
for sp in 8 4 2 16; do
  RG_QR_HEAD_SPLITS=$sp timeout 600 python bench.py --config c3 --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --no-parity --sustained-steps 0 --launch eager --no-graph > $OUT/b21.json 2> $OUT/b21.err || tail -5 $OUT/b21.err
  python - "$prec splits=$sp" <<'PY'
import json, sys
r = json.load(open("/root/repo/gpurun_out/b21.json"))
calls = r.get("per_call_ms_per_step", {})
short = {k.split("(")[0].replace("rg_mlp_", "").replace("rg_", ""): v for k, v in calls.items()}
print(f"[{sys.argv[1]:18s}] ms/step {r['ms_per_step']:.4f} | head_wgrad {short.get('group_head_wgrad',0)*1e3:.0f} trunk wgrad {short.get('wgrad_fused',0)*1e3:.0f}")
PY
done; done; done
