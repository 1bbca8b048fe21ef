#!/bin/bash
# round 4, batch 20: the free-running next-batch sampler (--prefetch) on C3 / C2 / C4, eager launches, same box
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
for rep in 1 2; do
for cfg in c3 c2; do
for pf in "" "--prefetch"; do
  timeout 600 python bench.py --config $cfg --precision bf16 --steps 40 --warmup 5 --no-cpu-baseline --no-accurate --no-also --no-parity --sustained-steps 0 --launch eager --no-graph --no-kernel-profile $pf > $OUT/b20.json 2> $OUT/b20.err || tail -5 $OUT/b20.err
  python - "$cfg $pf" <<'PY'
import json, sys
r = json.load(open("/root/repo/gpurun_out/b20.json"))
print(f"[{sys.argv[1]:16s}] ms/step {r['ms_per_step']:.4f} host {r.get('host_enqueue_ms_per_step',0):.3f}")
PY
done; done; done
