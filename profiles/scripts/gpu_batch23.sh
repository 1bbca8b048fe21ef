#!/bin/bash
# round 5, batch 23: the update of step k on the side stream beside step k+1's sampler (bench.py --side-update on / off),
# C2 (both modes), C3, C4 — same box, interleaved; first the bit-for-bit tests of the switch
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
timeout 900 python -m pytest tests/test_graph_replay.py -m gpu -q --no-header -p no:cacheprovider -k "side_stream" > $OUT/pytest_side_b23.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_side_b23.log
for rep in 1 2; do
for spec in "c2 bf16" "c3 bf16" "c4 bf16" "c2 bf16x3"; do
for side in off on; do
  set -- $spec
  timeout 600 python bench.py --config $1 --precision $2 --side-update $side --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --no-parity --sustained-steps ${SUS:-1000} --launch eager --no-graph --no-kernel-profile > $OUT/b23.json 2> $OUT/b23.err || tail -5 $OUT/b23.err
  python - "$1 $2 side=$side" <<'PY'
import json, sys
r = json.load(open("/root/repo/gpurun_out/b23.json"))
s = r.get("sustained") or {}
print(f"[{sys.argv[1]:24s}] ms/step {r['ms_per_step']:.4f}  sustained {s.get('ms_per_step', 0):.4f}  regions {r['region_ms']}  host {r['host_enqueue_ms_per_step']:.3f} | {r['config']['launch'][:60]}")
PY
done; done; done
