#!/bin/bash
# quick perf iteration: fused-path GPU tests + bf16 bench only
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; TAG=${1:-it}
# preflight: a node whose first device touch faults (seen once: "Memory access fault by GPU" on tensor.to)
# would otherwise burn the whole GPU budget in core dumps and timeouts
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
timeout 600 python -m pytest tests/test_fused_mlp.py tests/test_dqn_trainer.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "rc=$?"
python - <<PY
import json
r=json.load(open("$OUT/bench_$TAG.json"))
print("value %.3e  ms/step %.3f host_enqueue %.3f fc_frac %.4f  dom %s" % (r["value"], r["ms_per_step"], r.get("host_enqueue_ms_per_step",0), r.get("fc_roofline",{}).get("frac",0), r.get("roofline",{}).get("kernel")))
for k,v in r["per_call_ms_per_step"].items(): print("  %-70s %.4f" % (k,v))
PY
tail -3 $OUT/bench_$TAG.err
