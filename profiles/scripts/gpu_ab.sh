#!/bin/bash
# same-box A/B of kernel variants: profiles/scripts/gpu_ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument = one environment)
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
CFG=${AB_CONFIG:-c2}; PREC=${AB_PREC:-bf16}
for rep in 1 2; do
for envs in "$@"; do
  env $envs timeout 600 python bench.py --config $CFG --precision $PREC --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-accurate --no-also --sustained-steps 0 --launch eager --no-graph ${AB_EXTRA} > $OUT/ab.json 2> $OUT/ab.err || tail -5 $OUT/ab.err
  python - "$envs" <<'PY'
import json, sys
r = json.load(open("/root/repo/gpurun_out/ab.json"))
calls = r.get("per_call_ms_per_step", {})
short = {k.split("(")[0].replace("rg_mlp_", "").replace("rg_", "") + ("+save" if ", 1, (" in k else ""): v for k, v in calls.items()}
top = " ".join(f"{k}={v*1e3:.0f}" for k, v in list(short.items())[:7])
print(f"[{sys.argv[1]:32s}] ms/step {r['ms_per_step']:.4f}  fc_frac {r.get('fc_roofline',{}).get('frac',0):.4f}  dom {r.get('roofline',{}).get('frac',0):.4f} | us/step: {top}")
PY
done; done
