#!/bin/bash
# round 6 same-box A/B: profiles/scripts/gpu_ab6.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument = one environment; "-" = none)
# AB_CONFIG (c2) AB_PREC (bf16) AB_REPS (2) AB_EXTRA (more bench flags).  Reads the full record (--report), not the printed digest.
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
if [ -z "$AB_NO_PREFLIGHT" ]; then
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
fi
CFG=${AB_CONFIG:-c2}; PREC=${AB_PREC:-bf16}
for rep in $(seq 1 ${AB_REPS:-2}); do
for envs in "$@"; do
  [ "$envs" = "-" ] && envs="RG_NONE=1"
  env $envs RG_SKIP_PREFLIGHT=1 timeout 600 python bench.py --config $CFG --precision $PREC --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --sustained-steps 0 --report $OUT/ab.json ${AB_EXTRA} > $OUT/ab.line 2> $OUT/ab.err || tail -5 $OUT/ab.err
  python - "$CFG $PREC $envs" <<'PY'
import json, sys
r = json.load(open("/root/repo/gpurun_out/ab.json"))
calls = r.get("per_call_ms_per_step", {})
def short(k):
    n = k.split("(")[0].replace("rg_mlp_", "").replace("rg_", "")
    if ", 1, (" in k: n += "+save"
    if ", 2, (" in k: n += "+save2"
    return n
agg = {}
for k, v in calls.items():
    agg[short(k)] = agg.get(short(k), 0.0) + v
top = " ".join(f"{k}={v*1e3:.0f}" for k, v in list(agg.items())[:12])
p = r.get("parity") or {}
pk = [k for k in ("max_abs_dq", "max_abs_dquantile", "max_abs_dlogits") if k in p]
print(f"[{sys.argv[1]:40s}] ms/step {r['ms_per_step']:.4f} fc {r.get('fc_roofline',{}).get('frac',0):.4f} dom {r.get('roofline',{}).get('frac',0):.4f} parity {p.get('ok')}/{p.get('sane')} {[p.get(k) for k in pk]} | {top}")
PY
done; done
