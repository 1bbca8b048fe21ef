#!/bin/bash
# the driver's command (default flags) with its wall time: bash profiles/scripts/gpu_default_bench.sh [tag]
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; TAG=${1:-def}
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
t0=$(date +%s.%N)
timeout 900 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-5} > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "rc=$? wall=$(python -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s"
python - <<PY
import json
r=json.load(open("$OUT/bench_$TAG.json"))
def line(tag, d):
    p = d.get("parity") or {}
    print("%-10s %.3e tr/s  %.3f ms/step  fc_frac %.4f exec %.4f | parity dq %s dquant %s dlogits %s meets_north_star %s ok %s | %s" % (
        tag, d["value"], d["ms_per_step"], d.get("fc_roofline",{}).get("frac",0), d.get("fc_roofline",{}).get("executed_frac",0),
        p.get("max_abs_dq"), p.get("max_abs_dquantile"), p.get("max_abs_dlogits"), p.get("meets_north_star"), p.get("ok"), p.get("error", "")))
line("c2 bf16", r)
if "accurate" in r: line("c2 x3", r["accurate"])
for k, d in r.get("also_measured", {}).items():
    if "error" in d: print(k, d)
    else:
        line(k, d)
        if "accurate" in d:
            if "error" in d["accurate"]: print(k, "x3", d["accurate"])
            else: line(k + " x3", d["accurate"])
for tag, d in (("c2 bf16", r), ("c2 x3", r.get("accurate", {}))):
    s = d.get("sustained")
    if s: print("sustained %-8s %d steps %.3f ms/step  sclk %s  power %s  (%s)" % (tag, s["steps"], s["ms_per_step"], s.get("sclk_mhz"), s.get("power_w"), s.get("source")))
print("launch_calibration", r.get("launch_calibration"), "| instrumented_pass", r.get("instrumented_pass"))
print("cpu_baseline", r.get("cpu_baseline"))
for k,v in r["per_call_ms_per_step"].items(): print("  %-70s %.4f" % (k,v))
PY
tail -3 $OUT/bench_$TAG.err
