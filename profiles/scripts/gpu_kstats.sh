#!/bin/bash
# per-kernel rocprofv3 averages of a few environments on ONE box:
#   bash profiles/scripts/gpu_kstats.sh "<config> <precision> <steps>" "ENV=.. ENV=.." "ENV=.." ...
# prints, per environment, the average duration of every rg:: kernel (us) and the timed ms/step
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
set -- $1 "${@:2}"; CFG=$1; PREC=$2; STEPS=$3; shift 3
i=0
for envs in "$@"; do
  i=$((i+1)); D=$OUT/kstats_$i; rm -rf $D
  (cd /tmp && env $envs timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o k -- python /root/repo/bench.py --config $CFG --precision $PREC --repeats 1 --steps $STEPS --warmup 20 --no-cpu-baseline --no-kernel-profile --no-parity --no-accurate --no-also --sustained-steps 0 --launch eager > $D.json 2> $D.err) || tail -3 $D.err
  python - "$envs" $D <<'PY'
import csv, glob, json, sys, re
envs, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
ks = {re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("rg::", "")[:28]: float(r["AverageNs"]) / 1e3 for r in rows if "rg::" in r["Name"] and int(r["Calls"]) > 20}
try: ms = json.load(open(d + ".json"))["ms_per_step"]
except Exception: ms = float("nan")
print(f"[{envs[-60:]:60s}] ms/step {ms:.4f} | " + "  ".join(f"{k}={v:.1f}" for k, v in ks.items()))
PY
  find $D -name "*kernel_trace.csv" -delete
done
