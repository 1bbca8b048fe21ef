#!/bin/bash
# per-kernel timeline of ONE step (rocprofv3 kernel trace): bash profiles/scripts/gpu_timeline.sh <config> <precision> [ENV=..]
cd /root/repo; OUT=/root/repo/gpurun_out; export TMPDIR=/tmp; CFG=${1:-c2}; PREC=${2:-bf16}; shift 2
D=$OUT/timeline_${CFG}_${PREC}
(cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python /root/repo/bench.py --config $CFG --precision $PREC --steps 4 --warmup 3 --repeats 1 --no-cpu-baseline --no-kernel-profile --no-parity --no-accurate --no-also ${TL_LAUNCH:---launch eager --no-graph} > $D.log 2>&1; echo rc=$?)
python - $D <<'PY'
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
first = next((k for k in ("replay_dqn_batch", "replay_policy_batch", "replay_nstep") if any(k in r["Kernel_Name"] for r in rows)), "replay_nstep")
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
print(f"one step = {(int(rows[b]['Start_Timestamp']) - t0)/1e3:.1f} us, {b-a} kernels")
prev_end = 0
for r in rows[a:b]:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rg::", "")[:50]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"  {s/1e3:8.1f} -> {e/1e3:8.1f}  ({(e-s)/1e3:7.1f} us, gap {max(0,(s-prev_end))/1e3:5.1f}) q{r.get('Queue_Id','?')} grid {r.get('Grid_Size','?'):>8} {nm}")
    prev_end = max(prev_end, e)
PY
find $D -name "*kernel_trace.csv" -size +20M -delete
