#!/bin/bash
# what power cap / clock limits does this box run under, and where do clock and power sit during the C2 step and a pure MFMA loop?
cd /root/repo; O=/root/repo/gpurun_out/r06_power; mkdir -p $O
{ rocm-smi --showmaxpower --showpower --showclocks --showperflevel 2>&1 | grep -vE "^=|^$"; for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap_max /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap_default; do echo "$f $(cat $f 2>/dev/null)"; done; for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $f; cat $f; done; } > $O/caps.txt 2>&1
cat $O/caps.txt | head -40
sample() { for i in $(seq 1 $1); do sleep 0.25; echo "$(cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_average 2>/dev/null | head -1) $(grep '\*' /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -1) $(cat /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input 2>/dev/null | head -1)"; done; }
( sample 60 > $O/step_samples.txt & ) ; RG_SKIP_PREFLIGHT=1 timeout 300 python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --no-parity --no-kernel-profile --sustained-steps 20000 --report $O/step.json > $O/step.line 2>/dev/null; sleep 1
python -c "
import json; r=json.load(open('$O/step.json')); s=r['sustained']; print('C2 step: sustained', {k:v for k,v in s.items() if not isinstance(v,(list,dict))})"
echo "samples during the step (uW, sclk level, Hz):"; sort $O/step_samples.txt | uniq -c | sort -rn | head -8
cd profiles/microbench; ( sample 40 > $O/mfma_samples.txt & ) ; timeout 120 ./mfma_feed > $O/mfma_feed.txt 2>&1; head -3 $O/mfma_feed.txt
echo "samples during mfma_feed:"; sort $O/mfma_samples.txt | uniq -c | sort -rn | head -8
