#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py tests/test_graph_replay.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | tail -5
AB_CONFIG=c3 bash profiles/scripts/gpu_ab.sh "RG_QR_WGRAD_STREAMS=0" "RG_QR_WGRAD_STREAMS=1" 2>&1
AB_CONFIG=c3 AB_PREC=bf16x3 bash profiles/scripts/gpu_ab.sh "RG_QR_WGRAD_STREAMS=0" "RG_QR_WGRAD_STREAMS=1" 2>&1
python - <<'PY'
import json
r=json.load(open('/root/repo/gpurun_out/ab.json'))
for k,v in r['per_call_ms_per_step'].items(): print("  %-80s %.4f"%(k[:80],v))
PY
