#!/bin/bash
cd /root/repo
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed"; exit 97; }
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -6
echo "--- under load ---"
(timeout 300 python bench.py --steps 6000 --warmup 5 --no-cpu-baseline --no-parity --no-kernel-profile --launch eager --no-graph > /tmp/long.json 2>/dev/null) &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Socket" | tr '\n' ' '; echo; sleep 0.4; done
wait $BP
python -c "import json; r=json.load(open('/tmp/long.json')); print('long run ms/step', r['ms_per_step'])"
