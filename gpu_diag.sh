#!/bin/bash
cd /root/repo
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed"; exit 97; }
timeout 600 python profiles/microbench/diag_c4_actor.py 2>&1 | tail -15
timeout 900 python -m pytest tests/test_graph_replay.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -15
timeout 600 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-kernel-profile | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c4 ms/step %.3f host %.3f' % (r['ms_per_step'], r['host_enqueue_ms_per_step']), r['config']['launch'][:200])"
timeout 600 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-kernel-profile --no-graph | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c4 eager ms/step %.3f host %.3f' % (r['ms_per_step'], r['host_enqueue_ms_per_step']))"
