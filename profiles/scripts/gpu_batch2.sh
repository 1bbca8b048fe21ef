#!/bin/bash
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed"; exit 97; }
timeout 900 python -m pytest tests/test_baseline_shapes.py tests/test_full_size.py tests/test_qrdqn_trainer.py tests/test_data_parallel.py tests/test_model_autograd.py tests/test_torch_ops.py -m gpu -q -s --no-header -p no:cacheprovider > $OUT/pytest_gpu_b2.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu_b2.log | tail -3; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu_b2.log | head; grep "baseline_c3 bf16x3" $OUT/pytest_gpu_b2.log | head -4
L=/root/repo/reagent_amd
bash profiles/scripts/gpu_ab.sh "RG_LIB=$L/lib/libreagent_hip.so" "RG_LIB=$L/lib_m6/libreagent_hip.so" "RG_LIB=$L/lib_m1/libreagent_hip.so" "RG_LIB=$L/lib_m8/libreagent_hip.so" "RG_LIB=$L/lib_m9/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
# telemetry probe: what do the tools say while the chip is busy?
(python bench.py --no-cpu-baseline --no-parity --no-accurate --no-also --no-kernel-profile --sustained-steps 12000 > $OUT/tele_bench.json 2>/dev/null &) ; sleep 9
for i in 1 2 3; do amd-smi metric -g 0 --clock --power 2>&1 | grep -E -i "clk|power|socket" | head -24; echo ---; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power"; cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -8; sleep 0.7; done > $OUT/tele_probe.log 2>&1
wait; sleep 8; python -c "
import json; r=json.load(open('$OUT/tele_bench.json')); print('sustained', r.get('sustained'))"
head -60 $OUT/tele_probe.log
