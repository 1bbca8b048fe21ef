#!/bin/bash
# round 5, call 2: same-box A/B of (a) the gather's structure-of-arrays descriptors (lib_noreg = the old array-of-structures
# reads) and (b) bf16 split partials of the stack's weight gradient (RG_WGRAD_BF16_PART), then the GPU tests that cover both
cd /root/repo
N=/root/repo/reagent_amd/lib_noreg/libreagent_hip.so
bash profiles/scripts/gpu_ab.sh "RG_LIB=$N RG_WGRAD_BF16_PART=0" "RG_WGRAD_BF16_PART=0" "RG_LIB=$N RG_WGRAD_BF16_PART=1" "RG_WGRAD_BF16_PART=1"
timeout 900 python -m pytest tests/test_fused_mlp.py tests/test_replay_buffer.py tests/test_baseline_shapes.py tests/test_full_size.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -5
