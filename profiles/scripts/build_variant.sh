#!/bin/bash
# a second build of the library with extra compile flags, for same-box A/B runs through RG_LIB:
#   bash profiles/scripts/build_variant.sh NAME "-DRG_SOMETHING=1"   ->  reagent_amd/lib_NAME/libreagent_hip.so
set -e
cd /root/repo
make -C reagent_amd/csrc -j8 OUT=/root/repo/reagent_amd/lib_$1 EXTRA="$2" 2>&1 | grep -E "error|Error" || true
ls -la reagent_amd/lib_$1/libreagent_hip.so
