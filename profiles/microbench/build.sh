#!/bin/bash
# cross-compile every microbenchmark for gfx950 (runs in the build container; the binaries travel with the gpurun snapshot)
set -e
cd "$(dirname "$0")"
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../reagent_amd/csrc -I../../include -Wno-unused-value"
$HIPCC mfma_peak.hip -o mfma_peak
$HIPCC mfma_feed.hip -o mfma_feed
$HIPCC mfma_feed_dma.hip -o mfma_feed_dma
$HIPCC fwd_phases.hip -o fwd_phases
$HIPCC wgrad_phases.hip -o wgrad_phases
$HIPCC chain_fwd.hip -o chain_fwd
$HIPCC -DCH_KK=2 -DCH_NSLOT=8 chain_fwd.hip -o chain_fwd_k2s8
$HIPCC -DCH_KK=2 -DCH_NSLOT=9 chain_fwd.hip -o chain_fwd_k2s9
$HIPCC -DCH_PF=4 chain_fwd.hip -o chain_fwd_k4s4p4

$HIPCC x3_phases.hip -o x3_phases
$HIPCC bwd_phases.hip -o bwd_phases
ls -la mfma_peak mfma_feed fwd_phases wgrad_phases x3_phases chain_fwd*
