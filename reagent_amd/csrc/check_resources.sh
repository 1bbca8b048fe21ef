#!/bin/bash
# Per-kernel registers / scratch / LDS of every .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
# Anything with ScratchSize > 0 or an LDS size nobody declared (LLVM promotes dynamically indexed
# private arrays to LDS) is a performance bug: run this after touching a kernel.
cd "$(dirname "$0")"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. -I../../include -c "$f" -o /dev/null \
      -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|VGPRs:|ScratchSize|LDS Size" | paste - - - - |
    sed -E 's/\[-Rpass[^]]*\]//g; s/[a-z_]+\.hip:[0-9:]+ remark://g; s/ +/ /g' | sed "s/^/$f /"
done
