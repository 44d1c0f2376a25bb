#!/bin/bash
# Scan every kernel of csrc/*.hip for register spills / scratch (hipcc -Rpass-analysis=kernel-resource-usage).
# A spilling fp32-MFMA kernel pays for every scratch access out of the matrix pipe's time: run this
# after touching a kernel.  Prints only kernels with spills or scratch.
cd "$(dirname "$0")/../poem-v2_amd/csrc" || exit 1
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c "$f" -o /tmp/scan_spills.o -Rpass-analysis=kernel-resource-usage 2>&1 |
    awk -v F="$f" '/Function Name/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
                   /VGPRs:/ {v=$4} /ScratchSize/ {sc=$5}
                   /VGPRs Spill/ {sp=$5; if (sp+0>0 || sc+0>0) print F, substr(name,1,72), "VGPRs", v, "scratch", sc, "spilled", sp}'
done
rm -f /tmp/scan_spills.o
