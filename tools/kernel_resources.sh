#!/bin/bash
# Per-kernel register / scratch / LDS / occupancy figures of hip/pt_host.hip as hipcc reports them
# (-Rpass-analysis=kernel-resource-usage).  Usage: tools/kernel_resources.sh [extra hipcc flags] ; filter with grep.
cd "$(dirname "$0")/../amber_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -c hip/pt_host.hip -o /dev/null \
      -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       /TotalSGPRs:/ {sg=$(NF-1)} / VGPRs:/ {vg=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {oc=$(NF-1)}
       /LDS Size/ {lds=$(NF-1); printf "%-70s vgpr %3s sgpr %3s scratch %4s occ %s lds %s\n", name, vg, sg, sc, oc, lds}'
