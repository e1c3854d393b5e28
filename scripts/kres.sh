#!/bin/bash
# kernel resource table of one translation unit: scripts/kres.sh attention.hip [extra flags]
cd "$(dirname "$0")/../pea_diffusion_amd/csrc"
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres.o 2>&1 |
  grep -E "remark: +(Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|VGPRs Spill)" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  awk '/^Function Name/{if(n)print n, v, a, s, o, sp; n=$3} /^VGPRs:/{v="vgpr="$2} /^AGPRs/{a="agpr="$2} /^ScratchSize/{s="scratch="$4} /^Occupancy/{o="occ="$4} /^VGPRs Spill/{sp="spill="$3} END{print n,v,a,s,o,sp}' | c++filt | cut -c1-220
