#!/bin/bash
# tools/kres.sh <file.hip> [contract]: registers / LDS / occupancy of every kernel in one translation unit
# (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
cd "$(dirname "$0")/../splat_renderer_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -I../../include -ffp-contract=${2:-off} -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 |
  grep -E "Function Name|VGPRs:|SGPRs:|Occupancy|LDS Size|ScratchSize" |
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - - - - 2>/dev/null | sed 's/Function Name: //' | c++filt | cut -c1-300
