#!/bin/bash
# compact kernel resource report: tools/kres.sh file.hip [extra flags]  (compiles to /tmp)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno -Wall -Wno-unused-function -DHR_FAST_EXP -DHR_FAST_POST "$@" -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
  grep -E "error|warning|Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste -sd' ' | sed 's/Function Name: /\n/g'
