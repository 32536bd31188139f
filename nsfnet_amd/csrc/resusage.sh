#!/bin/bash
# usage: resusage.sh file.hip  -> one line per kernel: name VGPR AGPR scratch occupancy spills
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/$(basename "$1").o 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|VGPRs Spill|error" \
 | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - - | sed -e 's/Function Name: //'
