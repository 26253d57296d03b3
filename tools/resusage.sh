#!/bin/bash
# usage: tools_resusage.sh file.hip...  -> one line per kernel: name VGPRs scratch occupancy spills
hipcc --offload-arch=gfx950 -O3 -fPIC -c -Rpass-analysis=kernel-resource-usage -o /dev/null "$@" 2>&1 \
 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|SGPRs Spill|Occupancy|ScratchSize|LDS Size" \
 | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - - - - | sed -e 's/Function Name: //' | awk '{$1=$1};1'
