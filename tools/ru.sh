#!/bin/bash
# kernel resource usage of one source file (VGPRs, spills, scratch, LDS, occupancy): bash tools/ru.sh strip [kernel-name-filter] [extra flags]
cd "$(dirname "$0")/../phylo_hmrf_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -fno-honor-nans -mno-amdgpu-ieee $3 -Rpass-analysis=kernel-resource-usage --cuda-device-only -c -o /dev/null $1.hip 2>&1 | grep -E "Function Name|VGPRs:|Occupancy|ScratchSize|SGPRs:|Spill|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - - - - | grep "${2:-.}" | sed 's/_ZN5phmrf12_GLOBAL__N_1[0-9]*//; s/EvNS0_9StripGeom[^ \t]*//'
