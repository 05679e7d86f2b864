#!/bin/bash
# kernel resource usage of one HIP source: name, SGPR, VGPR, AGPR, scratch, occupancy
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "$1" -I/root/repo/livetalking_amd/csrc -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|TotalSGPRs|  VGPRs:|AGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n, s; n=$3; s=""} !/Function Name/{s=s" | "$0} END{print n, s}' | sed -E 's/_ZN3ltk[0-9]+//'
