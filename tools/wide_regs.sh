#!/bin/bash
# Register / scratch use of the 256-wide role kernels (compile only, no GPU).
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $SAMX_EXTRA_FLAGS -c /root/repo/sam_tf_amd/csrc/fused_wide.hip -o /tmp/fw_regs.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs|ScratchSize|Occupancy" | \
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/_ZN4samx4wide12_GLOBAL__N_1//' | awk '/Function Name/{if (l) print l; l=$0; next}{l=l" | "$0}END{print l}'
