#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
bash tools/pmc_one.sh e_two "$C" --no-frame-kernel 2>&1 | grep -v amdgpu.ids
bash tools/pmc_one.sh e_frame "$C" --sample-waves 8 2>&1 | grep -v amdgpu.ids
