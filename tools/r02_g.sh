#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02g; O=gpurun_out/r02g
timeout 900 python -m pytest tests/test_gpu_frame_kernel.py -q -p no:cacheprovider -k "bit_for_bit or ragged" 2>&1 | grep -v amdgpu.ids | tail -30
