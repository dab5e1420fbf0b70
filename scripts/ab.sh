#!/bin/bash
# A/B of two builds of the library on one box: scripts/ab.sh <script.py> [args]  (base = pbnet_amd/libpbnet_hip_base.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
echo "== base"; PBNET_HIP_LIB=$R/pbnet_amd/libpbnet_hip_base.so timeout 300 python "$@" 2>&1 | grep -v amdgpu.ids
echo "== new";  timeout 300 python "$@" 2>&1 | grep -v amdgpu.ids
