#!/bin/bash
# the same script under several builds of the library: scripts/ab_libs.sh "base new d3" scripts/probe_unet.py [args]
R=${GRAFT_REPO_ROOT:-/root/repo}
LIBS=$1; shift
for l in $LIBS; do
  f=$R/pbnet_amd/libpbnet_hip_$l.so; [ "$l" = new ] && f=$R/pbnet_amd/libpbnet_hip.so
  echo "== $l"; PBNET_HIP_LIB=$f timeout 300 python "$@" 2>&1 | grep -v amdgpu.ids
done
