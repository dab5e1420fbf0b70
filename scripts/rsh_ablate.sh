# staged row-stationary kernel: timing under the PBN_CONV_DBG switches that survive in the fast loop (8 no weight DMA, 16 no main
# loop, 32 no epilogue, 64 no staging, 128 slow path), then correctness spot checks in fp32 (K = 8 maps, segments, slow path)
export PBN_PROBE_CFGS=${PBN_PROBE_CFGS:-32,12000,11000}
PBN_PROBE_CASES="0,96,96;0,128,96;1,96,96;1,128,96;1,32,32;0,96,96,-2;0,32,32,2" python scripts/probe_rs.py 2>&1 | grep -v amdgpu.ids
for d in 16 48 64 8; do echo "DBG=$d"; PBN_PROBE_CFGS=11000 PBN_CONV_DBG=$d PBN_PROBE_CASES="0,96,96;1,96,96" python scripts/probe_rs.py 2>&1 | grep -v amdgpu.ids; done
echo F32; PBN_PROBE_CFGS=32,11000 PBN_PROBE_DTYPE=f32 PBN_PROBE_CASES="0,96,96;1,32,32;0,96,96,-2;0,32,32,2" python scripts/probe_rs.py 2>&1 | grep -v amdgpu.ids
echo SEG; PBN_RSH_SLOTS=704 PBN_PROBE_CFGS=32,11000 PBN_PROBE_DTYPE=f32 PBN_PROBE_CASES="0,96,96" python scripts/probe_rs.py 2>&1 | grep -v amdgpu.ids
echo DIRECT; PBN_CONV_DBG=128 PBN_PROBE_CFGS=32,11000 PBN_PROBE_DTYPE=f32 PBN_PROBE_CASES="1,96,96;0,96,96,-2" python scripts/probe_rs.py 2>&1 | grep -v amdgpu.ids
