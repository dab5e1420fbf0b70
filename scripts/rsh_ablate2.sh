export PBN_PROBE_CFGS=11000
for d in 4096 4098 4100 4352 4102 4358 4608 5126; do echo "DBG=$d"; PBN_CONV_DBG=$d PBN_PROBE_CASES="0,96,96;1,96,96" python scripts/probe_rs.py 2>&1 | grep -v amdgpu.ids; done
