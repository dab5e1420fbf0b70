for mr in 20000 30000 60000; do echo "== max_rows $mr"; PBN_WAVE_MAX_ROWS=$mr timeout 300 python scripts/probe_unet.py 2>&1 | grep -v amdgpu; done
for mg in 6.3 20 40; do echo "== max_gmacs $mg (rows 30000)"; PBN_WAVE_MAX_ROWS=30000 PBN_WAVE_MAX_GMACS=$mg timeout 300 python scripts/probe_unet.py 2>&1 | grep "three"; done
