cd /root/repo; timeout 800 python tools/dbg_oracle_big.py 13312 3328 4992 16 2>&1 | grep -v amdgpu.ids; timeout 800 python tools/dbg_oracle_big.py 6144 1536 2304 16 2>&1 | grep -v amdgpu.ids
