timeout -k 10 120 python tools/debug_u2.py 2>&1 | grep -v amdgpu.ids
