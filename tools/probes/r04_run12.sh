DMM_HOST_PROF=1 python3 tools/host_bound.py c2@64x96 2>&1 | grep -v amdgpu.ids
python3 tools/host_bound.py c5@64x96 c2@128x192 2>&1 | grep -v amdgpu.ids
