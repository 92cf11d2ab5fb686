for LN in 98304 131072 196608 262144 393216 589824; do echo "== lanes $LN"; S2K_MSM_LANES=$LN python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids; done
