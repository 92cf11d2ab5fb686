#!/bin/bash
# MSM / RLC parity tests, then timing A/B over the reduction chunk size and prebuilt library variants.
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -x -m gpu -k "msm or rlc or bisect" 2>&1 | tail -5
for CL in 2 3 4; do
  echo "== chunk_log2 $CL"; S2K_MSM_CHUNK_LOG2=$CL python3 tools/msm_time.py
done
for V in "$@"; do
  echo "== variant $V"; S2K_LIB=$PWD/secp256k1_voi_amd/libsecp256k1_voi_amd.$V.so python3 tools/msm_time.py
done
