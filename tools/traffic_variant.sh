#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_verify_fast for a compile-time variant:  bash tools/traffic_variant.sh "<flags>"
# whatever variant ran last, leave the DEFAULT build behind (build() also rebuilds when the recorded flags differ)
trap 'S2K_EXTRA_FLAGS="" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1' EXIT
S2K_EXTRA_FLAGS="$1" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1
REPO=$PWD; cd /tmp && export TMPDIR=/tmp; cd $REPO
rm -rf gpurun_out/tv_fetch gpurun_out/tv_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/tv_fetch -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/tv_write -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > /dev/null 2>&1
python3 tools/collect_traffic.py gpurun_out/tv_fetch gpurun_out/tv_write | python3 -c "import sys,json; d=json.load(sys.stdin); print('$1', d['k_verify_fast'])"
