#!/bin/bash
# Same-box A/B of PREBUILT library variants (python -c "import secp256k1_voi_amd as S; S.build_variant(name, flags)"
# on the build host; selected with S2K_LIB, so no GPU time goes into compiling):
#   bash tools/ab_libs.sh default pack ...      ("default" = the shipped library)
# Per variant: parity tests of the hot path, three bench runs (step, ladder kernel HIP-event time, clock), and one
# FETCH_SIZE / WRITE_SIZE pass for the fabric traffic of the ladder kernel.
REPO=$PWD
O=$REPO/gpurun_out/ab; mkdir -p $O
for v in "$@"; do
  if [ "$v" = default ]; then unset S2K_LIB; else export S2K_LIB=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.$v.so; fi
  T=$(timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hotpath.py -m gpu -q -x -k "random_batches or exceptional or wycheproof_ecdsa or ladder or recover_random or schnorr_random" 2>&1 | tail -1)
  echo "[$v] $T"
  case "$T" in *failed*|*error*|"") echo "[$v] parity failed: variant skipped"; continue;; esac
  for rep in 1 2 3; do
    timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v]', 'ms_per_step=%.3f kernel_ms=%.3f median=%.3f clock=%.0f build=%s' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_median'], r['shader_clock_mhz'], d['config']['build']))"
  done
  (cd /tmp && export TMPDIR=/tmp && cd $REPO && \
   timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-extras > /dev/null 2>&1; \
   timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$v -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-extras > /dev/null 2>&1)
  python3 tools/collect_traffic.py $O/fetch_$v $O/write_$v | python3 -c "import sys,json; d=json.load(sys.stdin); k=d.get('k_verify_fast',{}); print('[$v] k_verify_fast fetch_raw=%.2f GB write=%.2f GB' % (k.get('fetch_bytes_raw',0)/1e9, k.get('write_bytes',0)/1e9))"
done
