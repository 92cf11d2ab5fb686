#!/bin/bash
# same-box A/B of the scalar preparation variants: kernel time of k_scalar_prep at 2^20
# whatever variant ran last, leave the DEFAULT build behind (build() also rebuilds when the recorded flags differ)
trap 'S2K_EXTRA_FLAGS="" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1' EXIT
for v in "$@"; do
  S2K_EXTRA_FLAGS="$v" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
  rm -rf gpurun_out/ab_prep
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_prep -o run -- python3 tools/tail_probe.py > /dev/null 2>&1
  python3 - "$v" <<'PY'
import csv, sys
for r in csv.DictReader(open('gpurun_out/ab_prep/run_kernel_stats.csv')):
    if 'k_scalar_prep' in r['Name'] or 'k_verify_fast' in r['Name']:
        print(sys.argv[1], r['Name'].split('(')[0][-16:], 'min_us', float(r['MinNs'])/1e3, 'avg_us', round(float(r['AverageNs'])/1e3,1))
PY
done
