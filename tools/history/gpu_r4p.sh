#!/bin/bash
O=gpurun_out/r4p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_c_harness.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/two_contexts_probe.py | tail -2
for one in 0 1; do echo "one_lane=$one"; S2K_SUBMIT_ONE_LANE=$one timeout 300 python tools/boundary_probe.py 20 16 12 > $O/boundary_$one.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/boundary_$one.json')); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if 'ms' in k and 'all' not in k and 'stats' not in k})"; done
