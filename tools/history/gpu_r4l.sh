#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_c_harness.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python tools/two_contexts_probe.py | tail -2
python tools/pipeline_trace.py 24 | tail -1
python tools/pipeline_trace.py 24 | tail -1
timeout 300 python tools/boundary_probe.py 20 16 12 > $O/boundary.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/boundary.json')); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if 'ms' in k and 'all' not in k and 'stats' not in k})"
