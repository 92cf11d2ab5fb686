export PROBE_MODES=auto
for rep in 1 2; do
for P in 45 60 75 90 100; do
  echo "== first percent $P"
  S2K_GP_FIRST_PERCENT=$P python3 tools/keyed_probe.py 20 16 2>&1 | grep '"mode"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   keys 2^%d: %.3f ms  stages %s' % (d['keys_log2'], d['ms'], {k: round(v, 2) for k, v in d['stages_ms'].items()}))"
done
done
