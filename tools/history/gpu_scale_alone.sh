cd /tmp && export TMPDIR=/tmp
for P in 60 100; do
S2K_GP_FIRST_PERCENT=$P rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/scale_alone_$P -o run -- python3 $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-extras > /dev/null 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OLDPWD/gpurun_out/scale_alone_$P/run_kernel_stats.csv")))
print("first percent $P")
for r in rows:
    if any(k in r["Name"] for k in ("k_key_scale","k_key_odd","k_key_chain","k_generator_part","k_key_cofactors","k_verify_fast<4>")):
        print("  %-50s calls %4s avg %8.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:50], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
