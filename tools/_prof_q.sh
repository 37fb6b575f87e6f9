export TMPDIR=/tmp
p=$1; shift
build/batched_${p}_test /tmp/cg_$p 4096 0 > /dev/null 2>&1
for mode in "$@"; do
rm -rf gpurun_out/bprof_e; (cd /tmp && env $mode timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/bprof_e -o b -- $OLDPWD/build/batched_${p}_test /tmp/cg_$p 4096 0 > /dev/null 2>&1)
f=$(find gpurun_out/bprof_e -name "b_kernel_stats.csv" | head -1)
echo "== $p $mode"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    print(f"{float(r['Percentage']):6.2f}%  calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
PY
done
