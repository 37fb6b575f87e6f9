#!/usr/bin/env bash
# Headline kernel only: tile tests, bench line, rocprofv3 kernel stats and HBM counters (separate passes) of `bench.py --no-cpu-baseline --no-sub-results`.
# Outputs under gpurun_out/ (tools/collect_profiles.py <tag> anymal:4096:tiles copies the summaries into profiles/).
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
[ "${SKIP_TESTS:-0}" = 1 ] || timeout 1500 python -m pytest tests/test_tiles.py -x -q -m gpu --durations=5 2>&1 | tail -12 | tee gpurun_out/pytest_tiles.log
B="python bench.py --no-cpu-baseline --no-sub-results"
for i in 1 2 3; do timeout 300 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench: value %.4g frac %.4f kernel_ms %.4f ms_per_step %.4f store-only %.4f memset %.4f checksum %.9f' % (d['value'], r['frac'], r['kernel_ms'], d['ms_per_step'], r['box_store_only_ms'], r['box_memset_ms'], d['checksum']))"; done | tee gpurun_out/bench_headline.log
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('driver command: value %.4g frac %.4f kernel_ms %.4f' % (d['value'], r['frac'], r['kernel_ms']))" | tee -a gpurun_out/bench_headline.log
rm -rf gpurun_out/prof gpurun_out/pmc1 gpurun_out/pmc2
cd /tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o anymal -- $B > gpurun_out/prof_bench.log 2>&1)
(cd $R && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc1 -o anymal -- $B --steps 5 --warmup 1 > gpurun_out/pmc1.log 2>&1)
(cd $R && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc2 -o anymal -- $B --steps 5 --warmup 1 > gpurun_out/pmc2.log 2>&1)
cd $R
head -3 gpurun_out/prof/anymal_kernel_stats.csv | cut -c1-220
