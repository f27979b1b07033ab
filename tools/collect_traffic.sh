#!/bin/bash
# tools/collect_traffic.sh -- on the GPU box: the two PMC passes behind bench.py's `roofline.traffic` (FETCH_SIZE and
# WRITE_SIZE, one counter per pass as MI355X_MICROARCH.md prescribes), the kernel-trace stats of the same command, and
# the profiles/hbm_traffic.json record stamped with this build's kernel-source hash (tools/update_hbm_traffic.py).
# Outputs under gpurun_out/ (scratch) -- copy the summaries into profiles/rNN/.
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r02}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
args="--steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$root/gpurun_out/${tag}_pmc_fetch" -o fetch -- python3 "$root/bench.py" $args > "$root/gpurun_out/${tag}_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$root/gpurun_out/${tag}_pmc_write" -o write -- python3 "$root/bench.py" $args > "$root/gpurun_out/${tag}_pmc_write.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/${tag}_bench_trace" -o bench -- python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$root/gpurun_out/${tag}_bench_under_profiler.json" 2> "$root/gpurun_out/${tag}_bench_trace.log"
cd "$root"
python3 tools/update_hbm_traffic.py fwht_f32_D4096_rows1048576 "gpurun_out/${tag}_pmc_fetch" "gpurun_out/${tag}_pmc_write" > "gpurun_out/${tag}_hbm_traffic_record.json"
cp profiles/hbm_traffic.json "gpurun_out/${tag}_hbm_traffic.json"
python3 tools/pmc_summary.py "gpurun_out/${tag}_pmc_fetch" "gpurun_out/${tag}_pmc_FETCH_SIZE_headline.csv"
python3 tools/pmc_summary.py "gpurun_out/${tag}_pmc_write" "gpurun_out/${tag}_pmc_WRITE_SIZE_headline.csv"
