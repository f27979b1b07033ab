#!/bin/bash
# tools/collect_f16_profile.sh -- on the GPU box: the three rocprofv3 passes of tools/profile_f16.py (config 5 on one GPU,
# fp16 and bf16, finite data) and their reduction into gpurun_out/f16_config5_rocprof_summary.csv (copy into profiles/rNN/).
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$root/gpurun_out/f16_trace" -o f16 -- python3 "$root/tools/profile_f16.py" 16 > "$root/gpurun_out/f16_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$root/gpurun_out/f16_fetch" -o f16 -- python3 "$root/tools/profile_f16.py" 4 > "$root/gpurun_out/f16_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$root/gpurun_out/f16_write" -o f16 -- python3 "$root/tools/profile_f16.py" 4 > "$root/gpurun_out/f16_write.log" 2>&1
cd "$root"
python3 tools/summarize_stream_profile.py gpurun_out/f16_trace gpurun_out/f16_fetch gpurun_out/f16_write 17179869184 gpurun_out/f16_config5_rocprof_summary.csv
