#!/bin/bash
# tools/pmc_cmd.sh COUNTER NAME SCRIPT [ARGS...] -- one rocprofv3 --pmc pass (one counter per pass, no trace domains
# besides the kernel dispatch rows the counter collection needs) of `python3 SCRIPT ARGS` into gpurun_out/NAME/.
counter=$1; name=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
script=$1; shift
rocprofv3 --pmc "$counter" --output-format csv -d "$root/gpurun_out/$name" -o "$name" -- python3 "$root/$script" "$@" > "$root/gpurun_out/$name.log" 2>&1
