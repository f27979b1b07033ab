#!/bin/bash
# tools/prof_cmd.sh NAME SCRIPT [ARGS...] -- rocprofv3 kernel trace + stats of `python3 SCRIPT ARGS` into
# gpurun_out/NAME/ (run on the GPU box; rocprofv3 wants a writable cwd and TMPDIR).
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
script=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/$name" -o "$name" -- python3 "$root/$script" "$@" > "$root/gpurun_out/$name.log" 2>&1
