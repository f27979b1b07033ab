# tools/ab_fused_r03.sh -- A/B of fused-kernel build variants (make -C whvi_amd/csrc tuning TAG=ab_<name> DEFS=-D...), each in
# its own process on the same shapes (tools/probe_fused_inst.py); run on the GPU box: bash tools/ab_fused_r03.sh
python tools/probe_fused_inst.py batch-major,sample-major > gpurun_out/r03_ab_prod.log 2>&1
for lib in whvi_amd/_exp/libwhvi_hip_ab_*.so; do t=$(basename $lib .so); t=${t#libwhvi_hip_ab_}; python tools/_tuning.py --run $lib tools/probe_fused_inst.py batch-major,sample-major > gpurun_out/r03_ab_$t.log 2>&1; done
for f in gpurun_out/r03_ab_*.log; do echo "== $f"; grep -E "^float" $f | awk '{print $1, $2, $3, $4, $5, $6, $9, $12, $NF}' | cut -c1-150; done
