set -e
python tools/probe_fused_inst.py batch-major,sample-major > gpurun_out/r03_ab_prod.log 2>&1
for t in aux1 aux16 aux2 up8 up6; do WHVI_HIP_LIB=whvi_amd/_exp/libwhvi_hip_$t.so python tools/probe_fused_inst.py batch-major > gpurun_out/r03_ab_$t.log 2>&1; done
for tune in 0011 0013 0010; do WHVI_FUSED_TUNE=$tune WHVI_HIP_LIB=whvi_amd/_exp/libwhvi_hip_tuning.so python tools/probe_fused_inst.py sample-major > gpurun_out/r03_ab_tune$tune.log 2>&1; done
for f in prod aux1 aux16 aux2 up8 up6 tune0011 tune0013 tune0010; do echo "== $f"; grep -E "^float" gpurun_out/r03_ab_$f.log | awk '{print $1, $2, $3, $4, $5, $6, $9, $12, $NF}' | cut -c1-150; done
