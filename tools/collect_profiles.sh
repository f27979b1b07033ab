#!/bin/bash
# tools/collect_profiles.sh TAG PART -- ON THE GPU BOX: regenerate every rocprofv3 summary kept under profiles/TAG/ from the
# build in this tree, in one lease per part (PART = 1: headline + configs 3 and 5 with their HBM-traffic records; 2: the
# round-3 fused instantiations, config 2 / config 4 kernel-time shares, the weight-kernel traces, probe logs and the bench
# line; 3 (round 4): configs 2 and 4 on the shipped route and on the faithful dataflow, whvi_diag_apply's trace + PMC passes).  The program is always directly after `--`; counters are collected in passes of their own (one TCC counter per
# pass, MI355X_MICROARCH.md), never together with a trace domain.  Everything lands in gpurun_out/TAG_profiles/ (scratch,
# merged back by gpurun): copy it to profiles/TAG/ and profiles/hbm_traffic.json.
#     gpurun --timeout 1200 -- bash tools/collect_profiles.sh r03 1
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r03}; part=${2:-1}
out="$root/gpurun_out/${tag}_profiles"; raw="$root/gpurun_out/${tag}_raw"
mkdir -p "$out" "$raw"
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"
trace() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d "$raw/${name}_trace" -o "$name" -- python3 "$@" > "$raw/${name}_trace.log" 2>&1; echo "trace $name done"; }
pmc() { name=$1; ctr=$2; shift 2; rocprofv3 --pmc $ctr --output-format csv -d "$raw/${name}_$(echo $ctr | cut -d' ' -f1)" -o "$name" -- python3 "$@" > "$raw/${name}_$(echo $ctr | cut -d' ' -f1).log" 2>&1; echo "pmc $name $(echo $ctr | cut -d' ' -f1) done"; }
bench_args="--no-cpu-baseline --no-extras"
if [ "$part" = "1" ]; then
  # ---- headline: D = 4096 f32, 2^20 rows in place
  rocprofv3 --kernel-trace --stats --output-format csv -d "$raw/bench_trace" -o bench -- python3 "$root/bench.py" --steps 20 --warmup 5 $bench_args > "$out/bench_N1_under_profiler.json" 2> "$raw/bench_trace.log"; echo "trace bench done"
  pmc bench FETCH_SIZE "$root/bench.py" --steps 3 --warmup 1 $bench_args
  pmc bench WRITE_SIZE "$root/bench.py" --steps 3 --warmup 1 $bench_args
  pmc bench "$SQ" "$root/bench.py" --steps 3 --warmup 1 $bench_args
  # ---- config 3 (fused, D = 2048, 64 MC, batch 8192) and config 5 (fp16 / bf16 D = 4096, 2^20 rows)
  trace fused "$root/tools/profile_fused.py" 40
  pmc fused FETCH_SIZE "$root/tools/profile_fused.py" 4
  pmc fused WRITE_SIZE "$root/tools/profile_fused.py" 4
  pmc fused "$SQ" "$root/tools/profile_fused.py" 4
  trace f16 "$root/tools/profile_f16.py" 16
  pmc f16 FETCH_SIZE "$root/tools/profile_f16.py" 4
  pmc f16 WRITE_SIZE "$root/tools/profile_f16.py" 4
  cd "$root"
  python3 tools/filter_stats.py "$(find $raw/bench_trace -name "*kernel_stats.csv" | head -1)" "$out/bench_kernel_stats_noextras.csv"
  python3 tools/summarize_profile.py "$out/headline_rocprof_summary.csv" 34359738368 "$raw/bench_trace" "$raw/bench_FETCH_SIZE" "$raw/bench_WRITE_SIZE" "$raw/bench_SQ_WAVES"
  python3 tools/summarize_profile.py "$out/fused_config3_rocprof_summary.csv" 8589934592 "$raw/fused_trace" "$raw/fused_FETCH_SIZE" "$raw/fused_WRITE_SIZE" "$raw/fused_SQ_WAVES"
  python3 tools/summarize_profile.py "$out/f16_config5_rocprof_summary.csv" 17179869184 "$raw/f16_trace" "$raw/f16_FETCH_SIZE" "$raw/f16_WRITE_SIZE"
  python3 tools/update_hbm_traffic.py fwht_f32_D4096_rows1048576 "$raw/bench_FETCH_SIZE" "$raw/bench_WRITE_SIZE" "fwht_rows_kernel<float, 12" 34359738368 > "$raw/hbm_headline.json"
  python3 tools/update_hbm_traffic.py fused_shs_f32_D2048_S64_B8192 "$raw/fused_FETCH_SIZE" "$raw/fused_WRITE_SIZE" "fused_shs_kernel<float, 11" 8589934592 > "$raw/hbm_fused.json"
  python3 tools/update_hbm_traffic.py fwht_f16_D4096_rows1048576 "$raw/f16_FETCH_SIZE" "$raw/f16_WRITE_SIZE" "fwht_rows_kernel<__half, 12" 17179869184 > "$raw/hbm_f16.json"
  cp profiles/hbm_traffic.json "$out/hbm_traffic.json"
elif [ "$part" = "2" ]; then
  trace fused_inst "$root/tools/profile_fused_inst.py" 30
  pmc fused_inst FETCH_SIZE "$root/tools/profile_fused_inst.py" 4
  pmc fused_inst WRITE_SIZE "$root/tools/profile_fused_inst.py" 4
  pmc fused_inst "$SQ" "$root/tools/profile_fused_inst.py" 4
  trace config2 "$root/tools/profile_config2.py"
  trace config4 "$root/tools/profile_config4.py"
  trace wbar_bwd "$root/tools/profile_wbar_bwd.py" 10
  trace wbar_fwd "$root/tools/profile_wbar_fwd.py"
  trace train_graph "$root/tools/profile_train_graph.py" 300
  cd "$root"
  python3 tools/summarize_profile.py "$out/fused_instantiations_rocprof_summary.csv" 8589934592 "$raw/fused_inst_trace" "$raw/fused_inst_FETCH_SIZE" "$raw/fused_inst_WRITE_SIZE" "$raw/fused_inst_SQ_WAVES"
  python3 tools/filter_stats.py "$(find $raw/config2_trace -name "*kernel_stats.csv" | head -1)" "$out/config2_kernel_time_shares.csv"
  python3 tools/filter_stats.py "$(find $raw/config4_trace -name "*kernel_stats.csv" | head -1)" "$out/config4_train_step_kernel_shares.csv"
  python3 tools/filter_stats.py "$(find $raw/train_graph_trace -name "*kernel_stats.csv" | head -1)" "$out/train_graph_kernel_shares.csv"
  find "$raw/train_graph_trace" -name "*kernel_trace.csv" -delete     # 20 k rows: only the per-kernel statistics are kept
  python3 tools/summarize_profile.py "$out/wbar_bwd_kernel_trace_summary.csv" 0 "$raw/wbar_bwd_trace"
  python3 tools/summarize_profile.py "$out/wbar_fwd_kernel_trace_summary.csv" 0 "$raw/wbar_fwd_trace"
  cp "$raw/wbar_bwd_trace.log" "$out/wbar_bwd_shapes.log"; cp "$raw/wbar_fwd_trace.log" "$out/wbar_fwd_shapes.log"; cp "$raw/fused_inst_trace.log" "$out/fused_instantiations_symbols.log"
  python3 tools/probe_fused_inst.py batch-major,sample-major > "$out/fused_instantiations_hip_events.log" 2>&1
  python3 tools/probe_wbar_mean.py > "$out/wbar_mean_one_vs_two_launches.log" 2>&1
  python3 tools/config4_train_step.py 2> /dev/null | tail -1 > "$out/config4_train_step_recipe.json"
  python3 tools/probe_wbar_fwd_stream.py > "$out/wbar_fwd_stream_hip_events.log" 2>&1
  python3 bench.py > "$out/bench_N1.json" 2> "$raw/bench_N1.err"
fi
if [ "$part" = "3" ]; then
  # ---- round 4: the shipped route of configs 2 and 4 (whvi_diag_apply) beside the faithful dataflow, and the new kernel's traffic
  trace config2 "$root/tools/profile_config2.py"
  trace config2_faithful "$root/tools/profile_config2.py" faithful
  trace config4_predict "$root/tools/profile_config4.py" predict
  trace config4_predict_faithful "$root/tools/profile_config4.py" predict faithful
  trace diag_apply "$root/tools/profile_diag_apply.py" 20
  pmc diag_apply FETCH_SIZE "$root/tools/profile_diag_apply.py" 4
  pmc diag_apply WRITE_SIZE "$root/tools/profile_diag_apply.py" 4
  cd "$root"
  for n in config2 config2_faithful config4_predict config4_predict_faithful; do
    python3 tools/filter_stats.py "$(find $raw/${n}_trace -name "*kernel_stats.csv" | head -1)" "$out/${n}_kernel_time_shares.csv"
    cp "$raw/${n}_trace.log" "$out/${n}_timing.log"
  done
  python3 tools/summarize_profile.py "$out/diag_apply_config4_rocprof_summary.csv" 0 "$raw/diag_apply_trace" "$raw/diag_apply_FETCH_SIZE" "$raw/diag_apply_WRITE_SIZE"
  python3 tools/diag_apply_rate.py > "$out/diag_apply_rates.log" 2>&1
  python3 tools/stream_forms_r04.py > "$out/stream_forms_final_build.log" 2>&1
  python3 tools/config4_train_step.py 2> /dev/null | tail -1 > "$out/config4_train_step_recipe.json"
fi
ls -la "$out"
