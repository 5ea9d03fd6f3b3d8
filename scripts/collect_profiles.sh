# Regenerates everything under profiles/rNN from one GPU box (run through gpurun; results land in
# gpurun_out/rNN and are copied into profiles/rNN by hand).  Usage: bash scripts/collect_profiles.sh r03
set -x
RN=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$RN; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_n1_default_run.json 2> $O/bench.err
python3 $R/bench.py --steps 20 --warmup 5 --side-wrw 0 --no-cpu-baseline > $O/bench_n1_one_stream.json 2>> $O/bench.err
# kernel trace of the default command, and of the one-stream run (the durations of `roofline`
# are taken with the weight-gradient kernels serialised: compare with the second table)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-batch-sweep --no-retrieval --no-telemetry > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-batch-sweep --no-retrieval --no-telemetry --side-wrw 0 > $O/bench_under_rocprof_one_stream.json 2>/dev/null
# HBM traffic of the backbone kernels (separate passes per counter, no tracing)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_bb/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch-sweep --no-retrieval --no-telemetry --side-wrw 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_bb/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch-sweep --no-retrieval --no-telemetry --side-wrw 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_bb/sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch-sweep --no-retrieval --no-telemetry --side-wrw 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_nv/sq -- python3 $R/scripts/microbench.py --what netvlad,loss --iters 3 --loss-batches 24,192 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_nv/fetch -- python3 $R/scripts/microbench.py --what netvlad,loss --iters 3 --loss-batches 24,192 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_nv/write -- python3 $R/scripts/microbench.py --what netvlad,loss --iters 3 --loss-batches 24,192 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_nv -- python3 $R/scripts/microbench.py --what netvlad,loss --iters 20 --loss-batches 24,192 --netvlad-variants 920 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_tn/sq -- python3 $R/scripts/microbench.py --what topn --iters 20 --topn-score f32,bf16x3 > /dev/null 2>&1
cd $R
cp $(ls $O/trace_nv/*/*kernel_stats.csv | head -1) $O/microbench_netvlad_loss_kernel_stats_rocprofv3.csv
python3 scripts/pmc_summary.py $O/pmc_tn --only kernel --out $O/pmc_topn_sq_counters.csv > /dev/null 2>&1
python3 scripts/microbench.py --iters 20 --topn-score f32,bf16x3 --json $O/microbench_netvlad_loss_topn.json > $O/microbench.log 2>&1
python3 tests/tools/parity_report.py --json $O/parity_report.json > $O/parity.log 2>&1
python3 bench.py --workload retrieval --steps 5 --warmup 2 > $O/bench_retrieval_n1_f32.json 2>> $O/bench.err
python3 bench.py --workload retrieval --steps 5 --warmup 2 --score bf16x3 > $O/bench_retrieval_n1_bf16x3.json 2>> $O/bench.err
python3 scripts/vlad_stamps.py --kernel fwd > $O/vlad_stamps_fwd.txt 2>/dev/null
python3 scripts/vlad_stamps.py --kernel fwd8 > $O/vlad_stamps_fwd8.txt 2>/dev/null
python3 scripts/vlad_stamps.py --kernel dx > $O/vlad_stamps_dx.txt 2>/dev/null
python3 scripts/null_bracket.py > $O/null_bracket_events.txt 2>/dev/null
python3 tests/tools/netvlad_accuracy_probe.py > $O/netvlad_two_plane_accuracy.txt 2>/dev/null
python3 scripts/lds_conflicts.py > $O/lds_conflicts.txt 2>/dev/null
python3 scripts/conv_ab.py --rounds 2 > $O/conv_lds_kernels_32x32x16_vs_16x16x32.jsonl 2>/dev/null
python3 scripts/conv_layers.py > $O/conv_layers_own_vs_library.txt 2>/dev/null
# round 5: wrw64 staging A/B, the fused first-layer gradients (ablations + stamps, step A/B)
python3 scripts/wrw_ab.py --variants 0,2200 --rounds 2 > $O/wrw64_buffer_path_vs_round4_staging_same_box.txt 2>/dev/null
python3 scripts/first_wrw_fused_ablate.py --stamps > $O/first_wrw_fused_ablations_and_stamps.txt 2>/dev/null
bash scripts/env_ab.sh SCL_FUSED_FIRST_WRW 2 > $O/fused_first_wrw_step_ab.txt 2>/dev/null
# round 6: the one-rank RCCL step, the loss forward forms + stamps, NetVLAD inference sweep, retrieval forms
python3 bench.py --steps 20 --warmup 5 --force-dist --no-cpu-baseline --no-retrieval --no-batch-sweep > $O/bench_n1_force_dist_rccl.json 2>> $O/bench.err
python3 scripts/force_dist_ab.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $O/force_dist_same_process_ab.txt
python3 scripts/loss_fwd_ab.py 2>/dev/null | grep "^{" > $O/loss_fwd_forms_ab.jsonl
python3 scripts/loss_stamps.py --batch 192 2>/dev/null > $O/loss_stamps_b192.txt
python3 scripts/loss_stamps.py --batch 48 2>/dev/null > $O/loss_stamps_b48.txt
python3 scripts/vlad_inference_sweep.py 2>/dev/null | grep "^{" > $O/netvlad_inference_sweep.jsonl
python3 scripts/topn_kernels_ab.py --variants 1000,9001 2>/dev/null | grep "^{" > $O/topn_kernels_ab.jsonl
hipcc --offload-arch=gfx950 -O3 scripts/grid_barrier_probe.hip -o /tmp/gbp 2>/dev/null && /tmp/gbp > $O/grid_barrier_probe.txt 2>/dev/null
python3 scripts/trace_summary.py $O/trace --steps 8 --out $O/bench_n1_steady_state_per_step.csv > $O/trace_summary.log 2>&1
python3 scripts/trace_summary.py $O/trace1 --steps 8 --out $O/bench_n1_one_stream_per_step.csv >> $O/trace_summary.log 2>&1
python3 scripts/pmc_summary.py $O/pmc_nv --only kernel --out $O/pmc_netvlad_loss_b24_n1200.csv > /dev/null 2>&1
python3 scripts/pmc_summary.py $O/pmc_bb --only kernel --merge-templates --out $O/pmc_bench_backbone_b24_640x480.csv > /dev/null 2>&1
python3 scripts/kstats.py $(ls $O/trace/*/*kernel_trace.csv | head -1) > $O/bench_kernel_trace_by_shape.txt
python3 scripts/kstats.py $(ls $O/trace1/*/*kernel_trace.csv | head -1) > $O/bench_one_stream_kernel_trace_by_shape.txt
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/bench_n1_kernel_stats_rocprofv3.csv
cp $(ls $O/trace1/*/*kernel_stats.csv | head -1) $O/bench_n1_one_stream_kernel_stats_rocprofv3.csv
rm -rf $O/trace $O/trace1 $O/pmc_nv $O/pmc_bb $O/trace_nv $O/pmc_tn
tail -3 $O/parity.log; tail -6 $O/trace_summary.log; head -c 600 $O/bench_n1_default_run.json
