set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_n1_default_run.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_nv/sq -- python3 $R/scripts/microbench.py --what netvlad,loss --iters 3 --loss-batches 24,192 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_nv/fetch -- python3 $R/scripts/microbench.py --what netvlad,loss --iters 3 --loss-batches 24,192 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_nv/write -- python3 $R/scripts/microbench.py --what netvlad,loss --iters 3 --loss-batches 24,192 > /dev/null 2>&1
cd $R
python3 scripts/microbench.py --iters 20 --topn-score f32,bf16x3 --json $O/microbench_netvlad_loss_topn.json > $O/microbench.log 2>&1
python3 scripts/parity_report.py --json $O/parity_report.json > $O/parity.log 2>&1
python3 scripts/trace_summary.py $O/trace --steps 8 --out $O/bench_n1_steady_state_per_step.csv > $O/trace_summary.log 2>&1
python3 scripts/pmc_summary.py $O/pmc_nv --only kernel --out $O/pmc_netvlad_loss_b24_n1200.csv > /dev/null 2>&1
python3 scripts/kstats.py $(ls $O/trace/*/*kernel_trace.csv | head -1) > $O/bench_kernel_trace_by_shape.txt
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/bench_n1_kernel_stats_rocprofv3.csv
rm -rf $O/trace $O/pmc_nv
tail -3 $O/parity.log; tail -5 $O/trace_summary.log; head -c 600 $O/bench_n1_default_run.json
