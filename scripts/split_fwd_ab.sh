# Same-box A/B of the forward pass as two half-batches on two streams (nets.USE_SPLIT_FWD) inside the
# whole training step: bench.py --split-fwd 0 / 1, alternating.  Usage (on the GPU box):
#   bash scripts/split_fwd_ab.sh 3 > gpurun_out/split_fwd_ab.txt
R=${GRAFT_REPO_ROOT:-.}
N=${1:-3}
for rep in $(seq 1 $N); do
  for v in 0 1; do
    python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-retrieval --no-batch-sweep --no-telemetry --split-fwd $v 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('split_fwd $v  ms_per_step', d['ms_per_step'], ' median', d['ms_per_step_stats']['median'], ' p10', d['ms_per_step_stats']['p10'], ' images/s', d['value'])"
  done
done
