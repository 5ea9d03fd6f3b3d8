# Same-box A/B of an environment switch inside the whole training step: bench.py with VAR=0 / VAR=1,
# alternating.  Usage (on the GPU box): bash scripts/env_ab.sh SCL_FUSED_FIRST_WRW 3
R=${GRAFT_REPO_ROOT:-.}
VAR=${1:-SCL_FUSED_FIRST_WRW}
N=${2:-3}
for rep in $(seq 1 $N); do
  for v in 0 1; do
    env $VAR=$v python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-retrieval --no-batch-sweep --no-telemetry 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$VAR=$v  ms_per_step', d['ms_per_step'], ' median', d['ms_per_step_stats']['median'], ' p10', d['ms_per_step_stats']['p10'], ' images/s', d['value'], ' loss', d['config']['loss'])"
  done
done
