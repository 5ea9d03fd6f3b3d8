# Same-box A/B of the static-priority experiment (s_setprio 1 for waves 4..7 of the eight-wave
# convolution kernels; MI355X_MICROARCH.md "Two waves per SIMD", item 4).  Both arms pin the kernel
# through scl_debug_set_variant (which also turns the packed-weights launch off), so they differ in
# the priority only.  Usage (on the GPU box): bash scripts/setprio_ab.sh > gpurun_out/setprio_ab.txt
R=${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for v in 50000 53040 60000 60004; do
    python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-batch-sweep --variant $v 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('variant $v  ms_per_step', d['ms_per_step'], ' median', d['ms_per_step_stats']['median'], ' convh', [k['us'] for k in d['kernels'] if k['kernel']=='convh_kernel'], ' conv3x3', [k['us'] for k in d['kernels'] if k['kernel'].startswith('conv3x3_kernel')])"
  done
done
