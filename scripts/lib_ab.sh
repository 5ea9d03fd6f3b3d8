# Same-box A/B of the train step between the product library and a PREVIOUS build of it: put the other build at
# soft_contrastive_learning_amd/libscl_hip_prev.so (e.g. `git show HEAD~1:...conv64.hip`, compiled with the
# Makefile's flags and linked with the current objects), then `gpurun -- 'bash scripts/lib_ab.sh'`.
cd $GRAFT_REPO_ROOT
P=soft_contrastive_learning_amd
cp $P/libscl_hip.so /tmp/new.so; cp $P/libscl_hip_prev.so /tmp/prev.so
for r in 1 2 3; do
  for v in new prev; do
    cp /tmp/$v.so $P/libscl_hip.so
    echo -n "$v  "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-batch-sweep --no-telemetry 2>/dev/null | python scripts/bench_field.py value ms_per_step
  done
done
cp /tmp/new.so $P/libscl_hip.so
