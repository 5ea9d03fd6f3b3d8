# Same-box A/B of two builds of the library: soft_contrastive_learning_amd/libscl_hip_base.so (a copy of an earlier
# build) against libscl_hip_new.so (a copy of the current one), alternating, same command.  Run under gpurun:
#   bash scripts/lib_ab.sh 3 python3 scripts/convh_variants.py --layer 4_2 --variants 53008
P=soft_contrastive_learning_amd
N=$1; shift
for k in $(seq 1 $N); do
  for which in base new; do
    cp $P/libscl_hip_$which.so $P/libscl_hip.so
    "$@" --tag $which 2>/dev/null
  done
done
cp $P/libscl_hip_new.so $P/libscl_hip.so
