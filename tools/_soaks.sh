cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06v2
for S in "soak_random 480" "soak_shards 420" "soak_mirror 360" "soak_knn 360"; do
  set -- $S
  timeout $(( $2 + 120 )) python tools/$1.py --seconds $2 --seed 9941 > gpurun_out/r06v2/$1.log 2>&1
  echo "$1: $(tail -1 gpurun_out/r06v2/$1.log)"
  grep -c "^FAIL" gpurun_out/r06v2/$1.log
done
