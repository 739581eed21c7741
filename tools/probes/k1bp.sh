cd $GRAFT_REPO_ROOT
for bp in 1024 512 256; do
  echo "== NDT_K1_BUCKET_POINTS=$bp"
  NDT_K1_BUCKET_POINTS=$bp bash tools/prof_k1.sh 1e6 0 1.0 new 2>&1 | tail -5
  NDT_K1_BUCKET_POINTS=$bp bash tools/prof_k1.sh 1e6 100 1.0 new 2>&1 | grep finalize
done
