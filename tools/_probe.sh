cd $GRAFT_REPO_ROOT
for m in 0 1; do ./tools/micro/isort_time256 5888 300 8192 $m 1 | tail -2 | cut -c1-420; done
