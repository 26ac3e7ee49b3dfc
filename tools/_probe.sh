cd $GRAFT_REPO_ROOT
for m in 0 1; do ./tools/micro/isort_time256 5888 300 1024 $m 1 | tail -2 | cut -c1-400; done
./tools/micro/isort_time256 5888 300 4096 1 1 | tail -2 | cut -c1-400
