cd $GRAFT_REPO_ROOT
./tools/micro/isort_gtime 195000 256 0 | tail -1
./tools/micro/isort_gtime 195000 1024 0 | tail -1
./tools/micro/isort_gtime 100000 256 1 | tail -1
