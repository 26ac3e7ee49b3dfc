cd $GRAFT_REPO_ROOT
./tools/micro/isort_gtime 195000 1024 0 | tail -1 | cut -c1-330
./tools/micro/isort_gtime 120000 1024 1 | tail -1 | cut -c1-330
