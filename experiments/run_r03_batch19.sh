cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_tile_spmm.py -q -m gpu -x --tb=short 2>&1 | tail -15
