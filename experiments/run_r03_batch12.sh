cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp
bash profiles/collect.sh r03 2>&1 | tail -5
cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
