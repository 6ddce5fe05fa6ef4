cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p0 -- python3 $GRAFT_REPO_ROOT/dgl-0.5-benchmark_amd/kernel_controls.py --graphs products --widths 64,100 --reps 3 --scale 0.05 > /tmp/p0.log 2>&1; echo rc=$?
tail -5 /tmp/p0.log
find /tmp/p0 -type f | head
f=$(find /tmp/p0 -name "*counter_collection.csv" | head -1)
head -c 600 $f; echo
grep -c spmm $f; grep -c gather_rows $f
python3 - <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT/dgl-0.5-benchmark_amd")
import kernel_controls as kc
b = kc.parse_pmc_dir("/tmp/p0")
print(b)
print(kc.pmc_to_traffic([b], 3))
PY
