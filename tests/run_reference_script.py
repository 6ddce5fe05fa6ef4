"""TEST-ONLY launcher: runs one of the reference's scripts UNMODIFIED (read from /root/reference at run time,
never copied) against this repo's `dgl` package.  Without a GPU the script falls back to CPU tensors
(e.g. main_dgl_product_sage.py:149), for which the test-only oracle backend is registered here, so what the
run checks is the API surface / plumbing of BASELINE config 0, not the HIP kernels.

  python tests/run_reference_script.py <script.py> [script args...]
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(HERE, "shims"), os.path.join(ROOT, "dgl-0.5-benchmark_amd"), ROOT, HERE):
    sys.path.insert(0, p)

import torch  # noqa: E402

if not torch.cuda.is_available() and os.environ.get("MGX_CPU_BACKEND", "0") != "1":
    import oracle_backend  # noqa: E402
    oracle_backend.install()
# (MGX_CPU_BACKEND=1: `import dgl` registers the PRODUCT's CPU (OpenMP) variants -- csrc/cpu_ops.cpp -- and the oracle stays out)

script = os.path.abspath(sys.argv[1])
sys.argv = [script] + sys.argv[2:]
sys.path.insert(0, os.path.dirname(script))  # the scripts import their sibling utils.py
runpy.run_path(script, run_name="__main__")
