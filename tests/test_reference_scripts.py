"""CPU suite (runs only where /root/reference exists): the reference's own training scripts executed
UNMODIFIED on this repo's `dgl` package -- the plumbing check of BASELINE config 0 ("2-layer GraphSAGE on cora
... runs without a GPU").  Arithmetic on CPU tensors goes through the test-only oracle backend."""
import os
import re
import subprocess
import sys

import pytest

REF = "/root/reference/end_to_end/full_graph"
REF_SAMPLING = "/root/reference/end_to_end/sampling/node-classification"
HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def run(script, *args, root=REF, **extra_env):
    env = dict(os.environ, OMP_NUM_THREADS="4", **extra_env)
    p = subprocess.run([sys.executable, os.path.join(HERE, "run_reference_script.py"), os.path.join(root, script)] + list(args),
                       capture_output=True, text=True, timeout=600, env=env, cwd=HERE)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return p.stdout


SCRIPTS = [
    ("node_classification/main_dgl_citation_sage.py", ["--dataset", "cora", "--epochs", "5", "--runs", "1"]),
    ("node_classification/main_dgl_citation_sage.py", ["--dataset", "cora", "--epochs", "5", "--runs", "1", "--aggr", "sum", "--eval"]),
    ("node_classification/main_dgl_citation_gat.py", ["--dataset", "cora", "--epochs", "5", "--runs", "1"]),
    ("node_classification/main_dgl_product_sage.py", ["--epochs", "5", "--runs", "1", "--eval"]),
    ("node_classification/main_dgl_arxiv_sage.py", ["--epochs", "5", "--runs", "1"]),
    ("node_classification/main_dgl_arxiv_gat.py", ["--epochs", "5", "--runs", "1"]),
    ("node_classification/main_dgl_arxiv_sage_nn.py", ["--epochs", "5", "--runs", "1"]),
    ("node_classification/main_dgl_proteins_rgcn_for.py", ["--epochs", "4", "--runs", "1"]),
    ("graph_classification/main_dgl_molhiv_gcn.py", ["--epochs", "2", "--runs", "1", "--batch_size", "16", "--num_workers", "0"]),
]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("script,args", SCRIPTS, ids=[s[0].split("/")[-1] + ("-" + s[1][-1].strip("-") if s[1][-1].startswith("--") else "") for s in SCRIPTS])
def test_reference_script_runs_unmodified(script, args):
    out = run(script, *args)
    times = re.findall(r"Training time/epoch ([0-9.eE+-]+)", out)
    if "molhiv" not in script:
        assert times, out[-1500:]


@pytest.mark.timeout(900)
def test_neighbor_sampling_script_runs_unmodified():
    """reddit/ns-sage-dgl.py (MultiLayerNeighborSampler + NodeDataLoader + dglnn.SAGEConv on
    blocks, full-neighbour inference) on a 1 %-scale reddit stand-in."""
    out = run("reddit/ns-sage-dgl.py", "--gpu", "-1", "--num-epochs", "7", "--num-workers", "0", "--eval-every", "5",
              "--batch-size", "256", root=REF_SAMPLING, MGX_DATASET_SCALE="0.01")
    assert "Avg epoch time" in out and "Test Acc" in out, out[-1500:]
