"""Result driver: runs the launchers of this package one after another, each in its own process, and tabulates
epoch time / final accuracies -- the counterpart of the reference's generate_result.py
(end_to_end/full_graph/node_classification/generate_result.py:29-63, graph_classification/generate_result.py).

Parsing rule kept from the reference (`parse_results`, :29-44): every printed `Training time/epoch X` line is collected and
the LAST TEN are averaged; `Final Train:` / `Final Test:` lines give the accuracies.  The launchers here print one
steady-state mean per run (the reference prints a running mean every epoch), so the table holds that mean.

  python dgl-0.5-benchmark_amd/generate_result.py [--only sage_products,gat_reddit] [--out r.csv]
Writes JSON + a markdown table to stdout and a CSV file, as the reference driver does.
"""
import argparse
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

RUNS = [  # name, launcher, arguments (reference script each one stands for)
    ("sage_cora", "full_graph.py", ["--model", "sage", "--dataset", "cora", "--epochs", "50"]),          # main_dgl_citation_sage.py
    ("sage_pubmed", "full_graph.py", ["--model", "sage", "--dataset", "pubmed", "--epochs", "50"]),
    ("gat_cora", "full_graph.py", ["--model", "gat", "--dataset", "cora", "--epochs", "50"]),            # main_dgl_citation_gat.py
    ("sage_arxiv", "full_graph.py", ["--model", "sage", "--dataset", "arxiv", "--epochs", "30"]),        # main_dgl_arxiv_sage.py
    ("sage_arxiv_hipgraph", "full_graph.py", ["--model", "sage", "--dataset", "arxiv", "--epochs", "30", "--hipgraph"]),  # captured step
    ("gat_arxiv", "full_graph.py", ["--model", "gat", "--dataset", "arxiv", "--epochs", "30"]),          # main_dgl_arxiv_gat.py
    ("sage_reddit", "full_graph.py", ["--model", "sage", "--dataset", "reddit", "--epochs", "20"]),      # main_dgl_reddit_sage.py
    ("gat_reddit", "full_graph.py", ["--model", "gat", "--dataset", "reddit", "--heads", "1", "--num-layers", "3",
                                     "--num-hidden", "16", "--epochs", "20"]),                            # main_dgl_reddit_gat.py
    ("sage_products", "full_graph.py", ["--model", "sage", "--dataset", "products", "--epochs", "20"]),  # main_dgl_product_sage.py
    ("gat8_reddit_small", "full_graph.py", ["--model", "gat", "--dataset", "reddit-small", "--heads", "8", "--num-layers", "2",
                                            "--epochs", "20"]),                                          # BASELINE config 3 (ns-gat shape)
    ("gat8_reddit_small_hipgraph", "full_graph.py", ["--model", "gat", "--dataset", "reddit-small", "--heads", "8", "--num-layers", "2",
                                                     "--epochs", "20", "--hipgraph"]),                   # same step, one captured HIP graph
    ("gcn_molhiv", "graph_classification.py", ["--epochs", "3"]),                                         # main_dgl_molhiv_gcn.py
    ("gcn_molhiv_hipgraph", "graph_classification.py", ["--epochs", "4", "--hipgraph"]),                  # same loop, one captured HIP graph
    ("gin_molhiv", "graph_classification.py", ["--model", "gin", "--epochs", "3"]),                       # BASELINE config 5 wording
    ("gin_molhiv_hipgraph", "graph_classification.py", ["--model", "gin", "--epochs", "4", "--hipgraph"]),  # same loop, one captured HIP graph
    ("ns_sage_reddit", "sampling_sage.py", ["--num-epochs", "8"]),                                        # reddit/ns-sage-dgl.py
    ("gcmc_ml-1m", "link_prediction.py", ["--data_name", "ml-1m", "--train_max_iter", "30"]),             # gcmc_dgl/train.py
]


def parse_results(output):
    """Same rule as the reference driver: mean of the last ten `Training time/epoch` values, last Final Train/Test."""
    times, train_acc, test_acc = [], "", ""
    for line in output.split("\n"):
        line = line.strip()
        if line.startswith("Training time/epoch"):
            times.append(float(line.split(" ")[-1]))
        if line.startswith("Final Train"):
            train_acc = line.split(":")[-1]
        if line.startswith("Final Test"):
            test_acc = line.split(":")[-1]
    tail = times[-10:]
    return {"epoch_time": sum(tail) / len(tail) if tail else float("nan"), "final_train_acc": train_acc,
            "final_test_acc": test_acc}


def to_markdown(table):
    cols = ["epoch_time", "final_train_acc", "final_test_acc"]
    rows = ["| run | " + " | ".join(cols) + " |", "|---|" + "---|" * len(cols)]
    for name, rec in table.items():
        rows.append("| %s | " % name + " | ".join(str(rec[c]) for c in cols) + " |")
    return "\n".join(rows)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--only", default="", help="comma-separated run names")
    p.add_argument("--out", default="r.csv")
    p.add_argument("--extra", default="", help="extra arguments appended to every launcher (e.g. '--scale 0.01')")
    args = p.parse_args()
    only = set(x for x in args.only.split(",") if x)
    table = {}
    for name, launcher, argv in RUNS:
        if only and name not in only:
            continue
        cmd = [sys.executable, os.path.join(HERE, launcher)] + argv + args.extra.split()
        print("Run %s: %s" % (name, " ".join(cmd[1:])), flush=True)
        proc = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=HERE))
        if proc.returncode != 0:
            print("Failed to run %s\n%s" % (name, proc.stderr[-1500:]))
            continue
        table[name] = parse_results(proc.stdout)
    print(json.dumps(table))
    print(to_markdown(table))
    with open(args.out, "w") as f:
        f.write(",epoch_time,final_train_acc,final_test_acc\n")
        for name, rec in table.items():
            f.write("%s,%s,%s,%s\n" % (name, rec["epoch_time"], rec["final_train_acc"], rec["final_test_acc"]))
    return 0 if table else 1


if __name__ == "__main__":
    sys.exit(main())
