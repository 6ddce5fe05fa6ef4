"""PyTorch TunableOp selections for the dense layers of the products model -- a user-level PyTorch setting, not part of
the message-passing library.

hipBLASLt's default heuristics pick ~3 ms kernels for the tall-skinny weight-gradient GEMMs (K = 2.45 M rows).  The
selections in `tunableop_products.csv` were recorded once on MI355X by experiments/tune_dense.py; they are only LOADED
(tuning stays off).  PyTorch looks for `<name><device ordinal>.csv`, so the ONE committed file is staged under a
per-user temp directory once per visible ordinal.  PyTorch silently ignores a file whose `Validator` lines (PyTorch /
HIP / hipBLASLt / rocBLAS versions, GCN arch) differ from the running build: `status()` reports whether they match, so
a bench line says when the selections were rejected instead of just running slower.
"""
import os
import shutil
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE = os.path.join(HERE, "tunableop_products.csv")


def setup(max_devices=8):
    """Call BEFORE the first GEMM (safest: before importing torch).  MGX_BENCH_TUNABLEOP=0 turns it off."""
    if os.environ.get("MGX_BENCH_TUNABLEOP", "1") != "1" or not os.path.exists(SOURCE):
        return None
    stage = os.path.join(tempfile.gettempdir(), "mgx_tunableop_%d" % os.getuid())
    try:
        os.makedirs(stage, exist_ok=True)
        for d in range(max_devices):
            dst = os.path.join(stage, "tunableop_products%d.csv" % d)
            if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(SOURCE):
                tmp = dst + ".tmp%d" % os.getpid()
                shutil.copyfile(SOURCE, tmp)
                os.replace(tmp, dst)  # atomic: several ranks stage at once
    except OSError:
        return None
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "0")
    os.environ.setdefault("PYTORCH_TUNABLEOP_RECORD_UNTUNED", "0")
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(stage, "tunableop_products.csv"))
    return stage


def file_validators(path=SOURCE):
    out = {}
    with open(path) as f:
        for line in f:
            parts = line.strip().split(",")
            if len(parts) >= 3 and parts[0] == "Validator":
                out[parts[1]] = ",".join(parts[2:])
    return out


def status():
    """After the device is initialised: {"enabled", "accepted", "mismatch"}.  accepted=None when it cannot be told."""
    enabled = os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1"
    st = {"enabled": enabled, "accepted": None, "mismatch": []}
    if not enabled or not os.path.exists(SOURCE):
        return st
    try:
        import torch
        running = {k: v for k, v in torch.cuda.tunable.get_validators()}
        want = file_validators()
        st["mismatch"] = ["%s: file %s, running %s" % (k, v, running.get(k)) for k, v in want.items() if running.get(k) != v]
        st["accepted"] = not st["mismatch"]
    except Exception as err:  # an API difference must not lose the bench line
        st["error"] = str(err)[:120]
    return st
