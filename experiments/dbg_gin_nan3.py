"""First non-finite step of the graphed GIN loop: snapshot parameters before every step; at the failing step re-run the
same padded batch EAGERLY from the snapshot with finiteness checks on every module output."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "dgl-0.5-benchmark_amd"))
import torch, torch.nn as nn
torch.set_num_threads(4)
import graph_classification as gc
from mi355x_graph.datasets import molhiv_like
from dgl.dataloading import GraphDataLoader
dev = torch.device("cuda:0")
data = molhiv_like(32901)
loader = GraphDataLoader(data, batch_size=256, shuffle=True, num_workers=0)
torch.manual_seed(0)
model = gc.convert_masked_batchnorm(gc.GIN(256, 1, 5, 0.5).to(dev))
opt = torch.optim.Adam(model.parameters(), lr=0.001, capturable=True)
n_pad, e_pad = gc.GraphedBatchTrainer.static_shape(data, 256)
tr = gc.GraphedBatchTrainer(model, opt, nn.BCEWithLogitsLoss(), dev, 256, n_pad, e_pad)
model.train()
def finite_params():
    return all(bool(torch.isfinite(p).all()) for p in model.parameters()) and all(bool(torch.isfinite(b).all()) for b in model.buffers())
for ep in range(1, 4):
    for i, (bg, lab) in enumerate(loader):
        snap = [p.detach().clone() for p in model.parameters()]
        bsnap = [b.detach().clone() for b in model.buffers()]
        l = tr.step(bg, lab)
        torch.cuda.current_stream().synchronize()
        ok = finite_params() and bool(torch.isfinite(l))
        if not ok:
            print("first non-finite: epoch", ep, "step", i, "n", bg.number_of_nodes(), "loss", float(l))
            gnan = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
            print("non-finite params:", [n for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())][:6])
            st = [(n, [k for k, v in opt.state[p].items() if torch.is_tensor(v) and not bool(torch.isfinite(v).all())]) for n, p in model.named_parameters()]
            print("non-finite adam state:", [s for s in st if s[1]][:6])
            with torch.no_grad():
                for p, s in zip(model.parameters(), snap): p.copy_(s)
                for b, s in zip(model.buffers(), bsnap): b.copy_(s)
            hooks = []
            def mk(name):
                def hook(mod, inp, out):
                    o = out if torch.is_tensor(out) else out[0]
                    if not bool(torch.isfinite(o[:bg.number_of_nodes()] if o.shape[0] == n_pad else o).all()):
                        print("   non-finite REAL rows out of", name)
                    elif not bool(torch.isfinite(o).all()):
                        print("   non-finite ghost rows out of", name)
                return hook
            for name, mod in model.named_modules():
                if name: hooks.append(mod.register_forward_hook(mk(name)))
            loss = tr._forward_loss(tr.buf)
            print("eager replay of the failing step from the snapshot: loss", float(loss))
            model.zero_grad(set_to_none=True)
            loss.backward()
            print("   non-finite grads:", [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())][:8])
            print("   max |grad|:", max(float(p.grad.abs().max()) for p in model.parameters() if p.grad is not None))
            sys.exit(0)
    l.item(); torch.cuda.synchronize()
    print("epoch", ep, "ok", flush=True)
