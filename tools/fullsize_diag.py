"""Per-component gradient error of the full-width net against the oracle (diagnostic for tests/test_gpu_fullsize.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
from tests.oracle_net import OracleNet, component_table
from tests.gpu_util import dev, host, rel_l2
kw = dict(a.split("=") for a in sys.argv[1:])
B = int(kw.pop("B", 8)); ng = int(kw.pop("ng", 0)); rep = float(kw.pop("repair", 1e-5)); osd = float(kw.pop("osd", 0.05)); prec = int(kw.pop("prec", 0)); steps = int(kw.pop("steps", 1))
cfg = pkg.trainer.make_config(frames_per_chunk=150, num_sequences=B, use_natural_gradient=ng, relu_self_repair_scale=rep, gemm_precision=prec)
net = pkg.trainer.ChainNet(cfg)
table, n = component_table(cfg)
params = net.init_params_numpy(seed=0, output_stddev=osd)
net.set_params(params)
ref = OracleNet(pkg, cfg, table)
ref.relu_tie_tol = float(kw.pop("tietol", 1e-4))
feats, iv = pkg.trainer.synthetic_egs(net, seed=100)
den = pkg.synth.make_den_graph(4000, cfg.num_pdfs, mean_out_degree=12.0, seed=1)
sup = pkg.synth.make_supervision_from_den(den, B, 50, num_paths=2, seed=200)
net.grads.zero_()
net.set_capture(True)
r = host(net.forward_backward(dev(feats), dev(iv), pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup), step=0)).copy()
relu_names = ["tdnn1.relu"] + ["tdnnf%d.relu" % (l + 2) for l in range(cfg.num_layers)] + ["prefinal-chain.relu", "prefinal-xent.relu"]
relus = {k: host(net.activation(k)) for k in relu_names} if int(kw.pop("ties", 1)) else None
res_ref, g_ref, acts = ref.forward_backward(params, feats, iv, den, sup, step=0, relu_like=relus)
print("relu ties", ref.relu_ties)
g = host(net.grads)
print("objf", r[0], res_ref["objf"], "total", rel_l2(g, g_ref))
for k in ["tdnn1.batchnorm", "tdnnf2.relu", "tdnnf8.noop", "tdnnf15.noop", "output", "output.deriv"]:
    print(k, rel_l2(host(net.activation(k)), acts[k]))
names = ["output-xent.deriv"]
for hn in ("xent", "chain"):
    names += ["prefinal-%s.%s.deriv" % (hn, k) for k in ("batchnorm2", "linear", "batchnorm1", "affine")]
names += ["prefinal-l.deriv"]
for l in range(cfg.num_layers + 1, 1, -1):
    names += ["tdnnf%d.%s.deriv" % (l, k) for k in ("noop", "affine", "linear")]
for k in names:
    a, b = host(net.activation(k)), acts[k]
    print("%-34s %.2e   |ref| %.3e  max|ref| %.2e" % (k, rel_l2(a, b), np.linalg.norm(b), np.abs(b).max()))
for c in net.components[1:]:
    nW = c["rows"] * c["cols"]
    w = slice(c["begin"], c["begin"] + nW)
    b = slice(c["begin"] + nW + c["num_alpha"], c["begin"] + nW + c["num_alpha"] + (c["rows"] if c["has_bias"] else 0))
    eb = rel_l2(g[b], g_ref[b]) if c["has_bias"] else 0.0
    print("%-24s W %.2e (|g| %.3e)  bias %.2e" % (c["name"], rel_l2(g[w], g_ref[w]), np.linalg.norm(g_ref[w]), eb))
# ---- column-level look at the first stage where the error appears
k = kw.get("col", "prefinal-chain.affine.deriv")
a, b = host(net.activation(k)).astype(np.float64), acts[k].astype(np.float64)
err = np.linalg.norm(a - b, axis=0)
tot = np.linalg.norm(a - b)
order = np.argsort(-err)[:12]
print("column errors of", k, "total", tot, "top-12 share", np.sqrt((err[order] ** 2).sum()) / tot)
for c in order:
    r = int(np.argmax(np.abs(a[:, c] - b[:, c])))
    print("col %4d err %.3e |ref col| %.3e  max|diff| %.3e at row %d (ref %.4e hip %.4e) nonzero rows ref %d hip %d" % (
        c, err[c], np.linalg.norm(b[:, c]), np.abs(a[:, c] - b[:, c]).max(), r, b[r, c], a[r, c], (b[:, c] != 0).sum(), (a[:, c] != 0).sum()))
