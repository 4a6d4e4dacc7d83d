"""Per-component gradient comparison of the C++ trainer against tests/oracle_net.py for one test_gpu_net case.
usage (GPU box): python tools/debug_net.py <case-name> [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
sys.modules.setdefault("tdnnf_nas_amd", pkg)
from tests.gpu_util import dev, host, rel_l2
from tests.oracle_net import OracleNet
from tests import test_gpu_net as T

name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kw, H = [(c[1], c[2]) for c in T.CASES if c[0] == name][0]
cfg = pkg.trainer.make_config(**kw)
net = pkg.trainer.ChainNet(cfg)
params = net.init_params_numpy(seed=3, output_stddev=0.3)
rng = np.random.default_rng(19)
for c in net.components:
    if c["name"].endswith((".alpha", ".softmax")):
        params[c["begin"]:c["begin"] + c["rows"]] = rng.standard_normal(c["rows"]).astype(np.float32) * 0.7
net.set_params(params)
ref = OracleNet(pkg, cfg, net.components)
feats, iv = pkg.trainer.synthetic_egs(net, seed=4)
den = pkg.synth.make_den_graph(H, cfg.num_pdfs, mean_out_degree=4.0, seed=5)
sup = pkg.synth.make_supervision(cfg.num_sequences, cfg.frames_per_chunk // 3, cfg.num_pdfs, seed=6)
dg, ds = pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup)
for step in range(steps):
    draws = np.random.default_rng(100 + step).uniform(1e-3, 1 - 1e-3, max(net.num_draws, 1)).astype(np.float32)
    net.set_random_draws(draws)
    res_ref, g_ref, acts = ref.forward_backward(params, feats, iv, den, sup, step=step, draws=draws)
    net.grads.zero_()
    r = host(net.forward_backward(dev(feats), dev(iv), dg, ds, step=step))
    g = host(net.grads)
    print("step", step, "objf", r[0], res_ref["objf"])
    for c in net.components[1:]:
        n = c["rows"] * c["cols"]
        w = slice(c["begin"], c["begin"] + n)
        b = slice(c["begin"] + n + c["num_alpha"], c["begin"] + n + c["num_alpha"] + (c["rows"] if c["has_bias"] else 0))
        print(f"{c['name']:24s} W {rel_l2(g[w], g_ref[w]):.3e} |g|={np.linalg.norm(g[w]):.3e} |ref|={np.linalg.norm(g_ref[w]):.3e}",
              f"b {rel_l2(g[b], g_ref[b]):.3e}" if c["has_bias"] else "")
        if n <= 8:
            print("    ", g[w], g_ref[w])
    params = ref.update(params, g_ref, 1e-3, float(cfg.num_sequences), step)
    net.update(1e-3, step=step)
    net.set_params(params)
