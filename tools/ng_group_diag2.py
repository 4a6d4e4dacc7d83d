"""Grouped and per-object natural-gradient chains against the oracle over several minibatches (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from tests.test_gpu_ng_group import CASES
from tests.gpu_util import rel_l2, dev, host
from tests.oracle_net import OracleNet
name = sys.argv[1] if len(sys.argv) > 1 else "rank80"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kw = dict(CASES)[name]
for grouped in (0, 1):
    os.environ["TDNNF_NG_GROUPED"] = str(grouped)
    cfg = pkg.trainer.make_config(use_natural_gradient=1, **kw)
    net = pkg.trainer.ChainNet(cfg)
    params = net.init_params_numpy(seed=3, output_stddev=0.3)
    net.set_params(params)
    ref = OracleNet(pkg, cfg, net.components)
    feats, iv = pkg.trainer.synthetic_egs(net, seed=4)
    den = pkg.synth.make_den_graph(40, cfg.num_pdfs, mean_out_degree=4.0, seed=5)
    sup = pkg.synth.make_supervision(cfg.num_sequences, cfg.frames_per_chunk // 3, cfg.num_pdfs, seed=6)
    dg, ds = pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup)
    fd, ivd = dev(feats), dev(iv)
    for step in range(steps):
        draws = np.random.default_rng(100 + step).uniform(1e-3, 1 - 1e-3, max(net.num_draws, 1)).astype(np.float32)
        net.set_random_draws(draws)
        net.grads.zero_()
        r = host(net.forward_backward(fd, ivd, dg, ds, step=step))
        g = host(net.grads).copy()
        _, g_ref, _ = ref.forward_backward(params, feats, iv, den, sup, step=step, draws=draws)
        per = {c["name"]: rel_l2(g[c["begin"]:c["begin"] + c["rows"] * c["cols"]], g_ref[c["begin"]:c["begin"] + c["rows"] * c["cols"]]) for c in net.components[1:]}
        worst = max(per, key=per.get)
        print("grouped", grouped, "step", step, "%.2e" % rel_l2(g, g_ref), "worst", worst, "%.2e" % per[worst], flush=True)
        # both follow the ORACLE's trajectory, so the comparison stays a one-step comparison
        params = ref.update(params, g_ref, 1e-3, float(cfg.num_sequences), step)
        net.set_params(params)
    net.close()
