import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
from tests.gpu_util import dev, host, rel_l2
from tests.oracle_net import OracleNet
kw = dict(frames_per_chunk=48, num_sequences=8, strides=[1, 1, 1, 0, 3, 3, 3], bottleneck=24, feat_dim=40, ivector_dim=100, num_pdfs=150, hidden_dim=96, small_dim=48, use_natural_gradient=1)
cfg = pkg.trainer.make_config(**kw)
net = pkg.trainer.ChainNet(cfg)
params = net.init_params_numpy(seed=3, output_stddev=0.3)
net.set_params(params)
ref = OracleNet(pkg, cfg, net.components)
feats, iv = pkg.trainer.synthetic_egs(net, seed=4)
den = pkg.synth.make_den_graph(60, cfg.num_pdfs, mean_out_degree=4.0, seed=5)
sup = pkg.synth.make_supervision(cfg.num_sequences, cfg.frames_per_chunk // 3, cfg.num_pdfs, seed=6)
dg, ds = pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup)
res_ref, g_ref, acts = ref.forward_backward(params, feats, iv, den, sup, step=0)
net.grads.zero_()
r = host(net.forward_backward(dev(feats), dev(iv), dg, ds, step=0))
g = host(net.grads)
for c in net.components[1:]:
    n = c["rows"] * c["cols"]
    w = slice(c["begin"], c["begin"] + n)
    b = slice(c["begin"] + n, c["begin"] + n + (c["rows"] if c["has_bias"] else 0))
    print(f"{c['name']:24s} W {rel_l2(g[w], g_ref[w]):.3e} |g|={np.linalg.norm(g[w]):.3e} |ref|={np.linalg.norm(g_ref[w]):.3e}", f"b {rel_l2(g[b], g_ref[b]):.3e}" if c["has_bias"] else "")
