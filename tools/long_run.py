#!/usr/bin/env python3
"""Sanity of many consecutive steps at the 7q net's size: four fixed minibatches cycled, natural gradient on; the objective
per frame must rise and stay finite.  usage (GPU box): python tools/long_run.py [steps] [chunk] [minibatch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 150
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = pkg.trainer.make_config(frames_per_chunk=chunk, num_sequences=mb, use_natural_gradient=1)
net = pkg.trainer.ChainNet(cfg)
net.set_params(net.init_params_numpy(seed=0))
den = pkg.synth.make_den_graph(2000, cfg.num_pdfs, mean_out_degree=8.0, seed=2)
dg = pkg.hipabi.DenGraph(den)
data = []
for k in range(4):
    feats, iv = pkg.trainer.synthetic_egs(net, seed=10 + k)
    sup = pkg.synth.make_supervision_from_den(den, mb, chunk // 3, num_paths=1, seed=20 + k)
    data.append((torch.from_numpy(feats).cuda(), torch.from_numpy(iv).cuda(), pkg.hipabi.Supervision(sup)))
t0 = time.perf_counter()
hist = []
for i in range(steps):
    f, v, s = data[i % 4]
    r = net.forward_backward(f, v, dg, s, step=i)
    net.update(pkg.trainer.learning_rate(i, 1, steps, i, steps, 1e-3, 1e-4), l2_regularize_scale=float(mb), step=i)
    if i % 4 == 3:
        r = r.cpu().numpy()
        hist.append(r[0] / r[2])
        # every numerator path is a denominator path of the same weight (synth.make_supervision_from_den): log p_num <= log p_den
        assert r[0] / r[2] <= 1e-3, ("LF-MMI objective per frame above 0", i, r[0] / r[2])
        if not np.isfinite(r[0]) or r[5] != 1.0:
            print("step", i, "objf", r[0], "ok flag", r[5])
            sys.exit(1)
        if i % 40 == 39:
            print("step %4d  objf/frame %.4f  xent/frame %.4f  (%.1f ms/step)" % (i, r[0] / r[2], r[6] / r[2], 1e3 * (time.perf_counter() - t0) / (i + 1)), flush=True)
assert np.mean(hist[-5:]) > np.mean(hist[:5]), (hist[:5], hist[-5:])
print("ok: objective per frame %.4f -> %.4f" % (np.mean(hist[:5]), np.mean(hist[-5:])))
