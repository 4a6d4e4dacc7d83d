#!/usr/bin/env python3
"""Is the step launch-bound?  Times the host side of forward_backward + update (return of the calls, no sync) against the
synchronised step.  usage (GPU box): python tools/host_launch_time.py [chunk] [minibatch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 150
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = pkg.trainer.make_config(frames_per_chunk=chunk, num_sequences=mb, use_natural_gradient=1)
net = pkg.trainer.ChainNet(cfg)
net.set_params(net.init_params_numpy(seed=0, output_stddev=0.05))
feats, iv = pkg.trainer.synthetic_egs(net, seed=1)
den = pkg.hipabi.DenGraph(pkg.synth.make_den_graph(4000, cfg.num_pdfs, mean_out_degree=12.0, seed=2))
sup = pkg.hipabi.Supervision(pkg.synth.make_supervision(mb, chunk // 3, cfg.num_pdfs, seed=3))
fd, ivd = torch.from_numpy(feats).cuda(), torch.from_numpy(iv).cuda()
for i in range(14):
    net.forward_backward(fd, ivd, den, sup, step=i)
    net.update(1e-4, step=i)
torch.cuda.synchronize()
host, total = [], []
for i in range(14, 30):
    t0 = time.perf_counter()
    net.forward_backward(fd, ivd, den, sup, step=i)
    net.update(1e-4, step=i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0)
    total.append(t2 - t0)
print("per-step synchronised ms:", " ".join("%.1f" % (1e3 * t) for t in total))
print("per-step host enqueue ms:", " ".join("%.1f" % (1e3 * t) for t in host))
print("chunk %d x %d: host enqueue %.2f ms / step (median), synchronised step %.2f ms (median; mean over the refresh cycle %.2f)"
      % (chunk, mb, 1e3 * np.median(host), 1e3 * np.median(total), 1e3 * np.mean(total)))
