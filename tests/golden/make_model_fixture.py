#!/usr/bin/env python3
"""Writes tests/golden/r01_tiny_model.txt: an nnet3 raw text model of a tiny TDNN-F net, produced by THIS library's
writer (tdnnf_net_write_model) on an MI355X.  It pins the on-disk text format for the CPU tests of the reader side
(tdnnf_net_config_from_model needs no GPU).  Not a reference artefact: the reference ships no model files.
    gpurun -- 'python tests/golden/make_model_fixture.py gpurun_out/r01_tiny_model.txt'   # then copy into tests/golden/"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
cfg = pkg.trainer.make_config(frames_per_chunk=12, num_sequences=2, strides=[1, 0, 3], bottleneck=[4, 8, 4], feat_dim=8, ivector_dim=4, num_pdfs=10,
                              hidden_dim=16, small_dim=8, relu_self_repair_scale=2.0e-5)
net = pkg.trainer.ChainNet(cfg)
net.set_params(np.round(net.init_params_numpy(seed=11, output_stddev=0.3), 3))
st = np.round(np.random.default_rng(12).random(net.get_stats().size) + 0.5, 3)
net.set_stats(st * 4.0)
net.write_model(sys.argv[1], binary=False, learning_rate=1e-3)
print("wrote", sys.argv[1], os.path.getsize(sys.argv[1]), "bytes")
