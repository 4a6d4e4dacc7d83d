#!/usr/bin/env python3
"""Generates tests/golden/r01_golden.npz: small known-answer vectors for the hot path.

The reference (skhu101/TDNN-F_NAS) ships no tests or golden vectors and cannot be built or imported here (no Kaldi), so
these vectors do NOT come from the reference.  Groups tdnn_*, bn_*, den_* are computed by an independent float64
PyTorch-CPU formulation (autograd for every derivative), i.e. by neither the oracle nor the HIP path; group net_* is a
regression pin of the oracle's whole training step (oracle-generated, labelled as such in the file).
    python tests/golden/make_golden.py        # rewrites the .npz next to this script
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
F = np.float32
out = {}


def tdnn_case(tag, offs, nt, B, Di, Do, step, seed):
    rng = np.random.default_rng(seed)
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B, t_step_out=step)
    K = len(offs)
    x = rng.standard_normal((rows_in, Di)).astype(F)
    W = (rng.standard_normal((Do, K * Di)) / np.sqrt(K * Di)).astype(F)
    b = rng.standard_normal(Do).astype(F)
    c = (rng.random(K) + 0.25).astype(F)
    dy = rng.standard_normal((N, Do)).astype(F)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Wt = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    views = [xt[o:o + rho * (N - 1) + 1:rho] for o in ro]  # GetInputPart, nnet-tdnn-component.cc:806-820
    yt = bt + sum(float(c[i]) * v @ Wt[:, i * Di:(i + 1) * Di].T for i, v in enumerate(views))
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    out.update({f"tdnn_{tag}_{k}": v for k, v in dict(
        offsets=np.asarray(offs, np.int32), dims=np.asarray([nt, B, Di, Do, step, rho, rows_in, N], np.int32),
        row_offsets=np.asarray(ro, np.int32), x=x, W=W, b=b, c=c, dy=dy, y=yt.detach().numpy(), dx=xt.grad.numpy(),
        dW=Wt.grad.numpy(), db=bt.grad.numpy()).items()})


tdnn_case("k3", [-1, 0, 1], 10, 4, 40, 160, 1, 11)         # the shape family of BASELINE configs[0]
tdnn_case("stride3", [0, 3], 6, 3, 160, 96, 3, 12)         # rho = 3 row order (tdnnf .affine after the subsampling layer)

# BatchNormComponent (train mode, epsilon 1e-3, target-rms 1)
rng = np.random.default_rng(21)
x = (rng.standard_normal((37, 24)) * 1.7 + 0.3).astype(F)
dz = rng.standard_normal((37, 24)).astype(F)
xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
zt = torch.nn.functional.batch_norm(xt, None, None, training=True, eps=1e-3)
zt.backward(torch.tensor(dz, dtype=torch.float64))
out.update(bn_x=x, bn_dz=dz, bn_z=zt.detach().numpy(), bn_dx=xt.grad.numpy())

# leaky-HMM denominator: dense float64 recursion without renormalisation, autograd for the occupancies
H, P, B, T, leaky = 17, 9, 3, 12, 0.1
g = pkg.synth.make_den_graph(H, P, mean_out_degree=3.0, seed=5)
y = np.random.default_rng(1).standard_normal((T * B, P)).astype(F)
yt = torch.tensor(y, dtype=torch.float64, requires_grad=True)
init = torch.tensor(g["init"], dtype=torch.float64)
src, dst = torch.tensor(g["src"], dtype=torch.long), torch.tensor(g["dst"], dtype=torch.long)
pdf, prob = torch.tensor(g["pdf"], dtype=torch.long), torch.tensor(g["prob"], dtype=torch.float64)
tot = []
for s in range(B):
    a = init + leaky * init.sum() * init
    for t in range(T):
        contrib = a[src] * prob * torch.exp(yt[t * B + s])[pdf]
        a = torch.zeros(H, dtype=torch.float64).index_add(0, dst, contrib)
        a = a + leaky * a.sum() * init
    tot.append(torch.log(a.sum()))
lp = torch.stack(tot).sum()
lp.backward()
out.update(den_dims=np.asarray([H, P, B, T], np.int32), den_leaky=np.asarray([leaky]), den_src=g["src"], den_dst=g["dst"],
           den_pdf=g["pdf"], den_prob=g["prob"], den_init=g["init"], den_y=y, den_logprob=np.asarray([float(lp.detach())]),
           den_occupancy=yt.grad.numpy())

# whole training step of a tiny 7q-shaped net: ORACLE-GENERATED regression pin (not an independent answer)
from tests.test_oracle_net import tiny_setup  # noqa: E402
cfg, comps, params, net, feats, iv, den, sup = tiny_setup(pkg, strides=(1, 0, 3), T=12, B=2, seed=3)
res, grads, _ = net.forward_backward(params, feats, iv, den, sup, step=0)
p2 = net.update(params, grads, 1e-3, float(cfg.num_sequences), 0)
out.update(net_objf=np.asarray([res["objf"], res["xent_objf"], res["weight"]]),
           net_grad_norms=np.asarray([np.linalg.norm(grads[c["begin"]:c["begin"] + c["rows"] * c["cols"]].astype(np.float64)) for c in comps]),
           net_update_norm=np.asarray([np.linalg.norm((p2 - params).astype(np.float64))]))

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r01_golden.npz")
out = {k: (v.astype(np.float32) if v.dtype == np.float64 and v.size > 8 else v) for k, v in out.items()}  # answers rounded to f32
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
