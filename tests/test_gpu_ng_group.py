"""The grouped natural-gradient side chain (csrc/ng_group.hip: one launch per stage for all components of a gradient bucket)
against the per-object chain it replaces (option ng_grouped = 0: the round-2 path, itself held to the oracle's literal
formulation in test_gpu_net.py) over a whole refresh schedule: first 10 minibatches refresh every time, then every 4th."""
import numpy as np
import pytest

from tests.gpu_util import dev, host, rel_l2

pytestmark = pytest.mark.gpu


def make_net(pkg, grouped, **kw):
    with pkg.hipabi.option("ng_grouped", 1 if grouped else 0):  # read when the net is created
        return pkg.trainer.ChainNet(pkg.trainer.make_config(use_natural_gradient=1, **kw))


def run_steps(pkg, nets, steps):
    """Runs `steps` minibatches on every net of `nets`, all along the FIRST net's parameter trajectory (so that a comparison stays
    a one-step comparison of the preconditioned gradients: left to themselves, two trajectories of these tiny nets drift apart
    through flipped ReLU masks within a few steps).  Returns per net the list of gradients."""
    cfg = nets[0].cfg
    params = nets[0].init_params_numpy(seed=3, output_stddev=0.3)
    feats, iv = pkg.trainer.synthetic_egs(nets[0], seed=4)
    den = pkg.synth.make_den_graph(40, cfg.num_pdfs, mean_out_degree=4.0, seed=5)
    sup = pkg.synth.make_supervision(cfg.num_sequences, cfg.frames_per_chunk // 3, cfg.num_pdfs, seed=6)
    dg, ds = pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup)
    fd, ivd = dev(feats), dev(iv)
    out = [[] for _ in nets]
    for step in range(steps):
        draws = np.random.default_rng(100 + step).uniform(1e-3, 1 - 1e-3, max(nets[0].num_draws, 1)).astype(np.float32)
        for k, net in enumerate(nets):
            net.set_params(params)
            net.set_random_draws(draws)
            net.grads.zero_()
            r = host(net.forward_backward(fd, ivd, dg, ds, step=step))
            assert r[5] == 1.0
            out[k].append(host(net.grads).copy())
        nets[0].update(1e-3, step=step)
        params = host(nets[0].params).copy()
        for net in nets[1:]:
            net.grads.zero_()
    return out


CASES = [
    # (rows per minibatch well above the preconditioners' ranks: with N <= R the reference itself falls back to a randomised basis)
    ("7q-small", dict(frames_per_chunk=60, num_sequences=16, strides=[1, 0, 3], bottleneck=32, feat_dim=40, ivector_dim=100, num_pdfs=150,
                      hidden_dim=96, small_dim=48)),
    ("rank80", dict(frames_per_chunk=90, num_sequences=32, strides=[1, 3], bottleneck=64, feat_dim=40, ivector_dim=100, num_pdfs=200,
                    hidden_dim=192, small_dim=64)),
    ("darts-softmax", dict(frames_per_chunk=60, num_sequences=16, strides=[1, 3], bottleneck=16, feat_dim=16, ivector_dim=8, num_pdfs=64,
                           hidden_dim=64, small_dim=32, darts_num_offsets=4, darts_flags=0)),
    ("darts-uniform", dict(frames_per_chunk=60, num_sequences=16, strides=[1, 3], bottleneck=16, feat_dim=16, ivector_dim=8, num_pdfs=64,
                           hidden_dim=64, small_dim=32, darts_num_offsets=4)),
]


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_grouped_chain_matches_the_per_object_chain(pkg, name, kw):
    steps = 15  # refreshes at t = 0..10 and 14
    nets = [make_net(pkg, False, **kw), make_net(pkg, True, **kw)]
    g_ref, g = run_steps(pkg, nets, steps)
    for i in range(steps):
        # same arithmetic in another summation order; the eigen-decompositions feed the differences back
        assert rel_l2(g[i], g_ref[i]) < 1e-4, (i, rel_l2(g[i], g_ref[i]))
    for net in nets:
        net.close()


def test_grouped_chain_is_bit_reproducible(pkg):
    a = run_steps(pkg, [make_net(pkg, True, **CASES[0][1])], 13)[0]
    b = run_steps(pkg, [make_net(pkg, True, **CASES[0][1])], 13)[0]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
