"""-m gpu: examples read from an archive drive the trainer exactly as the arrays they were written from."""
import numpy as np
import pytest

from tests.gpu_util import dev, host

pytestmark = pytest.mark.gpu


def test_training_step_from_an_archive(pkg, tmp_path):
    E = pkg.egs
    cfg = pkg.trainer.make_config(frames_per_chunk=24, num_sequences=4, strides=[1, 0, 3], bottleneck=16, feat_dim=40, ivector_dim=100, num_pdfs=60,
                                  hidden_dim=64, small_dim=32)
    net = pkg.trainer.ChainNet(cfg)
    net.set_params(net.init_params_numpy(seed=1, output_stddev=0.3))
    B, extra = cfg.num_sequences, 2  # the archive carries 2 more frames of context on each side than the net needs
    den = pkg.hipabi.DenGraph(pkg.synth.make_den_graph(30, cfg.num_pdfs, mean_out_degree=4.0, seed=5))
    path = tmp_path / "cegs.1.ark"
    batches = []
    with E.Writer(path) as w:
        for m in range(2):
            feats, iv = pkg.trainer.synthetic_egs(net, seed=10 + m)
            sup = pkg.synth.make_supervision(B, cfg.frames_per_chunk // 3, cfg.num_pdfs, max_alt=2, seed=20 + m)
            batches.append((feats, iv, sup))
            rng = np.random.default_rng(30 + m)
            for b in range(B):
                x = feats[b::B]  # this sequence's frames first_t .. first_t + num_t - 1
                wide = np.concatenate([rng.standard_normal((extra, x.shape[1])).astype(np.float32), x, rng.standard_normal((extra, x.shape[1])).astype(np.float32)])
                w.write("u%d-%d" % (m, b), wide, net.first_t - extra, E.sequence_of(sup, b), cfg.num_pdfs, ivector=iv[b], compress=False)
    got = list(E.minibatches(path, net))
    assert len(got) == 2
    for (feats, iv, sup), (f_dev, iv_dev, sup_dev) in zip(batches, got):
        assert np.array_equal(host(f_dev), feats) and np.array_equal(host(iv_dev), iv)
        net.grads.zero_()
        r1 = host(net.forward_backward(f_dev, iv_dev, den, sup_dev, step=3)).copy()
        g1 = host(net.grads).copy()
        net.grads.zero_()
        r2 = host(net.forward_backward(dev(feats), dev(iv), den, pkg.hipabi.Supervision(sup), step=3))
        assert r1[5] == 1.0 and np.array_equal(r1, r2) and np.array_equal(g1, host(net.grads))
    ahead = list(E.minibatches(path, net, prefetch=2))  # the same minibatches through the worker thread
    assert len(ahead) == 2 and all(np.array_equal(host(a[0]), host(g[0])) for a, g in zip(ahead, got))
    with pytest.raises(pkg.hipabi.HipAbiError, match="the net needs"):
        list(E.minibatches(path, net, frame_shift=3, prefetch=1))  # errors of the worker reach the caller
    # a frame shift moves the window; shifting by more than the spare context is refused
    shifted = list(E.minibatches(path, net, frame_shift=1))
    assert not np.array_equal(host(shifted[0][0]), batches[0][0])
    with pytest.raises(pkg.hipabi.HipAbiError, match="the net needs"):
        list(E.minibatches(path, net, frame_shift=3))
    net.close()


def test_archive_with_several_chunk_widths(pkg, tmp_path):
    E, T = pkg.egs, pkg.trainer
    base = dict(num_sequences=2, strides=[1, 0, 3], bottleneck=16, feat_dim=40, ivector_dim=100, num_pdfs=60, hidden_dim=64, small_dim=32)
    a = T.ChainNet(T.make_config(frames_per_chunk=24, **base))
    b = T.ChainNet(T.make_config(frames_per_chunk=18, **base), share=a)
    a.set_params(a.init_params_numpy(seed=1, output_stddev=0.3))
    den = pkg.hipabi.DenGraph(pkg.synth.make_den_graph(30, 60, mean_out_degree=4.0, seed=5))
    path = tmp_path / "mixed.ark"
    order = [a, b, b, a, b, a, a, b, b]  # widths interleaved as after nnet3-chain-shuffle-egs; the last example fills no minibatch
    with E.Writer(path) as w:
        for i, net in enumerate(order):
            feats, iv = T.synthetic_egs(net, seed=40 + i)
            sup = pkg.synth.make_supervision(net.cfg.num_sequences, net.cfg.frames_per_chunk // 3, 60, seed=50 + i)
            w.write("e%d" % i, feats[0::2], net.first_t, E.sequence_of(sup, 0), 60, ivector=iv[0], compress=True)
    got = list(E.minibatches_by_width(path, [a, b]))
    assert [g[0] is a for g in got] == [False, True, True, False]  # b fills first (examples 1, 2), then a (0, 3), a (5, 6), b (4, 7)
    for net, f, iv, sup in got:
        r = host(net.forward_backward(f, iv, den, sup, step=0))
        assert r[5] == 1.0 and np.isfinite(r[0]) and f.shape[0] == net.num_t_in * 2
    b.close()
    a.close()
