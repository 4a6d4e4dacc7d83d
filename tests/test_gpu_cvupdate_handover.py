"""-m gpu: the literal pretrain -> cv-update hand-over of run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:128-142:

    nnet3-am-copy --raw --binary=false --edits="set-learning-rate-factor learning-rate-factor=0" final.mdl - | sed ... x7 | nnet3-copy - 0.raw

i.e. the edit applied to the model in memory, the model written as TEXT, the seven substitutions applied to the text, the result
read back -- against the route the trainer offers directly (a cv_update = 1 configuration + parameters + statistics): the graphs,
flags and learning-rate factors must be the same and one cv-update step must agree bit for bit."""
import numpy as np
import pytest

from tests.gpu_util import dev, host

pytestmark = pytest.mark.gpu

_D = dict(frames_per_chunk=18, num_sequences=3, strides=[1, 1, 1, 1], bottleneck=16, feat_dim=40, ivector_dim=100,
          num_pdfs=120, hidden_dim=64, small_dim=32, use_natural_gradient=1)


@pytest.mark.parametrize("use_gumbel", [True, False], ids=["gumbel", "softmax"])
def test_sed_pipeline_equals_the_cv_update_configuration(pkg, tmp_path, use_gumbel):
    T = pkg.trainer
    pre = T.ChainNet(T.make_config(darts_num_offsets=4, darts_flags=T.DARTS_UNIFORM_SAMPLE, **_D))
    pre.set_params(pre.init_params_numpy(seed=3, output_stddev=0.3))
    feats, iv = T.synthetic_egs(pre, seed=4)
    den = pkg.synth.make_den_graph(40, pre.cfg.num_pdfs, mean_out_degree=4.0, seed=5)
    sup = pkg.synth.make_supervision(pre.cfg.num_sequences, pre.cfg.frames_per_chunk // 3, pre.cfg.num_pdfs, seed=6)
    dg, ds = pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup)
    fd, ivd = dev(feats), dev(iv)
    for step in (0, 1):  # two pretrain steps: non-trivial BatchNorm / ReLU statistics and architecture logits
        pre.set_random_draws(np.random.default_rng(300 + step).uniform(1e-3, 1 - 1e-3, pre.num_draws).astype(np.float32))
        pre.grads.zero_()
        pre.forward_backward(fd, ivd, dg, ds, step=step)
        pre.update(1e-3, step=step)
    # ---- route B: the recipe's pipeline
    edited = pre.apply_edits('nnet3-am-copy --raw --binary=false --edits="set-learning-rate-factor learning-rate-factor=0" final.mdl -')
    assert edited == [("set-learning-rate-factor", len(pre.components) - 1)]  # every updatable component: all but the fixed lda layer
    assert all(c["lr_factor"] == 0.0 for c in pre.components)
    parent, child = tmp_path / "parent.txt", tmp_path / "0.raw.txt"
    pre.write_model(parent, binary=False)
    text = open(parent).read()
    assert text.count("<TdnnDARTSV3Component> <LearningRateFactor> 0 ") == 8 and "<uniform-sample> T" in text and "<use-gumbel> F" in text
    open(child, "w").write(T.apply_cvupdate_seds(text, use_gumbel=use_gumbel))
    cfg_b = T.config_from_model(child, pre.cfg.frames_per_chunk, pre.cfg.num_sequences)
    flags = (T.DARTS_USE_GUMBEL if use_gumbel else 0) | T.DARTS_UPDATE_ALPHA
    assert cfg_b.cv_update == 1 and cfg_b.darts_num_offsets == 4 and cfg_b.darts_flags == flags
    net_b = T.ChainNet(cfg_b)
    net_b.read_model(child)
    # ---- route A: the cv-update configuration, the parent's parameters and the statistics as the parent's file holds them
    cfg_a = T.make_config(darts_num_offsets=4, darts_flags=flags, cv_update=1, **_D)
    net_a = T.ChainNet(cfg_a)
    tmp = T.ChainNet(T.make_config(darts_num_offsets=4, darts_flags=T.DARTS_UNIFORM_SAMPLE, **_D))
    tmp.read_model(parent)  # (a text model stores BatchNorm means / variances in float: the statistics a reader reconstructs)
    net_a.set_params(host(pre.params))
    net_a.set_stats(tmp.get_stats())
    tmp.close()
    assert T.config_text(cfg_a) == T.config_text(cfg_b)
    fa, fb = {c["name"]: c["lr_factor"] for c in net_a.components}, {c["name"]: c["lr_factor"] for c in net_b.components}
    assert fa == fb and fb["tdnnf2.linear"] == np.float32(1e-4) and fb["tdnn1.affine"] == 0.0 and fb["output.affine"] == 0.0
    assert np.array_equal(host(net_a.params), host(net_b.params)) and np.array_equal(net_a.get_stats(), net_b.get_stats())
    # ---- one cv-update step
    out = []
    for net in (net_a, net_b):
        net.apply_edits(T.temperature_edit_string(0.4))  # the per-iteration edit train.py prepends under --temperature_schedule
        net.set_random_draws(np.random.default_rng(500).uniform(1e-3, 1 - 1e-3, net.num_draws).astype(np.float32))
        net.grads.zero_()
        r = host(net.forward_backward(fd, ivd, dg, ds, step=0)).copy()
        g = host(net.grads).copy()
        net.update(1e-3, step=0)
        out.append((r, g, host(net.params).copy()))
        net.close()
    (ra, ga, pa), (rb, gb, pb) = out
    assert ra[5] == 1.0 and np.array_equal(ra, rb) and np.array_equal(ga, gb) and np.array_equal(pa, pb)
    assert np.abs(ga).sum() > 0  # the architecture logits (and, at 1e-4, theta) do receive a gradient
    # only TdnnDARTSV3 components moved -- and the matrices ConstrainOrthonormal steps whatever their learning rate
    # (nnet-utils.cc:1040-1077 does not look at it; the DARTS .linear components are not constrained, :1047-1061)
    p0 = host(pre.params)
    for c in net_a.components:
        if c["orthonormal"] != 0.0:
            continue
        sl = slice(c["begin"], c["begin"] + c["rows"] * c["cols"] + c["num_alpha"] + (c["rows"] if c["has_bias"] else 0))
        assert (c["num_alpha"] > 0) == bool((pa[sl] != p0[sl]).any()), c["name"]
    pre.close()


def test_apply_edits_errors(pkg):
    T = pkg.trainer
    net = T.ChainNet(T.make_config(**{k: v for k, v in _D.items() if k != "use_natural_gradient"}))
    with pytest.raises(ValueError, match="not currently supported"):
        net.apply_edits("apply-svd name=* bottleneck-dim=10")
    with pytest.raises(ValueError, match="expected learning-rate-factor"):
        net.apply_edits("set-learning-rate-factor name=*")
    with pytest.raises(ValueError, match="Could not interpret"):
        net.apply_edits("set-learning-rate-factor learning-rate-factor=1 colour=blue")
    assert net.apply_edits("set-learning-rate-factor name=tdnnf*.affine learning-rate-factor=0.5") == [("set-learning-rate-factor", 4)]
    assert [c["lr_factor"] for c in net.components if c["name"].endswith(".affine") and c["name"].startswith("tdnnf")] == [0.5] * 4
    assert net.apply_edits("set-learning-rate-factor name=lda learning-rate-factor=3") == [("set-learning-rate-factor", 0)]  # not updatable
    net.close()
