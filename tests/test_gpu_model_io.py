"""-m gpu: nnet3 raw model files written / read by the trainer (csrc/model_io.hip).
Round trips in both encodings, the token order of the reference's Write() functions, and a Kaldi-formatted
(6 significant digits, free-form white space) text file."""
import re

import numpy as np
import pytest

from tests.gpu_util import dev, host, rel_l2

pytestmark = pytest.mark.gpu

SMALL = dict(frames_per_chunk=24, num_sequences=3, strides=[1, 0, 3], bottleneck=16, feat_dim=40, ivector_dim=100, num_pdfs=50,
             hidden_dim=64, small_dim=32)
VARIANTS = [
    ("tdnnf", dict(SMALL)),
    ("darts-k3", dict(SMALL, darts_num_offsets=3, darts_flags=1 | 16, darts_temp_proportion=0.8)),
    ("bn-supernet-onehot", dict(SMALL, bn_choice_dims=[4, 4, 8], bn_mode=0)),
    ("bn-supernet-gumbel-cv", dict(SMALL, bn_choice_dims=[4, 4, 8], bn_mode=2, bn_flops_scale=0.25, bn_temp_proportion=0.9, cv_update=1)),
    # a derived child: per-layer X.linear {-a, 0} / X.affine {0, b} and per-layer bottleneck dims
    ("child", dict(SMALL, strides=None, layer_offsets=[(2, 1), (0, 3), (5, 0), (1, 2)], bottleneck=[16, 8, 24, 16])),
]


def trained_net(pkg, kw, seed=3):
    """A net with non-trivial parameters and BatchNorm / ReLU statistics (one training step, cv nets get them loaded)."""
    cfg = pkg.trainer.make_config(**kw)
    net = pkg.trainer.ChainNet(cfg)
    params = net.init_params_numpy(seed=seed, output_stddev=0.3)
    rng = np.random.default_rng(seed)
    for c in net.components:  # architecture parameters away from zero
        n = c["rows"] * c["cols"]
        params[c["begin"] + n:c["begin"] + n + c["num_alpha"]] = rng.standard_normal(c["num_alpha"]).astype(np.float32)
        if c["name"].endswith((".alpha", ".softmax")):
            params[c["begin"]:c["begin"] + c["rows"]] = rng.standard_normal(c["rows"]).astype(np.float32)
    net.set_params(params)
    if cfg.cv_update:
        st = rng.random(net.get_stats().size) + 0.5
        net.set_stats(st * 10.0)
    else:
        feats, iv = pkg.trainer.synthetic_egs(net, seed=4)
        den = pkg.synth.make_den_graph(20, cfg.num_pdfs, mean_out_degree=4.0, seed=5)
        sup = pkg.synth.make_supervision(cfg.num_sequences, cfg.frames_per_chunk // 3, cfg.num_pdfs, seed=6)
        net.set_random_draws(rng.uniform(0.01, 0.99, max(net.num_draws, 1)).astype(np.float32))
        net.forward_backward(dev(feats), dev(iv), pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup), step=0)
        net.grads.zero_()
    return cfg, net


@pytest.mark.parametrize("name,kw", VARIANTS, ids=[v[0] for v in VARIANTS])
@pytest.mark.parametrize("binary", [True, False], ids=["binary", "text"])
def test_model_round_trip(pkg, tmp_path, name, kw, binary):
    cfg, a = trained_net(pkg, kw)
    path = tmp_path / ("m.raw" if binary else "m.txt")
    a.write_model(path, binary=binary, learning_rate=1e-3)
    head = open(path, "rb").read(9)
    assert head.startswith(b"\0B<Nnet3>") if binary else head.startswith(b"<Nnet3> \n")
    b = pkg.trainer.ChainNet(pkg.trainer.make_config(**kw))
    b.set_params(np.full(b.num_params, 7.0, np.float32))
    b.read_model(path)
    pa, pb = host(a.params), host(b.params)
    for c in a.components:  # every parameter of every component, bit for bit (text mode prints 9 significant digits)
        end = c["begin"] + c["rows"] * c["cols"] + c["num_alpha"] + (c["rows"] if c["has_bias"] else 0)
        assert np.array_equal(pa[c["begin"]:end], pb[c["begin"]:end]), c["name"]
    sa, sb = a.get_stats(), b.get_stats()
    assert sa[0] > 0 and np.allclose(sa, sb, rtol=2e-5, atol=1e-5 * np.abs(sa).max())
    a.close()
    b.close()


def test_text_model_follows_the_reference_token_order(pkg, tmp_path):
    cfg, net = trained_net(pkg, SMALL)
    path = tmp_path / "m.txt"
    net.write_model(path, binary=False, learning_rate=2e-3)
    text = open(path).read()
    cfg_part, comp_part = text.split("\n\n", 1)
    lines = cfg_part.split("\n")
    assert lines[0] == "<Nnet3> " and lines[1] == "input-node name=ivector dim=100" and lines[2] == "input-node name=input dim=40"
    assert "component-node name=lda component=lda input=Append(Offset(input, -1), input, Offset(input, 1), ReplaceIndex(ivector, t, 0))" in lines
    assert "component-node name=tdnnf2.noop component=tdnnf2.noop input=Sum(Scale(0.66, tdnn1.dropout), tdnnf2.dropout)" in lines
    assert "component-node name=tdnnf3.linear component=tdnnf3.linear input=tdnnf2.noop" in lines
    assert "output-node name=output input=output.affine objective=linear" in lines
    assert "output-node name=output-xent input=output-xent.log-softmax objective=linear" in lines
    m = re.match(r"<NumComponents> (\d+) \n", comp_part)
    names = re.findall(r"<ComponentName> (\S+) <(\w+)> ", comp_part)
    assert int(m.group(1)) == len(names) == 5 + 6 * 3 + 1 + 2 * 6 + 1
    assert names[0] == ("lda", "FixedAffineComponent") and names[1] == ("tdnn1.affine", "NaturalGradientAffineComponent")
    assert ("tdnnf2.linear", "TdnnComponent") in names and ("tdnnf2.noop", "NoOpComponent") in names
    assert ("prefinal-l", "LinearComponent") in names and names[-1] == ("output-xent.log-softmax", "LogSoftmaxComponent")

    def tokens_of(name):
        blk = comp_part.split(f"<ComponentName> {name} ", 1)[1].split("<ComponentName>", 1)[0]
        return re.findall(r"</?[A-Za-z][\w-]*>", blk)
    # TdnnDARTSV3Component::Write minus the DARTS tokens (nnet-tdnn-component.cc:659-700), after WriteUpdatableCommon
    assert tokens_of("tdnnf2.linear") == ["<TdnnComponent>", "<MaxChange>", "<L2Regularize>", "<LearningRate>", "<TimeOffsets>", "<LinearParams>",
                                          "<BiasParams>", "<OrthonormalConstraint>", "<UseNaturalGradient>", "<NumSamplesHistory>", "<AlphaInOut>",
                                          "<RankInOut>", "</TdnnComponent>"]
    # NaturalGradientAffineComponent::Write (nnet-simple-component.cc:2935-2958); output-xent has learning-rate-factor 5
    assert tokens_of("output-xent.affine") == ["<NaturalGradientAffineComponent>", "<LearningRateFactor>", "<MaxChange>", "<L2Regularize>",
                                               "<LearningRate>", "<LinearParams>", "<BiasParams>", "<RankIn>", "<RankOut>", "<UpdatePeriod>",
                                               "<NumSamplesHistory>", "<Alpha>", "</NaturalGradientAffineComponent>"]
    # LinearComponent::Write :3161-3188 (orthonormal constraint present)
    assert tokens_of("prefinal-l") == ["<LinearComponent>", "<MaxChange>", "<L2Regularize>", "<LearningRate>", "<Params>", "<OrthonormalConstraint>",
                                       "<UseNaturalGradient>", "<RankInOut>", "<Alpha>", "<NumSamplesHistory>", "<UpdatePeriod>", "</LinearComponent>"]
    # BatchNormComponent::Write nnet-normalize-component.cc:616-642, NonlinearComponent::Write nnet-component-itf.cc:630-686
    assert tokens_of("tdnnf2.batchnorm") == ["<BatchNormComponent>", "<Dim>", "<BlockDim>", "<Epsilon>", "<TargetRms>", "<TestMode>", "<Count>",
                                             "<StatsMean>", "<StatsVar>", "</BatchNormComponent>"]
    assert tokens_of("tdnnf2.relu") == ["<RectifiedLinearComponent>", "<Dim>", "<ValueAvg>", "<DerivAvg>", "<Count>", "<OderivRms>", "<OderivCount>",
                                        "<NumDimsSelfRepaired>", "<NumDimsProcessed>", "<SelfRepairScale>", "</RectifiedLinearComponent>"]
    lr = float(re.search(r"<LearningRate> (\S+) ", comp_part.split("<ComponentName> output-xent.affine ", 1)[1]).group(1))
    assert abs(lr - 2e-3 * 5) < 1e-8  # learning rate times the component's factor
    assert "<TimeOffsets> [ -1 0 ]" in comp_part and "<TimeOffsets> [ 0 3 ]" in comp_part and text.endswith("</Nnet3> ")
    net.close()


def test_reads_kaldi_formatted_text(pkg, tmp_path):
    """Kaldi prints 6 significant digits and breaks lines freely; such a file loads to within that precision."""
    cfg, a = trained_net(pkg, SMALL)
    path = tmp_path / "m.txt"
    a.write_model(path, binary=False, learning_rate=1e-3)
    text = open(path).read()
    cfg_part, comp_part = text.split("\n\n", 1)
    comp_part = re.sub(r"-?\d+\.\d+(e[-+]?\d+)?", lambda m: "%g" % float(m.group(0)), comp_part)  # 6 digits, as operator<< prints
    comp_part = comp_part.replace("<BiasParams>  [", "<BiasParams>\t[").replace("] \n<", "]\n<")
    open(path, "w").write(cfg_part + "\n\n" + comp_part)
    b = pkg.trainer.ChainNet(pkg.trainer.make_config(**SMALL))
    b.read_model(path)
    assert rel_l2(host(b.params), host(a.params)) < 2e-6
    assert not np.array_equal(host(b.params), host(a.params))  # (it really was rounded)
    a.close()
    b.close()


@pytest.mark.parametrize("name,kw", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_config_from_model_rebuilds_the_net(pkg, tmp_path, name, kw):
    """A net created only from the model file (tdnnf_net_config_from_model) computes what the original computes."""
    cfg, a = trained_net(pkg, kw)
    path = tmp_path / "m.raw"
    a.write_model(path, binary=True)
    cfg2 = pkg.trainer.config_from_model(path, frames_per_chunk=cfg.frames_per_chunk, num_sequences=cfg.num_sequences)
    for f in ("feat_dim", "ivector_dim", "num_pdfs", "hidden_dim", "prefinal_small_dim", "num_layers", "darts_num_offsets", "darts_flags",
              "bn_num_choices", "bn_mode", "cv_update"):
        assert getattr(cfg2, f) == getattr(cfg, f), f
    floats = ["bypass_scale", "l2_hidden", "l2_output", "max_change_hidden", "max_change_output", "xent_regularize", "relu_self_repair_scale"]
    floats += ["darts_temp_proportion"] if cfg.darts_num_offsets else []
    floats += ["bn_flops_scale"] if cfg.bn_mode else []
    floats += ["bn_temp_proportion"] if cfg.bn_mode == 2 else []
    for f in floats:
        assert abs(getattr(cfg2, f) - getattr(cfg, f)) < 1e-6, f
    L = cfg.num_layers
    assert list(cfg2.bottleneck_dim[:L]) == list(cfg.bottleneck_dim[:L])
    if not cfg.darts_num_offsets:
        assert list(cfg2.time_stride[:L]) == list(cfg.time_stride[:L])
    assert cfg2.use_layer_offsets == cfg.use_layer_offsets
    if cfg.use_layer_offsets:
        assert list(cfg2.offset_left[:L]) == list(cfg.offset_left[:L]) and list(cfg2.offset_right[:L]) == list(cfg.offset_right[:L])
    assert list(cfg2.bn_choice_dims[:cfg.bn_num_choices]) == list(cfg.bn_choice_dims[:cfg.bn_num_choices])
    b = pkg.trainer.ChainNet(cfg2)
    b.read_model(path)
    feats, iv = pkg.trainer.synthetic_egs(a, seed=14)
    den = pkg.synth.make_den_graph(20, cfg.num_pdfs, mean_out_degree=4.0, seed=5)
    sup = pkg.synth.make_supervision(cfg.num_sequences, cfg.frames_per_chunk // 3, cfg.num_pdfs, seed=6)
    dg, ds = pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup)
    draws = np.random.default_rng(8).uniform(0.01, 0.99, max(a.num_draws, 1)).astype(np.float32)
    outs = []
    for net in (a, b):
        net.set_random_draws(draws)
        net.grads.zero_()
        r = host(net.forward_backward(dev(feats), dev(iv), dg, ds, step=5)).copy()
        outs.append((r, host(net.activation("output")).copy(), host(net.grads).copy()))
    if cfg.cv_update:  # BatchNormTest: the statistics come back through float mean / variance
        assert rel_l2(outs[1][1], outs[0][1]) < 1e-4
    else:
        assert np.array_equal(outs[0][1], outs[1][1]) and outs[0][0][0] == outs[1][0][0]
    if not cfg.use_natural_gradient:  # (the rebuilt net preconditions its gradients; the raw-gradient original does not)
        assert rel_l2(outs[1][2], outs[0][2]) > 0
    a.close()
    b.close()


def test_read_model_rejects_mismatches(pkg, tmp_path):
    cfg, a = trained_net(pkg, SMALL)
    path = tmp_path / "m.raw"
    a.write_model(path)
    other = pkg.trainer.ChainNet(pkg.trainer.make_config(**dict(SMALL, hidden_dim=96)))
    with pytest.raises(pkg.hipabi.HipAbiError, match="in the file"):
        other.read_model(path)
    with pytest.raises(pkg.hipabi.HipAbiError, match="cannot open"):
        other.read_model(tmp_path / "missing.raw")
    data = open(path, "rb").read()
    open(path, "wb").write(data[:len(data) // 2])
    with pytest.raises(pkg.hipabi.HipAbiError):
        a.read_model(path)
    a.close()
    other.close()
