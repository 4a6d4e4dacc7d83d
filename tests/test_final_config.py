"""final.config ingestion: the texts the reference's own scripts WRITE -- outputs of
generate_config.py, generate_bottleneckCB8share_onehottrain_config.py and its optimal-offsets variant
(tests/golden/r01_configs_golden.json, made by running the scripts in the build container) -- and the xconfig-style texts of
configs.final_config() are parsed back into the trainer's configuration, the graph wiring is checked against the node
lines the library emits, and the initial parameters follow the components' InitFromConfig.  No GPU needed up to the last test."""
import json
import os

import numpy as np
import pytest

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "r01_configs_golden.json")))


def test_plain_7q_and_child_configs(pkg):
    c, t = pkg.configs, pkg.trainer
    cfg = c.net_config_from_final_config(c.final_config(), frames_per_chunk=150, num_sequences=64)
    ref = t.make_config(use_natural_gradient=1, use_dropout=1)
    for f, _ in t.NetConfig._fields_:
        a, b = getattr(cfg, f), getattr(ref, f)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == pytest.approx(b)), f
    # the manual-offset recipe (run_tdnn_7q_fbk_40_manual.sh --offset 6 --bottleneckdim 160)
    cfg = c.net_config_from_final_config(c.final_config(strides=[1, 1, 1, 0] + [6] * 10))
    assert [cfg.time_stride[i] for i in range(14)] == [1, 1, 1, 0] + [6] * 10 and not cfg.use_layer_offsets
    # a derived child: per-layer offsets and bottleneck dims (what generate_top_list.py writes)
    offs, dims = [(5, 0), (5, 2), (0, 2), (4, 0), (3, 6)], [160, 100, 25, 240, 80]
    cfg = c.net_config_from_final_config(c.final_config(layer_offsets=offs, bottleneck=dims))
    assert cfg.use_layer_offsets and [(cfg.offset_left[i], cfg.offset_right[i]) for i in range(5)] == offs
    assert [cfg.bottleneck_dim[i] for i in range(5)] == dims


@pytest.mark.parametrize("case", G["darts"], ids=lambda c: "K%d" % c["K"])
def test_offset_supernet_config_written_by_the_reference_script(pkg, case):
    """generate_config.py's own final.config -> K taps, the flags of the TdnnDARTSV3Component lines"""
    lines = case["out"]["final.config"].split("\n")
    cfg = pkg.configs.net_config_from_final_config(lines)
    fl = case["flags"]
    want = sum(bit for key, bit in (("use-gumbel", 1), ("free-select", 2), ("uniform-sample", 4), ("use-entropy", 8), ("update-alpha", 16))
               if fl.get(key, "true") == "true")
    assert cfg.darts_num_offsets == case["K"] and cfg.darts_flags == want and cfg.num_layers == 14
    assert cfg.bottleneck_dim[0] == 160 and cfg.hidden_dim == 1536


def test_bottleneck_supernet_configs_written_by_the_reference_scripts(pkg):
    t = pkg.trainer
    cfg = pkg.configs.net_config_from_final_config(G["bottleneck"]["final.config"].split("\n"))
    assert cfg.bn_num_choices == 8 and [cfg.bn_choice_dims[k] for k in range(8)] == t.BN_CHOICE_DIMS and cfg.bn_mode == t.BN_ONEHOT
    assert all(cfg.bottleneck_dim[i] == 240 for i in range(14)) and [cfg.time_stride[i] for i in range(14)] == t.STRIDES_7Q
    b = G["bottleneck_offsets"]
    cfg = pkg.configs.net_config_from_final_config(b["out"]["final.config"].split("\n"))
    o = b["offsets"]
    assert cfg.use_layer_offsets and cfg.bn_num_choices == 8
    assert [(cfg.offset_left[i], cfg.offset_right[i]) for i in range(14)] == [(-o[2 * i], o[2 * i + 1]) for i in range(14)]


def test_graphs_the_trainer_does_not_run_are_refused(pkg):
    c = pkg.configs
    lines = c.final_config()
    # another wiring: tdnnf3 reads tdnn1 instead of tdnnf2
    bad = [l.replace("component-node name=tdnnf3.linear component=tdnnf3.linear input=tdnnf2.noop", "component-node name=tdnnf3.linear component=tdnnf3.linear input=tdnn1.dropout")
           for l in lines]
    assert bad != lines
    with pytest.raises(ValueError, match="not a graph the trainer runs"):
        c.net_config_from_final_config(bad)
    with pytest.raises(ValueError, match="unknown line type"):
        c.parse_config(lines + ["componentnode name=x"])
    with pytest.raises(ValueError, match="time-offsets"):
        c.net_config_from_final_config([l.replace("time-offsets=-1,0", "time-offsets=-1,1") for l in lines])
    with pytest.raises(ValueError, match="no component output.affine"):
        c.net_config_from_final_config([l for l in lines if "name=output.affine " not in l])


def test_parse_keeps_descriptors_whole(pkg):
    p = pkg.configs.parse_config(["component-node name=lda component=lda input=Append(Offset(input, -1), input, Offset(input, 1), ReplaceIndex(ivector, t, 0))",
                                  "component name=a type=TdnnComponent input-dim=4 output-dim=2 time-offsets=-1,0  # comment"])
    assert p["nodes"] == ["component-node name=lda component=lda input=Append(Offset(input, -1), input, Offset(input, 1), ReplaceIndex(ivector, t, 0))"]
    assert p["components"]["a"] == {"type": "TdnnComponent", "input-dim": "4", "output-dim": "2", "time-offsets": "-1,0"}


@pytest.mark.gpu
def test_nnet3_init_parameters_from_the_config(pkg):
    """InitFromConfig statistics (nnet-tdnn-component.cc:139-176): stddev 1/sqrt(input-dim x taps), bias stddev 1, logits 0,
    output layers zero (param-stddev=0 bias-stddev=0 in the recipes' output-layer lines); and the net trains from it."""
    import torch
    c, t = pkg.configs, pkg.trainer
    lines = G["darts"][0]["out"]["final.config"].split("\n")
    cfg = c.net_config_from_final_config(lines, frames_per_chunk=30, num_sequences=4)
    net = t.ChainNet(cfg)
    p = c.init_params_from_final_config(lines, net, seed=1)
    comps = {x["name"]: x for x in net.components}
    lin = comps["tdnnf5.linear"]
    K = cfg.darts_num_offsets
    W = p[lin["begin"]:lin["begin"] + lin["rows"] * lin["cols"]]
    assert lin["cols"] == K * 1536 and abs(W.std() * np.sqrt(K * 1536) - 1.0) < 0.02 and abs(W.mean()) < 1e-3
    n = lin["rows"] * lin["cols"]
    assert not p[lin["begin"] + n:lin["begin"] + n + K].any()  # architecture logits start at 0 (:176)
    assert abs(p[lin["begin"] + n + K:lin["begin"] + n + K + lin["rows"]].std() - 1.0) < 0.2
    out = comps["output.affine"]
    assert not p[out["begin"]:out["begin"] + out["rows"] * (out["cols"] + 1)].any()
    lda = comps["lda"]
    Q = p[lda["begin"]:lda["begin"] + lda["rows"] * lda["cols"]].reshape(lda["rows"], lda["cols"])
    assert np.allclose(Q @ Q.T, np.eye(lda["rows"]), atol=1e-4)
    net.set_params(p)
    feats, iv = t.synthetic_egs(net, seed=2)
    den = pkg.hipabi.DenGraph(pkg.synth.make_den_graph(60, cfg.num_pdfs, mean_out_degree=4.0, seed=5))
    sup = pkg.hipabi.Supervision(pkg.synth.make_supervision(4, 10, cfg.num_pdfs, seed=6))
    net.set_random_draws(np.random.default_rng(3).uniform(0.01, 0.99, net.num_draws).astype(np.float32))
    r = net.forward_backward(torch.from_numpy(feats).cuda(), torch.from_numpy(iv).cuda(), den, sup, step=0).cpu().numpy()
    assert r[5] == 1.0 and np.isfinite(r[0]) and float(net.grads.abs().sum()) > 0
    net.close()
