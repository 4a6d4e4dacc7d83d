"""-m gpu: the search -> derive -> child pipeline on the trainer.  The derivation itself is pinned on
CPU against the reference scripts' outputs (tests/test_derive_child.py); here: the text model THIS trainer writes is what
those scripts parse, and the derived child is a net the trainer builds and steps."""
import numpy as np
import pytest

from tests.gpu_util import dev, host

pytestmark = pytest.mark.gpu

TINY = dict(frames_per_chunk=24, num_sequences=2, strides=[1] * 14, bottleneck=8, feat_dim=40, ivector_dim=100, num_pdfs=30, hidden_dim=32, small_dim=16)


def _step(pkg, net, cfg, step=0):
    feats, iv = pkg.trainer.synthetic_egs(net, seed=4)
    den = pkg.synth.make_den_graph(20, cfg.num_pdfs, mean_out_degree=4.0, seed=5)
    sup = pkg.synth.make_supervision(cfg.num_sequences, cfg.frames_per_chunk // 3, cfg.num_pdfs, seed=6)
    net.set_random_draws(np.random.default_rng(9).uniform(0.01, 0.99, max(net.num_draws, 1)).astype(np.float32))
    return host(net.forward_backward(dev(feats), dev(iv), pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup), step=step))


def test_offset_search_to_child(pkg, tmp_path):
    K = 5
    cfg = pkg.trainer.make_config(darts_num_offsets=K, darts_flags=pkg.trainer.DARTS_USE_GUMBEL | pkg.trainer.DARTS_UPDATE_ALPHA, **TINY)
    net = pkg.trainer.ChainNet(cfg)
    params = net.init_params_numpy(seed=2, output_stddev=0.3)
    rng = np.random.default_rng(3)
    for c in net.components:
        n = c["rows"] * c["cols"]
        params[c["begin"] + n:c["begin"] + n + c["num_alpha"]] = np.round(rng.standard_normal(c["num_alpha"]), 3).astype(np.float32)
    net.set_params(params)
    path = tmp_path / "final_txt.mdl"
    net.write_model(path, binary=False)  # nnet3-am-copy --binary=false of the recipes
    lines = open(path).read().split("\n")
    logits = pkg.derive.offset_logits(lines, K, "tdnn")  # the script's view of the file: 3rd..30th <BiasParams> row
    assert logits.shape == (28, K)
    np.testing.assert_allclose(logits, pkg.derive.logits_from_net(net), rtol=0, atol=1e-6)
    assert np.abs(logits).max() > 0.5
    chosen, offsets = pkg.derive.derive_offset_child(lines, "top", 1, K)
    assert [j for _, j in chosen] == list(np.argmax(pkg.derive.choice_probabilities(logits, "top"), axis=1))  # best path = row-wise arg max
    kw = pkg.derive.child_config_kwargs(offsets=offsets)
    net.close()
    child_cfg = pkg.trainer.make_config(**dict(TINY, strides=None, **kw))
    child = pkg.trainer.ChainNet(child_cfg)
    child.set_params(child.init_params_numpy(seed=5, output_stddev=0.3))
    for i, (a, b) in enumerate(kw["layer_offsets"]):  # the child's components have the taps that were chosen
        lin = next(c for c in child.components if c["name"] == f"tdnnf{i + 2}.linear")
        aff = next(c for c in child.components if c["name"] == f"tdnnf{i + 2}.affine")
        assert lin["cols"] == (2 if a else 1) * 32 and aff["cols"] == (2 if b else 1) * 8 and not lin["num_alpha"]
    r = _step(pkg, child, child_cfg)
    assert r[5] == 1.0 and np.isfinite(r[0]) and host(child.grads).any()
    # ... and its model file says so (what generate_top_list.py writes into final.config)
    cpath = tmp_path / "child.mdl"
    child.write_model(cpath, binary=False)
    text = open(cpath).read()
    want = pkg.derive.rewrite_offsets_config(["component name=tdnnf%d.%s type=TdnnComponent time-offsets=0,0 x=y" % (i // 2 + 2, "linear" if i % 2 == 0 else "affine")
                                              for i in range(28)], offsets)
    for i, line in enumerate(want):
        offs = line.split("time-offsets=")[1].split(" ")[0].replace(",", " ")
        name = "tdnnf%d.%s" % (i // 2 + 2, "linear" if i % 2 == 0 else "affine")
        seg = text.split("<ComponentName> " + name + " ")[1].split("</TdnnComponent>")[0]
        assert "<TimeOffsets> [ " + offs + " ]" in seg, (name, offs)
    c2 = pkg.trainer.config_from_model(cpath, child_cfg.frames_per_chunk, child_cfg.num_sequences)
    assert [(c2.offset_left[i], c2.offset_right[i]) for i in range(14)] == kw["layer_offsets"] or not child_cfg.use_layer_offsets
    child.close()


def test_bottleneck_search_to_child(pkg, tmp_path):
    dims = [2, 2, 4, 8]  # candidate bottleneck dims 2, 4, 8, 16 as block widths
    cand = list(np.cumsum(dims))
    cfg = pkg.trainer.make_config(bn_choice_dims=dims, bn_mode=pkg.trainer.BN_SOFTMAX_FLOPS, bn_flops_scale=1.0, cv_update=1, **TINY)
    net = pkg.trainer.ChainNet(cfg)
    params = net.init_params_numpy(seed=2, output_stddev=0.3)
    rng = np.random.default_rng(4)
    for c in net.components:
        if c["name"].endswith(".alpha"):
            params[c["begin"]:c["begin"] + c["rows"]] = np.round(rng.standard_normal(c["rows"]), 3).astype(np.float32)
    net.set_params(params)
    net.set_stats(np.random.default_rng(5).random(net.get_stats().size) * 10 + 5)
    path = tmp_path / "final_txt.mdl"
    net.write_model(path, binary=False)
    lines = open(path).read().split("\n")
    logits = pkg.derive.bottleneck_logits(lines, len(dims), "tdnn")  # 'X.alpha <ConstantFunctionComponent>' lines
    np.testing.assert_allclose(logits, pkg.derive.logits_from_net(net), rtol=0, atol=1e-6)
    chosen, layer_dims = pkg.derive.derive_bottleneck_child(lines, "top", 2, len(dims), "tdnn", dims=cand)
    assert len(layer_dims) == 14 and set(layer_dims) <= set(cand)
    net.close()
    child_cfg = pkg.trainer.make_config(**dict(TINY, **pkg.derive.child_config_kwargs(layer_dims=layer_dims)))
    child = pkg.trainer.ChainNet(child_cfg)
    child.set_params(child.init_params_numpy(seed=5, output_stddev=0.3))
    assert [c["rows"] for c in child.components if c["name"].endswith(".linear") and c["name"].startswith("tdnnf")] == layer_dims
    r = _step(pkg, child, child_cfg)
    assert r[5] == 1.0 and np.isfinite(r[0])
    child.close()
