"""Config text (tdnn-f_nas_amd/configs.py) against the outputs of the reference's own config-rewriting scripts
(tests/golden/r01_configs_golden.json, made by tests/golden/make_configs_golden.py), and its consistency with the
derivation scripts' rewriting (derive.py) and with the graph the trainer writes into model files."""
import json
import os

import pytest

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "r01_configs_golden.json")))


def text(lines):
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("case", G["darts"], ids=lambda c: "K%d" % c["K"])
def test_offset_supernet_config(pkg, case):
    c = pkg.configs
    tmpl = c.final_config(strides=[6] * 14, darts=case["flags"])
    assert case["out"]["returncode"] == 0
    assert text(c.darts_supernet_config(tmpl, case["K"])) == case["out"]["final.config"]
    assert text(c.darts_supernet_config(c.ref_config(tmpl), case["K"])) == case["out"]["ref.config"]
    # what it does to the searched components: bias on, K taps on each side of 0
    lin = [l for l in case["out"]["final.config"].split("\n") if "name=tdnnf2.linear type=" in l][0]
    aff = [l for l in case["out"]["final.config"].split("\n") if "name=tdnnf2.affine type=" in l][0]
    K = case["K"]
    assert "use-bias=true" in lin and "time-offsets=" + ",".join(str(i) for i in range(-(K - 1), 1)) + " orthonormal-constraint=-1.0" in lin
    assert aff.endswith("time-offsets=" + ",".join(str(i) for i in range(K)))


def test_bottleneck_supernet_config(pkg):
    c = pkg.configs
    plain = c.final_config()
    assert G["bottleneck"]["returncode"] == 0
    assert text(c.bottleneck_supernet_config(plain)) == G["bottleneck"]["final.config"]
    b = G["bottleneck_offsets"]
    assert b["out"]["returncode"] == 0
    assert text(c.bottleneck_supernet_config(plain, b["offsets"])) == b["out"]["final.config"]
    assert text(c.bottleneck_supernet_config(c.ref_config(plain), b["offsets"])) == b["out"]["ref.config"]


@pytest.mark.parametrize("case", G["flops"], ids=lambda c: "gumbel-" + c["use_gumbel"])
def test_flops_constraint_change_config(pkg, case):
    assert text(pkg.configs.flops_constraint_change_config(case["use_gumbel"], float(case["coef"]))) == case["out"]["change.config"]


@pytest.mark.parametrize("case", G["sizes"], ids=lambda c: c["child_type"])
def test_bottleneck_top5_model_sizes(pkg, case):
    assert text(pkg.configs.bottleneck_top5_model_sizes(case["model"], case["child_type"])) == case["out"]["arch.txt"]


def test_child_config_is_the_template_rewritten(pkg):
    # final_config(layer_offsets=...) writes directly what generate_top_list.py makes of the stride-6 template
    c, d = pkg.configs, pkg.derive
    offsets = [-5, 0, -5, 2, 0, 2, -4, 0, -4, 4, -2, 2, -2, 2, -1, 5, 0, 6, -5, 2, 0, 1, -2, 1, -3, 1, -2, 3]
    kw = d.child_config_kwargs(offsets=offsets)
    assert d.rewrite_offsets_config(c.final_config(strides=[6] * 14), offsets) == [l.strip() for l in c.final_config(**kw)]  # (the scripts strip)
    dims = [160, 100, 100, 120, 160, 80, 240, 120, 25, 100, 240, 200, 120, 100]
    assert d.rewrite_bottleneck_config(c.final_config(), dims) == [l.strip() for l in c.final_config(bottleneck=dims)]


def test_trainer_graph_lines_follow_the_config(pkg):
    """The node lines the trainer's model writer emits (csrc/model_io.hip config_lines) are the node lines of these configs;
    compared through tdnnf_net_config_text, which needs no GPU."""
    c, t = pkg.configs, pkg.trainer
    cases = [
        (t.make_config(), c.final_config()),
        (t.make_config(strides=[6] * 14, darts_num_offsets=7), c.darts_supernet_config(c.final_config(strides=[6] * 14, darts={}), 7)),
        (t.make_config(bn_choice_dims=t.BN_CHOICE_DIMS, bn_mode=t.BN_ONEHOT), c.bottleneck_supernet_config(c.final_config())),
        (t.make_config(layer_offsets=[(2, 1), (0, 3)], bottleneck=[8, 16]), c.final_config(layer_offsets=[(2, 1), (0, 3)], bottleneck=[8, 16])),
    ]
    for cfg, lines in cases:
        assert t.config_text(cfg).split("\n") == c.node_lines(lines)
    # cv-update of the bottleneck search: nnet3-copy --nnet-config=change.config puts X.alpha in front of X.softmax
    cfg = t.make_config(bn_choice_dims=t.BN_CHOICE_DIMS, bn_mode=t.BN_GUMBEL_SOFTMAX_FLOPS, bn_flops_scale=0.05, cv_update=1)
    got = t.config_text(cfg).split("\n")
    change = c.node_lines(c.flops_constraint_change_config("true", 0.05))
    assert len(change) == 28 and all(l in got for l in change)
    assert "component-node name=tdnnf2.softmax component=tdnnf2.softmax input=lda" not in got
