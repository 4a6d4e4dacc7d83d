"""Child derivation (tdnn-f_nas_amd/derive.py) against the outputs of the reference's own scripts
(tests/golden/r01_derive_golden.json, made by tests/golden/make_derive_golden.py running
local/chain_NAS/scripts/generate_top_list.py, generate_top_list_bottleneckdim.py and generate_optimal_stride.py)."""
import ast
import json
import os

import numpy as np
import pytest

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "r01_derive_golden.json")))
CASES = G["cases"]


def _ids():
    return ["%s-%s-%s" % (c["kind"], c.get("child_type", ""), c.get("top_id", "")) + ("-K%d" % c["K"] if "K" in c else "") + "-%d" % i
            for i, c in enumerate(CASES)]


@pytest.mark.parametrize("case", CASES, ids=_ids())
def test_matches_reference_script_outputs(pkg, case):
    d = pkg.derive
    out = case["out"]
    final_t, ref_t = G["final_temp"], G["ref_temp"]
    if case["kind"] == "optimal_stride":
        assert out["returncode"] == 0
        assert "\n".join(d.rewrite_offsets_config(final_t, case["offsets"], tdnn_only=False)) + "\n" == out["final.config"]
        assert "\n".join(d.rewrite_offsets_config(ref_t, case["offsets"], tdnn_only=False)) + "\n" == out["ref.config"]
        return
    if out["returncode"] != 0:  # the script fails when the beam holds fewer than top_id paths (all-equal logits)
        assert "IndexError" in out["error"]
        with pytest.raises(IndexError):
            if case["kind"] == "offset":
                d.derive_offset_child(case["model"], case["child_type"], case["top_id"], case["K"])
            else:
                d.derive_bottleneck_child(case["model"], case["child_type"], case["top_id"])
        return
    ref_path = ast.literal_eval(out["stdout"][0])
    if case["kind"] == "offset":
        path, offsets = d.derive_offset_child(case["model"], case["child_type"], case["top_id"], case["K"])
        assert path == ref_path
        assert out["stdout"][1] == "%s %s" % (case["child_type"], offsets)
        assert d.arch_txt_offsets(offsets) == out["arch.txt"]
        assert "\n".join(d.rewrite_offsets_config(final_t, offsets)) + "\n" == out["final.config"]
        assert "\n".join(d.rewrite_offsets_config(ref_t, offsets)) + "\n" == out["ref.config"]
        kw = d.child_config_kwargs(offsets=offsets)
        assert len(kw["layer_offsets"]) == 14 and all(a >= 0 and b >= 0 for a, b in kw["layer_offsets"])
    else:
        path, dims = d.derive_bottleneck_child(case["model"], case["child_type"], case["top_id"])
        assert path == ref_path
        assert out["stdout"][1] == "%s %s" % (case["child_type"], dims)
        assert d.arch_txt_bottleneck(dims) == out["arch.txt"]
        assert "\n".join(d.rewrite_bottleneck_config(final_t, dims)) + "\n" == out["final.config"]
        assert "\n".join(d.rewrite_bottleneck_config(ref_t, dims)) + "\n" == out["ref.config"]
        assert d.child_config_kwargs(layer_dims=dims)["bottleneck"] == dims


def test_beam_search_against_exhaustive_enumeration(pkg):
    # on a problem small enough to enumerate, with a beam wide enough not to prune, the beam is the exact top list
    rng = np.random.default_rng(3)
    prob = pkg.derive.choice_probabilities(rng.standard_normal((4, 3)), "top")
    paths = pkg.derive.beam_paths(prob, beam=81)
    scores = {}
    for a in range(3):
        for b in range(3):
            for c in range(3):
                for e in range(3):
                    scores[(a, b, c, e)] = float(prob[0, a]) * float(prob[1, b]) * float(prob[2, c]) * float(prob[3, e])
    best = sorted(scores.items(), key=lambda kv: -kv[1])
    assert len(paths) == 81
    for (s, p), (k, v) in zip(paths[:10], best[:10]):
        assert tuple(j for _, j in p) == k and abs(s - v) < 1e-12
    # 'last' ranks the least likely choices first
    last = pkg.derive.choice_probabilities(np.log(prob), "last")
    assert (np.argmax(last, axis=1) == np.argmin(prob, axis=1)).all()


def test_equal_logits_collapse_to_one_path(pkg):
    # the scripts key their beam by score: untrained (all-equal) logits leave exactly one path, the last choice everywhere
    paths = pkg.derive.beam_paths(pkg.derive.choice_probabilities(np.zeros((6, 4)), "top"))
    assert len(paths) == 1 and [j for _, j in paths[0][1]] == [3] * 6
