"""CPU: the reader side of the nnet3 model I/O (csrc/model_io.hip) on the committed text fixture
tests/golden/r01_tiny_model.txt (generator: tests/golden/make_model_fixture.py) -- tdnnf_net_config_from_model needs
no GPU.  Also a hand-edited Kaldi-style variant (6-digit floats, other line breaks) and error paths."""
import os
import re

import pytest

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r01_tiny_model.txt")


def test_config_from_text_model(pkg):
    c = pkg.trainer.config_from_model(FIX, frames_per_chunk=30, num_sequences=5)
    assert (c.feat_dim, c.ivector_dim, c.num_pdfs, c.hidden_dim, c.prefinal_small_dim, c.num_layers) == (8, 4, 10, 16, 8, 3)
    assert list(c.bottleneck_dim[:3]) == [4, 8, 4] and list(c.time_stride[:3]) == [1, 0, 3]
    assert (c.frames_per_chunk, c.num_sequences, c.frame_subsampling) == (30, 5, 3)
    assert abs(c.bypass_scale - 0.66) < 1e-6 and abs(c.l2_hidden - 0.01) < 1e-7 and abs(c.l2_output - 0.002) < 1e-7
    assert abs(c.max_change_hidden - 0.75) < 1e-6 and abs(c.max_change_output - 1.5) < 1e-6
    assert abs(c.xent_regularize - 0.1) < 1e-6 and abs(c.relu_self_repair_scale - 2e-5) < 1e-9
    assert (c.darts_num_offsets, c.bn_num_choices, c.cv_update, c.use_natural_gradient) == (0, 0, 0, 1)


def test_config_from_kaldi_style_text(pkg, tmp_path):
    text = open(FIX).read()
    head, comps = text.split("\n\n", 1)
    comps = re.sub(r"-?\d+\.\d+(e[-+]?\d+)?", lambda m: "%g" % float(m.group(0)), comps)
    comps = comps.replace("<BatchNormComponent>", "<BatchNormTestComponent>").replace("</BatchNormComponent>", "</BatchNormTestComponent>")
    comps = comps.replace("<TestMode> F", "<TestMode> T")  # the sed of the cv-update recipes (…cvupdate.sh:134-135)
    p = tmp_path / "cv.txt"
    p.write_text(head + "\n\n" + comps)
    c = pkg.trainer.config_from_model(p, 12, 2)
    assert c.cv_update == 1 and c.num_layers == 3 and c.hidden_dim == 16


def test_config_from_model_errors(pkg, tmp_path):
    with pytest.raises(pkg.hipabi.HipAbiError, match="cannot open"):
        pkg.trainer.config_from_model(tmp_path / "nope.txt")
    text = open(FIX).read()
    p = tmp_path / "bad.txt"
    p.write_text(text.replace("<LinearParams>", "<LinearParamz>", 1))
    with pytest.raises(pkg.hipabi.HipAbiError, match="unknown token"):
        pkg.trainer.config_from_model(p)
    p.write_text(text[:len(text) // 3])
    with pytest.raises(pkg.hipabi.HipAbiError):
        pkg.trainer.config_from_model(p)
    p.write_text(re.sub(r"<ComponentName> tdnnf2\.affine ", "<ComponentName> tdnnfX.affine ", text))
    with pytest.raises(pkg.hipabi.HipAbiError, match="no tdnnf2"):
        pkg.trainer.config_from_model(p)


def test_config_from_child_model(pkg, tmp_path):
    # a derived child (generate_top_list.py:97-141): asymmetric time offsets come back as layer offsets
    text = open(FIX).read()
    assert pkg.trainer.config_from_model(FIX).use_layer_offsets == 0
    p = tmp_path / "child.txt"
    p.write_text(text.replace("<TimeOffsets> [ -1 0 ]", "<TimeOffsets> [ -2 0 ]").replace("<TimeOffsets> [ 0 3 ]", "<TimeOffsets> [ 0 5 ]"))
    c = pkg.trainer.config_from_model(p, 12, 2)
    assert c.use_layer_offsets == 1 and list(c.offset_left[:3]) == [2, 0, 3] and list(c.offset_right[:3]) == [1, 0, 5]
    assert list(c.time_stride[:3]) == [2, 0, 5]
    p.write_text(text.replace("<TimeOffsets> [ -1 0 ]", "<TimeOffsets> [ 0 1 ]"))
    with pytest.raises(pkg.hipabi.HipAbiError, match="time offsets"):
        pkg.trainer.config_from_model(p)
