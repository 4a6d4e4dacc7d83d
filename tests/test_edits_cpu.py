"""CPU: the edit directives of ReadEditConfig (/root/reference/src/nnet3/nnet-utils.cc:1166-1415) as the trainer parses them, fed with
the very strings the reference's Python builds (temperature_schedule.py:51-60 through trainer.temperature_edit_string), and the
seven literal `sed`s of run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:128-142 on text."""
import pytest


def test_parse_edits_takes_the_commands_train_py_builds(pkg):
    T = pkg.trainer
    cmd = T.temperature_edit_string(0.25)
    assert cmd.startswith("nnet3-copy --edits='set-temperature-proportion name=* proportion=")
    (d, kv), = T.parse_edits(cmd)
    assert d == "set-temperature-proportion" and kv["name"] == "*" and abs(float(kv["proportion"]) - T.temperature_proportion(0.25)) < 1e-12
    # the option value itself; ';' separates directives (nnet3-copy replaces them by newlines), '#' starts a comment
    got = T.parse_edits("set-learning-rate-factor learning-rate-factor=0; set-learning-rate-factor name=tdnnf*.affine learning-rate-factor=0.5 # x\n"
                        "set-dropout-proportion name=* proportion=0.1")
    assert got == [("set-learning-rate-factor", {"learning-rate-factor": "0"}),
                   ("set-learning-rate-factor", {"name": "tdnnf*.affine", "learning-rate-factor": "0.5"}),
                   ("set-dropout-proportion", {"name": "*", "proportion": "0.1"})]
    # the recipe's own command line (…cvupdate.sh:129)
    assert T.parse_edits('nnet3-am-copy --raw --binary=false --edits="set-learning-rate-factor learning-rate-factor=0" final.mdl -') == \
        [("set-learning-rate-factor", {"learning-rate-factor": "0"})]
    with pytest.raises(ValueError, match="key=value"):
        T.parse_edits("set-learning-rate-factor oops")


def test_the_seven_seds_of_the_cvupdate_recipe(pkg):
    T = pkg.trainer
    block = ("<ComponentName> tdnnf2.linear <TdnnDARTSV3Component> <LearningRateFactor> 0 <MaxChange> 0.75 <L2Regularize> 0.01 <LearningRate> 0 "
             "<use-gumbel> F <use-entropy> F <free-select> F <update-alpha> F <update-theta> T <uniform-sample> T <Temp-Proportion> 1 \n"
             "<ComponentName> tdnn1.affine <NaturalGradientAffineComponent> <LearningRateFactor> 0 <MaxChange> 0.75 \n"
             "<ComponentName> tdnn1.batchnorm <BatchNormComponent> <Dim> 8 <BlockDim> 8 <Epsilon> 0.001 <TargetRms> 1 <TestMode> F <Count> 12 </BatchNormComponent> \n")
    out = T.apply_cvupdate_seds(block, use_gumbel=True)
    assert "<TdnnDARTSV3Component> <LearningRateFactor> 0.0001 <MaxChange>" in out
    assert "<NaturalGradientAffineComponent> <LearningRateFactor> 0 <MaxChange>" in out  # only the DARTS components are unfrozen
    assert "<use-gumbel> T" in out and "<update-alpha> T" in out and "<update-theta> F" in out and "<uniform-sample> F" in out
    assert "<BatchNormTestComponent> <Dim> 8" in out and "</BatchNormTestComponent>" in out and "<TestMode> T" in out
    assert "<use-gumbel> F" in T.apply_cvupdate_seds(block, use_gumbel=False)
