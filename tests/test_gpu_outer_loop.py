"""-m gpu: the outer loop (tdnn-f_nas_amd/outer_loop.py) on a tiny net: one model file per iteration,
the jobs of an iteration averaged, final combination, reproducible."""
import os

import numpy as np
import pytest

from tests.gpu_util import dev

pytestmark = pytest.mark.gpu

KW = dict(frames_per_chunk=24, num_sequences=4, strides=[1, 0, 3], bottleneck=16, feat_dim=40, ivector_dim=100, num_pdfs=40, hidden_dim=64, small_dim=32,
          use_natural_gradient=1)


def _setup(pkg):
    cfg = pkg.trainer.make_config(**KW)

    def factory():
        net = pkg.trainer.ChainNet(cfg)
        net.set_params(net.init_params_numpy(seed=1, output_stddev=0.1))
        return net

    probe = factory()
    den = pkg.hipabi.DenGraph(pkg.synth.make_den_graph(30, cfg.num_pdfs, mean_out_degree=4.0, seed=5))
    cache = {}

    def egs(archive, m):  # a fixed set of "archives": the same few minibatches come back every epoch
        key = (archive % 2, m)
        if key not in cache:
            feats, iv = pkg.trainer.synthetic_egs(probe, seed=100 + 10 * key[0] + m)
            sup = pkg.synth.make_supervision(cfg.num_sequences, cfg.frames_per_chunk // 3, cfg.num_pdfs, seed=200 + 10 * key[0] + m)
            cache[key] = (dev(feats), dev(iv), pkg.hipabi.Supervision(sup))
        f, i, s = cache[key]
        return f, i, den, s

    return cfg, factory, egs, probe


def test_outer_loop_runs_the_schedule(pkg, tmp_path):
    cfg, factory, egs, probe = _setup(pkg)
    o = pkg.outer_loop
    args = dict(num_epochs=3, num_archives=2, minibatches_per_archive=3, frame_subsampling_factor=3, num_jobs_initial=1, num_jobs_final=2,
                initial_effective_lrate=2e-3, final_effective_lrate=2e-4, max_models_combine=4)
    plan, combine, to_process = o.run(factory, egs, str(tmp_path / "a"), **args)
    assert to_process == 18 and len(plan) == 12 and combine == [9, 10, 11, 12]
    assert all(os.path.exists(tmp_path / "a" / ("%d.mdl" % i)) for i in range(13)) and os.path.exists(tmp_path / "a" / "final.mdl")
    objf = [p["objf_per_frame"] for p in plan]
    assert all(np.isfinite(objf)) and np.mean(objf[-3:]) > np.mean(objf[:3])  # it learns the repeated archives
    # final.mdl is the average of the combined models, statistics included
    models = []
    for i in combine:
        probe.read_model(tmp_path / "a" / ("%d.mdl" % i))
        models.append((probe.params.detach().cpu().numpy().copy(), probe.get_stats().copy()))
    p, s = o.average_models(models)
    probe.read_model(tmp_path / "a" / "final.mdl")
    np.testing.assert_allclose(probe.params.detach().cpu().numpy(), p, rtol=0, atol=1e-7)
    np.testing.assert_allclose(probe.get_stats(), s, rtol=1e-5)
    # an iteration with two jobs wrote the mean of two different models: neither job's model alone
    two = next(p for p in plan if p["num_jobs"] == 2)
    assert two["archives"][0] != two["archives"][1]
    # reproducible: the same call again gives the same final model
    o.run(factory, egs, str(tmp_path / "b"), **args)
    q = factory()
    q.read_model(tmp_path / "b" / "final.mdl")
    assert np.array_equal(q.params.detach().cpu().numpy(), probe.params.detach().cpu().numpy())
    q.close()
    probe.close()


def _rank_main(rank, world, port, work_dir):
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    cfg, factory, egs, probe = _setup(pkg)
    probe.close()
    pkg.outer_loop.run(factory, egs, work_dir, **ARGS2)
    dist.destroy_process_group()


ARGS2 = dict(num_epochs=2, num_archives=2, minibatches_per_archive=2, frame_subsampling_factor=3, num_jobs_initial=2, num_jobs_final=2,
             initial_effective_lrate=2e-3, final_effective_lrate=2e-4, max_models_combine=4)


def test_outer_loop_two_ranks_match_one(pkg, tmp_path):
    """Two processes (both on this GPU, gloo) share the two jobs of every iteration: the averaged models are the ones a
    single process computes by running the jobs one after the other."""
    import socket
    import torch.multiprocessing as mp
    cfg, factory, egs, probe = _setup(pkg)
    pkg.outer_loop.run(factory, egs, str(tmp_path / "one"), **ARGS2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path / "two")), nprocs=2, join=True)
    probe.read_model(tmp_path / "one" / "final.mdl")
    a, sa = probe.params.detach().cpu().numpy().copy(), probe.get_stats().copy()
    probe.read_model(tmp_path / "two" / "final.mdl")
    b, sb = probe.params.detach().cpu().numpy(), probe.get_stats()
    np.testing.assert_allclose(b, a, rtol=0, atol=1e-6)
    np.testing.assert_allclose(sb, sa, rtol=1e-5)
    probe.close()


def test_outer_loop_on_the_offset_supernet_with_temperature_schedule(pkg, tmp_path):
    """The search stage: Gumbel coefficients need random draws every minibatch and the temperature proportion of the iteration."""
    T = pkg.trainer
    cfg = T.make_config(**dict(KW, use_natural_gradient=0, darts_num_offsets=3, darts_flags=T.DARTS_USE_GUMBEL | T.DARTS_UPDATE_ALPHA, use_dropout=1))

    def factory():
        net = T.ChainNet(cfg)
        net.set_params(net.init_params_numpy(seed=1, output_stddev=0.1))
        return net

    probe = factory()
    den = pkg.hipabi.DenGraph(pkg.synth.make_den_graph(30, cfg.num_pdfs, mean_out_degree=4.0, seed=5))
    feats, iv = T.synthetic_egs(probe, seed=3)
    sup = pkg.hipabi.Supervision(pkg.synth.make_supervision(cfg.num_sequences, cfg.frames_per_chunk // 3, cfg.num_pdfs, seed=4))
    fd, ivd = dev(feats), dev(iv)
    plan, combine, _ = pkg.outer_loop.run(factory, lambda a, m: (fd, ivd, den, sup), str(tmp_path / "s"), num_epochs=1, num_archives=2, minibatches_per_archive=2,
                                          num_jobs_initial=1, num_jobs_final=1, temperature_schedule=True, initial_effective_lrate=1e-3, final_effective_lrate=1e-4,
                                          dropout_schedule='0,0@0.20,0.5@0.50,0')
    assert len(plan) == 6 and plan[0]["temperature_proportion"] == 1.0 and plan[-1]["temperature_proportion"] < 0.3
    assert all(np.isfinite(p["objf_per_frame"]) for p in plan) and max(p["dropout_proportion"] for p in plan) == 0.5
    a0 = pkg.derive.logits_from_net(probe)
    probe.read_model(tmp_path / "s" / "final.mdl")
    a1 = pkg.derive.logits_from_net(probe)
    assert a0.shape == (6, 3) and np.abs(a1 - a0).max() > 0  # the architecture logits moved
    probe.close()


def test_compute_prob_and_combination_by_objective(pkg, tmp_path):
    """Diagnostics on every iteration's model (test-mode BatchNorm, no update) and the final combination chosen by the objective
    on combine egs (running average, latest model first), BatchNorm statistics recomputed on those egs."""
    cfg, factory, egs, probe = _setup(pkg)
    o = pkg.outer_loop
    held_out = [egs(0, 0), egs(1, 1)]
    logs = []
    args = dict(num_epochs=2, num_archives=2, minibatches_per_archive=3, frame_subsampling_factor=3, num_jobs_initial=1, num_jobs_final=1,
                initial_effective_lrate=2e-3, final_effective_lrate=2e-4, max_models_combine=4)
    plan, combine, _ = o.run(factory, egs, str(tmp_path / "c"), combine_egs=held_out, diagnostic_egs={"valid": held_out}, log=logs.append, **args)
    # compute_prob leaves the model alone and is repeatable
    ev = pkg.trainer.ChainNet(o.evaluation_config(cfg, True))
    probe.read_model(tmp_path / "c" / "3.mdl")
    ev.params.copy_(probe.params)
    ev.set_stats(probe.get_stats())
    before = ev.params.detach().cpu().numpy().copy()
    a, b = o.compute_prob(ev, held_out), o.compute_prob(ev, held_out)
    assert a == b and np.array_equal(ev.params.detach().cpu().numpy(), before)
    assert abs(a["output"] - plan[3]["compute_prob"]["valid"]["output"]) < 1e-9 and a["weight"] == 2 * cfg.num_sequences * (cfg.frames_per_chunk // 3)
    valid = [p["compute_prob"]["valid"]["output"] for p in plan]
    assert all(np.isfinite(valid)) and valid[-1] > valid[0]  # the held-out minibatches are training data here
    ev.close()
    # the combination: redo it by hand from the model files
    models = []
    for i in sorted(combine, reverse=True):
        probe.read_model(tmp_path / "c" / ("%d.mdl" % i))
        models.append((probe.params.detach().cpu().numpy().astype(np.float64), np.asarray(probe.get_stats(), np.float64)))
    comb = pkg.trainer.ChainNet(o.evaluation_config(cfg, False))
    best = None
    for n in range(1, len(models) + 1):
        comb.set_params(np.mean([m[0] for m in models[:n]], axis=0).astype(np.float32))
        comb.set_stats(np.mean([m[1] for m in models[:n]], axis=0))
        t = np.zeros(3)
        for m, (f, iv, den, sup) in enumerate(held_out):
            t += comb.forward_backward(f, iv, den, sup, step=m).cpu().numpy()[:3]
        objf = (t[0] + t[1]) / t[2]
        if best is None or objf > best[0]:
            best = (objf, n)
    line = [x for x in logs if x.startswith("Combining %d nnets" % best[1])]
    assert line, logs
    comb.set_params(np.mean([m[0] for m in models[:best[1]]], axis=0).astype(np.float32))
    comb.set_stats(np.zeros_like(models[0][1]))
    for m, (f, iv, den, sup) in enumerate(held_out):
        comb.forward_backward(f, iv, den, sup, step=m)
    probe.read_model(tmp_path / "c" / "final.mdl")
    np.testing.assert_allclose(probe.params.detach().cpu().numpy(), comb.params.detach().cpu().numpy(), rtol=0, atol=1e-7)
    np.testing.assert_allclose(probe.get_stats(), comb.get_stats(), rtol=1e-5, atol=1e-7)
    comb.close()
    probe.close()


def test_shrink_scales_parameters_and_statistics(pkg, tmp_path):
    """nnet3-am-copy --scale = ScaleNnet: statistics sums and counts shrink with the parameters (nnet-component-itf.cc:533-541,
    nnet-normalize-component.cc:644-654).  One iteration of one minibatch at shrink value 0.75 from a model with statistics on board: a count in the
    written model is (0.75 x old + frames) [x 0.8 for BatchNorm, ScaleBatchnormStats] against (old + frames) [x 0.8] of the unshrunk run."""
    cfg, factory, egs, probe = _setup(pkg)
    o = pkg.outer_loop
    base = dict(num_epochs=1, num_archives=1, minibatches_per_archive=1, frame_subsampling_factor=1, num_jobs_initial=1, num_jobs_final=1,
                initial_effective_lrate=1e-3, final_effective_lrate=1e-3, max_models_combine=1)
    o.run(factory, egs, str(tmp_path / "seed"), **base)  # its 1.mdl carries one minibatch of statistics

    def seeded():
        net = factory()
        net.read_model(tmp_path / "seed" / "1.mdl")
        return net

    probe.read_model(tmp_path / "seed" / "1.mdl")
    old = probe.get_stats().copy()
    assert old[0] > 0  # the first entry of the block is a count (include/tdnnf_hip.h, tdnnf_net_get_stats)
    out = {}
    for name, shrink in (("plain", 0.0), ("shrunk", 250.0)):  # shrink value 1 - 250 x 1e-3 = 0.75
        plan, _, _ = o.run(seeded, egs, str(tmp_path / name), proportional_shrink=shrink, **base)
        assert plan[0]["shrink"] == pytest.approx(1.0 if shrink == 0 else 0.75)
        probe.read_model(tmp_path / name / "1.mdl")
        out[name] = (probe.params.detach().cpu().numpy().copy(), probe.get_stats().copy())
    assert not np.array_equal(out["plain"][0], out["shrunk"][0])
    d = (out["plain"][1][0] - out["shrunk"][1][0]) / old[0]
    assert d == pytest.approx(0.25, rel=1e-5) or d == pytest.approx(0.2, rel=1e-5), d
    probe.close()
