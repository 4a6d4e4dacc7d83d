"""CPU (gloo, world_size 2) coverage of the data-parallel path: sequences sharded over ranks, raw gradients
accumulated per shard with the CPU reference step, ONE all-reduce (trainer.allreduce_flat), identical update
on every rank.  Checks that the reduced gradient equals the sum of the shard gradients and that all ranks end
with the same parameters."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from tests.test_oracle_net import tiny_setup
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B_global, T = 4, 12
        b0, b1 = pkg.trainer.shard_sequences(B_global, rank, world)
        # every rank builds the same global minibatch and keeps its own sequences (t-major rows: column slice of [t, b])
        cfg, comps, params, net, feats, iv, den, sup_g = tiny_setup(pkg, T=T, B=b1 - b0, seed=7)
        rng = np.random.default_rng(123)
        num_t = net.num_t_in
        feats_g = rng.standard_normal((num_t, B_global, 8)).astype(np.float32)
        iv_g = rng.standard_normal((B_global, 4)).astype(np.float32)
        feats_l = np.ascontiguousarray(feats_g[:, b0:b1]).reshape(num_t * (b1 - b0), 8)
        sup = pkg.synth.make_supervision(b1 - b0, T // 3, 24, seed=50 + rank)
        res, grads, _ = net.forward_backward(params, feats_l, np.ascontiguousarray(iv_g[b0:b1]), den, sup, step=0)
        local = grads.copy()
        flat = torch.from_numpy(grads)
        pkg.trainer.allreduce_flat(flat)
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(local))
        expect = sum(g.numpy().astype(np.float64) for g in gathered)
        assert np.allclose(flat.numpy(), expect, rtol=1e-5, atol=1e-6)
        # identical update everywhere: the SUMMED gradient takes the effective learning rate (see
        # test_summed_gradient_step_is_kaldis_job_average), l2 scale = local sequences
        lr = pkg.trainer.learning_rate(0, 1, 10, 0, 10)
        p2 = net.update(params, flat.numpy(), lr, float(b1 - b0), step=0)
        ps = [torch.zeros(len(p2)) for _ in range(world)]
        dist.all_gather(ps, torch.from_numpy(p2))
        assert torch.equal(ps[0], ps[1])
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.asarray([res["objf"]]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_gloo(tmp_path, pkg):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}.npy") for r in range(world))


def test_shard_sequences_partitions(pkg):
    for B, W in [(128, 8), (7, 2), (5, 8), (64, 1)]:
        spans = [pkg.trainer.shard_sequences(B, r, W) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == B
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_learning_rate_schedule(pkg):
    lr = pkg.trainer.learning_rate
    assert abs(lr(0, 1, 100, 0, 100) - 2.5e-4) < 1e-12
    assert abs(lr(99, 1, 100, 99, 100) - 2.5e-5) < 1e-12          # last iteration: final rate
    assert abs(lr(50, 1, 100, 50, 100) - 2.5e-4 * np.sqrt(0.1)) < 1e-9
    assert lr(10, 8, 100, 10, 100) == 8 * lr(10, 1, 100, 10, 100)  # x num_jobs


def test_summed_gradient_step_is_kaldis_job_average(pkg):
    """Kaldi trains num_jobs jobs from the same model, each at learning rate lr_eff x num_jobs on its own minibatch with
    l2_regularize_factor = 1 / num_jobs, and averages the models (train.py / common.py:618).  The data-parallel step here sums
    the jobs' raw gradients and applies lr_eff once with the local sequence count as l2 scale: the same model, exactly, as long
    as max-change does not bind and no orthonormal step is scheduled (both are nonlinear in the step)."""
    from tests.oracle_net import decision
    from tests.test_oracle_net import tiny_setup
    J, T = 2, 12
    cfg, comps, params, net, feats, iv, den, sup = tiny_setup(pkg, T=T, B=2, seed=7)
    step = next(s for s in range(1000) if all(c["orthonormal"] == 0.0 or decision(s, 2 * i + 1) % 4 != 0 for i, c in enumerate(comps)))
    rng = np.random.default_rng(5)
    lr_eff = 1e-4  # small: max-change stays out of it
    job_models, grads = [], []
    for j in range(J):
        f = rng.standard_normal(feats.shape).astype(np.float32)
        v = rng.standard_normal(iv.shape).astype(np.float32)
        sp = pkg.synth.make_supervision(2, T // 3, 24, seed=60 + j)
        _, g, _ = net.forward_backward(params, f, v, den, sp, step=step)
        grads.append(g.copy())
        job_models.append(net.update(params, g, lr_eff * J, 2.0 / J, step))  # the job's own step: lr x J, l2 factor 1/J
    averaged = np.mean(job_models, axis=0)
    summed = net.update(params, np.sum(grads, axis=0), lr_eff, 2.0, step)  # what every rank does after allreduce_flat
    assert np.abs(averaged - params).max() > 0
    d1, d2 = (summed - params).astype(np.float64), (averaged - params).astype(np.float64)
    assert np.linalg.norm(d1 - d2) < 1e-4 * np.linalg.norm(d2)  # float32 rounding of the parameters only
    # ... and NOT the step the previous round took (sum applied with lr_eff x num_jobs): that one is num_jobs times too long
    too_long = net.update(params, np.sum(grads, axis=0), lr_eff * J, 2.0, step)
    assert np.linalg.norm(too_long - params) > 1.9 * np.linalg.norm(averaged - params)
