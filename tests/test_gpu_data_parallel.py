"""-m gpu: the data-parallel step on the HIP trainer itself, two processes on this one GPU (gloo, as bench.py's rehearsal
mode): SURVEY.md 8(e) "N-rank raw-gradient step == 1-rank step with the same global batch".

Sequences are independent through everything except train-mode BatchNorm statistics and the natural-gradient state, so the
exact check uses a net whose BatchNorm runs in test mode and that still forms gradients: the offset supernet in cv-update
mode (BatchNormTest from stored statistics, TdnnDARTSV3 components at learning-rate factor 1e-4, Gumbel coefficients from
shared draws).  Rank g takes sequences [4g, 4g + 4) of an 8-sequence minibatch (trainer.shard_rows / shard_supervision);
the summed gradient -- reduced per bucket behind the library's "bucket final" events, and as one flat buffer -- must be the
single process's gradient of all 8, and the update the same parameters."""
import os
import socket

import numpy as np
import pytest
import torch

from tests.gpu_util import dev, host, rel_l2

pytestmark = pytest.mark.gpu

KW = dict(frames_per_chunk=24, num_sequences=8, strides=[1, 1, 1], bottleneck=16, feat_dim=40, ivector_dim=100, num_pdfs=96, hidden_dim=64,
          small_dim=32, darts_num_offsets=3, darts_flags=1 | 16, darts_temp_proportion=0.8, cv_update=1,
          relu_self_repair_scale=0.0)  # (self-repair looks at ReLU statistics that include the shard just seen)


def _problem(pkg, B):
    T = pkg.trainer
    cfg = T.make_config(**dict(KW, num_sequences=B))
    net = T.ChainNet(cfg)
    full = T.ChainNet(T.make_config(**KW)) if B != KW["num_sequences"] else net
    rng = np.random.default_rng(3)
    params = full.init_params_numpy(seed=1, output_stddev=0.3)
    for c in full.components:  # trained-looking architecture logits
        n = c["rows"] * c["cols"]
        params[c["begin"] + n:c["begin"] + n + c["num_alpha"]] = rng.standard_normal(c["num_alpha"]).astype(np.float32) * 0.5
    stats = np.abs(rng.standard_normal(int(full.lib.tdnnf_net_stats_size(full.h)))) + 0.5
    # valid BatchNorm / ReLU statistics: [count, sum[D], sumsq[D]] with sumsq/count > (sum/count)^2
    o = 0
    Hd, S = cfg.hidden_dim, cfg.prefinal_small_dim
    dims = [(Hd, 0), (Hd, 1)] + [(Hd, 0), (Hd, 1)] * cfg.num_layers + [(Hd, 0), (Hd, 1), (S, 0)] * 2  # (dimension, is a ReLU block)
    for D, relu in dims:
        stats[o] = 100.0
        mean = rng.standard_normal(D) * 0.3
        var = rng.uniform(0.5, 1.5, D)
        stats[o + 1:o + 1 + D] = 100.0 * mean
        stats[o + 1 + D:o + 1 + 2 * D] = 100.0 * (var + mean * mean)
        o += 1 + 2 * D + ((1 + D) if relu else 0)  # a ReLU block also carries [oderiv_count, oderiv_sumsq[D]]
    assert o == len(stats)
    feats, iv = T.synthetic_egs(full, seed=4)
    den = pkg.synth.make_den_graph(40, cfg.num_pdfs, mean_out_degree=4.0, seed=5)
    sup = pkg.synth.make_supervision(KW["num_sequences"], cfg.frames_per_chunk // 3, cfg.num_pdfs, seed=6)
    draws = np.random.default_rng(9).uniform(0.05, 0.95, full.num_draws).astype(np.float32)
    if full is not net:
        full.close()
    net.set_params(params)
    net.set_stats(stats)
    net.set_random_draws(draws)  # every rank: the same architecture sample
    return net, feats, iv, den, sup


def _rank_main(rank, world, port, out_dir):
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    T = pkg.trainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        Bg = KW["num_sequences"]
        b0, b1 = T.shard_sequences(Bg, rank, world)
        net, feats, iv, den, sup = _problem(pkg, b1 - b0)
        fd, ivd = dev(T.shard_rows(feats, Bg, b0, b1)), dev(np.ascontiguousarray(iv[b0:b1]))
        dg, ds = pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(T.shard_supervision(sup, b0, b1))
        # (a) per bucket, behind the bucket events, on a communication stream
        comm = torch.cuda.Stream()
        r = host(net.forward_backward(fd, ivd, dg, ds, step=0)).copy()
        net.allreduce_grads_overlapped(comm)
        g_bucketed = host(net.grads).copy()
        buckets = net.grad_buckets()
        # (b) one flat all-reduce
        net.grads.zero_()
        net.forward_backward(fd, ivd, dg, ds, step=0)
        local = host(net.grads).copy()
        net.allreduce_grads()
        g_flat = host(net.grads).copy()
        # strong scaling: ONE minibatch of Bg sequences -> l2 scale = global count; effective learning rate (the sum is not x num_jobs)
        net.update(1e-3, l2_regularize_scale=float(Bg), step=0)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), g_bucketed=g_bucketed, g_flat=g_flat, local=local, params=host(net.params), res=r,
                 buckets=np.asarray(buckets))
        net.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_the_hip_trainer_match_one_process(pkg, tmp_path):
    import torch.multiprocessing as mp
    net, feats, iv, den, sup = _problem(pkg, KW["num_sequences"])
    r1 = host(net.forward_backward(dev(feats), dev(iv), pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup), step=0)).copy()
    g1 = host(net.grads).copy()
    assert r1[5] == 1.0 and np.linalg.norm(g1) > 0
    net.update(1e-3, l2_regularize_scale=float(KW["num_sequences"]), step=0)
    p1 = host(net.params).copy()
    buckets1 = net.grad_buckets()
    net.close()
    # buckets: contiguous, cover the buffer, highest addresses (the heads) first
    assert buckets1[0][1] == len(g1) and buckets1[-1][0] == 0 and all(a[0] == b[1] for a, b in zip(buckets1, buckets1[1:]))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(2)]
    for o in out:
        assert [tuple(b) for b in o["buckets"]] == buckets1
        assert np.array_equal(o["g_bucketed"], o["g_flat"])  # same sums either way, bit for bit (two addends)
        assert rel_l2(o["g_flat"], g1) < 1e-5, rel_l2(o["g_flat"], g1)  # == the single process's gradient of the whole minibatch
        assert rel_l2(o["params"], p1) < 1e-6
    assert np.array_equal(out[0]["g_flat"], out[1]["g_flat"]) and np.array_equal(out[0]["params"], out[1]["params"])
    assert not np.array_equal(out[0]["local"], out[1]["local"])  # the shards did differ
    # objective: the shards' sums add up
    assert abs(out[0]["res"][0] + out[1]["res"][0] - r1[0]) < 1e-5 * abs(r1[0]) and out[0]["res"][2] + out[1]["res"][2] == r1[2]


def _rccl_one_rank_main(rank, port, out_dir):
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        net, feats, iv, den, sup = _problem(pkg, KW["num_sequences"])
        fd, ivd, dg, ds = dev(feats), dev(iv), pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup)
        net.forward_backward(fd, ivd, dg, ds, step=0)
        torch.cuda.synchronize()
        before = host(net.grads).copy()
        net.grads.zero_()
        comm = torch.cuda.Stream()
        net.forward_backward(fd, ivd, dg, ds, step=0)
        net.allreduce_grads_overlapped(comm, min_world=1)  # one RCCL all-reduce per bucket behind the bucket events
        net.allreduce_grads(min_world=1)                   # and the flat one
        dist.barrier()
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "rccl.npz"), before=before, after=host(net.grads).copy())
        net.close()
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_accept_the_gradient_buffer(pkg, tmp_path):
    """bench.py --gpus N > 1 uses backend "nccl" (RCCL), which a one-GPU box cannot run with two ranks.  What it can run is the
    same call sequence in a ONE-rank RCCL group: init_process_group(device_id=...), an async all_reduce per bucket on slices of the
    library-owned gradient buffer behind the bucket events on a communication stream, the flat all_reduce, a barrier.  The sums of
    one rank leave the gradient as it was."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rccl_one_rank_main, args=(port, str(tmp_path)), nprocs=1, join=True)
    o = np.load(tmp_path / "rccl.npz")
    assert np.linalg.norm(o["before"]) > 0 and np.array_equal(o["before"], o["after"])


# ------------------------------------------------------------------------------------------------ synchronised BatchNorm
# Train-mode BatchNorm takes its statistics over all rows of the minibatch (nnet-normalize-component.cc:433-445), so a sharded
# minibatch only equals the whole one when the column sums are all-reduced (ChainNet.set_batchnorm_sync).
KW_BN = dict(frames_per_chunk=24, num_sequences=8, strides=[1, 0, 3], bottleneck=16, feat_dim=40, ivector_dim=100, num_pdfs=96, hidden_dim=64,
             small_dim=32, relu_self_repair_scale=0.0)


def _bn_problem(pkg, B):
    T = pkg.trainer
    net = T.ChainNet(T.make_config(**dict(KW_BN, num_sequences=B)))
    full = T.ChainNet(T.make_config(**KW_BN)) if B != KW_BN["num_sequences"] else net
    params = full.init_params_numpy(seed=1, output_stddev=0.3)
    feats, iv = T.synthetic_egs(full, seed=4)
    den = pkg.synth.make_den_graph(40, net.cfg.num_pdfs, mean_out_degree=4.0, seed=5)
    sup = pkg.synth.make_supervision(KW_BN["num_sequences"], net.cfg.frames_per_chunk // 3, net.cfg.num_pdfs, seed=6)
    if full is not net:
        full.close()
    net.set_params(params)
    return net, feats, iv, den, sup


def _bn_rank_main(rank, world, port, out_dir):
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    T = pkg.trainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        Bg = KW_BN["num_sequences"]
        b0, b1 = T.shard_sequences(Bg, rank, world)
        net, feats, iv, den, sup = _bn_problem(pkg, b1 - b0)
        fd, ivd = dev(T.shard_rows(feats, Bg, b0, b1)), dev(np.ascontiguousarray(iv[b0:b1]))
        dg, ds = pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(T.shard_supervision(sup, b0, b1))
        out = {}
        for sync in (True, False):
            assert net.set_batchnorm_sync(sync) == sync
            net.grads.zero_()
            r = host(net.forward_backward(fd, ivd, dg, ds, step=0)).copy()
            net.allreduce_grads()
            out["g_sync" if sync else "g_local"] = host(net.grads).copy()
            out["r_sync" if sync else "r_local"] = r
            out["y_sync" if sync else "y_local"] = host(net.activation("output")).copy()
        np.savez(os.path.join(out_dir, "bn_rank%d.npz" % rank), **out)
        net.close()
    finally:
        dist.destroy_process_group()


def test_synchronised_batchnorm_makes_two_shards_equal_the_whole_minibatch(pkg, tmp_path):
    import torch.multiprocessing as mp
    net, feats, iv, den, sup = _bn_problem(pkg, KW_BN["num_sequences"])
    r1 = host(net.forward_backward(dev(feats), dev(iv), pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup), step=0)).copy()
    g1 = host(net.grads).copy()
    y1 = host(net.activation("output")).copy()
    net.close()
    assert r1[5] == 1.0 and np.linalg.norm(g1) > 0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_bn_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = [np.load(tmp_path / ("bn_rank%d.npz" % r)) for r in range(2)]
    T = pkg.trainer
    Bg = KW_BN["num_sequences"]
    for r, o in enumerate(out):
        b0, b1 = T.shard_sequences(Bg, r, 2)
        # train-mode BatchNorm, raw-gradient path: the reduced gradient of two 4-sequence shards is the single process's of all 8
        assert rel_l2(o["g_sync"], g1) < 1e-5, rel_l2(o["g_sync"], g1)
        assert rel_l2(o["y_sync"], T.shard_rows(y1, Bg, b0, b1)) < 1e-5
        # ... which per-shard statistics (the default, Kaldi's per-job behaviour) do not give
        assert rel_l2(o["g_local"], g1) > 1e-3 and rel_l2(o["y_local"], T.shard_rows(y1, Bg, b0, b1)) > 1e-3
    assert np.array_equal(out[0]["g_sync"], out[1]["g_sync"])
    assert abs(out[0]["r_sync"][0] + out[1]["r_sync"][0] - r1[0]) < 1e-5 * abs(r1[0])


def _rccl_bn_sync_main(rank, port, out_dir):
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        net, feats, iv, den, sup = _bn_problem(pkg, KW_BN["num_sequences"])
        fd, ivd, dg, ds = dev(feats), dev(iv), pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup)
        net.forward_backward(fd, ivd, dg, ds, step=0)
        before = host(net.grads).copy()
        net.grads.zero_()
        assert net.set_batchnorm_sync(True, min_world=1)  # RCCL all-reduces of the BatchNorm sums on the compute stream (one rank: sums unchanged)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):  # ... also when the compute stream is not torch's default stream
            side.wait_stream(torch.cuda.default_stream())
            net.forward_backward(fd, ivd, dg, ds, step=0)
        side.synchronize()
        dist.barrier()
        np.savez(os.path.join(out_dir, "rccl_bn.npz"), before=before, after=host(net.grads).copy())
        net.close()
    finally:
        dist.destroy_process_group()


def test_rccl_runs_the_batchnorm_sync_collectives(pkg, tmp_path):
    """The synchronised-BatchNorm callback under backend "nccl" (RCCL) in a one-rank group: in-place all-reduces of 2 D / 3 D doubles
    at raw device pointers inside the library's arena, enqueued on the trainer's compute stream from inside tdnnf_net_forward_backward."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rccl_bn_sync_main, args=(port, str(tmp_path)), nprocs=1, join=True)
    o = np.load(tmp_path / "rccl_bn.npz")
    assert np.linalg.norm(o["before"]) > 0 and rel_l2(o["after"], o["before"]) < 1e-6


def _native_rccl_main(rank, out_dir):
    """One process = one rank = one GPU; the library creates the communicator itself (csrc/rccl_sync.hip), no torch process group."""
    import __graft_entry__ as ge
    pkg = ge.load_package()
    torch.cuda.set_device(0)
    comm = pkg.trainer.RcclComm(single=True)
    try:
        x = torch.randn(1000, device="cuda", dtype=torch.float64)
        x0 = x.clone()
        comm.allreduce_sum(x)
        y = torch.randn(777, device="cuda")
        y0 = y.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        comm.allreduce_sum(y, stream=side)
        side.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(x, x0) and torch.equal(y, y0)  # one rank: sums unchanged
        net, feats, iv, den, sup = _bn_problem(pkg, KW_BN["num_sequences"])
        fd, ivd, dg, ds = dev(feats), dev(iv), pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup)
        net.forward_backward(fd, ivd, dg, ds, step=0)
        before = host(net.grads).copy()
        net.grads.zero_()
        assert net.set_batchnorm_sync_rccl(comm)  # every train-mode BatchNorm: ncclAllReduce of its column sums, issued from C++ on the compute stream
        cs = torch.cuda.Stream()
        with torch.cuda.stream(side):  # (also when the compute stream is not torch's default stream)
            side.wait_stream(torch.cuda.default_stream())
            net.forward_backward(fd, ivd, dg, ds, step=0)
            net.allreduce_grads_rccl(comm, cs)  # the gradient buckets on their own stream behind the bucket events; `side` waits for them
            after = net.grads.clone()
        side.synchronize()
        net.set_batchnorm_sync_rccl(None)
        net.grads.zero_()
        net.forward_backward(fd, ivd, dg, ds, step=0)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "native.npz"), before=before, after=host(after), off_again=host(net.grads).copy())
        net.close()
    finally:
        comm.close()


def test_library_issued_rccl_exchanges_with_a_one_rank_communicator(pkg, tmp_path):
    """The exchanges the library issues itself (tdnnf_rccl_*, tdnnf_net_set_batchnorm_sync_rccl, tdnnf_net_allreduce_grads_rccl): librccl
    loaded with dlopen, a communicator created from a unique id, ncclAllReduce of doubles on the compute stream inside
    tdnnf_net_forward_backward and of the gradient buckets on a communication stream.  A one-GPU box can only run one rank: the
    sums must come back unchanged and the step must equal the unsynchronised one bit for bit."""
    import torch.multiprocessing as mp
    mp.spawn(_native_rccl_main, args=(str(tmp_path),), nprocs=1, join=True)  # (its own process: RCCL initialises its own GPU context state)
    o = np.load(tmp_path / "native.npz")
    assert np.linalg.norm(o["before"]) > 0
    assert np.array_equal(o["after"], o["before"]) and np.array_equal(o["off_again"], o["before"])


def test_unequal_shards_are_refused_by_synchronised_batchnorm(pkg):
    """rows x world_size is the global row count synchronised BatchNorm divides by: ranks with different sequence counts are an error
    (trainer._require_equal_shards), checked on the host with a fake two-rank gather."""
    import unittest.mock as mock
    import torch.distributed as dist
    T = pkg.trainer

    def fake_all_gather(out, mine, group=None):
        out[0].copy_(mine)
        out[1].copy_(mine + 1)

    with mock.patch.object(dist, "is_available", return_value=True), mock.patch.object(dist, "is_initialized", return_value=True), \
            mock.patch.object(dist, "get_backend", return_value="gloo"), mock.patch.object(dist, "all_gather", side_effect=fake_all_gather):
        with pytest.raises(ValueError, match="same number of sequences"):
            T._require_equal_shards(8, None, 2)


def _one_rccl_main(rank, port, out_dir):
    """torch.distributed's "nccl" backend first (it maps torch/lib/librccl.so), then the library's own communicators: they must bind
    the copy that is already in the process, not load /opt/rocm's beside it."""
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    torch.cuda.set_device(0)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    try:
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        comm = pkg.trainer.RcclComm(single=True)
        path = comm.library_path()
        x = torch.arange(8, device="cuda", dtype=torch.float32)
        comm.allreduce_sum(x)
        torch.cuda.synchronize()
        maps = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
        with open(os.path.join(out_dir, "one_rccl.txt"), "w") as f:
            f.write(path + "\n" + "\n".join(maps) + "\n")
        assert comm.h and comm.h_bn and comm.h.value != comm.h_bn.value  # two communicators: buckets / BatchNorm sums
        comm.close()
    finally:
        dist.destroy_process_group()


def test_the_library_binds_the_rccl_torch_already_loaded(pkg, tmp_path):
    """one RCCL per process.  After dist.init_process_group("nccl") the library's communicators come from the copy
    torch mapped (found among the loaded objects, re-opened with RTLD_NOLOAD); /proc/self/maps shows a single librccl."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_one_rccl_main, args=(port, str(tmp_path)), nprocs=1, join=True)
    lines = open(tmp_path / "one_rccl.txt").read().split("\n")
    assert "already mapped in the process" in lines[0], lines
    mapped = [ln for ln in lines[1:] if ln]
    assert len(mapped) == 1 and mapped[0] in lines[0], lines


def test_bench_starts_its_own_ranks(pkg):
    """`python3 bench.py --gpus 2` from a bare shell (no launcher, no WORLD_SIZE): bench.py starts the two ranks itself and exits 0
    with a two-rank line.  Both ranks on this one GPU over gloo (TDNNF_BENCH_REHEARSE_ON_ONE_GPU=1), tiny shape."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TDNNF_BENCH_REHEARSE_ON_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--ng-burn-in", "2", "--chunk", "150",
                        "--minibatch", "8", "--den-states", "500", "--no-also", "--no-alt", "--no-parity", "--no-strong"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["value"] > 0
    assert out["config"]["global_batch"] == 16
