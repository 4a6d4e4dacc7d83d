#!/usr/bin/env python3
"""bench.py -- LF-MMI training throughput of the SWBD 7q TDNN-F chain model on MI355X.

One "step" = one minibatch of nnet3-chain-train on synthetic egs already resident in HBM:
forward through every component, chain objective (denominator + numerator forward-backward),
backward with gradient accumulation (OnlineNaturalGradient-preconditioned, as the reference's recipes train),
[RCCL all-reduce of the gradient buffer when N > 1], L2 + max-change + parameter update + scheduled orthonormal constraint.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  metric = BASELINE.json's "LF-MMI training frames/sec per node".
Workload (config.workload): BASELINE.json configs[1], the fixed 7q TDNN-F (14 tdnnf layers, bottleneck
160, strides 1,1,1,0,3x10) on 40-dim fbank + 100-dim ivector egs, chunks of --chunk frames, --minibatch
sequences per GPU (weak scaling: per-GPU work is fixed as N grows).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0


def cpu_baseline(pkg, args):
    """The CPU restatement of the same training step (oracle, float-accumulating OpenMP build) timed on this
    box's host cores on a bounded sample: same net, chunk 150, a few sequences."""
    from tests.oracle_net import OracleNet
    B, T = args.cpu_sequences, 150
    cfg = pkg.trainer.make_config(frames_per_chunk=T, num_sequences=B, use_natural_gradient=args.natural_gradient)
    comps, begin = [], 0
    # component table without touching the GPU: same layout rule as the trainer (16-byte aligned blocks)
    lda_dim = 3 * cfg.feat_dim + cfg.ivector_dim

    def add(name, rows, cols, hb, lrf=1.0, l2=0.01, mc=0.75, orth=0.0):
        nonlocal begin
        comps.append(dict(name=name, begin=begin, rows=rows, cols=cols, has_bias=hb, lr_factor=lrf, l2=l2, max_change=mc, orthonormal=orth))
        begin = (begin + rows * cols + (rows if hb else 0) + 3) // 4 * 4

    add("lda", lda_dim, lda_dim, 1, lrf=0.0, l2=0.0, mc=0.0)
    add("tdnn1.affine", cfg.hidden_dim, lda_dim, 1)
    for i in range(cfg.num_layers):
        K = 2 if cfg.time_stride[i] > 0 else 1
        add(f"tdnnf{i + 2}.linear", cfg.bottleneck_dim[i], K * cfg.hidden_dim, 0, orth=-1.0)
        add(f"tdnnf{i + 2}.affine", cfg.hidden_dim, K * cfg.bottleneck_dim[i], 1)
    add("prefinal-l", cfg.prefinal_small_dim, cfg.hidden_dim, 0, orth=-1.0)
    for hn in ("chain", "xent"):
        add(f"prefinal-{hn}.affine", cfg.hidden_dim, cfg.prefinal_small_dim, 1)
        add(f"prefinal-{hn}.linear", cfg.prefinal_small_dim, cfg.hidden_dim, 0, orth=-1.0)
        add("output.affine" if hn == "chain" else "output-xent.affine", cfg.num_pdfs, cfg.prefinal_small_dim, 1,
            lrf=1.0 if hn == "chain" else 5.0, l2=0.002, mc=1.5)
    import numpy as np
    rng = np.random.default_rng(0)
    params = (rng.standard_normal(begin) * 0.02).astype(np.float32)
    net = OracleNet(pkg, cfg, comps, fast=True)
    feats = rng.standard_normal((net.num_t_in * B, cfg.feat_dim)).astype(np.float32)
    iv = rng.standard_normal((B, cfg.ivector_dim)).astype(np.float32)
    den = pkg.synth.make_den_graph(args.den_states, cfg.num_pdfs, mean_out_degree=args.den_degree, seed=1)
    sup = pkg.synth.make_supervision_from_den(den, B, T // 3, num_paths=2, seed=2)
    t0 = time.time()
    res, grads, _ = net.forward_backward(params, feats, iv, den, sup, step=0)
    net.update(params, grads, 1e-3, float(B), 0)
    dt = time.time() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    out = {"value": round(B * T / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"one full training step (natural gradient {'on' if args.natural_gradient else 'off'}) of the same 7q net on {B} sequences x "
                     f"{T} frames ({dt:.1f} s), CPU restatement of the reference path (oracle/, OpenMP float build), not Kaldi"}
    # stock Kaldi CPU nnet3 runs one thread per job ("nnet3 does not yet support multiple threads", train.py:251-252)
    try:
        gomp = C.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        B1 = 1
        cfg1 = pkg.trainer.make_config(frames_per_chunk=T, num_sequences=B1, use_natural_gradient=args.natural_gradient)
        net1 = OracleNet(pkg, cfg1, comps, fast=True)
        f1 = rng.standard_normal((net1.num_t_in * B1, cfg1.feat_dim)).astype(np.float32)
        sup1 = pkg.synth.make_supervision_from_den(den, B1, T // 3, num_paths=2, seed=3)
        t0 = time.time()
        _, g1, _ = net1.forward_backward(params, f1, iv[:B1], den, sup1, step=0)
        net1.update(params, g1, 1e-3, float(B1), 0)
        dt1 = time.time() - t0
        gomp.omp_set_num_threads(cores)
        out["single_thread"] = {"value": round(B1 * T / dt1, 2), "unit": "frames/s", "cores": 1,
                                "sample": f"the same step on {B1} sequences x {T} frames, one thread ({dt1:.1f} s)"}
    except OSError:
        pass
    return out


def workload_text(args):
    ng = "OnlineNaturalGradient-preconditioned" if args.natural_gradient else "raw-gradient (natural gradient off)"
    tail = (f"LF-MMI chain objective + xent head, {ng} SGD step with L2, max-change and the scheduled orthonormal constraint")
    if args.workload == "7q":
        return ("BASELINE configs[1]: run_tdnn_7q fixed TDNN-F (14 tdnnf layers, bottleneck 160, strides 1,1,1,0,3x10, 6034 pdfs, "
                "40-dim fbank + 100-dim ivector), " + tail)
    if args.workload == "darts-offset":
        return (f"BASELINE configs[3]: DARTS offset supernet, {args.darts_offsets} taps per TdnnDARTSV3 component, pretrain mode "
                "(uniform tap sample per layer and minibatch), otherwise as configs[1]; " + tail)
    dims = "25..240 in 8 blocks, the recipe's set" if args.bn_choices == "reference" else "80, 160, 240, 320 in 4 blocks"
    return (f"BASELINE configs[4]: bottleneck-dimension supernet (candidate dims {dims}, Onehot sample per layer and "
            "minibatch), otherwise as configs[1]; " + tail)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4, help="untimed steps (after the setup minibatches, see --ng-burn-in)")
    ap.add_argument("--ng-burn-in", type=int, default=10,
                    help="setup, with natural gradient on: minibatches run before the warmup so that the preconditioners are past their "
                         "first 10 calls, which refresh on EVERY call (OnlineNaturalGradient's num_initial_updates); afterwards every 4th "
                         "does, and that steady state -- refresh steps included -- is what is timed.  0: time a fresh process's first steps")
    ap.add_argument("--chunk", type=int, default=1500, help="frames per chunk (north_star: 1500-frame chunks)")
    ap.add_argument("--minibatch", type=int, default=128, help="sequences per GPU")
    ap.add_argument("--den-states", type=int, default=4000)
    ap.add_argument("--den-degree", type=float, default=12.0)
    ap.add_argument("--cpu-sequences", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="7q", choices=["7q", "darts-offset", "bn-supernet"],
                    help="7q = BASELINE configs[1] (default, the metric's config); darts-offset = configs[3], the K-tap offset "
                         "supernet of run_TDNN_DARTSV3_fbk_stride_pretrain.sh in pretrain (uniform-sample) mode; bn-supernet = "
                         "configs[4], the bottleneck-dimension supernet (8 candidate dims up to 240) in Onehot pretrain mode")
    ap.add_argument("--darts-offsets", type=int, default=7)
    ap.add_argument("--bn-choices", default="reference", choices=["reference", "baseline"],
                    help="bottleneck supernet candidates: reference = 25,50,80,100,120,160,200,240 (the recipe's 8); baseline = 80,160,240,320 "
                         "(BASELINE configs[4])")
    ap.add_argument("--gemm", default="f32", choices=["f32", "bf16x3", "bf16x6"],
                    help="GEMM arithmetic: f32 = exact v_mfma_f32_32x32x2_f32 (default, the reference's BaseFloat); bf16x3 = split-bf16 "
                         "(three bf16 MFMAs per product, f32 accumulate; 16 operand bits); bf16x6 = three planes, six MFMAs "
                         "(24 operand bits, f32-equivalent)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra split-bf16 measurement reported under \"alt\" (N = 1, --gemm f32 only)")
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="GeneralDropout proportion during the timed steps (the recipes' schedule 0,0@0.20,0.5@0.50,0 is at 0 for the first fifth "
                         "of training and peaks at 0.5); 0 = identity")
    ap.add_argument("--natural-gradient", type=int, default=1, choices=[0, 1],
                    help="1 (default, what the reference's recipes train with) = OnlineNaturalGradient preconditioning of every "
                         "updatable component's gradient; 0 = raw-gradient SGD step")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # rehearsal of the multi-rank path on a one-GPU box (never used by the driver): every rank on device 0, gloo collective
    rehearse = os.environ.get("TDNNF_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI

    lib = pkg.hipabi.load()  # raises if the HIP library is missing: there is no fallback path
    extra = {}
    if args.workload == "darts-offset":
        extra = dict(darts_num_offsets=args.darts_offsets)
    elif args.workload == "bn-supernet":
        extra = dict(bn_choice_dims=pkg.trainer.BN_CHOICE_DIMS if args.bn_choices == "reference" else [80, 80, 80, 80], bn_mode=pkg.trainer.BN_ONEHOT)
    cfg = pkg.trainer.make_config(frames_per_chunk=args.chunk, num_sequences=args.minibatch,
                                  use_natural_gradient=args.natural_gradient, gemm_precision={"f32": 0, "bf16x3": 1, "bf16x6": 2}[args.gemm],
                                  use_dropout=int(args.dropout > 0), **extra)
    net = pkg.trainer.ChainNet(cfg)
    if args.dropout > 0:
        net.set_dropout_proportion(args.dropout)
    # identical initial model on every rank (seed), different egs per rank (data parallel over sequences)
    net.set_params(net.init_params_numpy(seed=0, output_stddev=0.05))
    feats, iv = pkg.trainer.synthetic_egs(net, seed=100 + rank)
    den = pkg.synth.make_den_graph(args.den_states, cfg.num_pdfs, mean_out_degree=args.den_degree, seed=1)
    sup = pkg.synth.make_supervision_from_den(den, cfg.num_sequences, args.chunk // 3, num_paths=2, seed=200 + rank)
    dg, ds = pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup)
    fd, ivd = torch.from_numpy(feats).cuda(), torch.from_numpy(iv).cuda()
    lr = pkg.trainer.learning_rate(0, world, 100, 0, 100)  # 2.5e-4 * num_jobs

    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234)  # same architecture sample on every rank (seed-shared draws)

    def step(i):
        net.set_random_draws(generator=gen)
        net.forward_backward(fd, ivd, dg, ds, step=i)
        net.allreduce_grads()
        # l2 scale: GetNumNvalues * l2_regularize_factor(=1/num_jobs) -> per-GPU sequence count
        net.update(lr, l2_regularize_scale=float(cfg.num_sequences), step=i)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    burn = args.ng_burn_in if args.natural_gradient else 0
    for i in range(burn + args.warmup):
        step(i)
    sync()
    pkg.hipabi.check(lib.tdnnf_profile_enable(1))
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(burn + args.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    pkg.hipabi.check(lib.tdnnf_profile_enable(0))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res = net.results.cpu().numpy()

    # live roofline of the dominant kernel class (HIP events recorded on the launch stream)
    classes = []
    for k in range(4):
        n, ms, fl = C.c_double(), C.c_double(), C.c_double()
        pkg.hipabi.check(lib.tdnnf_profile_read(k, C.byref(n), C.byref(ms), C.byref(fl)))
        classes.append(dict(name=lib.tdnnf_profile_class_name(k).decode(), launches=n.value, ms=ms.value, flops=fl.value))
    dom = max(classes[:3], key=lambda c: c["ms"])  # the TDNN-F factored GEMMs (class 3 = natural-gradient statistics)
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom["name"], {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        frames = world * cfg.num_sequences * args.chunk * args.steps
        out = {
            "metric": "LF-MMI training frames/sec per node (SWBD 7q TDNN-F)",
            "value": round(frames / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"f32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA, 2 planes / 3 products, f32 accumulate)",
                                          "bf16x6": "bf16x6 (split-bf16 MFMA, 3 planes / 6 products, f32 accumulate; f32-equivalent)"}[args.gemm], "data": "synthetic",
            "config": {"workload": workload_text(args),
                       "frames_per_chunk": args.chunk, "sequences_per_gpu": cfg.num_sequences, "global_batch": world * cfg.num_sequences,
                       "den_graph": {"states": args.den_states, "arcs": int(len(den["src"]))},
                       "natural_gradient": {"on": bool(args.natural_gradient), "rank_in": 20, "rank_out": 80, "update_period": 4,
                                            "setup_minibatches_before_warmup": burn,
                                            "refresh_steps_in_timed_region": sum(1 for t in range(burn + args.warmup, burn + args.warmup + args.steps)
                                                                                 if t <= 10 or (t - 10) % 4 == 0) if args.natural_gradient else 0},
                       "parallelism": f"dp{world}", "objf_per_frame": float(res[0] / res[2]) if res[2] else None},
            "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic,
                         "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4), "launches": int(dom["launches"]),
                         "all_kernels": [{"kernel": c["name"], "launches": int(c["launches"]), "ms": round(c["ms"], 3),
                                          "tflops": round(c["flops"] / (c["ms"] * 1e-3) / 1e12, 2) if c["ms"] > 0 else 0.0}
                                         for c in classes]},
        }
        if world == 1 and args.gemm == "f32" and not args.no_alt:
            # the same step with the optional split-bf16 GEMM arithmetic (not the headline: its gradient parity sits AT the
            # 1e-3 bar), same workload, same steps
            net.close()
            cfg2 = pkg.trainer.make_config(frames_per_chunk=args.chunk, num_sequences=args.minibatch, use_natural_gradient=args.natural_gradient,
                                           gemm_precision=1, **extra)
            net = pkg.trainer.ChainNet(cfg2)
            net.set_params(net.init_params_numpy(seed=0, output_stddev=0.05))
            for i in range(burn + args.warmup):
                step(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step(burn + args.warmup + i)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            out["alt"] = {"gemm": "bf16x3 (split-bf16 MFMA, f32 accumulate; --gemm bf16x3)", "value": round(frames / dt2, 1), "unit": "frames/s",
                          "ms_per_step": round(1e3 * dt2 / args.steps, 3)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, args)
        print(json.dumps(out), flush=True)
    net.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
