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
sequences per GPU (--scaling weak, the default: per-GPU work is fixed as N grows) or per node (--scaling strong:
every rank takes minibatch / N sequences of one global minibatch).

At N = 1 the line also carries: "roofline" (live HIP-event timing of the dominant GEMM class + algorithmic bytes),
"parity" (the HIP step against the CPU oracle on a bounded sample of the same workload, asserted), "cpu_baseline"
(the oracle timed on this box's host cores), "alt" (split-bf16 GEMM arithmetic) and "also" (the reference's own egs
shape and SWBD-scale denominator graphs as further line items).
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0
BF16_PEAK_TFLOPS = 2500.0       # the same guide: dense bf16 MFMA peak (not the 2:1-sparsity figure)
# tools/make_profiles.py: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this command with THIS round's kernels; the f16x3 step has its own
TRAFFIC_FILE = os.path.join("profiles", "r05_pmc_traffic.json")
TRAFFIC_FILE_F16X3 = os.path.join("profiles", "r05_f16x3_pmc_traffic.json")
L2_GATHER_PEAK_GBS = 16800.0    # /opt/skills/guides/MI355X_MICROARCH.md, gather table: rows shared by every workgroup served by the XCDs' L2, 16.8-18.8 TB/s


def load_traffic(path):
    """The committed PMC summary (per kernel class: HBM bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) KB, the guide's gfx950 correction;
    per HBM-bound pass under "hbm_pass:<name>": bytes per pass over all its kernels), or {} when the file is missing."""
    full = os.path.join(ROOT, path)
    try:
        return json.load(open(full)) if os.path.exists(full) else {}
    except Exception:
        return {}


def hbm_entries(hbm_classes, event_steps, chunk, traffic, traffic_file):
    """roofline_hbm: the HBM-bound passes of the step, each timed live with HIP events on its launch stream.  achieved = algorithmic bytes /
    event time against the 8 TB/s peak; traffic = the measured PMC bytes per pass from the committed profile of this command.  The denominator is
    NOT an HBM pass: its arcs are gathered from L2 -- its entry carries the gather rate against the guide's L2 figure and, separately, the HBM
    bytes the PMC passes count against 8 TB/s; no fraction of an HBM peak is formed from L2 bytes."""
    what = {"bn_apply_bypass": "BatchNorm apply + dropout mask + Sum(Scale(0.66, bypass), .): reads x [and the bypass rows], writes the layer output [and, "
                               "--gemm f16x3, its f16 planes]",
            "bn_relu_bwd": "BatchNorm backward + ReLU backward + self-repair + ReLU statistics + bias column sums, two stages: (x, dz) read twice, d_aff written",
            "planes_split": "operand split into 16-bit planes (--gemm f16x3 / bf16x6): the matrix read once, each plane layout written once"}
    out = []
    for hc in hbm_classes:
        if hc["ms"] <= 0:
            continue
        sec = hc["ms"] * 1e-3
        tr = traffic.get("hbm_pass:" + hc["name"], {}).get("hbm_bytes_per_pass")
        per_pass_ms = hc["ms"] / max(hc["launches"], 1)
        if hc["name"] == "denominator":
            gbs = hc["bytes"] / sec / 1e9
            ent = {"bound": "l2", "kernel": "denominator",
                   "what": "chain denominator forward-backward, both recursions and the occupancies (fork .. join): per (frame, sequence) the arcs once per "
                           "recursion (8 B forward, 16 B backward) gathered from L2 (the graph's three sliced tables, 1.2 MB, stay there), the output row read "
                           "twice, the derivative row written; bound by the dependent gathers of a frame, not by bandwidth",
                   "achieved": round(gbs, 1), "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / L2_GATHER_PEAK_GBS, 4),
                   "peak_source": "MI355X_MICROARCH.md gather table: rows shared by every workgroup (the XCD's L2) 16.8-18.8 TB/s",
                   "launches": int(hc["launches"]), "ms_per_step": round(hc["ms"] / event_steps, 3),
                   "gather_bytes_per_step": round(hc["bytes"] / event_steps, 1),
                   "us_per_frame_step": round(1e3 * per_pass_ms / (chunk // 3), 2),
                   "hbm": {"traffic": tr, "traffic_source": traffic_file if tr else None,
                           "achieved": round(tr / (per_pass_ms * 1e-3) / 1e9, 1) if tr else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(tr / (per_pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tr else None,
                           "what": "HBM bytes of the pass (PMC: 2 FETCH_SIZE + WRITE_SIZE over its kernels) / event time of the pass"}}
        else:
            gbs = hc["bytes"] / sec / 1e9
            ent = {"bound": "hbm", "kernel": hc["name"], "what": what.get(hc["name"], ""), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(gbs / HBM_PEAK_GBS, 4), "launches": int(hc["launches"]), "ms_per_step": round(hc["ms"] / event_steps, 3),
                   "algorithmic_bytes_per_step": round(hc["bytes"] / event_steps, 1), "traffic": tr, "traffic_source": traffic_file if tr else None,
                   "traffic_over_algorithmic": round(tr / (hc["bytes"] / max(hc["launches"], 1)), 3) if tr and hc["bytes"] else None}
        out.append(ent)
    return out


# parity bars of BASELINE.json's north_star: objective 1e-4 relative, parameter-gradient L2 1e-3; with natural gradient the
# first step also held to 1e-3 (measured 1e-5 once ReLU ties are agreed; later steps feed the eigen-decompositions back)
OBJF_TOL, GRAD_TOL = 1e-4, 1e-3


def workload_kwargs(args, workload=None):
    workload = args.workload if workload is None else workload
    import __graft_entry__ as ge
    T = ge.load_package().trainer
    if workload == "darts-offset":
        return dict(darts_num_offsets=args.darts_offsets)
    if workload == "darts-offset-cvupdate":
        # run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:128-142: Gumbel coefficients on ALL K taps, update-alpha, BatchNormTest from the
        # parent's statistics, every learning-rate factor 0 except 1e-4 on the TdnnDARTSV3 components
        return dict(darts_num_offsets=args.darts_offsets, darts_flags=T.DARTS_USE_GUMBEL | T.DARTS_UPDATE_ALPHA, darts_temp_proportion=0.5, cv_update=1)
    if workload == "bn-supernet":
        return dict(bn_choice_dims=T.BN_CHOICE_DIMS if args.bn_choices == "reference" else [80, 80, 80, 80], bn_mode=T.BN_ONEHOT)
    return {}


def hip_step_against_oracle(pkg, args, gemm_precision, tie_tol, state=None):
    """One training step of the full-width net on the bounded sample (chunk 150 x --cpu-sequences) on the GPU against the
    double-accumulating oracle: relative objective and gradient differences, ReLU ties counted (taken over where the oracle's
    own pre-activation is within tie_tol x rms of zero)."""
    import numpy as np
    import torch
    from tests.oracle_net import OracleNet, component_table
    B, T = args.cpu_sequences, 150
    cfg = pkg.trainer.make_config(frames_per_chunk=T, num_sequences=B, use_natural_gradient=args.natural_gradient, gemm_precision=gemm_precision,
                                  **workload_kwargs(args))
    # (the pre-split plane kernels of gemm_precision 2 / 3 run on the one-stream schedule of large minibatches: the small parity sample asks for it)
    with pkg.hipabi.option("wgrad_stream", 0 if gemm_precision in (2, 3) else -1):
        net = pkg.trainer.ChainNet(cfg)
    routed0 = (C.c_longlong(), C.c_longlong())
    pkg.hipabi.load().tdnnf_planes_routed(C.byref(routed0[0]), C.byref(routed0[1]))
    comps, num_params = component_table(cfg)
    assert num_params == net.num_params and [c["begin"] for c in comps] == [c["begin"] for c in net.components]
    if state is None:
        params = net.init_params_numpy(seed=0, output_stddev=0.05)
        feats, iv = pkg.trainer.synthetic_egs(net, seed=100)
        den = pkg.synth.make_den_graph(args.den_states, cfg.num_pdfs, mean_out_degree=args.den_degree, seed=1)
        sup = pkg.synth.make_supervision_from_den(den, B, T // 3, num_paths=2, seed=2)
        draws = np.random.default_rng(5).uniform(1e-3, 1 - 1e-3, max(net.num_draws, 1)).astype(np.float32)
        state = dict(params=params, feats=feats, iv=iv, den=den, sup=sup, draws=draws, comps=comps)
    net.set_params(state["params"])
    net.set_random_draws(state["draws"])
    stats = None
    if cfg.cv_update:  # BatchNormTest needs stored statistics: those of one pretrain step of the same supernet on this sample
        kw = dict(workload_kwargs(args, "darts-offset"))
        pre = pkg.trainer.ChainNet(pkg.trainer.make_config(frames_per_chunk=T, num_sequences=B, use_natural_gradient=0, **kw))
        pre.set_params(state["params"])
        pre.set_random_draws(np.random.default_rng(6).uniform(1e-3, 1 - 1e-3, max(pre.num_draws, 1)).astype(np.float32))
        pre.forward_backward(torch.from_numpy(state["feats"]).cuda(), torch.from_numpy(state["iv"]).cuda(), pkg.hipabi.DenGraph(state["den"]),
                             pkg.hipabi.Supervision(state["sup"]), step=0)
        stats = pre.get_stats()
        pre.close()
        net.set_stats(stats)
    r = net.forward_backward(torch.from_numpy(state["feats"]).cuda(), torch.from_numpy(state["iv"]).cuda(), pkg.hipabi.DenGraph(state["den"]),
                             pkg.hipabi.Supervision(state["sup"]), step=0)
    torch.cuda.synchronize()
    r, g = r.cpu().numpy(), net.grads.cpu().numpy()
    relu_names = ["tdnn1.relu"] + ["tdnnf%d.relu" % (l + 2) for l in range(cfg.num_layers)] + ["prefinal-chain.relu", "prefinal-xent.relu"]
    relus = {k: net.activation(k).cpu().numpy() for k in relu_names}
    net.close()
    ref = OracleNet(pkg, cfg, comps)  # double-accumulating build
    ref.relu_tie_tol = tie_tol
    if stats is not None:
        ref.set_stats(stats)
    res_ref, g_ref, _ = ref.forward_backward(state["params"], state["feats"], state["iv"], state["den"], state["sup"], step=0, draws=state["draws"], relu_like=relus)
    objf_rel = abs(r[0] - res_ref["objf"]) / abs(res_ref["objf"])
    grad_rel = float(np.linalg.norm(g.astype(np.float64) - g_ref) / np.linalg.norm(g_ref.astype(np.float64)))
    out = {"objf_rel": float(f"{objf_rel:.3e}"), "grad_rel_l2": float(f"{grad_rel:.3e}"), "objf_tol": OBJF_TOL, "grad_tol": GRAD_TOL,
           # (the ties taken over from the GPU run must stay a vanishing fraction: 1e-4 of the ReLU elements for f32 arithmetic)
           "ok": bool(r[5] == 1.0 and objf_rel < OBJF_TOL and grad_rel < GRAD_TOL and np.isfinite(g).all() and
                      sum(ref.relu_ties.values()) <= (1e-4 if tie_tol <= 1e-4 else 1e-3) * sum(v.size for v in relus.values())),
           "objf_hip": float(r[0]), "objf_oracle": float(res_ref["objf"]),
           "relu_ties": int(sum(ref.relu_ties.values())), "relu_elements": int(sum(v.size for v in relus.values())), "relu_tie_tolerance_x_rms": tie_tol}
    if gemm_precision in (2, 3):  # how many GEMMs / weight gradients of this step ran on the plane kernels (the rest: tap coefficients, row strides -> their own kernels)
        routed1 = (C.c_longlong(), C.c_longlong())
        pkg.hipabi.load().tdnnf_planes_routed(C.byref(routed1[0]), C.byref(routed1[1]))
        out["plane_kernel_launches"] = {"rows_gemms": routed1[0].value - routed0[0].value, "weight_gradients": routed1[1].value - routed0[1].value}
    return out, state


def parity_and_cpu_baseline(pkg, args, want_baseline=True):
    """A bounded sample of the SAME workload (same net at full width, chunk 150, --cpu-sequences sequences, same denominator
    graph family): (1) one training step on the GPU through the C-ABI against the double-accumulating CPU oracle -- objective
    and parameter gradient, asserted against BASELINE.json's bars (objective 1e-4, gradient L2 1e-3), for the arithmetic of
    this run and, beside it, for the split-bf16 arithmetic of "alt"; (2) the float/OpenMP build of the oracle timed on this
    box's host cores (2 warm-ups, median of 5), a probed thread count and one thread."""
    import numpy as np
    from tests.oracle_net import OracleNet
    B, T = args.cpu_sequences, 150
    prec = {"f32": 0, "bf16x3": 1, "bf16x6": 2, "f16x3": 3}[args.gemm]
    parity, state = hip_step_against_oracle(pkg, args, prec, 2e-3 if prec == 1 else 1e-4)
    parity["sample"] = (f"one training step (natural gradient {'on' if args.natural_gradient else 'off'}, --gemm {args.gemm}) of the full-width net on {B} sequences x "
                        f"{T} frames, {args.den_states}-state denominator graph; HIP through the C-ABI against oracle/ (double-accumulating CPU "
                        f"restatement, parity unpinned vs Kaldi).  ReLU pre-activations within rounding of zero flip their derivative mask between "
                        f"any two correct implementations: those ties are taken over from the GPU run and counted")
    if args.gemm == "f32" and not args.no_alt:
        parity["f16x3"], _ = hip_step_against_oracle(pkg, args, 3, 1e-4, state)  # the arithmetic of "alt", same sample, same bars, the exact-f32 tie tolerance
    params, den, draws, comps = state["params"], state["den"], state["draws"], state["comps"]
    if not want_baseline:
        return parity, None

    gomp = C.CDLL("libgomp.so.1")
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if os.environ.get("OMP_NUM_THREADS"):
        avail = min(avail, int(os.environ["OMP_NUM_THREADS"]))

    def timed_steps(B_s, threads, warm, timed, seed):
        gomp.omp_set_num_threads(threads)
        cfg_s = pkg.trainer.make_config(frames_per_chunk=T, num_sequences=B_s, use_natural_gradient=args.natural_gradient, **workload_kwargs(args))
        o = OracleNet(pkg, cfg_s, comps, fast=True)
        rng = np.random.default_rng(seed)
        f = rng.standard_normal((o.num_t_in * B_s, cfg_s.feat_dim)).astype(np.float32)
        v = rng.standard_normal((B_s, cfg_s.ivector_dim)).astype(np.float32)
        sp = pkg.synth.make_supervision_from_den(den, B_s, T // 3, num_paths=2, seed=seed)
        p, ts = params.copy(), []
        for i in range(warm + timed):
            t0 = time.perf_counter()
            _, gg, _ = o.forward_backward(p, f, v, den, sp, step=i, draws=draws)
            p = o.update(p, gg, 1e-3, float(B_s), i)
            ts.append(time.perf_counter() - t0)
        return ts[warm:]

    # thread count: the oracle's OpenMP loops stop scaling long before 256 threads on matrices this small, so one step is
    # timed at a few counts and the timed runs use the fastest
    cands = sorted({c for c in (min(avail, 64), 32, 16) if c <= avail}, reverse=True)  # (256 threads: 27 s per step on these matrices)
    probe = {c: timed_steps(B, c, 1, 1, 7)[0] for c in cands} if len(cands) > 1 else {avail: 0.0}
    cores = min(probe, key=probe.get)
    ts = timed_steps(B, cores, 2, 5, 7)
    med = statistics.median(ts)
    out = {"value": round(B * T / med, 2), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"the same training step (natural gradient {'on' if args.natural_gradient else 'off'}, forward + objective + backward + update) of the "
                     f"full-width net on {B} sequences x {T} frames: median of 5 after 2 warm-ups ({med:.2f} s per step, min {min(ts):.2f} max {max(ts):.2f}); "
                     f"{cores} OpenMP threads of {avail} usable (fastest of {sorted(probe)} on a probe step); CPU restatement of the reference path "
                     f"(oracle/, float build), not Kaldi"}
    # stock Kaldi CPU nnet3 runs one thread per job ("nnet3 does not yet support multiple threads", train.py:251-252)
    t1 = timed_steps(1, 1, 1, 3, 9)
    m1 = statistics.median(t1)
    out["single_thread"] = {"value": round(T / m1, 2), "unit": "frames/s", "cores": 1,
                            "sample": f"the same step on 1 sequence x {T} frames, one thread: median of 3 after 1 warm-up ({m1:.2f} s per step)"}
    gomp.omp_set_num_threads(avail)
    return parity, out


def workload_text(args):
    ng = "OnlineNaturalGradient-preconditioned" if args.natural_gradient else "raw-gradient (natural gradient off)"
    tail = (f"LF-MMI chain objective + xent head, {ng} SGD step with L2, max-change and the scheduled orthonormal constraint")
    if args.workload == "7q":
        return ("BASELINE configs[1]: run_tdnn_7q fixed TDNN-F (14 tdnnf layers, bottleneck 160, strides 1,1,1,0,3x10, 6034 pdfs, "
                "40-dim fbank + 100-dim ivector), " + tail)
    if args.workload == "darts-offset":
        return (f"BASELINE configs[3]: DARTS offset supernet, {args.darts_offsets} taps per TdnnDARTSV3 component, pretrain mode "
                "(uniform tap sample per layer and minibatch), otherwise as configs[1]; " + tail)
    if args.workload == "darts-offset-cvupdate":
        return (f"BASELINE configs[3], cv-update stage (run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:128-142): DARTS offset supernet, Gumbel coefficients "
                f"over all {args.darts_offsets} taps, BatchNormTest, learning-rate factor 0 except 1e-4 on the TdnnDARTSV3 components; " + tail)
    dims = "25..240 in 8 blocks, the recipe's set" if args.bn_choices == "reference" else "80, 160, 240, 320 in 4 blocks"
    return (f"BASELINE configs[4]: bottleneck-dimension supernet (candidate dims {dims}, Onehot sample per layer and "
            "minibatch), otherwise as configs[1]; " + tail)


class Job:
    """One replica of the trainer on this rank's GPU with its own synthetic egs resident in HBM."""

    def __init__(self, pkg, args, chunk, sequences, den_states, rank, world, gemm=None, first_sequence=0, natural_gradient=None, workload=None,
                 stats=None, archive_minibatches=0, scaling=None):
        import torch
        self.pkg, self.args, self.world = pkg, args, world
        gemm = args.gemm if gemm is None else gemm
        ng = args.natural_gradient if natural_gradient is None else natural_gradient
        kw = workload_kwargs(args, workload)
        self.cfg = pkg.trainer.make_config(frames_per_chunk=chunk, num_sequences=sequences, use_natural_gradient=ng,
                                           gemm_precision={"f32": 0, "bf16x3": 1, "bf16x6": 2, "f16x3": 3}[gemm], use_dropout=int(args.dropout > 0 and not kw.get("cv_update")),
                                           **kw)
        self.net = pkg.trainer.ChainNet(self.cfg)
        if args.dropout > 0 and not kw.get("cv_update"):
            self.net.set_dropout_proportion(args.dropout)
        if stats is not None:  # cv-update: the BatchNormTest components normalise with the parent's stored statistics
            self.net.set_stats(stats)
        # identical initial model on every rank (seed), different egs per rank (data parallel over sequences)
        self.net.set_params(self.net.init_params_numpy(seed=0, output_stddev=0.05))
        feats, iv = pkg.trainer.synthetic_egs(self.net, seed=100 + rank)
        self.den = pkg.synth.make_den_graph(den_states, self.cfg.num_pdfs, mean_out_degree=args.den_degree, seed=1)
        sup = pkg.synth.make_supervision_from_den(self.den, sequences, chunk // 3, num_paths=2, seed=200 + rank)
        self.dg, self.ds = pkg.hipabi.DenGraph(self.den), pkg.hipabi.Supervision(sup)
        self.fd, self.ivd = torch.from_numpy(feats).cuda(), torch.from_numpy(iv).cuda()
        # Effective learning rate.  The gradient buffers are SUMMED over ranks, which to first order is Kaldi's scheme of
        # num_jobs jobs at learning rate lr_eff x num_jobs followed by model averaging (common.py:618): mean_j(lr_eff J g_j) =
        # lr_eff sum_j g_j -- so the summed gradient takes lr_eff itself, NOT lr_eff x num_jobs.
        self.lr = pkg.trainer.learning_rate(0, 1, 100, 0, 100)
        # l2 scale = GetNumNvalues x l2_regularize_factor (= 1 / num_jobs): weak scaling = every rank is a Kaldi job with its own
        # minibatch -> its sequence count; strong scaling = ONE minibatch sharded over the ranks -> the global sequence count
        scaling = args.scaling if scaling is None else scaling
        self.l2_scale = float(sequences * (world if scaling == "strong" else 1))
        # strong scaling shards ONE minibatch: train-mode BatchNorm statistics are all-reduced so that it normalises as the whole
        # minibatch does on one GPU (--sync-batchnorm auto); weak scaling = independent jobs, statistics per job as in Kaldi
        self.sync_bn = world > 1 and (args.sync_batchnorm == "on" or (args.sync_batchnorm == "auto" and scaling == "strong"))
        # the exchanges are issued by the library itself on RCCL (csrc/rccl_sync.hip) unless the group is the gloo rehearsal
        import torch.distributed as dist
        self.rccl = None
        if (world > 1 and dist.get_backend() != "gloo") or (world == 1 and args.sync_batchnorm == "on"):
            try:
                self.rccl = pkg.trainer.RcclComm(single=world == 1)
            except Exception as e:  # (no librccl.so to dlopen, or the communicator could not be created: torch.distributed issues the same exchanges)
                print("bench.py: the library's own RCCL communicator is not available (%s): falling back to torch.distributed collectives" % e, file=sys.stderr)
                self.rccl = None
        self.rccl_path = self.rccl.library_path() if self.rccl is not None else None
        if self.sync_bn or (world == 1 and args.sync_batchnorm == "on"):
            if self.rccl is not None:
                self.net.set_batchnorm_sync_rccl(self.rccl)
                self.sync_bn = True
            elif world > 1:
                self.net.set_batchnorm_sync(True)
            else:
                self.sync_bn = False
        self.gen = torch.Generator(device="cuda")
        self.gen.manual_seed(1234)  # same architecture sample on every rank (seed-shared draws)
        self.comm = torch.cuda.Stream() if world > 1 and not args.no_overlap else None
        self.i = 0
        # archive-fed: every step's minibatch comes out of a cegs archive (16-bit compressed features, as the recipes' egs) through
        # egs.minibatches(prefetch=2) -- read, decompress, merge on a worker thread, then the copy to the device
        self.archive, self.feed, self.keep = None, None, []
        if archive_minibatches > 0:
            import tempfile
            E = pkg.egs
            self.archive = os.path.join(tempfile.mkdtemp(prefix="tdnnf_bench_"), "cegs.1.ark")
            with E.Writer(self.archive) as w:
                for m in range(archive_minibatches):
                    f_m, iv_m = pkg.trainer.synthetic_egs(self.net, seed=1000 + m)
                    sup_m = pkg.synth.make_supervision_from_den(self.den, sequences, chunk // 3, num_paths=2, seed=2000 + m)
                    for b in range(sequences):
                        w.write("u%d-%d" % (m, b), f_m[b::sequences], self.net.first_t, E.sequence_of(sup_m, b), self.cfg.num_pdfs, ivector=iv_m[b], compress=True)
            self.archive_bytes = os.path.getsize(self.archive)

    def next_minibatch(self):
        while True:
            if self.feed is None:
                self.feed = self.pkg.egs.minibatches(self.archive, self.net, prefetch=2)
            try:
                item = next(self.feed)
                break
            except StopIteration:  # next epoch: the archive again
                self.feed = None
        self.keep = (self.keep + [item])[-4:]  # the GPU is up to a step behind the host: keep what it may still read
        return item

    def step(self):
        self.net.set_random_draws(generator=self.gen)
        if self.archive is not None:
            f, v, sp = self.next_minibatch()
            self.net.forward_backward(f, v, self.dg, sp, step=self.i)
        else:
            self.net.forward_backward(self.fd, self.ivd, self.dg, self.ds, step=self.i)
        if self.comm is not None and self.rccl is not None and self.world > 1:  # ... issued from C++ (tdnnf_net_allreduce_grads_rccl)
            self.net.allreduce_grads_rccl(self.rccl, self.comm)
        elif self.comm is not None:  # one collective per gradient bucket, each behind its "bucket final" event: overlaps the backward pass
            self.net.allreduce_grads_overlapped(self.comm)
        else:
            self.net.allreduce_grads()
        self.net.update(self.lr, l2_regularize_scale=self.l2_scale, step=self.i)
        self.i += 1

    def run(self, burn, warmup, steps, sync, profile=False):
        for _ in range(burn + warmup):
            self.step()
        sync()
        lib = self.pkg.hipabi.load()
        # Live roofline: HIP events around every GEMM launch of the first `event_steps` timed steps (on the launch streams).  Each
        # event is a marker packet between two launches -- about 1 000 per step, ~3 ms of a 130 ms step -- so the remaining timed
        # steps run without them (--roofline-steps 0: events on every timed step).
        event_steps = steps if (profile and self.args.roofline_steps <= 0) else min(steps, self.args.roofline_steps)
        if profile:
            self.pkg.hipabi.check(lib.tdnnf_profile_enable(1))
            sync()
        r0 = (C.c_longlong(), C.c_longlong())
        lib.tdnnf_planes_routed(C.byref(r0[0]), C.byref(r0[1]))
        lead = getattr(self.args, "host_lead", False)
        if lead:  # diagnostics: how far ahead of the GPU does the host return from step()?  (an event per step; read after the loop)
            import torch
            ev0 = torch.cuda.Event(enable_timing=True)
            evs, ret = [torch.cuda.Event(enable_timing=True) for _ in range(steps)], []
            ev0.record()
        t0 = time.perf_counter()
        for i in range(steps):
            if profile and i == event_steps:
                self.pkg.hipabi.check(lib.tdnnf_profile_enable(0))  # (host side only: stops recording, nothing is synchronised)
            self.step()
            if lead:
                evs[i].record()
                ret.append(time.perf_counter() - t0)
        sync()
        dt = time.perf_counter() - t0
        if getattr(self.args, "phases", False):  # diagnostics: the LAST step's phases on the caller's stream (option phase_events)
            ms, cnt = (C.c_double * 8)(), C.c_int()
            self.pkg.hipabi.check(lib.tdnnf_net_phase_times(self.net.h, ms, 8, C.byref(cnt)))
            names = ["fwd_trunk", "heads_fwd", "heads_bwd", "trunk_bwd", "join", "between", "update"]
            self.phase_ms = {names[i]: round(ms[i], 3) for i in range(cnt.value)}
            print("phases: " + json.dumps(self.phase_ms), file=sys.stderr)
        if lead:
            done = [ev0.elapsed_time(e) for e in evs]
            self.host_lead_ms = {"gpu_done_minus_host_returned_ms": [round(d - 1e3 * r, 3) for d, r in zip(done, ret)],
                                 "host_issue_ms_per_step": round(1e3 * ret[-1] / steps, 3), "gpu_ms_per_step": round(done[-1] / steps, 3)}
            print("host lead: " + json.dumps(self.host_lead_ms), file=sys.stderr)
        r1 = (C.c_longlong(), C.c_longlong())
        lib.tdnnf_planes_routed(C.byref(r1[0]), C.byref(r1[1]))
        # GEMMs per step that really ran on the pre-split plane kernels (with the weight-gradient stream on -- minibatches of
        # <= 32 768 rows -- gemm_precision 3 keeps the exact-f32 kernels; a line must say which arithmetic it measured)
        self.planes_routed_per_step = {"rows_gemms": (r1[0].value - r0[0].value) / max(steps, 1), "weight_gradients": (r1[1].value - r0[1].value) / max(steps, 1)}
        if profile:
            self.pkg.hipabi.check(lib.tdnnf_profile_enable(0))
            self.event_steps = event_steps
        return dt

    def arithmetic_run(self, gemm):
        """What the timed steps computed in: the requested plane arithmetic only if GEMMs were routed to the plane kernels."""
        if gemm in ("f16x3", "bf16x6"):
            r = getattr(self, "planes_routed_per_step", None)
            if r is not None and r["rows_gemms"] + r["weight_gradients"] == 0:
                return "f32 (no GEMM was routed to the plane kernels at this minibatch size: the weight-gradient stream is on; exact v_mfma_f32_32x32x2_f32)"
        return gemm

    def close(self):
        import torch
        torch.cuda.synchronize()
        if self.feed is not None:
            self.feed.close()
        self.keep = []
        self.net.close()
        if self.rccl is not None:
            self.rccl.close()
        if self.archive is not None:
            import shutil
            shutil.rmtree(os.path.dirname(self.archive), ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4, help="untimed steps (after the setup minibatches, see --ng-burn-in)")
    ap.add_argument("--ng-burn-in", type=int, default=10,
                    help="setup, with natural gradient on: minibatches run before the warmup so that the preconditioners are past their "
                         "first 10 calls, which refresh on EVERY call (OnlineNaturalGradient's num_initial_updates); afterwards every 4th "
                         "does, and that steady state -- refresh steps included -- is what is timed.  0: time a fresh process's first steps")
    ap.add_argument("--roofline-steps", type=int, default=4,
                    help="timed steps whose GEMM launches carry HIP events for the live roofline: the first 4 = one whole refresh cycle of the "
                         "preconditioners (1 refresh step + 3), so the sample has the timed region's own mix (0: all of them)")
    ap.add_argument("--chunk", type=int, default=1500, help="frames per chunk (north_star: 1500-frame chunks)")
    ap.add_argument("--minibatch", type=int, default=128, help="sequences per GPU (--scaling weak) or per node (--scaling strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every GPU runs --minibatch sequences, the global batch grows with N; strong: ONE minibatch of "
                         "--minibatch sequences is sharded, rank g takes minibatch / N of them (SURVEY.md 8(e))")
    ap.add_argument("--sync-batchnorm", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: all-reduce the train-mode BatchNorm column sums over the ranks (ChainNet.set_batchnorm_sync); auto = on for "
                         "--scaling strong (a sharded minibatch then equals the single-GPU one), off for weak (per-job statistics, as Kaldi)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1 with --scaling weak: skip the extra strong-scaling measurement reported under \"strong\"")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one all-reduce of the whole gradient buffer after the backward pass instead of "
                                                               "one per bucket overlapped with it")
    ap.add_argument("--den-states", type=int, default=4000)
    ap.add_argument("--den-degree", type=float, default=12.0)
    ap.add_argument("--cpu-sequences", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the timed CPU oracle leg (the parity check stays)")
    ap.add_argument("--no-parity", action="store_true", help="skip the HIP-against-oracle check of the bounded sample as well")
    ap.add_argument("--no-also", action="store_true", help="skip the further line items (recipe egs shape, 10 000 / 30 000-state denominator graphs)")
    ap.add_argument("--workload", default="7q", choices=["7q", "darts-offset", "darts-offset-cvupdate", "bn-supernet"],
                    help="7q = BASELINE configs[1] (default, the metric's config); darts-offset = configs[3], the K-tap offset "
                         "supernet of run_TDNN_DARTSV3_fbk_stride_pretrain.sh in pretrain (uniform-sample) mode; bn-supernet = "
                         "configs[4], the bottleneck-dimension supernet (8 candidate dims up to 240) in Onehot pretrain mode")
    ap.add_argument("--darts-offsets", type=int, default=7)
    ap.add_argument("--bn-choices", default="reference", choices=["reference", "baseline"],
                    help="bottleneck supernet candidates: reference = 25,50,80,100,120,160,200,240 (the recipe's 8); baseline = 80,160,240,320 "
                         "(BASELINE configs[4])")
    ap.add_argument("--gemm", default="f32", choices=["f32", "bf16x3", "bf16x6", "f16x3"],
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
    ap.add_argument("--also-only", default=None, metavar="REGEX", help="run only the further line items whose description matches (diagnostics)")
    ap.add_argument("--phases", action="store_true", help="diagnostics (stderr): the last timed step's time per phase on the caller's stream (sets option phase_events)")
    ap.add_argument("--host-lead", action="store_true", help="diagnostics (stderr): per timed step, when the GPU finished it minus when the host returned from issuing it")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="a tuning option of the library (tdnnf_set_option: ng_grouped, ng_fuse, ng_early_in, wgrad_stream, gemm_ring, planes); "
                         "repeatable -- same-box A/B runs of two code paths")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python3 bench.py --gpus N` from a bare shell: start the N ranks ourselves, one process per GPU, as the launcher would.  Nothing
        # above has touched the device (a process that has is never re-exec'ed); the children are ordinary subprocesses, this process only
        # waits and hands their status on.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
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

    n_ranks_seen = 1
    if world > 1:  # every rank adds 1: what the process group really spans
        one = torch.ones(1, dtype=torch.float32, device="cpu" if rehearse else "cuda")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        n_ranks_seen = int(one.item())

    lib = pkg.hipabi.load()  # raises if the HIP library is missing: there is no fallback path
    if args.phases:
        args.option = list(args.option) + ["phase_events=1"]
    for spec in args.option:
        name, _, value = spec.partition("=")
        pkg.hipabi.check(lib.tdnnf_set_option(name.encode(), int(value)))
    if args.scaling == "strong":
        b0, b1 = pkg.trainer.shard_sequences(args.minibatch, rank, world)
        if args.minibatch % world:
            raise SystemExit(f"--scaling strong: {args.minibatch} sequences do not divide over {world} ranks")
        seqs = b1 - b0
    else:
        seqs = args.minibatch

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    burn = args.ng_burn_in if args.natural_gradient else 0
    stats0 = None
    if args.workload == "darts-offset-cvupdate":  # the parent's BatchNorm / ReLU statistics: a few pretrain steps of the same supernet
        pj = Job(pkg, args, args.chunk, seqs, args.den_states, rank, world, workload="darts-offset")
        pj.run(0, 0, 3, sync)
        stats0 = pj.net.get_stats()
        pj.close()
    job = Job(pkg, args, args.chunk, seqs, args.den_states, rank, world, stats=stats0)
    cfg = job.cfg
    dt = job.run(burn, args.warmup, args.steps, sync, profile=True)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res = job.net.results.cpu().numpy()
    den_arcs = int(len(job.den["src"]))
    sync_bn_main = job.sync_bn
    rccl_path_main = job.rccl_path
    arithmetic_main, routed_main = job.arithmetic_run(args.gemm), job.planes_routed_per_step
    # N > 1, weak scaling (what the driver runs): the same node also timed on ONE minibatch of --minibatch sequences sharded over the
    # ranks (strong scaling, synchronised BatchNorm) -- the regime north_star's ">= 6x at 8 GPUs" is the harder question in
    strong = None
    if world > 1 and args.scaling == "weak" and not args.no_strong and args.minibatch % world == 0:
        sj = Job(pkg, args, args.chunk, args.minibatch // world, args.den_states, rank, world, scaling="strong")
        sdt = sj.run(burn, args.warmup, args.steps, sync)
        t = torch.tensor([sdt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sdt = float(t.item())
        strong = {"scaling": "strong", "value": round(args.minibatch * args.chunk * args.steps / sdt, 1), "unit": "frames/s",
                  "ms_per_step": round(1e3 * sdt / args.steps, 3), "global_batch": args.minibatch, "sequences_per_gpu": args.minibatch // world,
                  "sync_batchnorm": bool(sj.sync_bn)}
        sj.close()

    # live roofline of the dominant kernel class (HIP events recorded on the launch stream)
    classes = []
    for k in range(8):  # 0..3 the GEMM classes, 4..7 the HBM-bound passes (bn_apply_bypass, bn_relu_bwd, denominator, planes_split)
        n, ms, fl, by = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        pkg.hipabi.check(lib.tdnnf_profile_read(k, C.byref(n), C.byref(ms), C.byref(fl)))
        pkg.hipabi.check(lib.tdnnf_profile_read_bytes(k, C.byref(by)))
        classes.append(dict(name=lib.tdnnf_profile_class_name(k).decode(), launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value))
    hbm_classes, classes = classes[4:], classes[:4]
    # "dominant" = the TDNN-F factored GEMM class with the most time (north_star's kernel); the natural-gradient class (HBM-bound
    # passes over the layer inputs / output derivatives) is reported beside it as roofline_secondary
    dom = max(classes[:3], key=lambda c: c["ms"])
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    # HBM traffic per launch from the PMC passes of THIS command (tools/make_profiles.py writes the file; null when the
    # file is not from this round's kernels)
    tfile = TRAFFIC_FILE_F16X3 if args.gemm == "f16x3" else TRAFFIC_FILE
    tj = load_traffic(tfile) if args.workload == "7q" and args.chunk == 1500 and args.minibatch == 128 else {}
    traffic = tj.get(dom["name"], {}).get("hbm_bytes_per_launch")
    traffic_src = ("committed PMC profile of this command with this round's kernels (%s: separate rocprofv3 --pmc passes, FETCH_SIZE / WRITE_SIZE as the MI355X "
                   "guide prescribes, 2 FETCH + WRITE), NOT counters of this process" % tfile) if traffic else None
    alg_per_launch = dom["bytes"] / max(dom["launches"], 1)
    job.event_steps_ = getattr(job, "event_steps", 1)
    ngc = classes[3]
    job.close()

    if rank == 0:
        frames = world * seqs * args.chunk * args.steps
        out = {
            "metric": "LF-MMI training frames/sec per node (SWBD 7q TDNN-F)",
            "value": round(frames / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": args.scaling, "n_ranks_seen": n_ranks_seen,
            "vs_baseline": None, "dtype": {"f32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA, 2 planes / 3 products, f32 accumulate)",
                                          "bf16x6": "bf16x6 (split-bf16 MFMA, 3 planes / 6 products, f32 accumulate; f32-equivalent)",
                                          "f16x3": "f16x3 (operands pre-split into 2 scaled f16 planes, 3 products on the f16 MFMA, f32 accumulate; f32-equivalent)"}[args.gemm]
                     if arithmetic_main == args.gemm else arithmetic_main,
            "data": "synthetic",
            "config": {"workload": workload_text(args),
                       "frames_per_chunk": args.chunk, "sequences_per_gpu": seqs, "global_batch": world * seqs,
                       "den_graph": {"states": args.den_states, "arcs": den_arcs},
                       "natural_gradient": {"on": bool(args.natural_gradient), "rank_in": 20, "rank_out": 80, "update_period": 4,
                                            "setup_minibatches_before_warmup": burn,
                                            "refresh_steps_in_timed_region": sum(1 for t in range(burn + args.warmup, burn + args.warmup + args.steps)
                                                                                 if t <= 10 or (t - 10) % 4 == 0) if args.natural_gradient else 0},
                       "parallelism": f"dp{world}", "sync_batchnorm": bool(sync_bn_main), "rccl_library": rccl_path_main, "plane_gemms_routed_per_step": routed_main if args.gemm != "f32" else None, "allreduce": None if world == 1 else ("one collective after backward" if args.no_overlap else
                                                                                           "per gradient bucket, overlapped with backward"), "objf_per_frame": float(res[0] / res[2]) if res[2] else None},
            "roofline": {"bound": "mfma", "kernel": dom["name"], "kernel_choice": "the TDNN-F factored-GEMM class with the most time "
                         "(forward / backward-data 128x128, 128x160, weight gradient); the natural-gradient class is roofline_secondary",
                         "note": "launch durations are event-timed in the real step, i.e. with whatever runs beside the launch on other streams: since "
                         "round 3 the natural-gradient input-side statistics (HBM-bound) run on another stream while the caller's stream waits for the "
                         "denominator and beside the first GEMMs of the backward pass (DESIGN.md 4n)",
                         "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": round(alg_per_launch, 1),
                         "traffic_over_algorithmic": round(traffic / alg_per_launch, 3) if traffic and alg_per_launch else None,
                         "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4), "launches": int(dom["launches"]),
                         "all_kernels": [{"kernel": c["name"], "launches": int(c["launches"]), "ms": round(c["ms"], 3),
                                          "tflops": round(c["flops"] / (c["ms"] * 1e-3) / 1e12, 2) if c["ms"] > 0 else 0.0,
                                          "algorithmic_gb_per_s": round(c["bytes"] / (c["ms"] * 1e-3) / 1e9, 1) if c["ms"] > 0 else 0.0,
                                          "flops_per_step": round(c["flops"] / job.event_steps_, 1), "algorithmic_bytes_per_step": round(c["bytes"] / job.event_steps_, 1)}
                                         for c in classes],
                         "event_steps": job.event_steps_,
                         "event_steps_note": "HIP events bracket every GEMM launch of the first event_steps of the timed steps (the launch "
                                             "durations above are theirs); the other timed steps run without the ~1 000 marker packets per step"},
        }

        if args.natural_gradient and ngc["ms"] > 0:
            gbs = ngc["bytes"] / (ngc["ms"] * 1e-3) / 1e9
            out["roofline_secondary"] = {"bound": "hbm", "kernel": ngc["name"], "what": "natural-gradient statistics passes (H = X W^T either side, J = H^T X on a "
                                         "refresh, the rank-R products): algorithmic bytes / HIP-event time of the class", "achieved": round(gbs, 1),
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "tflops": round(ngc["flops"] / (ngc["ms"] * 1e-3) / 1e12, 2),
                                         "launches": int(ngc["launches"]), "ms_per_step": round(ngc["ms"] / job.event_steps_, 3),
                                         "traffic": tj.get(ngc["name"], {}).get("hbm_bytes_per_launch"), "algorithmic_bytes": round(ngc["bytes"] / max(ngc["launches"], 1), 1),
                                         # [r5] these passes are NOT purely HBM-bound in exact f32: H = X W^T at rank 80 is 48 FLOP per operand byte (rank 20
                                         # padded to a 32-column MFMA tile: 16), and the f32 matrix cores give 157 TFLOP/s -- the rank-80 pass over the
                                         # 6034-wide output derivative runs at 75-100 TFLOP/s of MFMA work, i.e. at the matrix cores' rate; longer K steps
                                         # (twice the bytes in flight) change nothing (docs/experiments.md r5-g).  Both fractions, algorithmic FLOPs:
                                         "mfma": {"achieved": round(ngc["flops"] / (ngc["ms"] * 1e-3) / 1e12, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                  "frac": round(ngc["flops"] / (ngc["ms"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                                  "note": "algorithmic FLOPs (rank 20 / 80); the kernels' MFMA tiles are 32 / 96 columns wide: 1.6 x / 1.2 x this in issued work"}}

        # "fraction of HBM roofline on elementwise / denominator work" (hbm_entries above)
        out["roofline_hbm"] = hbm_entries(hbm_classes, job.event_steps_, args.chunk, tj, tfile)

        if strong is not None:
            out["strong"] = strong

        def line_item(name, chunk, sequences, den_states, gemm=None, steps=None, natural_gradient=None, workload=None, stats=None, archive_minibatches=0,
                      want_stats=False, profile=False):
            if args.also_only and not name.startswith("--gemm") and not __import__("re").search(args.also_only, name):
                return {"what": name, "skipped": True, "value": 0.0, "ms_per_step": float("nan"), "_stats": None}
            j = Job(pkg, args, chunk, sequences, den_states, rank, world, gemm=gemm, natural_gradient=natural_gradient, workload=workload, stats=stats,
                    archive_minibatches=archive_minibatches)
            k = args.steps if steps is None else steps
            d = j.run(burn, args.warmup, k, lambda: torch.cuda.synchronize(), profile=profile)
            arcs = int(len(j.den["src"]))
            r = j.net.results.cpu().numpy()
            it = {"what": name, "value": round(sequences * chunk * k / d, 1), "unit": "frames/s", "ms_per_step": round(1e3 * d / k, 3),
                  "frames_per_chunk": chunk, "sequences": sequences, "den_graph": {"states": den_states, "arcs": arcs},
                  "objective_finite": bool(r[5] == 1.0 and np.isfinite(r[0]))}
            g_ = args.gemm if gemm is None else gemm
            if g_ != "f32":
                it["plane_gemms_routed_per_step"] = j.planes_routed_per_step
                it["arithmetic_run"] = j.arithmetic_run(g_)
            if archive_minibatches:
                it["archive"] = {"minibatches": archive_minibatches, "bytes": j.archive_bytes, "features": "16-bit compressed (CompressedMatrix kTwoByteAuto)"}
            if want_stats:
                it["_stats"] = j.net.get_stats()
            if profile:  # the event-timed GEMM classes of this job (as the headline's "roofline")
                it["_classes"], it["_hbm_classes"], it["_event_steps"] = [], [], j.event_steps
                for kc in range(8):
                    n_, ms_, fl_, by_ = C.c_double(), C.c_double(), C.c_double(), C.c_double()
                    pkg.hipabi.check(lib.tdnnf_profile_read(kc, C.byref(n_), C.byref(ms_), C.byref(fl_)))
                    pkg.hipabi.check(lib.tdnnf_profile_read_bytes(kc, C.byref(by_)))
                    it["_classes" if kc < 4 else "_hbm_classes"].append(dict(name=lib.tdnnf_profile_class_name(kc).decode(), launches=n_.value, ms=ms_.value, flops=fl_.value,
                                                                             bytes=by_.value))
            j.close()
            torch.cuda.empty_cache()
            return it

        if world == 1 and args.gemm == "f32" and not args.no_alt:
            # the same step with the f32-equivalent arithmetic on the 16-bit matrix cores (gemm_precision 3, "f16x3": every GEMM operand pre-split
            # into two scaled f16 planes in HBM, three f16 MFMA products, f32 accumulation): same workload, same steps.  Not the
            # headline -- the reference computes in f32 (BaseFloat) and the headline stays the exact-f32 MFMA -- but held to the same parity bars
            it = line_item("--gemm f16x3", args.chunk, seqs, args.den_states, gemm="f16x3", profile=True)
            out["alt"] = {"gemm": "f16x3 (operands pre-split into 2 scaled f16 planes, 3 products on v_mfma_f32_32x32x16_f16, f32 accumulate; f32-equivalent; --gemm f16x3)",
                          "value": it["value"], "unit": "frames/s", "ms_per_step": it["ms_per_step"]}
            # its roofline against the 16-bit matrix-core peak (BASELINE configs[4]: "fp32 objf / bf16 MFMA GEMM"): every f32 multiply-add of the
            # algorithm is three f16 MFMA multiply-adds here (h h' + h l' + l h'), so the matrix cores do 3 x the algorithmic FLOPs; both rates are
            # given, the fraction is of the DENSE 16-bit peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s for bf16 and f16 alike)
            cl = [c for c in it.get("_classes", [])[:3] if c["ms"] > 0]
            tj16 = load_traffic(TRAFFIC_FILE_F16X3) if args.workload == "7q" and args.chunk == 1500 and args.minibatch == 128 else {}
            # the PMC summary names the plane kernels by their own classes (tools/make_profiles.py klass())
            pmc16 = {"rows_gemm_f32_128x128": "planes_gemm_f16x3_256x256", "rows_gemm_f32_128x160": "planes_gemm_f16x3_256x160", "wgrad_f32": "planes_gemm_f16x3_wgrad"}
            if cl:
                dm = max(cl, key=lambda c: c["ms"])
                eq = dm["flops"] / (dm["ms"] * 1e-3) / 1e12
                nm = {"rows_gemm_f32_128x128": "planes_gemm_f16x3 (256-/128-column tiles) + the GEMMs left on their own kernels",
                      "rows_gemm_f32_128x160": "planes_gemm_f16x3 (160-column tiles)", "wgrad_f32": "planes_gemm_f16x3 (weight gradients, split-K)"}
                out["alt"]["roofline"] = {"bound": "mfma", "kernel": nm.get(dm["name"], dm["name"]), "achieved": round(3.0 * eq, 2), "peak": BF16_PEAK_TFLOPS,
                                          "unit": "TFLOP/s", "frac": round(3.0 * eq / BF16_PEAK_TFLOPS, 4), "f32_equivalent_tflops": round(eq, 2),
                                          "f16_mfma_flops_per_algorithmic_flop": 3, "traffic": tj16.get(pmc16.get(dm["name"], ""), {}).get("hbm_bytes_per_launch"),
                                          "traffic_source": ("%s, class %s: PMC bytes per launch of the plane kernels of this event class (the event class also holds "
                                                             "the few launches left on the f32 kernels)" % (TRAFFIC_FILE_F16X3, pmc16.get(dm["name"])))
                                          if tj16.get(pmc16.get(dm["name"], "")) else None,
                                          "algorithmic_bytes": round(dm["bytes"] / max(dm["launches"], 1), 1), "event_steps": it.get("_event_steps"),
                                          "all_kernels": [{"kernel": nm.get(c["name"], c["name"]), "launches": int(c["launches"]), "ms": round(c["ms"], 3),
                                                           "f32_equivalent_tflops": round(c["flops"] / (c["ms"] * 1e-3) / 1e12, 2),
                                                           "frac_of_16bit_peak": round(3.0 * c["flops"] / (c["ms"] * 1e-3) / 1e12 / BF16_PEAK_TFLOPS, 4)} for c in cl],
                                          "note": "event classes as in the headline's roofline (the class names are the f32 kernels'): forward / backward-data GEMMs by "
                                                  "output width, weight gradients; the plane splits (HBM passes, class planes_split) are not in these times but in ms_per_step"}
                out["alt"]["roofline_hbm"] = hbm_entries(it.get("_hbm_classes", []), max(it.get("_event_steps") or 1, 1), args.chunk, tj16, TRAFFIC_FILE_F16X3)
        if world == 1 and not args.no_also:
            # further line items, same net and step: the reference's own egs shape (chunk 150 x 64, ...pretrain.sh:46,197) and
            # SWBD-scale denominator graphs (10 000 and 30 000 states)
            recipe = line_item("the recipes' egs shape (--chunk 150 --minibatch 64)", 150, 64, args.den_states, steps=40)
            shard = line_item("the per-GPU shard of a 128-sequence minibatch at 8 GPUs, --scaling strong (--chunk 1500 --minibatch 16)", args.chunk, 16, args.den_states,
                              steps=16)
            shard["vs_one_eighth_of_the_headline_step"] = round(shard["ms_per_step"] / (1e3 * dt / args.steps / 8.0), 3)
            shard["ideal_speedup_at_8_gpus_before_any_collective"] = round(8.0 / shard["vs_one_eighth_of_the_headline_step"], 2)
            out["also"] = [recipe, shard,
                           line_item("10 000-state denominator graph (--den-states 10000)", args.chunk, seqs, 10000, steps=4),
                           line_item("30 000-state denominator graph (--den-states 30000)", args.chunk, seqs, 30000, steps=4)]
            if args.natural_gradient:  # what the preconditioning costs: the same step with raw gradients
                it = line_item("natural gradient off (--natural-gradient 0)", args.chunk, seqs, args.den_states, natural_gradient=0)
                it["natural_gradient_cost_ms_per_step"] = round(1e3 * dt / args.steps - it["ms_per_step"], 3)
                out["also"].append(it)
            if args.workload == "7q":
                # the supernets north_star's scaling target is quoted on, same shape and step (BASELINE configs[3] / [4]):
                # offset supernet pretrain (uniform tap sample), its cv-update (Gumbel over all K taps, BatchNormTest from the
                # pretrain net's statistics, run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:128-142), bottleneck-dimension supernet
                pre = line_item("DARTS offset supernet, pretrain (--workload darts-offset)", args.chunk, seqs, args.den_states, steps=6, workload="darts-offset", want_stats=True)
                stats = pre.pop("_stats")
                out["also"].append(pre)
                out["also"].append(line_item("DARTS offset supernet, cv-update: Gumbel over all %d taps, BatchNormTest (--workload darts-offset-cvupdate)" % args.darts_offsets,
                                             args.chunk, seqs, args.den_states, steps=6, workload="darts-offset-cvupdate", stats=stats))
                bnsup = line_item("bottleneck-dimension supernet, Onehot pretrain (--workload bn-supernet)", args.chunk, seqs, args.den_states, steps=6, workload="bn-supernet")
                out["also"].append(bnsup)
                # north_star quotes the >= 6x scaling target "on the SWBD 7q DARTS supernet": the per-GPU shards of the two supernets at 8 GPUs
                # (strong scaling of the 128-sequence minibatch), against an eighth of their own one-GPU step
                for nm, wl, full in (("DARTS offset supernet, pretrain", "darts-offset", pre), ("bottleneck-dimension supernet, Onehot pretrain", "bn-supernet", bnsup)):
                    it = line_item(nm + ": the per-GPU shard of a 128-sequence minibatch at 8 GPUs (--workload %s --chunk 1500 --minibatch 16)" % wl, args.chunk, 16,
                                   args.den_states, steps=12, workload=wl)
                    it["vs_one_eighth_of_its_128_sequence_step"] = round(it["ms_per_step"] / (full["ms_per_step"] / 8.0), 3)
                    it["ideal_speedup_at_8_gpus_before_any_collective"] = round(8.0 / it["vs_one_eighth_of_its_128_sequence_step"], 2)
                    out["also"].append(it)
                if not args.no_alt:  # the offset supernet on the plane kernels: tap coefficients folded into the weight planes, zero taps skipped in the kernels
                    out["also"].append(line_item("DARTS offset supernet, pretrain, f32-equivalent 16-bit matrix-core arithmetic (--workload darts-offset --gemm f16x3)",
                                                 args.chunk, seqs, args.den_states, steps=6, workload="darts-offset", gemm="f16x3"))
                    out["also"].append(line_item("DARTS offset supernet, cv-update, f32-equivalent 16-bit matrix-core arithmetic (--workload darts-offset-cvupdate --gemm f16x3)",
                                                 args.chunk, seqs, args.den_states, steps=6, workload="darts-offset-cvupdate", stats=stats, gemm="f16x3"))
                if not args.no_alt:  # configs[4] as BASELINE.json words it: "fp32 objf / bf16 MFMA GEMM" -- the 16-bit matrix cores with an f32 objective
                    out["also"].append(line_item("bottleneck-dimension supernet with the f32-equivalent 16-bit matrix-core arithmetic (--workload bn-supernet --gemm f16x3: "
                                                 "BASELINE configs[4], fp32 objective / 16-bit MFMA GEMMs)", args.chunk, seqs, args.den_states, steps=6,
                                                 workload="bn-supernet", gemm="f16x3"))
            # archive-fed: every minibatch read from a cegs archive, decompressed, merged and copied to the device inside the timed loop
            for (ch, sq, st, resident) in ((args.chunk, seqs, 8, 1e3 * dt / args.steps), (150, 64, 40, recipe["ms_per_step"])):
                it = line_item("archive-fed (egs.minibatches(prefetch=2): read + decompress + merge on a worker thread, H2D included), --chunk %d --minibatch %d" % (ch, sq),
                               ch, sq, args.den_states, steps=st, archive_minibatches=4)
                it["vs_resident_inputs"] = round(resident / it["ms_per_step"], 3)
                out["also"].append(it)
        ok = True
        if world == 1 and not args.no_parity:
            parity, base = parity_and_cpu_baseline(pkg, args, want_baseline=not args.no_cpu_baseline)
            out["parity"] = parity
            ok = parity["ok"]
            if base is not None:
                out["cpu_baseline"] = base
        if "alt" in out and "parity" in out and "f16x3" in out["parity"]:
            out["alt"]["parity"] = out["parity"].pop("f16x3")
        print(json.dumps(out), flush=True)
        if not ok:
            raise SystemExit("bench.py: the HIP step does not match the oracle on the parity sample: " + json.dumps(out["parity"]))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
