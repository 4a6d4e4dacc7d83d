"""The outer training loop around the chain trainer: what steps/nnet3/chain/train.py:405-560 does between the hot-path
calls -- iteration count, per-iteration number of jobs / learning rate / shrinkage / temperature proportion, one model
file per iteration, averaging of the jobs' models, final combination.  Host orchestration only;
the per-minibatch work is ChainNet.forward_backward + update.

`chain_lib.train_one_iteration` / `combine_models` (steps/libs/nnet3/train/chain_objf/acoustic_model.py) are upstream
Kaldi, not shipped with the reference; restated here from the command lines train.py passes them:
 * every job of an iteration starts from <iter>.mdl, trains on its own archive with the iteration's learning rate
   (nnet3-chain-train; a fresh process, so the natural-gradient state starts from scratch), writes <iter+1>.<job>.raw;
 * every job starts from <iter>.mdl scaled by the shrinkage value; the jobs' models are averaged (nnet3-average) into <iter+1>.mdl;
 * final.mdl = combination of the models of `model_combine_iters`: with combine egs given, nnet3-chain-combine's search
   (combine_models below: running average over the models, latest first, the average with the best objective on the
   combine egs wins, its BatchNorm statistics are recomputed); without, their plain average.  Restated from the upstream
   binary's published behaviour, not pinned.
 * diagnostics: compute_prob = nnet3-chain-compute-prob (BatchNorm in test mode, no dropout, no update).
On several GPUs the jobs of an iteration are spread over the ranks (one process per GPU, torch.distributed over RCCL) and
the average is one all-reduce of parameters and statistics -- the same arithmetic, no gradient exchange.
"""
import os

import numpy as np

from . import trainer as T


def num_iterations(num_epochs, num_archives, frame_subsampling_factor, num_jobs_initial, num_jobs_final):
    """(num_archives_to_process, num_iters) of train.py:406,443-447."""
    num_archives_expanded = num_archives * frame_subsampling_factor
    if num_jobs_final > num_archives_expanded:
        raise ValueError('num_jobs_final cannot exceed the expanded number of archives')  # train.py:408-410
    to_process = int(num_epochs * num_archives_expanded)
    return to_process, (to_process * 2) // (num_jobs_initial + num_jobs_final)


def model_combine_iters(num_iters, num_epochs, num_archives, max_models_combine, num_jobs_final):
    """The iterations whose models enter the final combination (steps/libs/nnet3/train/common.py:562-603); num_archives is
    the expanded count, as train.py:455-459 passes it."""
    approx_iters_per_epoch_final = float(num_archives) / num_jobs_final
    initial = min(int(approx_iters_per_epoch_final / 2) + 1, int(num_iters / 2))
    if initial > max_models_combine:
        factor = int(float(initial) / max_models_combine)
        models = set(range(num_iters - initial + 1, num_iters + 1, factor))
        models.add(num_iters)
    else:
        n = min(max_models_combine, num_iters // 2)
        models = set(range(num_iters - n + 1, num_iters + 1))
    return models


def shrinkage_value(learning_rate, proportional_shrink=0.0):
    """train.py:488-492 (the saturation-triggered --shrink-value needs nnet3-am-info and is left at its default 1.0)."""
    v = 1.0 - proportional_shrink * learning_rate
    if v <= 0.5:
        raise ValueError("proportional-shrink={0} is too large, it gives shrink-value={1}".format(proportional_shrink, v))
    return v


def iteration_plan(num_epochs, num_archives, frame_subsampling_factor=3, num_jobs_initial=1, num_jobs_final=1, initial_effective_lrate=2.5e-4,
                   final_effective_lrate=2.5e-5, proportional_shrink=0.0, temperature_schedule=False, dropout_schedule=None):
    """One dict per iteration with everything train.py:473-531 derives before it launches the jobs."""
    to_process, num_iters = num_iterations(num_epochs, num_archives, frame_subsampling_factor, num_jobs_initial, num_jobs_final)
    plan = []
    processed = 0
    for it in range(num_iters):
        jobs = int(0.5 + num_jobs_initial + (num_jobs_final - num_jobs_initial) * float(it) / num_iters)
        lr = T.learning_rate(it, jobs, num_iters, processed, to_process, initial_effective_lrate, final_effective_lrate)
        frac = float(processed) / to_process
        num_archives_expanded = num_archives * frame_subsampling_factor
        plan.append(dict(iteration=it, num_jobs=jobs, learning_rate=lr, shrink=shrinkage_value(lr, proportional_shrink), data_fraction=frac,
                         temperature_proportion=T.temperature_proportion(frac) if temperature_schedule else None,
                         # train.py:519-527: get_dropout_edit_string(dropout_schedule, data fraction, iter)
                         dropout_proportion=T.dropout_proportion(dropout_schedule, frac) if dropout_schedule else None,
                         # job j (1-based) of this iteration reads archive (processed + j - 1) % expanded + 1, i.e. archive k of the
                         # egs dir at frame shift (k_expanded // num_archives) -- the upstream convention, stated
                         archives=[(processed + j) % num_archives_expanded for j in range(jobs)]))
        processed += jobs
    return plan


def archive_and_frame_shift(expanded_index, num_archives, frame_subsampling_factor=3):
    """(1-based archive of the egs dir, frame shift) for the k-th processed archive, k = expanded_index (the numbers in
    iteration_plan()['archives']): chain_lib.train_new_models (upstream) reads archive k % num_archives + 1 with the inputs
    shifted by (archive + k // num_archives) % frame_subsampling_factor frames, so that every archive is seen at every shift
    over the epochs.  Pass the shift to egs.minibatches(..., frame_shift=...)."""
    archive = expanded_index % num_archives + 1
    return archive, (archive + expanded_index // num_archives) % frame_subsampling_factor


def average_models(models):
    """nnet3-average: the mean of parameters and of the stored statistics (BatchNorm count / sums, ReLU averages are linear
    in Component::Add / Scale).  models = [(params, stats)] numpy arrays."""
    p = np.mean(np.stack([m[0].astype(np.float64) for m in models]), axis=0).astype(np.float32)
    s = np.mean(np.stack([np.asarray(m[1], np.float64) for m in models]), axis=0)
    return p, s


def evaluation_config(config, batchnorm_test_mode):
    """The configuration of a net that only evaluates the objective of the same model: no natural gradient state, and with
    batchnorm_test_mode the BatchNorm components use the stored statistics (cv_update: BatchNormTest, every learning-rate
    factor of the weights 0, so no parameter gradient is formed).  Same parameter and statistics layout as `config`."""
    c = type(config).from_buffer_copy(bytes(config))
    c.use_natural_gradient = 0
    if batchnorm_test_mode:
        c.cv_update = 1
    return c


def _objective(net, minibatches, seed=0):
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    tot = np.zeros(7)
    for m, (feats, iv, den, sup) in enumerate(minibatches):
        if net.num_draws:
            net.set_random_draws(generator=g)
        tot += net.forward_backward(feats, iv, den, sup, step=m).cpu().numpy()[:7]
    return tot


def compute_prob(net, minibatches):
    """nnet3-chain-compute-prob on a net built from evaluation_config(cfg, True) that holds the model: per-frame objective of
    the chain output (objf + l2 term) and of the xent output, as the "Overall log-probability for 'output' / 'output-xent'"
    log lines give them."""
    if net.cfg.use_dropout:
        net.set_dropout_proportion(0.0)  # dropout test mode
    t = _objective(net, minibatches)
    return dict(output=(t[0] + t[1]) / t[2], output_xent=t[6] / t[2], weight=t[2])


def combine_models(net, models, num_models, minibatches, max_objective_evaluations=30, log=None):
    """nnet3-chain-combine (UPSTREAM, restated from its usage text and log lines): models = an iterable of num_models (parameters,
    statistics) pairs, latest first (train.py hands the files over in reversed order); a running average A_n = A_{n-1} (n-1)/n + M_n / n over parameters and statistics; the
    objective (objf + l2 term) / weight of 'output' on the combine egs -- BatchNorm in TRAINING mode, dropout in test mode, the
    binary's defaults -- is evaluated for n = 1 and then every `mod` models, mod = ceil(num_models / max_objective_evaluations);
    the best average wins and its BatchNorm statistics are recomputed on the same egs.  net: built from
    evaluation_config(cfg, False).  Leaves the winner in net (parameters + statistics); returns (number combined, objf before,
    objf after)."""
    minibatches = list(minibatches)
    if net.cfg.use_dropout:
        net.set_dropout_proportion(0.0)
    mod = (num_models + max_objective_evaluations - 1) // max_objective_evaluations
    avg_p = avg_s = best = None
    for n, (p, s) in enumerate(models):
        p, s = np.asarray(p, np.float64), np.asarray(s, np.float64)
        if n == 0:
            avg_p, avg_s = p, s
        else:
            avg_p = avg_p * (n / (n + 1.0)) + p / (n + 1.0)
            avg_s = avg_s * (n / (n + 1.0)) + s / (n + 1.0)
        if n == 0 or (n - 1) % mod == 0:
            net.set_params(avg_p.astype(np.float32))
            net.set_stats(avg_s)
            t = _objective(net, minibatches)
            objf = (t[0] + t[1]) / t[2]
            if log:
                log("Combining last %d models, objective function is %.6f" % (n + 1, objf))
            if n == 0:
                first = objf
            if best is None or objf > best[0]:
                best = (objf, n + 1, avg_p.copy(), avg_s.copy())
    objf, count, p, s = best
    net.set_params(p.astype(np.float32))
    # RecomputeStats: zeroed statistics, one training-mode pass over the egs accumulates them again
    net.set_stats(np.zeros_like(s))
    _objective(net, minibatches)
    if log:
        log("Combining %d nnets, objective function changed from %.6f to %.6f" % (count, first, objf))
    return count, first, objf


def run(net_factory, egs_for_archive, work_dir, num_epochs, num_archives, minibatches_per_archive, frame_subsampling_factor=3, num_jobs_initial=1,
        num_jobs_final=1, initial_effective_lrate=2.5e-4, final_effective_lrate=2.5e-5, proportional_shrink=0.0, temperature_schedule=False,
        do_final_combination=True, max_models_combine=20, srand=0, binary=True, log=None, dropout_schedule=None, combine_egs=None,
        diagnostic_egs=None):
    """Runs the whole schedule.  net_factory() -> a ChainNet with initial parameters set (called once per job: a fresh process in
    the reference, so fresh natural-gradient state; BatchNorm / ReLU statistics and parameters come from <iter>.mdl).
    As upstream's train_one_iteration (absent from the reference, restated from what train.py passes): every job starts from
    <iter>.mdl scaled by the iteration's shrink value (nnet3-am-copy --scale), trains at learning rate lr_eff x num_jobs with
    l2_regularize_factor = 1 / num_jobs, and the jobs' models are averaged.  NOT restated: upstream's special case for
    iteration 0 (minibatch halved, max-param-change / sqrt(2), the best job's model instead of the average) -- a ChainNet is
    built for one minibatch shape; the reference recipes run one job, where only the halved minibatch would differ.
    egs_for_archive(archive_index, minibatch_index) -> (feats, ivectors, den_graph, supervision) device objects for
    ChainNet.forward_backward.  Writes <work_dir>/<iter>.mdl for every iteration and final.mdl; returns the plan with the
    per-iteration mean objective added.  combine_egs: minibatches [(feats, ivectors, den_graph, supervision)] for the final
    combination (combine_models; without them the plain average); diagnostic_egs: {"valid": minibatches, "train": ...}
    evaluated on every iteration's model by compute_prob (train.py's compute_train_cv_probabilities), results in the plan.  Ranks of an initialised torch.distributed group share the jobs of an iteration."""
    import torch
    import torch.distributed as dist
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_available() and dist.is_initialized() else (1, 0)
    os.makedirs(work_dir, exist_ok=True)
    plan = iteration_plan(num_epochs, num_archives, frame_subsampling_factor, num_jobs_initial, num_jobs_final, initial_effective_lrate,
                          final_effective_lrate, proportional_shrink, temperature_schedule, dropout_schedule)
    num_iters = len(plan)
    to_process = int(num_epochs * num_archives * frame_subsampling_factor)
    combine = model_combine_iters(num_iters, num_epochs, num_archives * frame_subsampling_factor, max_models_combine, num_jobs_final) \
        if do_final_combination else None
    path = lambda it: os.path.join(work_dir, "%d.mdl" % it)  # noqa: E731
    net = net_factory()
    if rank == 0:
        net.write_model(path(0), binary=binary)
    step = 0
    prob_net = T.ChainNet(evaluation_config(net.cfg, True)) if diagnostic_egs and rank == 0 else None
    for it in plan:
        i, jobs, lr = it["iteration"], it["num_jobs"], it["learning_rate"]
        if world > 1:
            dist.barrier()
        if prob_net is not None:  # on <iter>.mdl, as train.py does before it launches the iteration's jobs
            net.read_model(path(i))
            prob_net.params.copy_(net.params)
            prob_net.set_stats(net.get_stats())
            it["compute_prob"] = {k: compute_prob(prob_net, v) for k, v in diagnostic_egs.items()}
            if log:
                log("iter %d  " % i + "  ".join("%s objf/frame %.5f (xent %.5f)" % (k, v["output"], v["output_xent"])
                                                  for k, v in it["compute_prob"].items()))
        acc_p = np.zeros(net.num_params, np.float64)
        acc_s = np.zeros(net.get_stats().size, np.float64)
        objf, weight = 0.0, 0.0
        for j in range(rank, jobs, world):
            if j != rank or i > 0:  # a fresh job: new natural-gradient state
                net.close()
                net = net_factory()
            net.read_model(path(i))
            if it["shrink"] != 1.0:
                # the job's input model: "nnet3-am-copy --scale=<shrink>" = ScaleNnet: parameters, and every component's statistics too --
                # NonlinearComponent::Scale (nnet-component-itf.cc:533-541: value / deriv / oderiv sums and both counts) and
                # BatchNormComponent::Scale (nnet-normalize-component.cc:644-654: count, sum, sumsq); every entry of the statistics
                # block is such a sum or count, so the averages they stand for are unchanged and old minibatches weigh less
                net.params.mul_(float(it["shrink"]))
                net.set_stats(net.get_stats() * float(it["shrink"]))
            if it["temperature_proportion"] is not None:
                net.set_temperature_proportion(it["temperature_proportion"])
            if it["dropout_proportion"] is not None:
                net.set_dropout_proportion(it["dropout_proportion"])
            g = torch.Generator(device="cuda")
            g.manual_seed(srand + 1000 * i + j)
            for m in range(minibatches_per_archive):
                feats, iv, den, sup = egs_for_archive(it["archives"][j], m)
                if net.num_draws:
                    net.set_random_draws(generator=g)
                r = net.forward_backward(feats, iv, den, sup, step=step + m)
                # l2 scale = GetNumNvalues x l2_regularize_factor, the factor being 1 / num_jobs (train.py -> train_one_iteration)
                net.update(lr, l2_regularize_scale=float(net.cfg.num_sequences) / jobs, step=step + m)
                r = r.cpu().numpy()
                objf += float(r[0])
                weight += float(r[2])
            acc_p += net.params.detach().cpu().numpy().astype(np.float64)
            acc_s += net.get_stats()
        step += minibatches_per_archive
        if world > 1:
            buf = torch.from_numpy(np.concatenate([acc_p, acc_s, [objf, weight]]))
            if dist.get_backend() != "gloo":  # RCCL reduces device buffers; gloo (rehearsals on one GPU) host ones
                buf = buf.cuda()
            dist.all_reduce(buf)
            buf = buf.cpu().numpy()
            acc_p, acc_s, objf, weight = buf[:acc_p.size], buf[acc_p.size:acc_p.size + acc_s.size], buf[-2], buf[-1]
        it["objf_per_frame"] = objf / weight if weight else float("nan")
        net.set_params((acc_p / jobs).astype(np.float32))  # nnet3-average
        net.set_stats(acc_s / jobs)
        if rank == 0:
            net.write_model(path(i + 1), binary=binary, learning_rate=lr)
            if log:
                log("iter %d/%d  jobs %d  lr %.6g  objf/frame %.5f" % (i, num_iters - 1, jobs, lr, it["objf_per_frame"]))
    if world > 1:
        dist.barrier()
    if rank == 0:
        if combine and combine_egs:
            comb = T.ChainNet(evaluation_config(net.cfg, False))

            def models_latest_first():
                for i in sorted(combine, reverse=True):
                    net.read_model(path(i))
                    yield net.params.detach().cpu().numpy().copy(), net.get_stats().copy()

            combine_models(comb, models_latest_first(), len(combine), combine_egs, log=log)
            net.params.copy_(comb.params)
            net.set_stats(comb.get_stats())
            comb.close()
        elif combine:
            models = []
            for i in sorted(combine):
                net.read_model(path(i))
                models.append((net.params.detach().cpu().numpy().copy(), net.get_stats().copy()))
            p, s = average_models(models)
            net.set_params(p)
            net.set_stats(s)
        else:
            net.read_model(path(num_iters))
        net.write_model(os.path.join(work_dir, "final.mdl"), binary=binary)
    if prob_net is not None:
        prob_net.close()
    net.close()
    return plan, (sorted(combine) if combine else None), to_process
