"""The outer training loop around the chain trainer: what steps/nnet3/chain/train.py:405-560 does between the hot-path
calls -- iteration count, per-iteration number of jobs / learning rate / shrinkage / temperature proportion, one model
file per iteration, averaging of the jobs' models, final combination.  Host orchestration only;
the per-minibatch work is ChainNet.forward_backward + update.

`chain_lib.train_one_iteration` / `combine_models` (steps/libs/nnet3/train/chain_objf/acoustic_model.py) are upstream
Kaldi, not shipped with the reference; restated here from the command lines train.py passes them:
 * every job of an iteration starts from <iter>.mdl, trains on its own archive with the iteration's learning rate
   (nnet3-chain-train; a fresh process, so the natural-gradient state starts from scratch), writes <iter+1>.<job>.raw;
 * the jobs' models are averaged (nnet3-average) into <iter+1>.mdl, scaled by the shrinkage value;
 * final.mdl = combination of the models of `model_combine_iters` -- here their plain average (nnet3-combine's default
   since Kaldi 5.4 when no held-out objective is evaluated: "--combine-sum-to-one-penalty" era weights are gone); stated,
   not pinned.
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


def run(net_factory, egs_for_archive, work_dir, num_epochs, num_archives, minibatches_per_archive, frame_subsampling_factor=3, num_jobs_initial=1,
        num_jobs_final=1, initial_effective_lrate=2.5e-4, final_effective_lrate=2.5e-5, proportional_shrink=0.0, temperature_schedule=False,
        do_final_combination=True, max_models_combine=20, srand=0, binary=True, log=None, dropout_schedule=None):
    """Runs the whole schedule.  net_factory() -> a ChainNet with initial parameters set (called once per job: a fresh process in
    the reference, so fresh natural-gradient state; BatchNorm / ReLU statistics and parameters come from <iter>.mdl).
    egs_for_archive(archive_index, minibatch_index) -> (feats, ivectors, den_graph, supervision) device objects for
    ChainNet.forward_backward.  Writes <work_dir>/<iter>.mdl for every iteration and final.mdl; returns the plan with the
    per-iteration mean objective added.  Ranks of an initialised torch.distributed group share the jobs of an iteration."""
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
    for it in plan:
        i, jobs, lr = it["iteration"], it["num_jobs"], it["learning_rate"]
        if world > 1:
            dist.barrier()
        acc_p = np.zeros(net.num_params, np.float64)
        acc_s = np.zeros(net.get_stats().size, np.float64)
        objf, weight = 0.0, 0.0
        for j in range(rank, jobs, world):
            if j != rank or i > 0:  # a fresh job: new natural-gradient state
                net.close()
                net = net_factory()
            net.read_model(path(i))
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
                net.update(lr, step=step + m)
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
        net.set_params((it["shrink"] * acc_p / jobs).astype(np.float32))  # nnet3-average ... | nnet3-copy --scale=shrink
        net.set_stats(acc_s / jobs)
        if rank == 0:
            net.write_model(path(i + 1), binary=binary, learning_rate=lr)
            if log:
                log("iter %d/%d  jobs %d  lr %.6g  objf/frame %.5f" % (i, num_iters - 1, jobs, lr, it["objf_per_frame"]))
    if world > 1:
        dist.barrier()
    if rank == 0:
        if combine:
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
    net.close()
    return plan, (sorted(combine) if combine else None), to_process
