"""Seeded synthetic inputs of the shapes SURVEY.md 8(d) prescribes: denominator
graphs, numerator (supervision) graphs, fbank/ivector egs and tdnn index sets.
numpy only; shared by tests/, bench.py and __graft_entry__.smoke().

Shapes follow the reference recipes: 40-dim fbank + 100-dim ivector inputs
(local/chain_NAS/run_tdnn_fbk_40_iv_sp_7q.sh:160-161), t-major row order
row = t_index * num_sequences + n (src/nnet3/nnet-tdnn-component.cc:642-646,897-902).
"""
import numpy as np


def make_den_graph(H, P, mean_out_degree=12.0, seed=0):
    """Strongly connected HMM: ring arcs h->h+1 plus Poisson extra arcs, pdf ids
    uniform, transition probs row-normalised Dirichlet(1).  init = average
    occupancy over 100 steps from state 0 (DenominatorGraph::SetInitialProbs, UPSTREAM)."""
    rng = np.random.default_rng(seed)
    src, dst = [], []
    for h in range(H):
        k = max(1, int(rng.poisson(mean_out_degree)))
        d = rng.integers(0, H, size=k)
        d[0] = (h + 1) % H
        src.append(np.full(k, h))
        dst.append(d)
    src = np.concatenate(src).astype(np.int32)
    dst = np.concatenate(dst).astype(np.int32)
    A = len(src)
    pdf = rng.integers(0, P, size=A).astype(np.int32)
    raw = rng.gamma(1.0, 1.0, size=A)
    tot = np.zeros(H)
    np.add.at(tot, src, raw)
    prob = (raw / tot[src]).astype(np.float32)
    # initial probs
    cur = np.zeros(H)
    cur[0] = 1.0
    avg = np.zeros(H)
    for _ in range(100):
        avg += cur / 100
        nxt = np.zeros(H)
        np.add.at(nxt, dst, cur[src] * prob)
        cur = nxt
    return {"H": H, "P": P, "src": src, "dst": dst, "pdf": pdf, "prob": prob,
            "init": avg.astype(np.float32)}


def make_supervision(B, T, P, max_alt=2, seed=0, weight=1.0):
    """Per-sequence time-synchronous numerator graphs: at every frame 1..max_alt
    alternative states, fully connected to the next frame's states, one random
    pdf and a small random log-weight per arc; every state at time T is final."""
    rng = np.random.default_rng(seed)
    state_time, final, a_src, a_dst, a_pdf, a_lp = [], [], [], [], [], []
    seq_state_begin, seq_arc_begin = [0], [0]
    ns = 0
    for _ in range(B):
        counts = [1] + [int(rng.integers(1, max_alt + 1)) for _ in range(T)]
        first = ns + np.concatenate([[0], np.cumsum(counts)[:-1]])
        for t, c in enumerate(counts):
            state_time += [t] * c
            final += [0.0 if t == T else -np.inf] * c
        for t in range(T):
            for i in range(counts[t]):
                for j in range(counts[t + 1]):
                    a_src.append(first[t] + i)
                    a_dst.append(first[t + 1] + j)
                    a_pdf.append(int(rng.integers(0, P)))
                    a_lp.append(float(-rng.random() * 0.5))
        ns += sum(counts)
        seq_state_begin.append(ns)
        seq_arc_begin.append(len(a_src))
    i32 = lambda x: np.asarray(x, dtype=np.int32)
    return {"B": B, "T": T, "seq_state_begin": i32(seq_state_begin),
            "seq_arc_begin": i32(seq_arc_begin), "state_time": i32(state_time),
            "final_logprob": np.asarray(final, dtype=np.float32), "arc_src": i32(a_src),
            "arc_dst": i32(a_dst), "arc_pdf": i32(a_pdf),
            "arc_logprob": np.asarray(a_lp, dtype=np.float32), "weight": float(weight)}


def make_supervision_from_den(den, B, T, num_paths=2, seed=0, weight=1.0):
    """Numerator graphs whose paths are paths of the denominator graph (as real chain supervisions are, after
    composition with the normalization FST): per sequence `num_paths` independent random walks on the den graph
    from one start state drawn from the initial distribution; arc log-weights are the den transition
    log-probs, and a path's first arc also carries the log of the denominator's initial probability of its start state -- the
    weight the denominator's own recursion gives that start (alpha(0, i) = init(i)).  Every numerator path is then a
    denominator path with the same weight, so log p_num <= log p_den and the LF-MMI objective stays <= 0 however long a net is
    trained on it (without the start weight the objective of a long run settled at +log(1 / init) / T per frame)."""
    rng = np.random.default_rng(seed)
    H = den["H"]
    order = np.argsort(den["src"], kind="stable")
    src_sorted = den["src"][order]
    begin = np.searchsorted(src_sorted, np.arange(H + 1))
    init = den["init"].astype(np.float64)
    init = init / init.sum()
    state_time, final, a_src, a_dst, a_pdf, a_lp = [], [], [], [], [], []
    seq_state_begin, seq_arc_begin = [0], [0]
    ns = 0
    for _ in range(B):
        h0 = int(rng.choice(H, p=init))
        # state ids: start, then per frame t = 1..T one state per path
        state_time.append(0)
        final.append(-np.inf)
        for t in range(1, T + 1):
            for _p in range(num_paths):
                state_time.append(t)
                final.append(0.0 if t == T else -np.inf)
        cur = [h0] * num_paths
        for t in range(T):
            for pth in range(num_paths):
                lo, hi = begin[cur[pth]], begin[cur[pth] + 1]
                a = order[int(rng.integers(lo, hi))]
                s_from = ns if t == 0 else ns + 1 + (t - 1) * num_paths + pth
                s_to = ns + 1 + t * num_paths + pth
                a_src.append(s_from)
                a_dst.append(s_to)
                a_pdf.append(int(den["pdf"][a]))
                lp = float(np.log(max(float(den["prob"][a]), 1e-30)))
                if t == 0:
                    lp += float(np.log(max(float(den["init"][h0]), 1e-30)))
                a_lp.append(lp)
                cur[pth] = int(den["dst"][a])
        ns += 1 + T * num_paths
        seq_state_begin.append(ns)
        seq_arc_begin.append(len(a_src))
    i32 = lambda x: np.asarray(x, dtype=np.int32)
    return {"B": B, "T": T, "seq_state_begin": i32(seq_state_begin), "seq_arc_begin": i32(seq_arc_begin),
            "state_time": i32(state_time), "final_logprob": np.asarray(final, dtype=np.float32), "arc_src": i32(a_src),
            "arc_dst": i32(a_dst), "arc_pdf": i32(a_pdf), "arc_logprob": np.asarray(a_lp, dtype=np.float32),
            "weight": float(weight)}


def tdnn_indexes(time_offsets, num_t_out, B, start_t_in=None, t_step_in=1, t_step_out=1,
                 start_t_out=0):
    """Restates TdnnDARTSV3Component::PrecomputeIndexes for a regular grid
    (src/nnet3/nnet-tdnn-component.cc:846-905).  Returns (row_stride, row_offsets,
    num_rows_in, num_rows_out)."""
    offs = list(time_offsets)
    if start_t_in is None:
        start_t_in = start_t_out + min(offs)
    rho = t_step_out // t_step_in
    assert t_step_out % t_step_in == 0
    last_t_in = start_t_out + (num_t_out - 1) * t_step_out + max(offs)
    num_t_in = (last_t_in - start_t_in) // t_step_in + 1
    num_t_in = rho * ((num_t_in + rho - 1) // rho)  # :841-843
    row_offsets = []
    for o in offs:
        req = start_t_out + o
        input_t = (req - start_t_in) // t_step_in
        assert req == start_t_in + t_step_in * input_t
        mult, rem = rho * (input_t // rho), input_t % rho
        row_offsets.append(mult * B + rem)  # :897-902
    return rho, np.asarray(row_offsets, dtype=np.int32), num_t_in * B, num_t_out * B
