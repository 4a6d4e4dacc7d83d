"""Child derivation: from the architecture logits of a searched supernet to the child network the recipes train next.

Host-side mirror (pure Python, no GPU) of the reference's derivation scripts, same inputs and outputs:
  local/chain_NAS/scripts/generate_top_list.py                (offset supernet   -> per-component time offsets)
  local/chain_NAS/scripts/generate_top_list_bottleneckdim.py  (bottleneck search -> per-layer bottleneck dims)
  local/chain_NAS/scripts/generate_optimal_stride.py          (explicit offsets  -> config)
Pinned by tests/golden/r01_derive_golden.json: outputs of those scripts run in the build container
(tests/golden/make_derive_golden.py).

The scripts work on text: the parent's `final_txt.mdl` (nnet3-am-copy --binary=false) and the xconfig-generated
`final.config_temp` / `ref.config_temp`; they write `final.config`, `ref.config` and `arch.txt`.  The same functions
are offered on lists of lines, plus `child_config_kwargs()` which turns a derived architecture into the keyword
arguments of `trainer.make_config` so that the child can be trained by the same chain trainer.
"""
import numpy as np

# candidate bottleneck dims of the bottleneck search (generate_top_list_bottleneckdim.py:59-60)
BOTTLENECK_DIMS = [25, 50, 80, 100, 120, 160, 200, 240]
BEAM = 10  # generate_top_list.py:70


# ---------------------------------------------------------------------------------------------- reading the logits
def _floats_after_bracket(line, n):
    # "... [ a b c ... ]": split on '[' then on single blanks; element 0 is the empty string before the first number
    # (generate_top_list.py:25, generate_top_list_bottleneckdim.py:25)
    toks = line.split('[')[1].split(' ')
    return [float(t) for t in toks[1:n + 1]]


def offset_logits(model_lines, num_offsets, model_type="tdnn"):
    """(components x K) float32 logits of the offset supernet: the first K entries of the <BiasParams> row of the 3rd..30th
    component that has one ('tdnn': 28 = 14 layers x (linear, affine)) or the 8th..25th ('cnn-tdnn': 18).
    generate_top_list.py:12-40."""
    first, last, rows = {"tdnn": (3, 30, 28), "cnn-tdnn": (8, 25, 18)}[model_type]
    out = np.zeros((rows, num_offsets), np.float32)
    count = 1
    for line in model_lines:
        line = line.strip()
        if '<BiasParams>' in line:
            if first <= count <= last:
                out[count - first, :] = np.asarray(_floats_after_bracket(line, num_offsets), np.float32)
            count += 1
    return out


def bottleneck_logits(model_lines, num_choices=8, network_type="tdnn"):
    """(layers x C) logits of the bottleneck search: every '<name>.alpha <ConstantFunctionComponent>' line in file order
    (14 layers for 'tdnn', 9 for 'cnn-tdnn').  generate_top_list_bottleneckdim.py:14-27."""
    rows = {"tdnn": 14, "cnn-tdnn": 9}[network_type]
    out = np.zeros((rows, num_choices), np.float32)
    count = 0
    for line in model_lines:
        if "alpha <ConstantFunctionComponent>" in line:
            out[count, :] = np.asarray(_floats_after_bracket(line, num_choices), np.float32)  # IndexError past `rows`, as the script
            count += 1
    return out


# ---------------------------------------------------------------------------------------------- the search
def choice_probabilities(logits, child_type):
    """Row-wise softmax of the logits ('top') or of their negation ('last'), in float32 (torch's implicit dim for a 2-D
    tensor is 1).  generate_top_list.py:42-47."""
    x = np.asarray(logits, np.float32)
    if child_type == 'last':
        x = -x
    elif child_type != 'top':
        raise ValueError("child_type must be 'top' or 'last'")
    e = np.exp(x - x.max(axis=1, keepdims=True), dtype=np.float32)
    return (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)


def beam_paths(prob, beam=BEAM):
    """The scripts' beam search over the product of per-row probabilities, with their data structure: a dict keyed by
    the path's score (a Python float).  Paths whose scores are exactly equal therefore replace each other (untrained,
    all-equal logits leave ONE path: the last choice everywhere), the first row is not pruned, later rows keep the
    `beam` best.  Returns [(score, [[row, choice], ...]), ...] best first (for a single row: in choice order).
    generate_top_list.py:49-72."""
    prob = np.asarray(prob, np.float32)
    nxt = {}
    for i in range(prob.shape[0]):
        cur = {}
        for j in range(prob.shape[1]):
            pij = float(prob[i, j])
            if i == 0:
                nxt[pij] = [[i, j]]
            else:
                for key, path in nxt.items():
                    cur[key * pij] = path + [[i, j]]
        if i >= 1:
            nxt = {k: cur[k] for k in sorted(cur.keys(), reverse=True)[:beam]}
    return list(nxt.items())


def pick_path(paths, top_id):
    """The top_id-th (1-based) path; IndexError if the beam holds fewer (as the scripts)."""
    return paths[top_id - 1][1]


def offset_child(path, num_offsets):
    """Per-component time offset of the child: even components (X.linear) choose from -(K-1)..0, odd ones (X.affine) from
    0..K-1.  Returns the list the scripts write to arch.txt.  generate_top_list.py:78-84,121-141."""
    left = list(range(-(num_offsets - 1), 1))
    right = list(range(0, num_offsets))
    return [(left if n % 2 == 0 else right)[path[n][1]] for n in range(len(path))]


def bottleneck_child(path, dims=BOTTLENECK_DIMS):
    """Per-layer bottleneck dim of the child.  generate_top_list_bottleneckdim.py:59-60,98-103."""
    return [dims[c[1]] for c in path]


# ---------------------------------------------------------------------------------------------- config rewriting
def rewrite_offsets_config(lines, offsets, tdnn_only=True):
    """final.config_temp -> final.config for explicit per-component offsets (offsets[2l] <= 0 for X.linear, offsets[2l+1]
    >= 0 for X.affine): 'time-offsets=-a,0' / '0' on the linear lines (the token after it -- orthonormal-constraint -- is
    kept), 'time-offsets=0,b' / '0' on the affine lines (everything after it is dropped, as the scripts do).
    tdnn_only: generate_top_list.py:107 looks at TdnnComponent lines only, generate_optimal_stride.py:22 at every line with
    'time-offsets'.  Lines are stripped, as the scripts do."""
    out = []
    n = 0
    for line in lines:
        line = line.strip()
        new = line
        if 'time-offsets' in line and (not tdnn_only or 'TdnnComponent' in line):
            head, tail = line.split('time-offsets')[0], line.split('time-offsets')[1]
            if n % 2 == 0:
                keep = tail.split(' ')[1]
                new = head + 'time-offsets=' + ('0' if offsets[n] == 0 else str(offsets[n]) + ',0') + ' ' + keep
            else:
                new = head + 'time-offsets=' + ('0' if offsets[n] == 0 else '0,' + str(offsets[n]))
            n += 1
        out.append(new)
    return out


def rewrite_bottleneck_config(lines, layer_dims):
    """final.config_temp -> final.config for per-layer bottleneck dims: output-dim of X.linear, input-dim of X.affine.
    generate_top_list_bottleneckdim.py:72-87."""
    out = []
    n = 0
    for line in lines:
        line = line.strip()
        new = line
        if 'time-offsets' in line and 'TdnnComponent' in line:
            d = str(layer_dims[n // 2])
            if n % 2 == 0:
                new = line.split('output-dim=')[0] + 'output-dim=' + d + ' l2-regularize' + line.split('l2-regularize')[1]
            else:
                new = line.split('input-dim=')[0] + 'input-dim=' + d + ' output-dim' + line.split('output-dim')[1]
            n += 1
        out.append(new)
    return out


def arch_txt_offsets(offsets):
    return ''.join(str(v) + ' ' for v in offsets)  # generate_top_list.py:146-148


def arch_txt_bottleneck(layer_dims, dims=BOTTLENECK_DIMS):
    return ''.join(str(v) + ' ' for v in layer_dims) + '\n' + ''.join(str(dims.index(v)) + ' ' for v in layer_dims)  # :111-116


# ---------------------------------------------------------------------------------------------- whole scripts
def derive_offset_child(model_lines, child_type, top_id, num_offsets, model_type="tdnn"):
    """generate_top_list.py end to end (without the files): returns (path, offsets)."""
    prob = choice_probabilities(offset_logits(model_lines, num_offsets, model_type), child_type)
    path = pick_path(beam_paths(prob), top_id)
    return path, offset_child(path, num_offsets)


def derive_bottleneck_child(model_lines, child_type, top_id, num_choices=8, network_type="tdnn", dims=BOTTLENECK_DIMS):
    """generate_top_list_bottleneckdim.py end to end (without the files): returns (path, layer_dims)."""
    prob = choice_probabilities(bottleneck_logits(model_lines, num_choices, network_type), child_type)
    path = pick_path(beam_paths(prob), top_id)
    return path, bottleneck_child(path, dims)


def child_config_kwargs(offsets=None, layer_dims=None):
    """Keyword arguments for trainer.make_config describing the child: `layer_offsets` = [(a_l, b_l)] (X.linear taps
    {-a_l, 0}, X.affine taps {0, b_l}; a single tap when the offset is 0) and / or `bottleneck` = per-layer dims."""
    kw = {}
    if offsets is not None:
        assert len(offsets) % 2 == 0 and all(offsets[2 * i] <= 0 <= offsets[2 * i + 1] for i in range(len(offsets) // 2))
        kw["layer_offsets"] = [(-offsets[2 * i], offsets[2 * i + 1]) for i in range(len(offsets) // 2)]
    if layer_dims is not None:
        kw["bottleneck"] = list(layer_dims)
    return kw


def logits_from_net(net, params=None):
    """The same logits straight from a ChainNet's flat parameter vector (no text model): (2L x K) for the offset supernet
    (K logits behind each searched component's weights), (L x C) for the bottleneck search (the X.alpha / X.softmax
    vectors)."""
    p = np.asarray(params if params is not None else net.params.detach().cpu().numpy())
    rows = []
    for c in net.components:
        if c["num_alpha"] > 0:
            b = c["begin"] + c["rows"] * c["cols"]
            rows.append(p[b:b + c["num_alpha"]])
        elif c["name"].endswith((".alpha", ".softmax")):
            rows.append(p[c["begin"]:c["begin"] + c["rows"]])
    return np.asarray(rows, np.float32)
