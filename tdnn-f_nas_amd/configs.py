"""nnet3 config text of the graphs the trainer runs, and the reference's config-rewriting scripts on lists of lines.

Host-side mirror (pure Python, no GPU) of
  steps/libs/nnet3/xconfig/composite_layers.py:135-215    tdnnf-layer            -> `final_config()` (plain layers)
  steps/libs/nnet3/xconfig/composite_layers.py:706-792    tdnnfdartsv3-layer     -> `final_config(darts=...)`
  steps/libs/nnet3/xconfig/composite_layers.py:1283-1331  prefinal-layer         -> `final_config()`
  local/chain_NAS/scripts/generate_config.py                                      -> `darts_supernet_config()`
  local/chain_NAS/scripts/generate_bottleneckCB8share_onehottrain_config.py       -> `bottleneck_supernet_config()`
  local/chain_NAS/scripts/generate_optimal_context_offset_bottleneckCB8share_onehottrain_config.py
                                                                                  -> `bottleneck_supernet_config(context_offset=...)`
  local/chain_NAS/scripts/add_flopsconstraint.py                                  -> `flops_constraint_change_config()`
  local/chain_NAS/scripts/bottleneckdim_search_top_model_size.py                  -> `bottleneck_top5_model_sizes()`
The five scripts are pinned by their own outputs (tests/golden/r01_configs_golden.json, made by
tests/golden/make_configs_golden.py running them in the build container on `final_config()` templates).  The lines
of the input, tdnn1, linear-component and output layers come from xconfig classes the reference does not ship
(basic_layers.py is upstream Kaldi): restated in the same style, not pinned.
"""
from . import derive

# block widths of the bottleneck supernet: candidate dims 25, 50, 80, 100, 120, 160, 200, 240
BN_BLOCKS = [25, 25, 30, 20, 20, 40, 40, 40]
STRIDES_7Q = [1, 1, 1, 0] + [3] * 10


def _py(v):
    """str() of a Python float / int / str as the xconfig '{}'.format() prints it (0.0, 0.01, 1e-05, 0.66, 1536)."""
    return repr(float(v)) if isinstance(v, float) else str(v)


def _offsets(s):
    return ('{0},0'.format(-s), '0,{0}'.format(s)) if s != 0 else ('0', '0')


def final_config(strides=None, bottleneck=160, hidden=1536, feat_dim=40, ivector_dim=100, num_pdfs=6034, small_dim=256, l2=0.01, max_change=0.75,
                 bypass_scale=0.66, dropout_proportion=0.0, self_repair_scale=1.0e-5, xent_regularize=0.1, l2_output=0.002, darts=None,
                 layer_offsets=None, lda_mat="configs/lda.mat"):
    """final.config of the TDNN-F net as xconfig_to_configs.py writes it for the recipes' network.xconfig
    (run_tdnn_fbk_40_iv_sp_7q.sh:160-186): per layer `component name=...` then `component-node name=...`.
    darts: None or a dict of the tdnnfdartsv3-layer flags ('use-gumbel', 'use-entropy', 'free-select', 'update-alpha',
    'update-theta', 'uniform-sample' -> 'true' / 'false'): TdnnDARTSV3Component lines (still with the two time-stride
    offsets: generate_config.py / darts_supernet_config() then writes the K taps).
    layer_offsets: [(a, b)] of a derived child instead of strides."""
    strides = list(STRIDES_7Q if strides is None else strides)
    if layer_offsets is not None:
        strides = [max(a, b) for a, b in layer_offsets]
    bns = list(bottleneck) if isinstance(bottleneck, (list, tuple)) else [bottleneck] * len(strides)
    lda_dim = 3 * feat_dim + ivector_dim
    L = ['input-node name=ivector dim=%d' % ivector_dim, 'input-node name=input dim=%d' % feat_dim]
    L.append('component name=lda type=FixedAffineComponent matrix=%s' % lda_mat)
    L.append('component-node name=lda component=lda input=Append(Offset(input, -1), input, Offset(input, 1), ReplaceIndex(ivector, t, 0))')
    # relu-batchnorm-dropout-layer name=tdnn1 (basic_layers.py, upstream)
    L.append('component name=tdnn1.affine type=NaturalGradientAffineComponent input-dim=%d output-dim=%d  max-change=%s l2-regularize=%s'
             % (lda_dim, hidden, _py(max_change), _py(l2)))
    L.append('component-node name=tdnn1.affine component=tdnn1.affine input=lda')
    L.append('component name=tdnn1.relu type=RectifiedLinearComponent dim=%d self-repair-scale=%s' % (hidden, _py(self_repair_scale)))
    L.append('component-node name=tdnn1.relu component=tdnn1.relu input=tdnn1.affine')
    L.append('component name=tdnn1.batchnorm type=BatchNormComponent dim=%d target-rms=1.0' % hidden)
    L.append('component-node name=tdnn1.batchnorm component=tdnn1.batchnorm input=tdnn1.relu')
    L.append('component name=tdnn1.dropout type=GeneralDropoutComponent dim=%d dropout-proportion=%s continuous=true' % (hidden, _py(dropout_proportion)))
    L.append('component-node name=tdnn1.dropout component=tdnn1.dropout input=tdnn1.batchnorm')
    prev = 'tdnn1.dropout'
    for i, (s, bn) in enumerate(zip(strides, bns)):
        name = 'tdnnf%d' % (i + 2)
        o1, o2 = _offsets(s)
        if layer_offsets is not None:
            a, b = layer_offsets[i]
            o1, o2 = ('%d,0' % -a if a else '0'), ('0,%d' % b if b else '0')
        if darts is None:  # composite_layers.py:156-171
            L.append('component name={0}.linear type=TdnnComponent input-dim={1} output-dim={2} l2-regularize={3} max-change={4} use-bias=false '
                     'time-offsets={5} orthonormal-constraint=-1.0'.format(name, hidden, bn, _py(l2), _py(max_change), o1))
            L.append('component-node name={0}.linear component={0}.linear input={1}'.format(name, prev))
            L.append('component name={0}.affine type=TdnnComponent input-dim={1} output-dim={2} l2-regularize={3} max-change={4} '
                     'time-offsets={5}'.format(name, bn, hidden, _py(l2), _py(max_change), o2))
        else:  # composite_layers.py:733-747
            fl = [darts.get(k, d) for k, d in (('use-gumbel', 'false'), ('use-entropy', 'false'), ('free-select', 'false'), ('update-alpha', 'false'),
                                               ('update-theta', 'true'), ('uniform-sample', 'false'))]
            flags = 'use-gumbel={0} use-entropy={1} free-select={2} update-alpha={3} update-theta={4} uniform-sample={5} Temp-Proportion=1.0'.format(*fl)
            L.append('component name={0}.linear type=TdnnDARTSV3Component input-dim={1} output-dim={2} l2-regularize={3} max-change={4} use-bias=false '
                     '{5} time-offsets={6} orthonormal-constraint=-1.0'.format(name, hidden, bn, _py(l2), _py(max_change), flags, o1))
            L.append('component-node name={0}.linear component={0}.linear input={1}'.format(name, prev))
            L.append('component name={0}.affine type=TdnnDARTSV3Component input-dim={1} output-dim={2} l2-regularize={3} max-change={4} '
                     '{5} time-offsets={6}'.format(name, bn, hidden, _py(l2), _py(max_change), flags, o2))
        L.append('component-node name={0}.affine component={0}.affine input={0}.linear'.format(name))
        L.append('component name={0}.relu type=RectifiedLinearComponent dim={1} self-repair-scale={2}'.format(name, hidden, _py(self_repair_scale)))
        L.append('component-node name={0}.relu component={0}.relu input={0}.affine'.format(name))
        L.append('component name={0}.batchnorm type=BatchNormComponent dim={1}'.format(name, hidden))
        L.append('component-node name={0}.batchnorm component={0}.batchnorm input={0}.relu'.format(name))
        L.append('component name={0}.dropout type=GeneralDropoutComponent dim={1} dropout-proportion={2} continuous=true'.format(name, hidden, _py(dropout_proportion)))
        L.append('component-node name={0}.dropout component={0}.dropout input={0}.batchnorm'.format(name))
        L.append('component name={0}.noop type=NoOpComponent dim={1}'.format(name, hidden))
        L.append('component-node name={0}.noop component={0}.noop input=Sum(Scale({1}, {2}), {0}.dropout)'.format(name, _py(bypass_scale), prev))
        prev = name + '.noop'
    # linear-component name=prefinal-l (trivial_layers.py)
    L.append('component name=prefinal-l type=LinearComponent input-dim=%d output-dim=%d l2-regularize=%s orthonormal-constraint=-1.0' % (hidden, small_dim, _py(l2)))
    L.append('component-node name=prefinal-l component=prefinal-l input=%s' % prev)
    for head, out in (('prefinal-chain', 'output'), ('prefinal-xent', 'output-xent')):  # composite_layers.py:1295-1329
        L.append('component name={0}.affine type=NaturalGradientAffineComponent input-dim={1} output-dim={2} l2-regularize={3} max-change={4}'.format(
            head, small_dim, hidden, _py(l2), _py(max_change)))
        L.append('component-node name={0}.affine component={0}.affine input=prefinal-l'.format(head))
        L.append('component name={0}.relu type=RectifiedLinearComponent dim={1} self-repair-scale={2}'.format(head, hidden, _py(self_repair_scale)))
        L.append('component-node name={0}.relu component={0}.relu input={0}.affine'.format(head))
        L.append('component name={0}.batchnorm1 type=BatchNormComponent dim={1}'.format(head, hidden))
        L.append('component-node name={0}.batchnorm1 component={0}.batchnorm1 input={0}.relu'.format(head))
        L.append('component name={0}.linear type=LinearComponent input-dim={1} output-dim={2} l2-regularize={3} max-change={4} orthonormal-constraint=-1 '.format(
            head, hidden, small_dim, _py(l2), _py(max_change)))
        L.append('component-node name={0}.linear component={0}.linear input={0}.batchnorm1'.format(head))
        L.append('component name={0}.batchnorm2 type=BatchNormComponent dim={1}'.format(head, small_dim))
        L.append('component-node name={0}.batchnorm2 component={0}.batchnorm2 input={0}.linear'.format(head))
        # output-layer (basic_layers.py, upstream): the xent head has a log-softmax and learning-rate-factor 0.5 / xent_regularize
        lrf = '' if out == 'output' else ' learning-rate-factor=%s' % _py(0.5 / xent_regularize)
        L.append('component name={0}.affine type=NaturalGradientAffineComponent input-dim={1} output-dim={2}{3} max-change=1.5 l2-regularize={4} param-stddev=0.0 '
                 'bias-stddev=0.0'.format(out, small_dim, num_pdfs, lrf, _py(l2_output)))
        L.append('component-node name={0}.affine component={0}.affine input={1}.batchnorm2'.format(out, head))
        if out == 'output':
            L.append('output-node name=output input=output.affine objective=linear')
        else:
            L.append('component name=output-xent.log-softmax type=LogSoftmaxComponent dim=%d' % num_pdfs)
            L.append('component-node name=output-xent.log-softmax component=output-xent.log-softmax input=output-xent.affine')
            L.append('output-node name=output-xent input=output-xent.log-softmax objective=linear')
    return L


def ref_config(lines):
    """ref.config has the same lines with the fixed lda matrix replaced by a dimension (xconfig 'ref' config): the scripts
    only ever look at the tdnnf lines, which are the same in both."""
    return [l.replace('matrix=configs/lda.mat', 'input-dim=220 output-dim=220') if l.startswith('component name=lda ') else l for l in lines]


# ------------------------------------------------------------------------------------------------ generate_config.py
def darts_supernet_config(lines, num_offsets):
    """final.config_temp -> final.config of the offset supernet: every TdnnDARTSV3Component gets use-bias=true and the K
    taps -(K-1)..0 (X.linear; the token after time-offsets is kept) or 0..K-1 (X.affine).  generate_config.py:8-44."""
    right = ','.join(str(i) for i in range(num_offsets))
    left = ','.join(str(i) for i in range(-(num_offsets - 1), 1))
    out = []
    n = 0
    for line in lines:
        line = line.strip()
        new = line
        if 'use-bias=false' in line and 'TdnnDARTSV3Component' in line:
            line = line.replace('use-bias=false', 'use-bias=true')
            new = line
        if 'time-offsets' in line and 'TdnnDARTSV3Component' in line:
            head, tail = line.split('time-offsets')[0], line.split('time-offsets')[1]
            new = head + 'time-offsets=' + (left + ' ' + tail.split(' ')[1] if n % 2 == 0 else right)
            n += 1
        out.append(new)
    return out


# ----------------------------------------------------------- generate_bottleneckCB8share_onehottrain_config.py (+ optimal offsets)
def _bn_block(name, prev_name, count, lin_offsets, aff_offsets):
    C = len(BN_BLOCKS)
    o = []
    o.append("component name=" + name + ".softmax type=OnehotFunctionComponent input-dim=220 output-dim=8 is-updatable=true use-natural-gradient=false")
    o.append("component-node name=" + name + ".softmax component=" + name + ".softmax input=lda")
    for k in range(C):
        o.append("dim-range-node name=%s.softmax%d input-node=%s.softmax dim-offset=%d dim=1" % (name, k, name, k))
    for k in range(C):
        o.append("component name=%s%d.copyn type=CopyNComponent input-dim=1 output-dim=%d" % (name, k, BN_BLOCKS[k]))
        terms = ",".join("%s.softmax%d" % (name, j) for j in range(k, C))
        o.append("component-node name=%s%d.copyn component=%s%d.copyn input=%s" % (name, k, name, k, "Sum(" + terms + ")" if k + 1 < C else terms))
    o.append("component name=" + name + ".linear type=TdnnComponent input-dim=1536 output-dim=240 l2-regularize=0.01 max-change=0.75 use-bias=false time-offsets="
             + lin_offsets + " orthonormal-constraint=-1.0")
    o.append("component-node name=" + name + ".linear component=" + name + ".linear input=" + ("tdnn1.dropout" if count == 2 else prev_name + ".noop"))
    off = 0
    for k in range(C):
        o.append("dim-range-node name=%s%d.linear input-node=%s.linear dim-offset=%d dim=%d" % (name, k, name, off, BN_BLOCKS[k]))
        off += BN_BLOCKS[k]
    for k in range(C):
        o.append("component name=%s%d.output type=ElementwiseProductComponent input-dim=%d output-dim=%d" % (name, k, 2 * BN_BLOCKS[k], BN_BLOCKS[k]))
        o.append("component-node name=%s%d.output component=%s%d.output input=Append(%s%d.copyn, %s%d.linear)" % (name, k, name, k, name, k, name, k))
    o.append("component name=" + name + ".affine type=TdnnComponent input-dim=240 output-dim=1536 l2-regularize=0.01 max-change=0.75 time-offsets=" + aff_offsets)
    o.append("component-node name=" + name + ".affine component=" + name + ".affine input=Append(" + ",".join("%s%d.output" % (name, k) for k in range(C)) + ")")
    o.append("component name=" + name + ".relu type=RectifiedLinearComponent dim=1536 self-repair-scale=1e-05")
    o.append("component-node name=" + name + ".relu component=" + name + ".relu input=" + name + ".affine")
    o.append("component name=" + name + ".batchnorm type=BatchNormComponent dim=1536")
    o.append("component-node name=" + name + ".batchnorm component=" + name + ".batchnorm input=" + name + ".relu")
    return o


def bottleneck_supernet_config(lines, context_offset=None):
    """final_ori.config (the 7q net) -> final.config of the bottleneck-dimension supernet in Onehot pretrain mode: each
    tdnnfN block is replaced by X.softmax (OnehotFunction on lda), its 8 one-column ranges, 8 CopyN of the running sums,
    the 240-wide X.linear cut in 8 blocks, 8 ElementwiseProducts, X.affine on their Append; relu / batchnorm / dropout
    re-emitted; everything else passes through.  The scripts hard-code the 7q dims (1536, 240, 220) and, without
    context_offset, the 7q time strides by layer number (generate_bottleneckCB8share_onehottrain_config.py:52-63,94-102);
    with context_offset (28 per-component offsets of a derived child) the variant
    generate_optimal_context_offset_bottleneckCB8share_onehottrain_config.py ('tdnn' branch)."""
    out = []
    count = 2
    name = prev_name = None
    for line in lines:
        line = line.strip()
        c = str(count)
        if 'component name=tdnnf' + c + '.linear' in line:
            name, prev_name = 'tdnnf' + c, 'tdnnf' + str(count - 1)
            if context_offset is None:
                lin, aff = ("-1,0", "0,1") if count <= 4 else (("0", "0") if count == 5 else ("-3,0", "0,3"))
            else:
                a, b = context_offset[(count - 2) * 2], context_offset[(count - 2) * 2 + 1]
                lin = "0" if a == 0 else str(a) + ",0"
                aff = "0" if b == 0 else "0," + str(b)
            out += _bn_block(name, prev_name, count, lin, aff)
        elif ('component-node name=tdnnf' + c + '.linear' in line or 'component name=tdnnf' + c + '.affine' in line
              or 'component-node name=tdnnf' + c + '.affine' in line):
            continue
        elif 'component-node name=tdnnf' + c + '.relu' in line or 'component name=tdnnf' + c + '.relu' in line:
            continue
        elif 'component-node name=tdnnf' + c + '.batchnorm' in line or 'component name=tdnnf' + c + '.batchnorm' in line:
            continue
        elif 'component name=tdnnf' + c + '.dropout' in line:
            continue
        elif 'component-node name=tdnnf' + c + '.dropout' in line:
            out.append("component name=" + name + ".dropout type=GeneralDropoutComponent dim=1536 dropout-proportion=0.0 continuous=true")
            out.append("component-node name=" + name + ".dropout component=" + name + ".dropout input=" + name + ".batchnorm")
            count += 1
        else:
            out.append(line)
    return out


# ------------------------------------------------------------------------------------------------ add_flopsconstraint.py
def flops_constraint_change_config(use_gumbel, flops_coef, network_type='tdnn'):
    """change.config of the cv-update stage: an X.alpha ConstantFunctionComponent (the architecture logits) in front of
    every layer's X.softmax, which becomes a (Gumbel)SoftmaxFlopsComponent.  use_gumbel is the script's string argument
    ("true" selects Gumbel).  add_flopsconstraint.py:9-30."""
    out = []
    count = 2 if network_type == 'tdnn' else 7
    while count <= 15:
        n = 'tdnnf' + str(count)
        if network_type == 'tdnn':
            out.append("component name=" + n + ".alpha type=ConstantFunctionComponent input-dim=220 output-dim=8 is-updatable=true use-natural-gradient=false")
            out.append("component-node name=" + n + ".alpha component=" + n + ".alpha input=lda")
        else:
            out.append("component name=" + n + ".alpha type=ConstantFunctionComponent input-dim=40 output-dim=8 is-updatable=true use-natural-gradient=false")
            out.append("component-node name=" + n + ".alpha component=" + n + ".alpha input=input")
        if use_gumbel == "true":
            out.append("component name=" + n + ".softmax type=GumbelSoftmaxFlopsComponent dim=8 scale=" + str(float(flops_coef)) + " temp-proportion=1.0")
        else:
            out.append("component name=" + n + ".softmax type=SoftmaxFlopsComponent dim=8 scale=" + str(float(flops_coef)))
        out.append("component-node name=" + n + ".softmax component=" + n + ".softmax input=" + n + ".alpha")
        count += 1
    return out


# ------------------------------------------------------------------------------ bottleneckdim_search_top_model_size.py
def bottleneck_top5_model_sizes(model_lines, child_type, dims=derive.BOTTLENECK_DIMS):
    """The five best bottleneck children with their parameter counts ('tdnn'): returns the lines the script appends to
    configs/arch.txt.  The count is the script's closed form for the 7q net with 6008 pdfs (:63-69): layer 5 (time-stride
    0) has single-tap matrices, the others two taps each."""
    prob = derive.choice_probabilities(derive.bottleneck_logits(model_lines, 8, 'tdnn'), child_type)
    paths = derive.beam_paths(prob)
    out = []
    for top_id in range(5):
        info = paths[top_id][1]
        lst = []
        param = 220 * 1536 + 1536 * 256 + (256 * 1536 + 1536 * 256 + 256 * 6008) * 2 + 1536 * 14
        for n in range(14):
            d = dims[info[n][1]]
            lst.append(d)
            param += 1536 * int(d) * 2 if n == 4 else 1536 * int(d) * 2 * 2
        size = str(param / 1000000) + 'M'
        out.append('top' + str(top_id) + ' ' + ''.join(str(v) + ' ' for v in lst) + size)
        out.append('top' + str(top_id) + ' ' + ''.join(str(dims.index(v)) + ' ' for v in lst) + size)
    return out


def node_lines(lines):
    """The graph part of a config (what Nnet::Write keeps in a model file): input / component / dim-range / output nodes."""
    return [l for l in lines if l.split(' ')[0] in ('input-node', 'component-node', 'dim-range-node', 'output-node')]


# ------------------------------------------------------------------------------------------------------------------
# final.config IN: the front-end of SURVEY.md 8(f) rank 2.  The recipes write network.xconfig, xconfig_to_configs.py turns it
# into configs/final.config (steps/libs/nnet3/xconfig/composite_layers.py:135-215,706-792,1283-1331), the NAS scripts rewrite
# that text (generate_config.py, generate_bottleneckCB8share_onehottrain_config.py, add_flopsconstraint.py) and nnet3-init
# (UPSTREAM) builds the initial model from it: Nnet::ReadConfig parses `component name=.. type=.. key=value ..`,
# `component-node`, `dim-range-node`, `input-node`, `output-node` lines and calls every component's InitFromConfig
# (TdnnDARTSV3Component::InitFromConfig nnet-tdnn-component.cc:109-212).  parse_config() / net_config_from_final_config() /
# init_params_from_final_config() are that step for the graphs of SURVEY.md 3.4, the ones the trainer runs.
def parse_config(lines):
    """The lines of a final.config as {'inputs': {name: dim}, 'components': {name: {'type': .., key: value, ..}} (in file order),
    'nodes': [node lines, whitespace-normalised, in file order], 'outputs': [names]}.  Values stay strings (a ConfigLine's
    GetValue converts on demand).  Raises ValueError for a line that is none of the five kinds."""
    out = {"inputs": {}, "components": {}, "nodes": [], "outputs": []}
    for raw in lines:
        line = raw.split("#")[0].strip()
        if not line:
            continue
        kind, _, rest = line.partition(" ")
        if kind not in ("input-node", "component", "component-node", "dim-range-node", "output-node"):
            raise ValueError("final.config: unknown line type %r in %r" % (kind, raw))
        # key=value pairs; a value runs to the next ' key=' (descriptors contain spaces and commas)
        import re
        pairs = re.findall(r"([A-Za-z][\w\-]*)=(.*?)(?=\s+[A-Za-z][\w\-]*=|\s*$)", rest)
        kv = {k: v.strip() for k, v in pairs}
        if "name" not in kv:
            raise ValueError("final.config: line without name=: %r" % raw)
        if kind == "input-node":
            out["inputs"][kv["name"]] = int(kv["dim"])
        elif kind == "component":
            if "type" not in kv:
                raise ValueError("final.config: component without type=: %r" % raw)
            if kv["name"] in out["components"]:
                raise ValueError("final.config: component %s defined twice" % kv["name"])
            out["components"][kv["name"]] = {k: v for k, v in kv.items() if k != "name"}
        else:
            if kind == "output-node":
                out["outputs"].append(kv["name"])
            out["nodes"].append(" ".join(line.split()))
    out["nodes"] = ["input-node name=%s dim=%d" % (k, v) for k, v in out["inputs"].items()] + out["nodes"]
    return out


def _b(v, default):
    return default if v is None else {"true": True, "t": True, "1": True, "false": False, "f": False, "0": False}[str(v).strip().lower()]


def net_config_kwargs_from_final_config(lines):
    """trainer.make_config keyword arguments for the network a final.config describes: dimensions and hyper-parameters from
    the component lines, the layer structure (time strides / per-layer offsets of a derived child / the K taps and flags of
    the offset supernet / the choice blocks and mode of the bottleneck supernet / BatchNormTestComponent = cv-update) from
    their types and time-offsets, the bypass scale from the tdnnfN.noop descriptor.  The graph WIRING is not taken on trust:
    net_config_from_final_config() checks it line by line."""
    import re
    cfg = parse_config(lines)
    comp = cfg["components"]

    def need(name):
        if name not in comp:
            raise ValueError("final.config: no component %s -- not one of the TDNN-F graphs the trainer runs" % name)
        return comp[name]

    kw = dict(feat_dim=cfg["inputs"].get("input"), ivector_dim=cfg["inputs"].get("ivector"))
    if kw["feat_dim"] is None or kw["ivector_dim"] is None:
        raise ValueError("final.config: needs input-node name=input and name=ivector")
    t1 = need("tdnn1.affine")
    kw["hidden_dim"] = int(t1["output-dim"])
    kw["l2_hidden"] = float(t1.get("l2-regularize", 0.0))
    layers = sorted({int(m.group(1)) for n in comp for m in [re.match(r"tdnnf(\d+)\.linear$", n)] if m})
    if not layers or layers != list(range(2, 2 + len(layers))):
        raise ValueError("final.config: tdnnf layers must be tdnnf2 .. tdnnfN, found %s" % layers)
    offs, bns, darts_K, flags, temp = [], [], 0, None, 1.0
    for n in layers:
        lin, aff = need("tdnnf%d.linear" % n), need("tdnnf%d.affine" % n)
        if lin["type"] != aff["type"] or lin["type"] not in ("TdnnComponent", "TdnnDARTSV3Component"):
            raise ValueError("final.config: tdnnf%d: unsupported component types %s / %s" % (n, lin["type"], aff["type"]))
        lo, ao = [int(v) for v in lin["time-offsets"].split(",")], [int(v) for v in aff["time-offsets"].split(",")]
        bns.append(int(lin["output-dim"]))
        if lin["type"] == "TdnnDARTSV3Component" and len(lo) > 2:  # the K-tap search space of generate_config.py
            K = len(lo)
            if lo != list(range(-(K - 1), 1)) or ao != list(range(0, K)):
                raise ValueError("final.config: tdnnf%d: offset supernet taps must be -(K-1)..0 / 0..K-1" % n)
            # InitFromConfig :150-163: every flag defaults to TRUE when the line does not give it
            f = ((1 if _b(lin.get("use-gumbel"), True) else 0) | (2 if _b(lin.get("free-select"), True) else 0) | (4 if _b(lin.get("uniform-sample"), True) else 0) |
                 (8 if _b(lin.get("use-entropy"), True) else 0) | (16 if _b(lin.get("update-alpha"), True) else 0))
            if darts_K not in (0, K) or flags not in (None, f):
                raise ValueError("final.config: the trainer runs one K and one set of flags for every layer")
            darts_K, flags, temp = K, f, float(lin.get("Temp-Proportion", 1.0))
            offs.append((0, 0))
            continue
        if not ((lo == [0] or (len(lo) == 2 and lo[1] == 0 and lo[0] < 0)) and (ao == [0] or (len(ao) == 2 and ao[0] == 0 and ao[1] > 0))):
            raise ValueError("final.config: tdnnf%d: time-offsets %s / %s are not {-a,0} / {0,b}" % (n, lo, ao))
        offs.append((-lo[0] if len(lo) == 2 else 0, ao[1] if len(ao) == 2 else 0))
    if darts_K:
        kw.update(darts_num_offsets=darts_K, darts_flags=flags, darts_temp_proportion=temp, strides=[1] * len(layers))
    elif all(a == b for a, b in offs):
        kw["strides"] = [a for a, _ in offs]
    else:
        kw["layer_offsets"] = offs
    kw["bottleneck"] = bns
    # bottleneck supernet: CopyN blocks tdnnf<N><k>.copyn, the C-vector component tdnnfN.softmax / tdnnfN.alpha
    first = "tdnnf%d" % layers[0]
    blocks = sorted(int(m.group(1)) for n in comp for m in [re.match(re.escape(first) + r"(\d)\.copyn$", n)] if m)
    if blocks:
        kw["bn_choice_dims"] = [int(comp["%s%d.copyn" % (first, k)]["output-dim"]) for k in blocks]
        sm = need(first + ".softmax")["type"]
        kw["bn_mode"] = {"OnehotFunctionComponent": 0, "SoftmaxFlopsComponent": 1, "GumbelSoftmaxFlopsComponent": 2}.get(sm)
        if kw["bn_mode"] is None:
            raise ValueError("final.config: %s.softmax of type %s" % (first, sm))
        if kw["bn_mode"]:
            kw["bn_flops_scale"] = float(comp[first + ".softmax"].get("scale", comp[first + ".softmax"].get("flops-scale", 0.0)))
            kw["bn_temp_proportion"] = float(comp[first + ".softmax"].get("Temp-Proportion", 1.0))
    relu = need("tdnn1.relu")
    kw["relu_self_repair_scale"] = float(relu.get("self-repair-scale", 1.0e-5))
    kw["use_dropout"] = int(any(c["type"] == "GeneralDropoutComponent" for c in comp.values()))
    kw["cv_update"] = int(need("tdnn1.batchnorm")["type"] == "BatchNormTestComponent")
    kw["small_dim"] = int(need("prefinal-l")["output-dim"])
    out, xent = need("output.affine"), need("output-xent.affine")
    kw["num_pdfs"] = int(out["output-dim"])
    kw["l2_output"] = float(out.get("l2-regularize", 0.0))
    lrf = float(xent.get("learning-rate-factor", 1.0))
    kw["xent_regularize"] = 0.5 / lrf if lrf != 1.0 else 0.1
    kw["use_natural_gradient"] = int(_b(t1.get("use-natural-gradient"), True))  # (the reference's components default to true)
    noop = [n for n in cfg["nodes"] if n.startswith("component-node name=%s.noop " % first)]
    m = re.search(r"Scale\(([0-9.eE+\-]+),", noop[0]) if noop else None
    kw["bypass_scale"] = float(m.group(1)) if m else 0.66
    return kw


def net_config_from_final_config(lines, frames_per_chunk=150, num_sequences=64, **overrides):
    """trainer.NetConfig for a final.config (what nnet3-init + the trainer's own option parsing would arrive at), verified: the
    node lines the library writes for that configuration (tdnnf_net_config_text -- themselves pinned by the reference scripts'
    outputs, tests/test_configs.py) must be the file's node lines, name for name and descriptor for descriptor.  A graph the
    trainer does not run (another wiring, extra nodes) is refused with the first differing line."""
    from . import trainer
    kw = net_config_kwargs_from_final_config(lines)
    kw.update(overrides)
    cfg = trainer.make_config(frames_per_chunk=frames_per_chunk, num_sequences=num_sequences, **kw)
    norm = lambda s: "".join(s.split())  # noqa: E731
    want = [norm(n) for n in parse_config(lines)["nodes"]]
    have = [norm(n) for n in trainer.config_text(cfg).split("\n") if n.strip()]
    if sorted(want) != sorted(have):
        extra = [n for n in want if n not in have][:1] + [n for n in have if n not in want][:1]
        raise ValueError("final.config: not a graph the trainer runs; first difference: %s" % extra)
    return cfg


def init_params_from_final_config(lines, net, seed=0, lda_matrix=None):
    """nnet3-init for `net` (a trainer.ChainNet of net_config_from_final_config): every component's InitFromConfig with the
    keys of its config line -- TdnnDARTSV3Component / TdnnComponent: linear_params ~ N(0, param-stddev^2), param-stddev
    default 1 / sqrt(input-dim x num-offsets); bias ~ N(bias-mean, bias-stddev^2), bias-stddev default 1, architecture logits 0
    (nnet-tdnn-component.cc:139-176); NaturalGradientAffineComponent: param-stddev default 1 / sqrt(input-dim), bias-stddev 1;
    LinearComponent: param-stddev default 1 / sqrt(input-dim); OnehotFunction / ConstantFunction output_: zeros.  The lda
    matrix comes from its file (`lda_matrix`: the D x (D + 1) array of configs/lda.mat; trainer.read_kaldi_matrix) or, without
    one, a random orthonormal transform.  Returns the flat float32 parameter vector (numpy's generator stands in for Kaldi's
    RandGauss: the distribution is restated, not the random stream)."""
    import numpy as np
    from . import trainer
    comp = parse_config(lines)["components"]
    rng = np.random.default_rng(seed)
    p = np.zeros(net.num_params, np.float32)
    for c in net.components:
        name, rows, cols = c["name"], c["rows"], c["cols"]
        if name == "lda":
            if lda_matrix is not None:
                trainer.set_lda(p, net.components, lda_matrix)
            else:
                q = np.linalg.qr(rng.standard_normal((rows, cols)))[0]
                p[c["begin"]:c["begin"] + rows * cols] = q.astype(np.float32).ravel()
            continue
        line = comp.get(name)
        if line is None:
            raise ValueError("final.config has no component %s" % name)
        if line["type"] in ("OnehotFunctionComponent", "ConstantFunctionComponent"):
            continue  # output_ starts at zero
        taps = len(line["time-offsets"].split(",")) if "time-offsets" in line else 1
        in_dim = int(line["input-dim"])
        if cols != in_dim * taps or rows != int(line["output-dim"]):
            raise ValueError("final.config: %s is %s x %s x %d taps, the net has %d x %d" % (name, line["output-dim"], line["input-dim"], taps, rows, cols))
        sd = float(line.get("param-stddev", -1.0))
        if sd < 0:
            sd = 1.0 / np.sqrt(in_dim * taps)
        n = rows * cols
        p[c["begin"]:c["begin"] + n] = (rng.standard_normal(n) * sd).astype(np.float32)
        if c["has_bias"]:
            bsd, bmean = float(line.get("bias-stddev", 1.0)), float(line.get("bias-mean", 0.0))
            o = c["begin"] + n + c["num_alpha"]  # the num_alpha logits in front stay 0 (:176)
            p[o:o + rows] = (rng.standard_normal(rows) * bsd + bmean).astype(np.float32)
    return p
