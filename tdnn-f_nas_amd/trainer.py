"""Python plumbing around the C++ chain trainer in libtdnnf_hip.so (include/tdnnf_hip.h,
"chain trainer" section): owns the flat parameter / gradient tensors (torch device memory),
initialises parameters the way the reference's components do, and shards minibatches
data-parallel with one RCCL all-reduce of the raw gradient buffer per step.

Reference graph and hyper-parameters: local/chain_NAS/run_tdnn_fbk_40_iv_sp_7q.sh:149-186,
local/chain_NAS/run_TDNN_DARTSV3_fbk_stride_pretrain.sh:185-203.
"""
import ctypes as C

import numpy as np

from . import hipabi

MAX_LAYERS = 32


class NetConfig(C.Structure):
    _fields_ = [("feat_dim", C.c_int), ("ivector_dim", C.c_int), ("num_pdfs", C.c_int),
                ("hidden_dim", C.c_int), ("prefinal_small_dim", C.c_int), ("num_layers", C.c_int),
                ("bottleneck_dim", C.c_int * MAX_LAYERS), ("time_stride", C.c_int * MAX_LAYERS),
                ("bypass_scale", C.c_float), ("frames_per_chunk", C.c_int), ("num_sequences", C.c_int),
                ("frame_subsampling", C.c_int), ("leaky_hmm", C.c_float), ("xent_regularize", C.c_float),
                ("chain_l2_regularize", C.c_float), ("l2_hidden", C.c_float), ("l2_output", C.c_float),
                ("max_change_hidden", C.c_float), ("max_change_output", C.c_float), ("max_param_change", C.c_float),
                ("relu_self_repair_scale", C.c_float), ("batchnorm_stats_scale", C.c_float),
                ("darts_num_offsets", C.c_int), ("darts_flags", C.c_int), ("darts_temp_proportion", C.c_float),
                ("use_natural_gradient", C.c_int),
                ("bn_num_choices", C.c_int), ("bn_choice_dims", C.c_int * 8), ("bn_mode", C.c_int),
                ("bn_flops_scale", C.c_float), ("bn_temp_proportion", C.c_float), ("cv_update", C.c_int), ("gemm_precision", C.c_int),
                ("use_layer_offsets", C.c_int), ("offset_left", C.c_int * MAX_LAYERS), ("offset_right", C.c_int * MAX_LAYERS),
                ("use_dropout", C.c_int)]

DARTS_USE_GUMBEL, DARTS_FREE_SELECT, DARTS_UNIFORM_SAMPLE, DARTS_USE_ENTROPY, DARTS_UPDATE_ALPHA = 1, 2, 4, 8, 16


# time strides of the fixed 7q net (run_tdnn_fbk_40_iv_sp_7q.sh:171-184) and of the "manual" variant
STRIDES_7Q = [1, 1, 1, 0] + [3] * 10
STRIDES_MANUAL_OFFSET6 = [1, 1, 1, 0] + [6] * 10   # run_tdnn_7q_fbk_40_manual.sh --offset 6
# block widths of the bottleneck-dimension supernet (generate_bottleneckCB8share_onehottrain_config.py:24-38):
# candidate bottleneck dims 25, 50, 80, 100, 120, 160, 200, 240
BN_CHOICE_DIMS = [25, 25, 30, 20, 20, 40, 40, 40]
BN_ONEHOT, BN_SOFTMAX_FLOPS, BN_GUMBEL_SOFTMAX_FLOPS = 0, 1, 2
GEMM_F32, GEMM_BF16X3, GEMM_BF16X6 = 0, 1, 2


def make_config(frames_per_chunk=150, num_sequences=64, strides=None, bottleneck=160, feat_dim=40, ivector_dim=100,
                num_pdfs=6034, hidden_dim=1536, small_dim=256, **kw):
    # layer_offsets = [(a, b)] per layer: a derived child (derive.child_config_kwargs): X.linear taps {-a, 0}, X.affine {0, b}
    layer_offsets = kw.get("layer_offsets")
    if layer_offsets is not None and strides is None:
        strides = [max(a, b) for a, b in layer_offsets]
    strides = list(STRIDES_7Q if strides is None else strides)
    bns = bottleneck if isinstance(bottleneck, (list, tuple)) else [bottleneck] * len(strides)
    c = NetConfig()
    c.feat_dim, c.ivector_dim, c.num_pdfs = feat_dim, ivector_dim, num_pdfs
    c.hidden_dim, c.prefinal_small_dim, c.num_layers = hidden_dim, small_dim, len(strides)
    for i, (s, b) in enumerate(zip(strides, bns)):
        c.time_stride[i], c.bottleneck_dim[i] = s, b
    c.bypass_scale = kw.get("bypass_scale", 0.66)
    c.frames_per_chunk, c.num_sequences, c.frame_subsampling = frames_per_chunk, num_sequences, kw.get("frame_subsampling", 3)
    c.leaky_hmm, c.xent_regularize, c.chain_l2_regularize = kw.get("leaky_hmm", 0.1), kw.get("xent_regularize", 0.1), kw.get("chain_l2", 0.0)
    c.l2_hidden, c.l2_output = kw.get("l2_hidden", 0.01), kw.get("l2_output", 0.002)
    c.max_change_hidden, c.max_change_output, c.max_param_change = 0.75, 1.5, 2.0
    c.relu_self_repair_scale = kw.get("relu_self_repair_scale", 1.0e-5)
    c.batchnorm_stats_scale = kw.get("batchnorm_stats_scale", 0.8)
    # DARTS offset supernet: run_TDNN_DARTSV3_fbk_stride_pretrain.sh <K> (pretrain flags :124 = uniform-sample)
    c.darts_num_offsets = kw.get("darts_num_offsets", 0)
    c.darts_flags = kw.get("darts_flags", DARTS_UNIFORM_SAMPLE if c.darts_num_offsets else 0)
    c.darts_temp_proportion = kw.get("darts_temp_proportion", 1.0)
    c.use_natural_gradient = int(kw.get("use_natural_gradient", 0))
    # bottleneck-dimension supernet: bn_choice_dims = block widths (their sum is every layer's linear output dim)
    dims = kw.get("bn_choice_dims")
    if dims:
        assert 2 <= len(dims) <= 8
        c.bn_num_choices = len(dims)
        for i, d in enumerate(dims):
            c.bn_choice_dims[i] = int(d)
        for i in range(len(strides)):
            c.bottleneck_dim[i] = int(sum(dims))
        c.bn_mode = int(kw.get("bn_mode", BN_ONEHOT))
        c.bn_flops_scale = float(kw.get("bn_flops_scale", 0.0))
        c.bn_temp_proportion = float(kw.get("bn_temp_proportion", 1.0))
    if layer_offsets is not None:
        assert len(layer_offsets) == len(strides)
        c.use_layer_offsets = 1
        for i, (a, b) in enumerate(layer_offsets):
            c.offset_left[i], c.offset_right[i] = int(a), int(b)
    c.use_dropout = int(kw.get("use_dropout", 0))  # reserve the GeneralDropout masks / draws; ChainNet.set_dropout_proportion
    c.cv_update = int(kw.get("cv_update", 0))
    c.gemm_precision = int(kw.get("gemm_precision", 0))  # GEMM_F32 exact f32 MFMA, GEMM_BF16X3 / GEMM_BF16X6 split-bf16
    return c


def config_from_model(path, frames_per_chunk=150, num_sequences=64):
    """NetConfig for the graph of an nnet3 raw model file (text or binary); see tdnnf_net_config_from_model."""
    c = NetConfig()
    hipabi.check(hipabi.load().tdnnf_net_config_from_model(str(path).encode(), int(frames_per_chunk), int(num_sequences), C.byref(c)))
    return c


def config_text(config):
    """The graph of a NetConfig as nnet3 config node lines (tdnnf_net_config_text; no GPU needed)."""
    lib = hipabi.load()
    need = C.c_size_t()
    hipabi.check(lib.tdnnf_net_config_text(C.byref(config), None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    hipabi.check(lib.tdnnf_net_config_text(C.byref(config), buf, need.value, None))
    return buf.value.decode().rstrip("\n")


class ChainNet:
    """One replica of the TDNN-F chain model on the current CUDA device."""

    def __init__(self, config, share=None):
        """share: a ChainNet of the same model built for another minibatch shape (chunk width / number of sequences): this one
        then uses its parameters, gradients, natural-gradient state and model statistics (tdnnf_net_create_shared)."""
        import torch
        self.lib = hipabi.load()
        self.cfg = config
        self.h = C.c_void_p()
        self.share = share  # keeps the primary alive
        if share is None:
            hipabi.check(self.lib.tdnnf_net_create(C.byref(config), C.byref(self.h)))
        else:
            hipabi.check(self.lib.tdnnf_net_create_shared(C.byref(config), share.h, C.byref(self.h)))
        self.num_params = int(self.lib.tdnnf_net_num_params(self.h))
        self.params = torch.zeros(self.num_params, dtype=torch.float32, device="cuda") if share is None else share.params
        self.grads = torch.zeros(self.num_params, dtype=torch.float32, device="cuda") if share is None else share.grads
        hipabi.check(self.lib.tdnnf_net_set_buffers(self.h, hipabi.ptr(self.params), hipabi.ptr(self.grads)))
        self.components = []
        for i in range(self.lib.tdnnf_net_num_components(self.h)):
            name = C.create_string_buffer(64)
            begin, rows, cols, hb = C.c_longlong(), C.c_int(), C.c_int(), C.c_int()
            lrf, l2, mc, orth = C.c_float(), C.c_float(), C.c_float(), C.c_float()
            hipabi.check(self.lib.tdnnf_net_component_info(self.h, i, name, C.byref(begin), C.byref(rows), C.byref(cols),
                                                           C.byref(hb), C.byref(lrf), C.byref(l2), C.byref(mc), C.byref(orth)))
            self.components.append(dict(name=name.value.decode(), begin=begin.value, rows=rows.value, cols=cols.value,
                                        has_bias=hb.value, lr_factor=lrf.value, l2=l2.value, max_change=mc.value,
                                        orthonormal=orth.value, num_alpha=self.lib.tdnnf_net_component_num_alpha(self.h, i)))
        nt, t0 = C.c_int(), C.c_int()
        hipabi.check(self.lib.tdnnf_net_input_frames(self.h, C.byref(nt), C.byref(t0)))
        self.num_t_in, self.first_t = nt.value, t0.value
        self.results = torch.zeros(8, dtype=torch.float64, device="cuda")
        self.num_draws = self.lib.tdnnf_net_num_random_draws(self.h)
        self.draws = torch.full((max(self.num_draws, 1),), 0.5, dtype=torch.float32, device="cuda")
        if self.num_draws:
            hipabi.check(self.lib.tdnnf_net_set_random_draws(self.h, hipabi.ptr(self.draws)))

    def close(self):
        if self.h:
            self.lib.tdnnf_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def init_params_numpy(self, seed=0, output_stddev=None):
        """Flat float32 vector initialised like the reference's InitFromConfig: weights N(0, 1/input_dim),
        biases N(0,1) (TdnnDARTSV3Component nnet-tdnn-component.cc:145-175; affine layers alike); the lda
        transform is a random orthonormal matrix; output layers are zero in the recipes (param-stddev=0),
        `output_stddev` overrides that for parity tests."""
        rng = np.random.default_rng(seed)
        p = np.zeros(self.num_params, np.float32)
        for c in self.components:
            n = c["rows"] * c["cols"]
            if c["name"] == "lda":
                q = np.linalg.qr(rng.standard_normal((c["rows"], c["cols"])))[0]
                W, b = q.astype(np.float32), np.zeros(c["rows"], np.float32)
            elif c["name"].endswith((".softmax", ".alpha")):  # ConstantFunction / Onehot output_: zeros (InitFromConfig)
                W, b = np.zeros((c["rows"], c["cols"]), np.float32), None
            elif c["name"].startswith("output"):
                sd = 0.0 if output_stddev is None else output_stddev
                W = (rng.standard_normal((c["rows"], c["cols"])) * sd).astype(np.float32)
                b = (rng.standard_normal(c["rows"]) * sd).astype(np.float32)
            else:
                W = (rng.standard_normal((c["rows"], c["cols"])) / np.sqrt(c["cols"])).astype(np.float32)
                b = rng.standard_normal(c["rows"]).astype(np.float32)
            p[c["begin"]:c["begin"] + n] = W.ravel()
            n += c.get("num_alpha", 0)  # architecture logits start at 0 (nnet-tdnn-component.cc:176)
            if c["has_bias"]:
                p[c["begin"] + n:c["begin"] + n + c["rows"]] = b
        return p

    def set_random_draws(self, draws=None, generator=None):
        """Uniform(0,1) draws for the DARTS components of the next step (inputs, for reproducibility)."""
        import torch
        if not self.num_draws:
            return
        if draws is None:
            self.draws.copy_(torch.rand(self.num_draws, device="cuda", generator=generator).clamp_(1e-6, 1 - 1e-6))
        else:
            self.draws.copy_(torch.as_tensor(draws, dtype=torch.float32))

    def get_stats(self):
        """BatchNorm / ReLU statistics of the model as one float64 vector (layout: include/tdnnf_hip.h, tdnnf_net_get_stats)."""
        out = np.zeros(int(self.lib.tdnnf_net_stats_size(self.h)), np.float64)
        hipabi.check(self.lib.tdnnf_net_get_stats(self.h, out.ctypes.data_as(C.POINTER(C.c_double)), hipabi.stream()))
        return out

    def set_stats(self, stats):
        st = np.ascontiguousarray(stats, dtype=np.float64)
        assert st.size == int(self.lib.tdnnf_net_stats_size(self.h))
        hipabi.check(self.lib.tdnnf_net_set_stats(self.h, st.ctypes.data_as(C.POINTER(C.c_double)), hipabi.stream()))

    def write_model(self, path, binary=True, learning_rate=0.0):
        """nnet3 raw model file (text or binary) of this net: parameters + BatchNorm / ReLU statistics."""
        hipabi.check(self.lib.tdnnf_net_write_model(self.h, str(path).encode(), int(bool(binary)), float(learning_rate), hipabi.stream()))

    def read_model(self, path):
        hipabi.check(self.lib.tdnnf_net_read_model(self.h, str(path).encode(), hipabi.stream()))

    def set_dropout_proportion(self, proportion):
        """nnet3-copy --edits='set-dropout-proportion name=* proportion=p' (train.py's --trainer.dropout-schedule)."""
        hipabi.check(self.lib.tdnnf_net_set_dropout_proportion(self.h, float(proportion)))

    def set_temperature_proportion(self, proportion):
        """nnet edit "set-temperature-proportion name=* proportion=p" (temperature_schedule.py:57-60)."""
        hipabi.check(self.lib.tdnnf_net_set_temperature_proportion(self.h, float(proportion)))

    def set_learning_rate_factor(self, factor, name="*"):
        """nnet edit "set-learning-rate-factor name=<pattern> learning-rate-factor=f" (nnet-utils.cc:1232-1256); returns the number of
        components set.  self.components is refreshed."""
        cnt = C.c_int()
        hipabi.check(self.lib.tdnnf_net_set_learning_rate_factor(self.h, str(name).encode(), float(factor), C.byref(cnt)))
        for i, c in enumerate(self.components):
            lrf = C.c_float()
            hipabi.check(self.lib.tdnnf_net_component_info(self.h, i, None, None, None, None, None, C.byref(lrf), None, None, None))
            c["lr_factor"] = lrf.value
        return cnt.value

    def apply_edits(self, edits):
        """ReadEditConfig (/root/reference/src/nnet3/nnet-utils.cc:1166-1415) for the directives the NAS recipes use, applied to this
        net as nnet3-copy / nnet3-am-copy --edits would apply them to the model in memory: `edits` is the option's value (directives
        separated by ';' or newlines) or a whole "nnet3-copy --edits='...' - - |" command as train.py builds them
        (temperature_edit_string).  Supported: set-learning-rate-factor [name=pattern] learning-rate-factor=f
        (run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:129), set-temperature-proportion [name=*] proportion=p (the reference's own
        addition, :1352-1405), set-dropout-proportion [name=*] proportion=p (:1295-1330).  Anything else is an error, as the
        reference's KALDI_ERR "Directive ... is not currently supported"; so is a key the directive does not read ("Could not
        interpret ...").  Returns [(directive, number of components set)]."""
        done = []
        for directive, kv in parse_edits(edits):
            if directive == "set-learning-rate-factor":
                if "learning-rate-factor" not in kv:
                    raise ValueError("In edits-config, expected learning-rate-factor to be set in line: %s" % directive)
                name, f = kv.pop("name", "*"), float(kv.pop("learning-rate-factor"))
                n = self.set_learning_rate_factor(f, name)
            elif directive in ("set-temperature-proportion", "set-dropout-proportion"):
                if "proportion" not in kv:
                    raise ValueError("In edits-config, expected proportion to be set in line: %s" % directive)
                name, prop = kv.pop("name", "*"), float(kv.pop("proportion"))
                if name != "*":
                    raise ValueError("%s: only name=* is supported (one proportion for the whole net), got %r" % (directive, name))
                if directive == "set-temperature-proportion":
                    self.set_temperature_proportion(prop)
                    n = sum(1 for c in self.components if c["num_alpha"] or c["name"].endswith(".alpha"))
                else:
                    self.set_dropout_proportion(prop)
                    n = (self.cfg.num_layers + 1) if self.cfg.use_dropout else 0
            else:
                raise ValueError("Directive '%s' is not currently supported (reading edit-config)." % directive)
            if kv:
                raise ValueError("Could not interpret '%s' in edit config line %s" % (" ".join("%s=%s" % i for i in kv.items()), directive))
            done.append((directive, n))
        return done

    def set_params(self, flat):
        import torch
        self.params.copy_(torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float32)))

    # ------------------------------------------------------------------------ steps
    def forward_backward(self, feats, ivectors, den_graph, supervision, step=0):
        hipabi.check(self.lib.tdnnf_net_forward_backward(self.h, hipabi.pmat(feats), hipabi.pmat(ivectors), den_graph.h,
                                                         supervision.h, hipabi.ptr(self.results), int(step), hipabi.stream()))
        return self.results

    def update(self, learning_rate, l2_regularize_scale=None, step=0):
        if l2_regularize_scale is None:  # GetNumNvalues(eg.inputs) * l2_regularize_factor (UPSTREAM trainer)
            l2_regularize_scale = float(self.cfg.num_sequences)
        hipabi.check(self.lib.tdnnf_net_update(self.h, float(learning_rate), float(l2_regularize_scale), int(step), hipabi.stream()))

    def set_capture(self, on=True):
        """Parity aid: keep copies of the backward pass's derivative matrices (tdnnf_net_set_capture); read them with activation()."""
        hipabi.check(self.lib.tdnnf_net_set_capture(self.h, int(bool(on))))

    def activation(self, name):
        import torch
        r, c = C.c_int(), C.c_int()
        hipabi.check(self.lib.tdnnf_net_activation_dims(self.h, name.encode(), C.byref(r), C.byref(c)))
        out = torch.zeros(r.value, c.value, dtype=torch.float32, device="cuda")
        hipabi.check(self.lib.tdnnf_net_get_activation(self.h, name.encode(), hipabi.pmat(out), hipabi.stream()))
        return out

    # ------------------------------------------------------------ data-parallel step
    def set_batchnorm_sync_rccl(self, comm):
        """Synchronised BatchNorm with the collective issued by the library itself (tdnnf_net_set_batchnorm_sync_rccl: ncclAllReduce
        on the compute stream between the two finalize launches, no host code in between).  comm: an RcclComm, or None = off."""
        if comm is not None:
            _require_equal_shards(self.cfg.num_sequences, comm.group, comm.world)
        hipabi.check(self.lib.tdnnf_net_set_batchnorm_sync_rccl(self.h, comm.h_bn if comm is not None else None, comm.world if comm is not None else 1))
        self._bn_sync_cb = None
        return comm is not None

    def allreduce_grads_rccl(self, comm, comm_stream):
        """The gradient exchange of one minibatch issued by the library (tdnnf_net_allreduce_grads_rccl): one ncclAllReduce per gradient
        bucket on `comm_stream`, each behind the event recorded when the bucket became final; the current stream waits for the last."""
        import torch
        hipabi.check(self.lib.tdnnf_net_allreduce_grads_rccl(self.h, comm.h, C.c_void_p(comm_stream.cuda_stream), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def set_batchnorm_sync(self, on=True, group=None, min_world=2):
        """Synchronised BatchNorm (tdnnf_net_set_batchnorm_sync): every train-mode BatchNorm all-reduces its column sums over the
        ranks of `group` on the compute stream, so that a minibatch sharded over the ranks (bench.py --scaling strong) normalises
        exactly as the whole minibatch does on one GPU (nnet-normalize-component.cc:433-445: statistics over all rows).  Off (the
        default): statistics per shard, like the per-job statistics of Kaldi's parallel jobs.  RCCL (backend "nccl") reduces the
        device buffer in place on that stream; gloo (CPU rehearsals, also two ranks on one GPU) stages through the host."""
        import torch
        import torch.distributed as dist
        if not on or not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) >= min_world):
            hipabi.check(self.lib.tdnnf_net_set_batchnorm_sync(self.h, None, None, 1))
            self._bn_sync_cb = None
            return False
        world = dist.get_world_size(group)
        _require_equal_shards(self.cfg.num_sequences, group, world)

        class _DevDoubles:  # zero-copy view of `count` doubles at a raw device pointer
            def __init__(self, ptr, count):
                self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        def allreduce(ctx, buf, count, stream):
            try:
                t = torch.as_tensor(_DevDoubles(int(buf), int(count)), device="cuda")
                cur = torch.cuda.current_stream()
                st = cur if (stream or 0) == cur.cuda_stream else torch.cuda.ExternalStream(int(stream or 0))
                with torch.cuda.stream(st):
                    if dist.get_backend(group) == "gloo":
                        host = t.cpu()  # (synchronises the stream: rehearsal only)
                        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                        t.copy_(host)
                    else:
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                return 0
            except Exception as e:  # never raise through the C frames
                import sys
                print("batchnorm sync all-reduce failed: %r" % (e,), file=sys.stderr)
                return 1

        self._bn_sync_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p)(allreduce)  # kept alive with the net
        hipabi.check(self.lib.tdnnf_net_set_batchnorm_sync(self.h, C.cast(self._bn_sync_cb, C.c_void_p), None, int(world)))
        return True

    def allreduce_grads(self, group=None, min_world=2):
        """Sum the raw parameter gradients over ranks (RCCL all-reduce over xGMI: backend "nccl" on ROCm)."""
        allreduce_flat(self.grads, group, min_world)

    def grad_buckets(self):
        """[(begin, end)] element ranges of the flat gradient buffer in the order the backward pass finishes them
        (tdnnf_net_grad_bucket): the heads and prefinal-l first, then groups of tdnnf layers from the top down."""
        out = []
        for i in range(self.lib.tdnnf_net_num_grad_buckets(self.h)):
            b, e = C.c_longlong(), C.c_longlong()
            hipabi.check(self.lib.tdnnf_net_grad_bucket(self.h, i, C.byref(b), C.byref(e)))
            out.append((b.value, e.value))
        return out

    def allreduce_grads_overlapped(self, comm_stream, group=None, min_world=2):
        """The same sum, one collective per bucket, each enqueued on `comm_stream` behind the event the library recorded when
        that bucket's gradients became final -- so the reductions of the upper layers run while the backward pass (already
        enqueued: the host is far ahead of the GPU) is still working on the lower ones.  Call right after forward_backward;
        the current stream waits for the reductions before anything else touches the gradients.
        min_world=1 issues the collectives in a one-rank group too (the RCCL rehearsal on a one-GPU box)."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) >= min_world):
            return
        works = []
        for i, (b, e) in enumerate(self.grad_buckets()):
            hipabi.check(self.lib.tdnnf_net_wait_grad_bucket(self.h, i, C.c_void_p(comm_stream.cuda_stream)))
            with torch.cuda.stream(comm_stream):
                if dist.get_backend(group) == "gloo":  # rehearsal on one GPU: host-staged, blocking
                    allreduce_flat(self.grads[b:e], group)
                else:
                    works.append(dist.all_reduce(self.grads[b:e], op=dist.ReduceOp.SUM, group=group, async_op=True))
        for w in works:
            w.wait()  # the current stream waits for the collective's stream
        torch.cuda.current_stream().wait_stream(comm_stream)


def _require_equal_shards(num_sequences, group, world):
    """Synchronised BatchNorm forms its global row count as rows x world_size: every rank must hold the same number of sequences."""
    import torch
    import torch.distributed as dist
    if world <= 1 or not (dist.is_available() and dist.is_initialized()):
        return
    mine = torch.tensor([int(num_sequences)], dtype=torch.int64)
    if dist.get_backend(group) != "gloo":
        mine = mine.cuda()
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    counts = [int(t.item()) for t in every]
    if len(set(counts)) != 1:
        raise ValueError("synchronised BatchNorm needs the same number of sequences on every rank, got %r (shard a minibatch that divides "
                         "by the world size)" % (counts,))


class RcclComm:
    """RCCL communicators owned by the library (csrc/rccl_sync.hip) for the exchanges it issues itself from C++: rank 0 takes the
    unique ids, torch.distributed's group (whatever its backend) carries them to the other ranks, every rank joins on its current
    device.  TWO communicators: `h` for the gradient buckets (communication stream) and `h_bn` for the synchronised BatchNorm sums
    (compute stream) -- RCCL runs the operations of one communicator in issue order, and the buckets of a step are issued behind all
    of its BatchNorm collectives: on one communicator no bucket could start before the backward pass has ended.
    single=True: one-rank communicators without any process group (tests, rehearsals on one GPU).
    The outcome is agreed collectively: a failure on any rank (no RCCL, ncclCommInitRank error) raises on EVERY rank, so the callers'
    fallback to torch.distributed's collectives is taken by all ranks or none."""

    def __init__(self, group=None, single=False):
        import torch.distributed as dist
        self.lib = hipabi.load()
        self.group = group
        self.world, self.rank = (1, 0) if single else (dist.get_world_size(group), dist.get_rank(group))
        self.h, self.h_bn = None, None
        err = None
        ids = None
        if self.rank == 0:
            try:
                if not self.lib.tdnnf_rccl_available():
                    raise RuntimeError("no RCCL in the process and none to dlopen")
                ids = []
                for _ in range(2):
                    buf = (C.c_char * 128)()
                    hipabi.check(self.lib.tdnnf_rccl_unique_id(buf))
                    ids.append(bytes(buf))
            except Exception as e:  # (the other ranks wait in the broadcast: tell them instead of raising here)
                err, ids = "rank 0: %s" % e, None
        if self.world > 1:
            box = [(ids, err)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ids, err = box[0]
        if err is None:
            try:
                if not self.lib.tdnnf_rccl_available():
                    raise RuntimeError("no RCCL in the process and none to dlopen")
                hs = []
                for ident in ids:  # (same order on every rank: communicator creation is itself collective)
                    h = C.c_void_p()
                    hipabi.check(self.lib.tdnnf_rccl_comm_create(ident, self.world, self.rank, C.byref(h)))
                    hs.append(h)
                self.h, self.h_bn = hs
            except Exception as e:
                err = "rank %d: %s" % (self.rank, e)
        if self.world > 1:
            errs = [None] * self.world
            dist.all_gather_object(errs, err, group=group)
            err = next((e for e in errs if e), None)
        if err is not None:
            self.close()
            raise RuntimeError("the library's RCCL communicators could not be created (%s)" % err)

    def library_path(self):
        buf = C.create_string_buffer(1200)
        hipabi.check(self.lib.tdnnf_rccl_library_path(buf, 1200))
        return buf.value.decode()

    def allreduce_sum(self, tensor, stream=None):
        """In-place sum of a contiguous float32 / float64 device tensor over the ranks, on `stream` (default: the current one)."""
        import torch
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.dtype in (torch.float32, torch.float64)
        st = stream if stream is not None else torch.cuda.current_stream()
        hipabi.check(self.lib.tdnnf_rccl_allreduce_sum(self.h, C.c_void_p(tensor.data_ptr()), tensor.numel(), int(tensor.dtype == torch.float64),
                                                       C.c_void_p(st.cuda_stream)))
        return tensor

    def close(self):
        for name in ("h", "h_bn"):
            h = getattr(self, name, None)
            if h:
                self.lib.tdnnf_rccl_comm_destroy(h)
            setattr(self, name, None)

    def __del__(self):
        self.close()


def allreduce_flat(flat, group=None, min_world=2):
    """The one exchange step of the data-parallel path: sequences (chunks) of a minibatch are sharded over
    ranks (they are independent through every component except BatchNorm statistics, which stay per-shard
    like Kaldi's per-job statistics), every rank accumulates the raw gradient of its shard, and the flat
    gradient buffer is SUMMED over ranks -- 74.8 MB of fp32 for the 7q net.
    Step size: to first order this is Kaldi's num_jobs jobs at learning rate lr_eff x num_jobs followed by model
    averaging (steps/libs/nnet3/train/common.py:618), mean_j(lr_eff J g_j) = lr_eff sum_j g_j -- so the summed gradient
    is applied with the EFFECTIVE learning rate, not lr_eff x num_jobs (tests/test_data_parallel_gloo.py).
    The reference has no counterpart (single process).  No-op for world size 1.
    gloo (CPU rehearsals, also with both ranks on one GPU) reduces host memory: device tensors are staged."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) >= min_world:
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            host = flat.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def parse_edits(edits):
    """[(directive, {key: value})] of an --edits value or of a whole "nnet3-copy --edits='...' - - |" command: the option's ';'
    separate config lines (nnet3-copy.cc replaces them by newlines), a line is "directive key=value ..." (ConfigLine)."""
    import re
    text = str(edits)
    m = re.search(r"--edits=(['\"])(.*?)\1", text, flags=re.S)
    if m:
        text = m.group(2)
    out = []
    for line in re.split(r"[;\n]", text):
        line = line.split("#", 1)[0].strip()
        if not line:
            continue
        toks = line.split()
        kv = {}
        for t in toks[1:]:
            if "=" not in t:
                raise ValueError("edit config line %r: expected key=value, got %r" % (line, t))
            k, v = t.split("=", 1)
            kv[k] = v
        out.append((toks[0], kv))
    return out


def apply_cvupdate_seds(text, use_gumbel=True):
    """The literal text substitutions of run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:128-142 on a TEXT model that already went through
    --edits="set-learning-rate-factor learning-rate-factor=0": the seven `sed "s/a/b/g"` of the pipeline, in its order (with
    use_gumbel false the first flag substitution is the recipe's no-op `s/<use-gumbel> F/<use-gumbel> F/g`)."""
    subs = [("<TdnnDARTSV3Component> <LearningRateFactor> 0", "<TdnnDARTSV3Component> <LearningRateFactor> 0.0001"),
            ("<use-gumbel> F", "<use-gumbel> T" if use_gumbel else "<use-gumbel> F"),
            ("<update-alpha> F", "<update-alpha> T"),
            ("<update-theta> T", "<update-theta> F"),
            ("<uniform-sample> T", "<uniform-sample> F"),
            ("<TestMode> F", "<TestMode> T"),
            ("BatchNormComponent", "BatchNormTestComponent")]
    for a, b in subs:
        text = text.replace(a, b)
    return text


def shard_sequences(num_sequences, rank, world_size):
    """Contiguous shard [begin, end) of a global minibatch's sequences for this rank."""
    base, rem = divmod(num_sequences, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_supervision(sup, begin, end):
    """The numerator graphs of sequences [begin, end) of a merged minibatch (the dict of synth.make_supervision /
    egs.merge): what rank g of a strong-scaling step hands to tdnnf_supervision_create."""
    sb, ab = np.asarray(sup["seq_state_begin"]), np.asarray(sup["seq_arc_begin"])
    s0, s1, a0, a1 = int(sb[begin]), int(sb[end]), int(ab[begin]), int(ab[end])
    out = dict(sup)
    out.update(B=end - begin, seq_state_begin=(sb[begin:end + 1] - s0).astype(np.int32), seq_arc_begin=(ab[begin:end + 1] - a0).astype(np.int32),
               state_time=np.asarray(sup["state_time"])[s0:s1], final_logprob=np.asarray(sup["final_logprob"])[s0:s1],
               arc_src=(np.asarray(sup["arc_src"])[a0:a1] - s0).astype(np.int32), arc_dst=(np.asarray(sup["arc_dst"])[a0:a1] - s0).astype(np.int32),
               arc_pdf=np.asarray(sup["arc_pdf"])[a0:a1], arc_logprob=np.asarray(sup["arc_logprob"])[a0:a1])
    return out


def shard_rows(mat, num_sequences, begin, end):
    """Rows of sequences [begin, end) of a t-major matrix (row = t * num_sequences + b), t-major again."""
    m = np.asarray(mat)
    return np.ascontiguousarray(m.reshape(-1, num_sequences, m.shape[1])[:, begin:end]).reshape(-1, m.shape[1])


def learning_rate(iteration, num_jobs, num_iters, num_archives_processed, num_archives_to_process,
                  initial_effective_lrate=2.5e-4, final_effective_lrate=2.5e-5):
    """get_learning_rate, steps/libs/nnet3/train/common.py:606-618: exponential decay in the fraction of
    archives processed, the last iteration at the final rate, times num_jobs (rates from
    run_TDNN_DARTSV3_fbk_stride_pretrain.sh:202-203)."""
    if iteration + 1 >= num_iters:
        eff = final_effective_lrate
    else:
        eff = initial_effective_lrate * np.exp(num_archives_processed *
                                               np.log(float(final_effective_lrate) / initial_effective_lrate) /
                                               num_archives_to_process)
    return float(num_jobs * eff)


def read_kaldi_matrix(path):
    """A Kaldi matrix file as float32 (text "[ a b\n c d ]" or binary "\0B" + "FM "/"DM " + rows + cols + data): the
    lda.mat of configs/ that the fixed-affine-layer reads (run_tdnn_fbk_40_iv_sp_7q.sh:167).  Formats upstream, restated."""
    import struct
    raw = open(path, "rb").read()
    if raw[:2] == b"\0B":
        tok, rest = raw[2:].split(b" ", 1)
        if tok not in (b"FM", b"DM"):
            raise ValueError("%s: unsupported matrix type %r" % (path, tok))
        if rest[0] != 4 or rest[5] != 4:
            raise ValueError("%s: bad matrix header" % path)
        rows, cols = struct.unpack("<i", rest[1:5])[0], struct.unpack("<i", rest[6:10])[0]
        dt = np.dtype("<f4") if tok == b"FM" else np.dtype("<f8")
        data = np.frombuffer(rest[10:10 + rows * cols * dt.itemsize], dtype=dt)
        if data.size != rows * cols:
            raise ValueError("%s: truncated matrix" % path)
        return data.reshape(rows, cols).astype(np.float32)
    text = raw.decode().strip()
    if not (text.startswith("[") and text.endswith("]")):
        raise ValueError("%s: not a Kaldi text matrix" % path)
    rows = [r.split() for r in text[1:-1].strip().split("\n") if r.strip()]
    if len({len(r) for r in rows}) != 1:
        raise ValueError("%s: ragged text matrix" % path)
    return np.asarray(rows, dtype=np.float32)


def set_lda(params, components, lda_matrix):
    """Puts a D x (D + 1) lda.mat ([linear | offset], as FixedAffineComponent::Init splits it) into a flat parameter vector."""
    c = next(c for c in components if c["name"] == "lda")
    m = np.asarray(lda_matrix, np.float32)
    if m.shape != (c["rows"], c["cols"] + 1):
        raise ValueError("lda matrix is %s, the net's lda layer wants %d x %d" % (m.shape, c["rows"], c["cols"] + 1))
    n = c["rows"] * c["cols"]
    params[c["begin"]:c["begin"] + n] = m[:, :-1].reshape(-1)
    params[c["begin"] + n:c["begin"] + n + c["rows"]] = m[:, -1]
    return params


def synthetic_egs(net, seed=0):
    """fbank ~ N(0,1) [num_t_in*B, feat_dim] t-major and ivector ~ N(0,1) [B, ivector_dim]."""
    rng = np.random.default_rng(seed)
    B = net.cfg.num_sequences
    feats = rng.standard_normal((net.num_t_in * B, net.cfg.feat_dim)).astype(np.float32)
    iv = rng.standard_normal((B, net.cfg.ivector_dim)).astype(np.float32)
    return feats, iv


def temperature_proportion(data_fraction):
    """get_temperature_edit_string, steps/libs/nnet3/train/temperature_schedule.py:51: the Gumbel temperature proportion
    falls linearly from 1 to 0.03 with the fraction of the data processed."""
    return (1.0 - data_fraction) * (1.0 - 0.03) + 0.03


def temperature(temperature_init, temperature_final, data_fraction):
    """get_temperature_edit_string_adapt, temperature_schedule.py:20: geometric interpolation."""
    return temperature_init * (temperature_final / temperature_init) ** data_fraction


def parse_dropout_schedule(schedule):
    """[(data_fraction, proportion)] in ascending data_fraction of a --trainer.dropout-schedule function such as
    '0,0@0.20,0.5@0.50,0' (run_tdnn_fbk_40_iv_sp_7q.sh:48): the parser the reference keeps (commented out) in
    steps/libs/nnet3/train/temperature_schedule.py:122-182 (_parse_dropout_string): at least two values; the first sits at
    data fraction 0, the last at 1; a middle value 'v@x' at x, a middle value WITHOUT '@x' at 0.5; data fractions must not
    decrease and must lie in [0, 1], proportions in [0, 1]."""
    parts = str(schedule).strip().split(",")
    if len(parts) < 2:
        raise ValueError("dropout schedule %r: at least the start and end proportions are needed" % (schedule,))
    pts = [(0.0, float(parts[0]))]
    for tok in parts[1:-1]:
        v, at, x = tok.strip().partition("@")
        frac = float(x) if at else 0.5
        if frac < pts[-1][0] or frac > 1.0:
            raise ValueError("dropout schedule %r: data fractions must be in increasing order and <= 1 (%r)" % (schedule, tok))
        pts.append((frac, float(v)))
    pts.append((1.0, float(parts[-1])))
    if not all(0.0 <= f <= 1.0 and 0.0 <= v <= 1.0 for f, v in pts):
        raise ValueError("dropout schedule %r: fractions and proportions must lie in [0, 1]" % (schedule,))
    return pts


def dropout_proportion(schedule, data_fraction):
    """Dropout proportion at data_fraction: piecewise linear on parse_dropout_schedule(), as _get_component_dropout
    (temperature_schedule.py:185-239): at a repeated data fraction the later point wins from there on.  One function for every
    component (the recipes give no per-pattern rules); None = no dropout."""
    if schedule is None:
        return 0.0
    pts = parse_dropout_schedule(schedule)
    f = min(max(float(data_fraction), 0.0), 1.0)
    if f == 0.0:
        return pts[0][1]
    lo = max(i for i, (x, _) in enumerate(pts) if x <= f)  # the last point at or below f (the reference searches from the top)
    if lo == len(pts) - 1:
        return pts[-1][1]
    (x0, v0), (x1, v1) = pts[lo], pts[lo + 1]
    return v0 if x1 == x0 else v0 + (v1 - v0) * (f - x0) / (x1 - x0)


def temperature_edit_string(data_fraction):
    """The nnet3-copy command train.py prepends to the model of an iteration under --temperature_schedule
    (get_temperature_edit_string, temperature_schedule.py:34-67); ChainNet.set_temperature_proportion is its effect."""
    return "nnet3-copy --edits='set-temperature-proportion name=* proportion={0}' - - |".format(temperature_proportion(data_fraction))


def temperature_adapt_edit_string(temperature_init, temperature_final, data_fraction):
    """get_temperature_edit_string_adapt, temperature_schedule.py:15-32 (None when either end is None)."""
    if temperature_init is None or temperature_final is None:
        return None
    return "nnet3-copy --edits='set-temperature temperature={0}' - - |".format(temperature(temperature_init, temperature_final, data_fraction))


def training_schedule(num_iters, num_archives_to_process, num_jobs_initial=1, num_jobs_final=1, use_temperature_schedule=False,
                      initial_effective_lrate=2.5e-4, final_effective_lrate=2.5e-5):
    """The per-iteration settings train.py derives before launching its jobs (steps/nnet3/chain/train.py:473-531):
    number of jobs (linear ramp, train.py:477-479 / common.py), learning rate (common.py:606-618), and -- with
    --temperature_schedule -- the temperature proportion of the data fraction processed so far.  Yields dicts."""
    processed = 0
    for it in range(num_iters):
        jobs = int(0.5 + num_jobs_initial + (num_jobs_final - num_jobs_initial) * float(it) / num_iters)
        frac = float(processed) / num_archives_to_process
        yield dict(iteration=it, num_jobs=jobs, data_fraction=frac,
                   learning_rate=learning_rate(it, jobs, num_iters, processed, num_archives_to_process, initial_effective_lrate,
                                               final_effective_lrate),
                   temperature_proportion=temperature_proportion(frac) if use_temperature_schedule else None)
        processed += jobs
