"""ctypes plumbing of the egs reader / writer / merge in csrc/egs_io.hip (include/tdnnf_hip.h, "egs" section): chain
example archives in Kaldi's binary format (restated, see the header of egs_io.hip) to and from numpy, and the merge of
single-sequence examples into the minibatch the trainer takes.  Host only, no GPU needed."""
import ctypes as C

import numpy as np

from . import hipabi


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Eg:
    """One NnetChainExample read from an archive."""

    def __init__(self, handle):
        self.lib = hipabi.load()
        self.h = handle
        self.key = self.lib.tdnnf_eg_key(self.h).decode()

    def close(self):
        if self.h:
            self.lib.tdnnf_eg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def input(self, name="input"):
        """(matrix rows x cols, time of the first row)"""
        r, c, t0 = C.c_int(), C.c_int(), C.c_int()
        hipabi.check(self.lib.tdnnf_eg_input_info(self.h, name.encode(), C.byref(r), C.byref(c), C.byref(t0)))
        out = np.zeros((r.value, c.value), np.float32)
        hipabi.check(self.lib.tdnnf_eg_input_copy(self.h, name.encode(), _p(out)))
        return out, t0.value

    def supervision_info(self):
        w = C.c_float()
        v = [C.c_int() for _ in range(7)]
        hipabi.check(self.lib.tdnnf_eg_supervision_info(self.h, C.byref(w), *[C.byref(x) for x in v]))
        keys = ("num_sequences", "frames_per_seq", "label_dim", "num_states", "num_arcs", "first_t", "t_step")
        return dict(weight=w.value, **{k: x.value for k, x in zip(keys, v)})


class Reader:
    """Iterates over the examples of one archive:  for eg in Reader(path): ..."""

    def __init__(self, path):
        self.lib = hipabi.load()
        self.h = C.c_void_p()
        hipabi.check(self.lib.tdnnf_egs_reader_open(str(path).encode(), C.byref(self.h)))

    def __iter__(self):
        return self

    def __next__(self):
        eg = C.c_void_p()
        hipabi.check(self.lib.tdnnf_egs_reader_next(self.h, C.byref(eg)))
        if not eg.value:
            self.close()
            raise StopIteration
        return Eg(eg)

    def close(self):
        if self.h:
            self.lib.tdnnf_egs_reader_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge(egs, first_t, num_t, frame_shift=0, with_ivectors=True):
    """nnet3-chain-merge-egs (+ --frame-shift of nnet3-chain-copy-egs) for single-sequence examples: returns
    (feats (num_t * n) x D t-major, ivectors n x Di or None, supervision dict in the form hipabi.Supervision takes)."""
    lib = hipabi.load()
    n = len(egs)
    arr = (C.c_void_p * n)(*[e.h for e in egs])
    ns, na, T, D, Di = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    hipabi.check(lib.tdnnf_egs_merge_sizes(arr, n, C.byref(ns), C.byref(na), C.byref(T), C.byref(D), C.byref(Di)))
    feats = np.zeros((num_t * n, D.value), np.float32)
    iv = np.zeros((n, Di.value), np.float32) if with_ivectors and Di.value else None
    sup = dict(B=n, T=T.value, seq_state_begin=np.zeros(n + 1, np.int32), seq_arc_begin=np.zeros(n + 1, np.int32),
               state_time=np.zeros(ns.value, np.int32), final_logprob=np.zeros(ns.value, np.float32), arc_src=np.zeros(na.value, np.int32),
               arc_dst=np.zeros(na.value, np.int32), arc_pdf=np.zeros(na.value, np.int32), arc_logprob=np.zeros(na.value, np.float32))
    w = C.c_float()
    hipabi.check(lib.tdnnf_egs_merge(arr, n, int(first_t), int(num_t), int(frame_shift), _p(feats), _p(iv) if iv is not None else None,
                                     _p(sup["seq_state_begin"]), _p(sup["seq_arc_begin"]), _p(sup["state_time"]), _p(sup["final_logprob"]),
                                     _p(sup["arc_src"]), _p(sup["arc_dst"]), _p(sup["arc_pdf"]), _p(sup["arc_logprob"]), C.byref(w)))
    sup["weight"] = float(w.value)
    return feats, iv, sup


def sequence_of(sup, b):
    """Sequence b of a minibatch supervision dict (synth.make_supervision) as one sequence's local arrays, state 0 = start."""
    s0, s1 = int(sup["seq_state_begin"][b]), int(sup["seq_state_begin"][b + 1])
    a0, a1 = int(sup["seq_arc_begin"][b]), int(sup["seq_arc_begin"][b + 1])
    assert sup["state_time"][s0] == 0 and (sup["state_time"][s0 + 1:s1] > 0).all(), "state 0 of the sequence must be its start state"
    return dict(final_logprob=np.ascontiguousarray(sup["final_logprob"][s0:s1], np.float32), arc_src=np.ascontiguousarray(sup["arc_src"][a0:a1] - s0, np.int32),
                arc_dst=np.ascontiguousarray(sup["arc_dst"][a0:a1] - s0, np.int32), arc_pdf=np.ascontiguousarray(sup["arc_pdf"][a0:a1], np.int32),
                arc_logprob=np.ascontiguousarray(sup["arc_logprob"][a0:a1], np.float32), frames=int(sup["T"]), weight=float(sup.get("weight", 1.0)))


class Writer:
    def __init__(self, path):
        self.lib = hipabi.load()
        self.h = C.c_void_p()
        hipabi.check(self.lib.tdnnf_egs_writer_open(str(path).encode(), C.byref(self.h)))

    def write(self, key, feats, first_t, seq, label_dim, ivector=None, compress=True, t_step=3):
        """seq: one sequence's supervision (sequence_of)."""
        feats = np.ascontiguousarray(feats, np.float32)
        iv = np.ascontiguousarray(ivector, np.float32).reshape(-1) if ivector is not None else None
        hipabi.check(self.lib.tdnnf_egs_writer_write(
            self.h, key.encode(), _p(feats), feats.shape[0], feats.shape[1], int(first_t), _p(iv) if iv is not None else None, iv.size if iv is not None else 0,
            int(bool(compress)), float(seq["weight"]), int(seq["frames"]), int(t_step), int(label_dim), seq["final_logprob"].size, seq["arc_src"].size,
            _p(seq["final_logprob"]), _p(seq["arc_src"]), _p(seq["arc_dst"]), _p(seq["arc_pdf"]), _p(seq["arc_logprob"])))

    def close(self):
        if self.h:
            h, self.h = self.h, None
            hipabi.check(self.lib.tdnnf_egs_writer_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def minibatches(path, net, frame_shift=0, discard_partial=True, prefetch=0):
    """Reads an archive and yields (feats, ivectors, hipabi.Supervision) device objects for ChainNet.forward_backward, merging
    net.cfg.num_sequences examples at a time in archive order (nnet3-chain-merge-egs --minibatch-size; the archive is expected to
    be shuffled already, as nnet3-chain-shuffle-egs leaves it).  prefetch > 0: reading, decompression and the merge run on a
    worker thread that many minibatches ahead (the C calls release the GIL), the copy to the device stays on the caller."""
    import torch
    B = net.cfg.num_sequences
    first_t, num_t, with_iv = net.first_t, net.num_t_in, net.cfg.ivector_dim > 0
    chunk, sub = net.cfg.frames_per_chunk, net.cfg.frame_subsampling

    def host_batches():
        group = []
        reader = Reader(path)
        try:
            for eg in reader:
                group.append(eg)
                if len(group) == B:
                    f, iv, sup = merge(group, first_t, num_t, frame_shift, with_ivectors=with_iv)
                    for e in group:
                        e.close()
                    group = []
                    if sup["T"] * sub != chunk:
                        raise ValueError("examples of %d output frames, the net was built for %d" % (sup["T"], chunk // sub))
                    yield f, iv, sup
            if group and not discard_partial:
                raise ValueError("the last %d examples do not fill a minibatch of %d" % (len(group), B))
        finally:  # also when the consumer stops early: no example handle or open archive is left behind
            for e in group:
                e.close()
            if hasattr(reader, "close"):
                reader.close()

    def to_device(f, iv, sup):
        return torch.from_numpy(f).cuda(), torch.from_numpy(iv).cuda() if iv is not None else None, hipabi.Supervision(sup)

    if prefetch <= 0:
        for item in host_batches():
            yield to_device(*item)
        return
    import queue
    import threading
    q = queue.Queue(maxsize=prefetch)
    stop = threading.Event()

    def put(item):
        """hand `item` to the consumer unless it has stopped listening (generator closed, exception in to_device)"""
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def worker():
        gen = host_batches()
        try:
            for item in gen:
                if not put(item):
                    return
            put(None)
        except BaseException as e:  # handed to the consumer
            put(e)
        finally:
            gen.close()  # closes the Reader and the examples of an unfinished group (host_batches' own finally)

    th = threading.Thread(target=worker, daemon=True)
    th.start()
    try:
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            yield to_device(*item)
    finally:
        stop.set()
        th.join(timeout=5)


def minibatches_by_width(path, nets, frame_shift=0):
    """An archive with examples of several chunk widths (--egs.chunk-width 150,110,100) and one ChainNet per width
    (ChainNet(cfg, share=primary)): examples are binned by their number of output frames and a minibatch is emitted for a
    width as soon as its bin holds that net's num_sequences examples (nnet3-chain-merge-egs).  Yields (net, feats, ivectors,
    hipabi.Supervision); what is left in the bins at the end of the archive is dropped."""
    import torch
    by_frames = {n.cfg.frames_per_chunk // n.cfg.frame_subsampling: n for n in nets}
    bins = {k: [] for k in by_frames}
    for eg in Reader(path):
        T = eg.supervision_info()["frames_per_seq"]
        if T not in by_frames:
            raise ValueError("example %s has %d output frames, nets were given for %s" % (eg.key, T, sorted(by_frames)))
        net, group = by_frames[T], bins[T]
        group.append(eg)
        if len(group) == net.cfg.num_sequences:
            f, iv, sup = merge(group, net.first_t, net.num_t_in, frame_shift, with_ivectors=net.cfg.ivector_dim > 0)
            for e in group:
                e.close()
            bins[T] = []
            yield net, torch.from_numpy(f).cuda(), torch.from_numpy(iv).cuda() if iv is not None else None, hipabi.Supervision(sup)
