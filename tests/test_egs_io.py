"""CPU: chain example archives (csrc/egs_io.hip, tdnn-f_nas_amd/egs.py).  The formats are upstream
Kaldi / OpenFst, restated -- nothing in the reference to pin them with -- so the tests are (1) round trips through the
library's own writer and (2) byte streams assembled HERE, field by field from the published formats, for the branches the
writer never produces (escaped indexes, per-column compressed matrices, 8-bit matrices, <DW> / <DW2> weights, the older
stream without <End2End>, <AlignmentPdfs>)."""
import struct

import numpy as np
import pytest


def _minibatch(pkg, B=3, T=8, P=20, feat_dim=6, iv_dim=4, ctx=5, seed=0):
    rng = np.random.default_rng(seed)
    sup = pkg.synth.make_supervision(B, T, P, max_alt=3, seed=seed + 1, weight=1.0)
    rows = 3 * T + 2 * ctx
    feats = [rng.standard_normal((rows, feat_dim)).astype(np.float32) for _ in range(B)]
    ivs = [rng.standard_normal(iv_dim).astype(np.float32) for _ in range(B)]
    return sup, feats, ivs, rows


@pytest.mark.parametrize("compress", [False, True], ids=["full", "cm2"])
def test_round_trip_and_merge(pkg, tmp_path, compress):
    E = pkg.egs
    B, T, P, ctx = 3, 8, 20, 5
    sup, feats, ivs, rows = _minibatch(pkg, B, T, P, ctx=ctx)
    path = tmp_path / "cegs.1.ark"
    with E.Writer(path) as w:
        for b in range(B):
            w.write("utt%d-0" % b, feats[b], -ctx, E.sequence_of(sup, b), P, ivector=ivs[b], compress=compress)
    egs = list(E.Reader(path))
    assert [e.key for e in egs] == ["utt0-0", "utt1-0", "utt2-0"]
    tol = 0.0 if not compress else (max(f.max() - f.min() for f in feats) / 65535.0)  # 16 bits over the matrix's range
    for b, e in enumerate(egs):
        x, t0 = e.input("input")
        assert t0 == -ctx and x.shape == feats[b].shape and np.abs(x - feats[b]).max() <= tol
        iv, _ = e.input("ivector")
        assert np.array_equal(iv.reshape(-1), ivs[b])
        info = e.supervision_info()
        assert (info["num_sequences"], info["frames_per_seq"], info["label_dim"], info["first_t"], info["t_step"]) == (1, T, P, 0, 3)
        assert info["num_arcs"] == sup["seq_arc_begin"][b + 1] - sup["seq_arc_begin"][b]
    # merge: the trainer's window [first_t, first_t + num_t) out of the examples' wider context, t-major
    first_t, num_t = -3, 3 * T + 4
    f, iv, s = E.merge(egs, first_t, num_t)
    for b in range(B):
        np.testing.assert_allclose(f[b::B], feats[b][first_t + ctx:first_t + ctx + num_t], rtol=0, atol=tol)
        assert np.array_equal(iv[b], ivs[b])
    for k in ("seq_state_begin", "seq_arc_begin", "state_time", "arc_src", "arc_dst", "arc_pdf"):
        assert np.array_equal(s[k], sup[k]), k
    assert np.array_equal(s["final_logprob"], sup["final_logprob"]) and np.array_equal(s["arc_logprob"], sup["arc_logprob"]) and s["weight"] == 1.0
    # frame shift: nnet3-chain-copy-egs --frame-shift=1 relabels the inputs one frame later, the net sees earlier rows
    f1, _, _ = E.merge(egs, first_t, num_t, frame_shift=1)
    np.testing.assert_allclose(f1[0::B], feats[0][first_t + ctx - 1:first_t + ctx - 1 + num_t], rtol=0, atol=tol)
    with pytest.raises(pkg.hipabi.HipAbiError, match="the net needs"):
        E.merge(egs, -ctx - 1, num_t)
    with pytest.raises(pkg.hipabi.HipAbiError, match="the net needs"):
        E.merge(egs, first_t, num_t, frame_shift=-5)


# ---------------------------------------------------------------- an independent writer of the published formats
def tok(s):
    return s.encode() + b" "


def i32(v):
    return b"\x04" + struct.pack("<i", v)


def f32(v):
    return b"\x04" + struct.pack("<f", v)


def index_vector(indexes):
    out = tok("<I1V>") + i32(len(indexes))
    prev = None
    for (n, t, x) in indexes:
        if prev is None:
            small = n == 0 and x == 0 and abs(t) < 125
            delta = t
        else:
            small = n == prev[0] and x == prev[2] and abs(t - prev[1]) < 125
            delta = t - prev[1]
        out += struct.pack("<b", delta) if small else b"\x7f" + i32(n) + i32(t) + i32(x)
        prev = (n, t, x)
    return out


def compact_acceptor(num_states, arcs, finals):
    """arcs: (src, label, cost, dst); finals: {state: cost}.  OpenFst CompactFst<StdArc, AcceptorCompactor>, version 2."""
    elems, off = [], []
    for s in range(num_states):
        off.append(len(elems))
        if s in finals:
            elems.append((-1, finals[s], -1))
        elems += [(l, c, d) for (a, l, c, d) in arcs if a == s]
    off.append(len(elems))
    def fs(x):
        return struct.pack("<i", len(x)) + x.encode()
    hdr = struct.pack("<i", 2125659606) + fs("compact_acceptor") + fs("standard") + struct.pack("<iiQqqq", 2, 0, 0, 0, num_states, len(arcs))
    return hdr + struct.pack("<%dI" % len(off), *off) + b"".join(struct.pack("<ifi", *e) for e in elems)


def eg_bytes(key, input_matrix_bytes, n_rows, first_t, fst, frames, label_dim, weight=1.0, end2end_token=True, dw=None, align=None, t_step=3,
             escaped_indexes=False):
    idx = [(0, first_t + i, 0) for i in range(n_rows)]
    if escaped_indexes:  # a first time outside the one-byte range forces the 127 escape
        idx = [(0, first_t - 400 + i, 0) for i in range(n_rows)]
    out = tok(key) + b"\0B" + tok("<Nnet3ChainEg>") + tok("<NumInputs>") + i32(1)
    out += tok("<NnetIo>") + tok("input") + index_vector(idx) + input_matrix_bytes + tok("</NnetIo>")
    out += tok("<NumOutputs>") + i32(1) + tok("<NnetChainSup>") + tok("output") + index_vector([(0, t_step * i, 0) for i in range(frames)])
    out += tok("<Supervision>") + tok("<Weight>") + f32(weight) + tok("<NumSequences>") + i32(1) + tok("<FramesPerSeq>") + i32(frames)
    out += tok("<LabelDim>") + i32(label_dim) + (tok("<End2End>") + b"F" if end2end_token else b"") + fst
    if align is not None:
        out += tok("<AlignmentPdfs>") + b"\x04" + struct.pack("<i", len(align)) + struct.pack("<%di" % len(align), *align)
    out += tok("</Supervision>")
    if dw == "chars":
        out += tok("<DW>") + b"\x01" + struct.pack("<i", frames) + bytes([255] * (frames - 1) + [0])
    elif dw == "floats":
        out += tok("<DW2>") + tok("FV") + i32(frames) + struct.pack("<%df" % frames, *[0.5] * frames)
    return out + tok("</NnetChainSup>") + tok("</Nnet3ChainEg>")


def two_frame_fst():
    # start 0 -(pdf 3, cost .25)-> 1, 0 -(pdf 7, cost .5)-> 2, 1 -(pdf 5, cost 0)-> 3, 2 -(pdf 5, cost 1)-> 3, final 3 with cost .125
    return compact_acceptor(4, [(0, 4, 0.25, 1), (0, 8, 0.5, 2), (1, 6, 0.0, 3), (2, 6, 1.0, 3)], {3: 0.125})


def test_reader_on_independently_assembled_streams(pkg, tmp_path):
    E = pkg.egs
    rng = np.random.default_rng(2)
    rows, cols = 9, 5
    X = rng.standard_normal((rows, cols)).astype(np.float32)
    full = tok("FM") + i32(rows) + i32(cols) + X.tobytes()
    # CompressedMatrix format 1: global header (min, range, rows, cols), per column 4 x uint16 percentiles, bytes column-major
    mn, rg = float(X.min()), float(X.max() - X.min())
    def to16(v):
        return int(round((v - mn) / rg * 65535))
    def from16(u):
        return mn + rg * u / 65535.0
    hdr, byts, expect1 = b"", b"", np.zeros_like(X)
    for j in range(cols):
        col = np.sort(X[:, j])
        p = [to16(col[0]), to16(col[rows // 4]), to16(col[3 * rows // 4]), to16(col[-1])]
        p = [p[0], max(p[1], p[0] + 1), 0, 0][:2] + [max(p[2], p[1] + 2), 0][:1] + [max(p[3], p[2] + 3)]
        hdr += struct.pack("<4H", *p)
        q = [from16(u) for u in p]
        for i in range(rows):
            v = X[i, j]
            if v < q[1]:
                c = int(np.clip(round((v - q[0]) / (q[1] - q[0]) * 64), 0, 64))
            elif v < q[2]:
                c = int(np.clip(64 + round((v - q[1]) / (q[2] - q[1]) * 128), 64, 192))
            else:
                c = int(np.clip(192 + round((v - q[2]) / (q[3] - q[2]) * 63), 192, 255))
            byts += bytes([c])
            expect1[i, j] = (q[0] + (q[1] - q[0]) * c / 64 if c <= 64 else q[1] + (q[2] - q[1]) * (c - 64) / 128 if c <= 192
                             else q[2] + (q[3] - q[2]) * (c - 192) / 63)
    cm1 = tok("CM") + struct.pack("<ffii", mn, rg, rows, cols) + hdr + byts
    b8 = np.clip(np.round((X - mn) / rg * 255), 0, 255).astype(np.uint8)
    cm3 = tok("CM3") + struct.pack("<ffii", mn, rg, rows, cols) + b8.tobytes()
    fst = two_frame_fst()
    blob = (eg_bytes("a", full, rows, -2, fst, 2, 10, weight=0.5, dw="chars") +
            eg_bytes("b", cm1, rows, -2, fst, 2, 10, end2end_token=False, dw="floats") +
            eg_bytes("c", cm3, rows, -2, fst, 2, 10, align=[3, 5], escaped_indexes=True))
    path = tmp_path / "hand.ark"
    path.write_bytes(blob)
    a, b, c = list(E.Reader(path))
    xa, t0 = a.input()
    assert t0 == -2 and np.array_equal(xa, X) and a.supervision_info()["weight"] == 0.5
    xb, _ = b.input()
    np.testing.assert_allclose(xb, expect1, rtol=0, atol=1e-6)
    assert np.abs(xb - X).max() < 0.05 * rg  # and it is a compression of X
    xc, t0c = c.input()
    np.testing.assert_allclose(xc, mn + rg * b8 / 255.0, rtol=0, atol=1e-6)
    assert t0c == -402
    # the supervision: labels are pdf-id + 1, weights are costs; states get their times
    _, _, s = E.merge([a], -2, rows, with_ivectors=False)
    assert s["T"] == 2 and list(s["state_time"]) == [0, 1, 1, 2] and list(s["arc_pdf"]) == [3, 7, 5, 5]
    np.testing.assert_allclose(s["arc_logprob"], [-0.25, -0.5, 0.0, -1.0])
    assert list(s["arc_src"]) == [0, 0, 1, 2] and list(s["arc_dst"]) == [1, 2, 3, 3]
    assert np.isneginf(s["final_logprob"][:3]).all() and s["final_logprob"][3] == -0.125 and s["weight"] == 0.5


def test_reader_errors(pkg, tmp_path):
    E = pkg.egs
    X = np.zeros((4, 2), np.float32)
    full = tok("FM") + i32(4) + i32(2) + X.tobytes()
    good = eg_bytes("k", full, 4, 0, two_frame_fst(), 2, 10)
    p = tmp_path / "x.ark"
    with pytest.raises(pkg.hipabi.HipAbiError, match="cannot open"):
        E.Reader(tmp_path / "missing.ark")
    p.write_bytes(good[:len(good) // 2])
    with pytest.raises(pkg.hipabi.HipAbiError, match="end of file"):
        list(E.Reader(p))
    p.write_bytes(good.replace(b"k \0B", b"k <N", 1))
    with pytest.raises(pkg.hipabi.HipAbiError, match="binary mode"):
        list(E.Reader(p))
    p.write_bytes(good.replace(b"<End2End> F", b"<End2End> T", 1))
    with pytest.raises(pkg.hipabi.HipAbiError, match="end-to-end"):
        list(E.Reader(p))
    # an acceptor that is not time-synchronous (a skip arc 0 -> 3) is refused
    bad = compact_acceptor(4, [(0, 4, 0.25, 1), (0, 8, 0.5, 3), (1, 6, 0.0, 3)], {3: 0.0})
    p.write_bytes(eg_bytes("k", full, 4, 0, bad, 2, 10))
    with pytest.raises(pkg.hipabi.HipAbiError, match="time-synchronous"):
        list(E.Reader(p))
    # a label beyond label-dim
    p.write_bytes(eg_bytes("k", full, 4, 0, two_frame_fst(), 2, 5))
    with pytest.raises(pkg.hipabi.HipAbiError, match="label"):
        list(E.Reader(p))
    p.write_bytes(good + good.replace(b"k ", b"k2 ", 1))
    assert [e.key for e in E.Reader(p)] == ["k", "k2"]
