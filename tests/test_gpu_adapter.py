"""-m gpu: the Kaldi-side adapter (include/tdnnf_nnet3_adapter.h) exercised from C++.

tests/adapter_driver.cc is compiled against the adapter header with a CuMatrixBase<float> stand-in over hipMalloc memory
and drives every component family of the hot path the way the edited Propagate / Backprop bodies of INTEGRATION.md would,
including the natural-gradient branch every recipe takes (UpdateNaturalGradient nnet-tdnn-component.cc:457-626,
NaturalGradientAffineComponent::Update nnet-simple-component.cc:2980-3024).  The outputs are compared with the oracle's
literal formulation."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from tests.gpu_util import F, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_blob(path, arrays):
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.atleast_2d(np.asarray(a, F))
            f.write(struct.pack("<i", len(name)) + name.encode() + struct.pack("<ii", *a.shape) + np.ascontiguousarray(a).tobytes())


def read_blob(path):
    out, raw, o = {}, open(path, "rb").read(), 0
    while o < len(raw):
        n, = struct.unpack_from("<i", raw, o)
        name = raw[o + 4:o + 4 + n].decode()
        r, c = struct.unpack_from("<ii", raw, o + 4 + n)
        o += 12 + n
        out[name] = np.frombuffer(raw, F, r * c, o).reshape(r, c).copy()
        o += 4 * r * c
    return out


@pytest.fixture(scope="module")
def driver(pkg, tmp_path_factory):
    exe = tmp_path_factory.mktemp("adapter") / "adapter_driver"
    lib_dir = os.path.dirname(pkg.hipabi.LIB_PATH)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "adapter_driver.cc"),
                           "-L", lib_dir, "-ltdnnf_hip", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)])

    def run(what, arrays, tmp_path):
        fin, fout = tmp_path / (what + "_in.bin"), tmp_path / (what + "_out.bin")
        write_blob(fin, arrays)
        p = subprocess.run([str(exe), what, str(fin), str(fout)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        return read_blob(fout)

    return run


def _rand(rng, *shape):
    return rng.standard_normal(shape).astype(F)


def _ng_update(L, ora, X, dy, ngi, ngo, lr, Wacc, bacc):
    """NaturalGradientAffineComponent::Update on an already spliced X (with the ones column when bacc is given)"""
    Y = dy.copy()
    a, b = C.c_float(1.0), C.c_float(1.0)
    L.oracle_ng_precondition(ngi, ora.omat(X), C.byref(a))
    L.oracle_ng_precondition(ngo, ora.omat(Y), C.byref(b))
    sc = F(a.value * b.value) * F(lr)
    nW = Wacc.shape[1]
    Xw = np.ascontiguousarray(X[:, :nW])
    L.oracle_affine_update_simple(ora.omat(Xw), ora.omat(Y), float(sc), ora.fptr(Wacc), nW, None)
    if bacc is not None:
        bacc += (sc * (Y.astype(np.float64) * X[:, -1:].astype(np.float64)).sum(0)).astype(F)


@pytest.mark.parametrize("flags", [-1, 0, 1 | 16, 4], ids=["plain-TdnnComponent", "darts-softmax", "darts-gumbel-updatealpha", "darts-uniform-pretrain"])
def test_tdnn_components_through_the_adapter(driver, ora, pkg, tmp_path, flags):
    L = ora.lib()
    rng = np.random.default_rng(50 + flags)
    darts = flags >= 0
    offs = [0, 1, 2] if darts else [-3, 0]
    K, B, nt, Di, Do, steps, lr, temp = len(offs), 8, 20, 48, 64, 3, 0.02, 0.8
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B)
    W = (_rand(rng, Do, K * Di) / np.sqrt(K * Di)).astype(F)
    bias = _rand(rng, (K if darts else 0) + Do) * 0.5
    arrays = {"cfg": [K, Di, Do, rho, flags, temp, lr, steps, float(offs[1] > 0)] + [float(v) for v in ro], "W": W, "bias": bias}
    bx, by = _rand(rng, 5, Di), _rand(rng, 6, Do)
    data = []
    for t in range(steps):
        x = (_rand(rng, rows_in, 5) @ bx + 0.4 * _rand(rng, rows_in, Di)).astype(F)
        dy = (_rand(rng, N, 6) @ by + 0.4 * _rand(rng, N, Do)).astype(F)
        draws = rng.uniform(0.05, 0.95, K + 1).astype(F)
        arrays.update({f"x{t}": x, f"dy{t}": dy, f"draws{t}": draws})
        data.append((x, dy, draws))
    got = driver("tdnn", arrays, tmp_path)
    Dx = K * Di + 1
    ngi, ngo = L.oracle_ng_create(min(20, (Dx + 1) // 2), 4, 2000.0, 4.0), L.oracle_ng_create(min(80, (Do + 1) // 2), 4, 2000.0, 4.0)
    share = 0 if offs[1] > 0 else K - 1
    for t, (x, dy, draws) in enumerate(data):
        coef = eff = None
        if darts:
            coef, eff = np.zeros(K, F), np.zeros(K, F)
            la = np.ascontiguousarray(bias[:K])
            L.oracle_tdnn_darts_coef(ora.fptr(la), K, flags, temp, ora.fptr(np.ascontiguousarray(draws[:K])), float(draws[K]), ora.fptr(coef))
            L.oracle_tdnn_darts_effective_coef(ora.fptr(coef), K, flags, share, ora.fptr(eff))
            assert np.allclose(got[f"memo{t}"].ravel(), np.concatenate([coef, eff]), rtol=1e-5, atol=1e-7)
        real_bias = np.ascontiguousarray(bias[K:] if darts else bias)
        y = np.zeros((N, Do), F)
        # DARTS: bias rows only when offsets[1] > 0 (here: yes); plain: always
        L.oracle_tdnn_propagate(ora.omat(x), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro), ora.fptr(real_bias), ora.fptr(eff) if darts else None, 1, ora.omat(y))
        assert rel_l2(got[f"y{t}"], y) < 2e-5
        dx = np.zeros((rows_in, Di), F)
        L.oracle_tdnn_backprop_data(ora.omat(dy), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro), ora.fptr(eff) if darts else None, ora.omat(dx))
        assert rel_l2(got[f"dx{t}"], dx) < 2e-5
        W_acc, b_acc = np.zeros_like(W), np.zeros_like(bias)
        if darts:
            s = np.zeros(K)
            if not (flags & 4):
                L.oracle_tdnn_darts_tap_dots(ora.omat(x), ora.omat(dy), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro), ora.dptr(s))
            aacc = np.zeros(K, F)
            L.oracle_tdnn_darts_alpha_update(ora.dptr(s), ora.fptr(coef), K, flags, share, temp, lr, ora.fptr(aacc))
            b_acc[:K] = aacc
        X = np.zeros((N, Dx), F)
        L.oracle_tdnn_splice(ora.omat(x), N, Di, K, rho, ora.iptr(ro), ora.fptr(eff) if darts else None, 1, ora.omat(X))
        bview = np.zeros(Do, F)
        _ng_update(L, ora, X, dy, ngi, ngo, lr, W_acc, bview)
        b_acc[(K if darts else 0):] = bview
        assert rel_l2(got[f"W_acc{t}"], W_acc) < 5e-3, (t, rel_l2(got[f"W_acc{t}"], W_acc))
        assert rel_l2(got[f"b_acc{t}"].ravel()[(K if darts else 0):], bview) < 5e-3
        if darts:
            assert rel_l2(got[f"b_acc{t}"].ravel()[:K], b_acc[:K]) < 1e-4 or not np.any(b_acc[:K])
    L.oracle_ng_destroy(ngi)
    L.oracle_ng_destroy(ngo)
    # the same minibatches through the Component CLASS (tdnnf_nnet3_components.h): created by factory name, driven through the
    # virtual Propagate / Backprop with a second instance as the delta-nnet, its RandUniform hook fed the same draws
    cls = driver("tdnn_classes", arrays, tmp_path)
    assert set(cls) == set(got)
    for k in got:
        assert rel_l2(cls[k], got[k]) < 1e-6, (k, rel_l2(cls[k], got[k]))


def test_affine_relu_batchnorm_linear_logsoftmax_through_the_adapter(driver, ora, tmp_path):
    L = ora.lib()
    rng = np.random.default_rng(77)
    N, Di, H, P, steps, lr, repair = 384, 40, 96, 50, 3, 0.05, 1e-3
    Wa, ba, Wl = (_rand(rng, H, Di) / np.sqrt(Di)).astype(F), _rand(rng, H), (_rand(rng, P, H) / np.sqrt(H)).astype(F)
    ba[:4] = -30.0  # dead units: self-repair has something to do
    arrays = {"cfg": [steps, lr, repair], "Wa": Wa, "ba": ba, "Wl": Wl}
    data = []
    for t in range(steps):
        x, d = _rand(rng, N, Di), _rand(rng, N, P)
        arrays.update({f"x{t}": x, f"d{t}": d})
        data.append((x, d))
    got = driver("stack", arrays, tmp_path)
    nga = (L.oracle_ng_create(min(20, (Di + 2) // 2), 4, 2000.0, 4.0), L.oracle_ng_create(min(80, (H + 1) // 2), 4, 2000.0, 4.0))
    ngl = (L.oracle_ng_create(min(20, (H + 1) // 2), 4, 2000.0, 4.0), L.oracle_ng_create(min(80, (P + 1) // 2), 4, 2000.0, 4.0))
    vs, ds, cnt = np.zeros(H), np.zeros(H), C.c_double(0.0)
    bcnt, bsum, bsq = C.c_double(0.0), np.zeros(H), np.zeros(H)
    for t, (x, d) in enumerate(data):
        a = np.zeros((N, H), F)
        L.oracle_affine_propagate(ora.omat(x), ora.fptr(Wa), Di, ora.fptr(ba), H, ora.omat(a))
        r = np.maximum(a, 0)
        L.oracle_relu_store_stats(ora.omat(r), ora.dptr(vs), ora.dptr(ds), C.byref(cnt))
        z, memo = np.zeros_like(r), np.zeros((5, H), F)
        L.oracle_batchnorm_propagate(ora.omat(r), 1e-3, 1.0, ora.omat(z), ora.fptr(memo))
        L.oracle_batchnorm_store_stats(ora.fptr(memo), H, N, C.byref(bcnt), ora.dptr(bsum), ora.dptr(bsq))
        l, lsm = np.zeros((N, P), F), np.zeros((N, P), F)
        L.oracle_affine_propagate(ora.omat(z), ora.fptr(Wl), H, None, P, ora.omat(l))
        L.oracle_log_softmax_propagate(ora.omat(l), ora.omat(lsm))
        assert rel_l2(got[f"z{t}"], z) < 2e-5 and rel_l2(got[f"lsm{t}"], lsm) < 2e-5
        dl = np.zeros_like(d)
        L.oracle_log_softmax_backprop(ora.omat(lsm), ora.omat(d), ora.omat(dl))
        dz = np.zeros((N, H), F)
        L.oracle_affine_backprop(ora.omat(dl), ora.fptr(Wl), H, H, ora.omat(dz))
        Wl_acc = np.zeros_like(Wl)
        _ng_update(L, ora, z.copy(), dl, ngl[0], ngl[1], lr, Wl_acc, None)
        dr = np.zeros_like(dz)
        L.oracle_batchnorm_backprop(ora.omat(z), ora.omat(dz), 1.0, ora.fptr(memo), ora.omat(dr))
        da = ((r > 0) * dr).astype(F)
        L.oracle_relu_repair(ora.dptr(ds), cnt.value, H, repair, 0.05, 0.95, ora.omat(da))
        assert rel_l2(got[f"da{t}"], da) < 5e-5
        dx = np.zeros((N, Di), F)
        L.oracle_affine_backprop(ora.omat(da), ora.fptr(Wa), Di, Di, ora.omat(dx))
        assert rel_l2(got[f"dx{t}"], dx) < 5e-5
        Wa_acc, ba_acc = np.zeros_like(Wa), np.zeros(H, F)
        X = np.ones((N, Di + 1), F)
        X[:, :Di] = x
        _ng_update(L, ora, X, da, nga[0], nga[1], lr, Wa_acc, ba_acc)
        assert rel_l2(got[f"Wl_acc{t}"], Wl_acc) < 5e-3, t
        assert rel_l2(got[f"Wa_acc{t}"], Wa_acc) < 5e-3, t
        assert rel_l2(got[f"ba_acc{t}"].ravel(), ba_acc) < 5e-3, t
    assert rel_l2(got["relu_stats"].ravel(), np.concatenate([[cnt.value], vs, ds])) < 1e-5
    assert rel_l2(got["bn_stats"].ravel(), np.concatenate([[bcnt.value], bsum, bsq])) < 1e-5
    for g in nga + ngl:
        L.oracle_ng_destroy(g)
    # the same stack as Component CLASSES created by factory name and driven through the virtual interface (StoreStats included)
    cls = driver("stack_classes", arrays, tmp_path)
    assert set(cls) == set(got)
    for k in got:
        assert rel_l2(cls[k], got[k]) < 1e-6, (k, rel_l2(cls[k], got[k]))


def test_darts_mixing_components_through_the_adapter(driver, ora, tmp_path):
    L = ora.lib()
    rng = np.random.default_rng(91)
    N, Cn, d, temp, fscale, lr = 600, 8, 40, 0.7, 2.0, 0.1
    alpha, u = _rand(rng, Cn) * 0.6, rng.uniform(0.05, 0.95, Cn).astype(F)
    flops = -np.cumsum([25, 25, 30, 20, 20, 40, 40, 40]).astype(F)
    lin, dmasked, sk, dP = _rand(rng, N, d), _rand(rng, N, d), rng.uniform(0.1, 1.0, (N, 1)).astype(F), _rand(rng, N, Cn)
    draw = np.asarray([0.4], F)
    got = driver("mixing", {"cfg": [N, Cn, d, 0, fscale, temp, lr], "alpha": alpha, "u": u, "flops": flops, "draw": draw, "lin": lin,
                            "dmasked": dmasked, "sk": sk, "dP": dP}, tmp_path)
    A, P = np.zeros((N, Cn), F), np.zeros((N, Cn), F)
    L.oracle_constant_function_propagate(ora.fptr(alpha), ora.omat(A))
    L.oracle_softmax_flops_propagate(ora.omat(A), ora.fptr(u), temp, ora.omat(P))
    assert rel_l2(got["P"], P) < 1e-5
    cop = np.zeros((N, d), F)
    L.oracle_copyn_propagate(ora.omat(sk), 1.0, ora.omat(cop))
    ew_in = np.ascontiguousarray(np.concatenate([cop, lin], axis=1))
    masked = np.zeros((N, d), F)
    L.oracle_elementwise_product_propagate(ora.omat(ew_in), d, ora.omat(masked))
    assert rel_l2(got["masked"], masked) < 1e-6
    d_ew = np.zeros((N, 2 * d), F)
    L.oracle_elementwise_product_backprop(ora.omat(ew_in), ora.omat(dmasked), d, ora.omat(d_ew))
    assert rel_l2(got["d_ew"], d_ew) < 1e-6
    d_sk, d_cop = np.zeros((N, 1), F), np.ascontiguousarray(d_ew[:, :d])  # (named: omat() keeps no reference)
    L.oracle_copyn_backprop(ora.omat(d_cop), 1.0, ora.omat(d_sk))
    assert rel_l2(got["d_sk"], d_sk) < 1e-5
    dP2, dA = dP.copy(), np.zeros((N, Cn), F)
    L.oracle_softmax_flops_backprop(ora.omat(P), ora.omat(dP2), fscale, ora.fptr(flops), Cn, temp, ora.omat(dA))
    assert rel_l2(got["dP_after"], dP2) < 1e-6 and rel_l2(got["dA"], dA) < 1e-4
    acc = np.zeros(Cn, F)
    L.oracle_constant_function_backprop(ora.omat(dA), lr, ora.fptr(acc))
    assert rel_l2(got["alpha_acc"].ravel(), acc) < 1e-4
    oh = np.zeros((N, Cn), F)
    L.oracle_onehot_propagate(float(draw[0]), ora.omat(oh))
    assert np.array_equal(got["onehot"], oh) and oh.sum() == N
    assert rel_l2(got["onehot_acc"].ravel(), lr * dP2.astype(np.float64).sum(0)) < 1e-5


@pytest.mark.parametrize("continuous", [1, 0], ids=["continuous-mask", "binary-mask"])
def test_remaining_factory_names_through_the_component_classes(driver, ora, tmp_path, continuous):
    """AffineComponent, FixedAffineComponent, NoOpComponent, GeneralDropoutComponent, FlopsConstraintComponent, GumbelSoftmaxComponent
    (/root/reference/src/nnet3/nnet-component-itf.cc:136,150,156,194,260,266) created by factory name and driven through the virtual
    interface; every output against the oracle."""
    L = ora.lib()
    rng = np.random.default_rng(123 + continuous)
    S, nt, Di, H, Cn, p, temp, fscale, lr, nbs = 6, 30, 24, 40, 5, 0.3, 0.6, 1.5, 0.05, 0.7
    N = S * nt
    Wf, bf = (_rand(rng, H, Di) / np.sqrt(Di)).astype(F), _rand(rng, H)
    Wa, ba = (_rand(rng, H, H) / np.sqrt(H)).astype(F), _rand(rng, H)
    flops = -np.cumsum(rng.integers(10, 40, Cn)).astype(F)
    x, d, d0_init, dx_init = _rand(rng, N, Di), _rand(rng, N, H), _rand(rng, N, H), _rand(rng, N, Di)
    xs, dgs = _rand(rng, N, Cn), _rand(rng, N, Cn)
    draws = rng.uniform(0.02, 0.98, S * H + Cn).astype(F)
    got = driver("rest_classes", {"cfg": [S, p, continuous, temp, fscale, lr, nbs], "Wf": Wf, "bf": bf, "Wa": Wa, "ba": ba, "flops": flops, "x": x, "d": d,
                                  "d0_init": d0_init, "dx_init": dx_init, "xs": xs, "dgs": dgs, "draws": draws}, tmp_path)
    y0, y1 = np.zeros((N, H), F), np.zeros((N, H), F)
    L.oracle_affine_propagate(ora.omat(x), ora.fptr(Wf), Di, ora.fptr(bf), H, ora.omat(y0))
    L.oracle_affine_propagate(ora.omat(y0), ora.fptr(Wa), H, ora.fptr(ba), H, ora.omat(y1))
    assert rel_l2(got["y0"], y0) < 2e-5 and rel_l2(got["y1"], y1) < 2e-5
    assert np.array_equal(got["y2"], got["y1"])  # NoOp: a copy
    u = draws[:S * H].reshape(S, H)
    mask = (u * F(p * 4.0) + F(1.0 - 2.0 * p)).astype(F) if continuous else ((u + F(-p) > 0).astype(F) * F(1.0 / (1.0 - p))).astype(F)
    y3 = np.zeros_like(y1)
    L.oracle_general_dropout_propagate(ora.omat(np.ascontiguousarray(got["y2"])), ora.fptr(np.ascontiguousarray(mask)), S, ora.omat(y3))
    assert rel_l2(got["y3"], y3) < 1e-6
    if not continuous:
        assert 0.15 < (got["y3"] == 0).mean() < 0.45  # a proportion p of the units is dropped
    assert np.array_equal(got["y3_test_mode"], got["y2"])
    d2 = np.zeros_like(d)
    L.oracle_general_dropout_propagate(ora.omat(d), ora.fptr(np.ascontiguousarray(mask)), S, ora.omat(d2))
    assert rel_l2(got["d2"], d2) < 1e-6
    assert rel_l2(got["d1"], d2 * F(nbs)) < 1e-6
    d1 = np.ascontiguousarray(got["d1"])
    back = np.zeros((N, H), F)
    L.oracle_affine_backprop(ora.omat(d1), ora.fptr(Wa), H, H, ora.omat(back))
    assert rel_l2(got["d0"], d0_init + back) < 2e-5  # kBackpropAdds: added to what in_deriv held
    Wa_acc, ba_acc = np.zeros_like(Wa), np.zeros(H, F)
    L.oracle_affine_update_simple(ora.omat(np.ascontiguousarray(got["y0"])), ora.omat(d1), lr, ora.fptr(Wa_acc), H, ora.fptr(ba_acc))
    assert rel_l2(got["Wa_acc"], Wa_acc) < 2e-5 and rel_l2(got["ba_acc"].ravel(), ba_acc) < 2e-5  # UpdateSimple: no preconditioning
    d0 = np.ascontiguousarray(got["d0"])
    backx = np.zeros((N, Di), F)
    L.oracle_affine_backprop(ora.omat(d0), ora.fptr(Wf), Di, Di, ora.omat(backx))
    assert rel_l2(got["dx"], dx_init + backx) < 2e-5
    P = np.zeros((N, Cn), F)
    L.oracle_softmax_flops_propagate(ora.omat(xs), ora.fptr(np.ascontiguousarray(draws[S * H:])), temp, ora.omat(P))
    assert rel_l2(got["P"], P) < 1e-5 and np.array_equal(got["F"], got["P"])
    dg, dxs = dgs.copy(), np.zeros((N, Cn), F)
    L.oracle_softmax_flops_backprop(ora.omat(P), ora.omat(dg), 0.0, None, 0, temp, ora.omat(dxs))
    assert rel_l2(got["dxs"], dxs) < 1e-4 and np.array_equal(got["dgs_after"], dgs)
    dP = np.zeros((N, Cn), F)
    L.oracle_flops_constraint_backprop(ora.fptr(flops), fscale, N, Cn, ora.omat(dP))
    assert rel_l2(got["dP"], dP) < 1e-6


def surface_lines(tmp_path):
    """One config line per factory name: the lines the reference's own scripts emit (tests/golden/r01_configs_golden.json holds the outputs of
    generate_config.py / generate_bottleneckCB8share_onehottrain_config.py / add_flopsconstraint.py run from /root/reference, make_configs_golden.py)
    for the 16 types they instantiate, written by hand after the reference's InitFromConfig for the four they do not."""
    import json
    import re
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "r01_configs_golden.json")))
    texts = []
    for key in ("darts", "bottleneck", "bottleneck_offsets", "flops"):
        items = gold[key] if isinstance(gold[key], list) else [gold[key]]
        for it in items:
            out = it.get("out", it)
            texts += [v for v in out.values() if isinstance(v, str) and "component name=" in v]
    first = {}
    for t in texts:
        for ln in t.split("\n"):
            if ln.startswith("component name="):
                first.setdefault(re.search(r"type=(\S+)", ln).group(1), ln[len("component "):])
    assert len(first) == 16, sorted(first)
    # the lda matrix the recipes point FixedAffineComponent at (matrix=configs/lda.mat): a Kaldi text matrix [W | b]
    lda = tmp_path / "lda.mat"
    m = np.random.default_rng(0).standard_normal((220, 221)).astype(F) * 0.1
    lda.write_text(" [\n" + "\n".join("  " + " ".join("%.9g" % x for x in row) for row in m) + " ]\n")
    first["FixedAffineComponent"] = first["FixedAffineComponent"].replace("configs/lda.mat", str(lda))
    first["AffineComponent"] = "name=a type=AffineComponent input-dim=64 output-dim=32 param-stddev=0.1 bias-stddev=0.5 max-change=0.75"
    first["BatchNormTestComponent"] = "name=b type=BatchNormTestComponent"
    first["FlopsConstraintComponent"] = "name=f type=FlopsConstraintComponent input-dim=8 output-dim=8 flops=25,50,80,100,120,160,200,240 scale=0.5"
    first["GumbelSoftmaxComponent"] = "name=g type=GumbelSoftmaxComponent dim=8 temp-proportion=0.7"
    order = sorted(first, key=lambda t: (t == "BatchNormTestComponent", t))  # (the test-mode component is made from the BatchNormComponent's text)
    return [(t, first[t]) for t in order]


def test_every_component_class_through_its_whole_virtual_surface(pkg, tmp_path):
    """for each of the 20 factory names NewComponentOfType -> InitFromConfig (the reference scripts' own config line) ->
    Write (text, binary) -> ReadNew -> byte-identical Write -> Copy -> identical Propagate on the GPU; Scale / Add / DotProduct /
    NumParameters / Vectorize / UnVectorize / PerturbParams / FreezeNaturalGradient for the updatable ones; PrecomputeIndexes /
    ReorderIndexes / GetInputIndexes for the Tdnn classes; the cv-update sed edits on the text form (tests/surface_driver.cc)."""
    exe = tmp_path / "surface_driver"
    lib_dir = os.path.dirname(pkg.hipabi.LIB_PATH)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "surface_driver.cc"),
                           "-L", lib_dir, "-ltdnnf_hip", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)])
    rows = surface_lines(tmp_path)
    assert len(rows) == 20
    lines = tmp_path / "lines.txt"
    lines.write_text("".join("%s\t%s\n" % r for r in rows))
    p = subprocess.run([str(exe), str(lines)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-3000:]
    ok = [ln.split()[1] for ln in p.stdout.splitlines() if ln.startswith("OK ") and len(ln.split()) == 2]
    assert sorted(ok) == sorted(t for t, _ in rows), p.stdout
    assert "OK cv-update edits" in p.stdout and "DONE 20" in p.stdout
