#!/usr/bin/env python3
"""Golden vectors for tdnn-f_nas_amd/derive.py: runs the reference's own child-derivation scripts
(local/chain_NAS/scripts/generate_top_list.py, generate_top_list_bottleneckdim.py, generate_optimal_stride.py) IN THIS
CONTAINER on synthetic inputs and records inputs + outputs as data in r01_derive_golden.json.  The scripts are executed
from /root/reference as they lie; nothing of their text is stored.  Usage: python tests/golden/make_derive_golden.py"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/local/chain_NAS/scripts"
L = 14  # tdnnf layers of the 'tdnn' model type (28 searched components / 14 alpha vectors)


def template_config(strides, bottleneck=160, hidden=1536):
    """final.config / ref.config lines in the form xconfig emits them for tdnnf-layer (composite_layers.py:135-215):
    only the shape of the TdnnComponent lines matters to the scripts."""
    out = ["input-node name=ivector dim=100", "input-node name=input dim=40",
           "component name=tdnn1.affine type=NaturalGradientAffineComponent input-dim=220 output-dim=%d  max-change=0.75 l2-regularize=0.01" % hidden]
    for i, s in enumerate(strides):
        n = "tdnnf%d" % (i + 2)
        o1, o2 = ("%d,0" % -s, "0,%d" % s) if s else ("0", "0")
        out.append("component name=%s.linear type=TdnnComponent input-dim=%d output-dim=%d l2-regularize=0.01 max-change=0.75 use-bias=false "
                   "time-offsets=%s orthonormal-constraint=-1.0" % (n, hidden, bottleneck, o1))
        out.append("component-node name=%s.linear component=%s.linear input=%s" % (n, n, "tdnn1.dropout" if i == 0 else "tdnnf%d.noop" % (i + 1)))
        out.append("component name=%s.affine type=TdnnComponent input-dim=%d output-dim=%d l2-regularize=0.01 max-change=0.75 time-offsets=%s"
                   % (n, bottleneck, hidden, o2))
        out.append("component-node name=%s.affine component=%s.affine input=%s.linear" % (n, n, n))
        out.append("component name=%s.relu type=RectifiedLinearComponent dim=%d self-repair-scale=1e-05" % (n, hidden))
    out.append("component name=prefinal-l type=LinearComponent input-dim=%d output-dim=256 l2-regularize=0.01 orthonormal-constraint=-1.0" % hidden)
    return out


def offset_model_text(alpha, K):
    """The lines of a text model generate_top_list.py looks at: the <BiasParams> rows; the 3rd..30th hold the searched
    components' K logits followed by the real bias (TdnnDARTSV3Component::Write, nnet-tdnn-component.cc:659-700)."""
    lines = ["<Nnet3> ", ""]
    rng = np.random.default_rng(1)
    def bias_line(vals):
        return "<BiasParams>  [ " + " ".join("%.7g" % v for v in vals) + " ]"
    for _ in range(2):  # two components with a bias in front of the searched ones (lda, tdnn1.affine)
        lines.append(bias_line(rng.standard_normal(5)))
    for row in alpha:
        lines.append("<ComponentName> x <TdnnDARTSV3Component> <MaxChange> 0.75 <LinearParams>  [")
        lines.append("  0.1 0.2 ]")
        lines.append(bias_line(list(row) + list(rng.standard_normal(3))))
    for _ in range(3):  # components after them
        lines.append(bias_line(rng.standard_normal(4)))
    return lines


def bottleneck_model_text(alpha):
    """The lines generate_top_list_bottleneckdim.py looks at: '<ComponentName> tdnnfN.alpha <ConstantFunctionComponent> ... [ v ]'"""
    lines = ["<Nnet3> "]
    for i, row in enumerate(alpha):
        lines.append("<ComponentName> tdnnf%d.alpha <ConstantFunctionComponent> <MaxChange> 0 <IsUpdatable> T <UseNaturalGradient> F <Output>  [ %s ]"
                     % (i + 2, " ".join("%.7g" % v for v in row)))
        lines.append("<ComponentName> tdnnf%d.linear <TdnnComponent> <MaxChange> 0.75" % (i + 2))
    return lines


def run(script, args, files):
    tmp = tempfile.mkdtemp(dir=HERE)
    try:
        parent, cfg = os.path.join(tmp, "parent"), os.path.join(tmp, "cfg")
        os.makedirs(parent)
        os.makedirs(cfg)
        for rel, lines in files.items():
            with open(os.path.join(tmp, rel), "w") as f:
                f.write("\n".join(lines) + "\n")
        a = [x.replace("@P", parent).replace("@C", cfg + "/") for x in args]
        r = subprocess.run([sys.executable, "-W", "ignore", os.path.join(REF, script)] + a, capture_output=True, text=True, cwd=tmp)
        out = {"returncode": r.returncode, "stdout": r.stdout.splitlines()}
        if r.returncode != 0:
            out["error"] = r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ""
        for name in ("final.config", "ref.config", "arch.txt"):
            p = os.path.join(cfg, name)
            if os.path.exists(p) and r.returncode == 0:
                out[name] = open(p).read()
        return out
    finally:
        shutil.rmtree(tmp)


def main():
    cases = []
    rng = np.random.default_rng(7)
    strides = [6] * L
    tmpl = template_config(strides)
    ref_tmpl = [l for l in tmpl if "component-node" not in l]
    for K, child_type, top_id, scale in [(7, "top", 1, 1.0), (7, "top", 3, 1.0), (7, "last", 2, 1.0), (4, "top", 10, 0.3), (7, "top", 1, 0.0)]:
        alpha = (rng.standard_normal((2 * L, K)) * scale).astype(np.float32)
        mdl = offset_model_text(alpha, K)
        out = run("generate_top_list.py", ["@P", child_type, str(top_id), "@C", str(K), "tdnn"],
                  {"parent/final_txt.mdl": mdl, "cfg/final.config_temp": tmpl, "cfg/ref.config_temp": ref_tmpl})
        cases.append({"kind": "offset", "K": K, "child_type": child_type, "top_id": top_id, "model": mdl, "out": out})
    dims = [25, 50, 80, 100, 120, 160, 200, 240]
    for child_type, top_id, scale in [("top", 1, 1.0), ("top", 4, 1.0), ("last", 1, 2.0), ("top", 2, 0.0)]:
        alpha = (rng.standard_normal((L, 8)) * scale).astype(np.float32)
        mdl = bottleneck_model_text(alpha)
        out = run("generate_top_list_bottleneckdim.py", ["@P", child_type, str(top_id), "@C", "8", "tdnn"],
                  {"parent/final_txt.mdl": mdl, "cfg/final.config_temp": tmpl, "cfg/ref.config_temp": ref_tmpl})
        cases.append({"kind": "bottleneck", "dims": dims, "child_type": child_type, "top_id": top_id, "model": mdl, "out": out})
    offs = [int(v) for v in np.where(np.arange(2 * L) % 2 == 0, -rng.integers(0, 7, 2 * L), rng.integers(0, 7, 2 * L))]
    out = run("generate_optimal_stride.py", ["@C"] + [str(v) for v in offs], {"cfg/final.config_temp": tmpl, "cfg/ref.config_temp": ref_tmpl})
    cases.append({"kind": "optimal_stride", "offsets": offs, "out": out})
    with open(os.path.join(HERE, "r01_derive_golden.json"), "w") as f:
        json.dump({"final_temp": tmpl, "ref_temp": ref_tmpl, "cases": cases}, f, indent=0)
    for c in cases:
        print(c["kind"], c.get("child_type"), c.get("top_id"), "rc", c["out"]["returncode"], c["out"].get("error", ""), (c["out"].get("arch.txt") or "").strip()[:80])


if __name__ == "__main__":
    main()
