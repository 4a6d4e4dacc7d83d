#!/usr/bin/env python3
"""Golden vectors for tdnn-f_nas_amd/configs.py: runs the reference's config-rewriting scripts
(local/chain_NAS/scripts/generate_config.py, generate_bottleneckCB8share_onehottrain_config.py,
generate_optimal_context_offset_bottleneckCB8share_onehottrain_config.py, add_flopsconstraint.py,
bottleneckdim_search_top_model_size.py) IN THIS CONTAINER on templates written by configs.final_config() and records
inputs + outputs as data in r01_configs_golden.json.  The scripts are executed from /root/reference as they lie;
nothing of their text is stored.  Usage: python tests/golden/make_configs_golden.py"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import __graft_entry__ as ge  # noqa: E402
from make_derive_golden import bottleneck_model_text  # noqa: E402

REF = "/root/reference/local/chain_NAS/scripts"
cfgs = ge.load_package().configs


def run(script, args, files, outputs):
    tmp = tempfile.mkdtemp(dir=HERE)
    try:
        for rel, lines in files.items():
            os.makedirs(os.path.dirname(os.path.join(tmp, rel)), exist_ok=True)
            with open(os.path.join(tmp, rel), "w") as f:
                f.write("\n".join(lines) + "\n")
        a = [x.replace("@T", tmp) for x in args]
        r = subprocess.run([sys.executable, "-W", "ignore", os.path.join(REF, script)] + a, capture_output=True, text=True, cwd=tmp)
        out = {"returncode": r.returncode}
        if r.returncode != 0:
            out["error"] = r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ""
        for rel in outputs:
            p = os.path.join(tmp, rel)
            if os.path.exists(p):
                out[os.path.basename(rel)] = open(p).read()
        return out
    finally:
        shutil.rmtree(tmp)


def main():
    G = {}
    # offset supernet: tdnnfdartsv3-layer template with the pretrain flags (run_TDNN_DARTSV3_fbk_stride_pretrain.sh:124), time-stride 6
    flags = {"use-gumbel": "false", "use-entropy": "false", "free-select": "false", "update-alpha": "false", "update-theta": "true", "uniform-sample": "true"}
    darts_t = cfgs.final_config(strides=[6] * 14, darts=flags)
    G["darts"] = []
    for K in (7, 4):
        out = run("generate_config.py", [str(K), "@T/"], {"final.config_temp": darts_t, "ref.config_temp": cfgs.ref_config(darts_t)}, ["final.config", "ref.config"])
        G["darts"].append({"K": K, "flags": flags, "out": out})
    # bottleneck supernet on the 7q net
    plain = cfgs.final_config()
    G["bottleneck"] = run("generate_bottleneckCB8share_onehottrain_config.py", ["@T"], {"final_ori.config": plain}, ["final.config"])
    rng = np.random.default_rng(11)
    offs = [int(v) for v in np.where(np.arange(28) % 2 == 0, -rng.integers(0, 7, 28), rng.integers(0, 7, 28))]
    G["bottleneck_offsets"] = {"offsets": offs,
                               "out": run("generate_optimal_context_offset_bottleneckCB8share_onehottrain_config.py", ["@T", "tdnn"] + [str(v) for v in offs],
                                          {"final_ori.config": plain, "ref_ori.config": cfgs.ref_config(plain)}, ["final.config", "ref.config"])}
    G["flops"] = []
    for use_gumbel, coef in (("true", "0.05"), ("false", "2")):
        G["flops"].append({"use_gumbel": use_gumbel, "coef": coef,
                           "out": run("add_flopsconstraint.py", ["@T", use_gumbel, coef, "tdnn"], {}, ["change.config"])})
    G["sizes"] = []
    for child_type in ("top", "last"):
        alpha = (rng.standard_normal((14, 8))).astype(np.float32)
        mdl = bottleneck_model_text(alpha)
        G["sizes"].append({"child_type": child_type, "model": mdl,
                           "out": run("bottleneckdim_search_top_model_size.py", ["@T", child_type, "tdnn"], {"final_txt.mdl": mdl, "configs/.keep": []}, ["configs/arch.txt"])})
    with open(os.path.join(HERE, "r01_configs_golden.json"), "w") as f:
        json.dump(G, f, indent=0)
    for k, v in G.items():
        for c in (v if isinstance(v, list) else [v]):
            o = c.get("out", c)
            print(k, "rc", o["returncode"], o.get("error", ""), [kk for kk in o if kk not in ("returncode", "error")])


if __name__ == "__main__":
    main()
