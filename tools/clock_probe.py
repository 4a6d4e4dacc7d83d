#!/usr/bin/env python3
"""Engine clock and power while one GEMM shape runs in a loop (rocm-smi sampled from a thread).
usage: clock_probe.py [seconds]"""
import ctypes as C, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); abi = pkg.hipabi; lib = abi.load()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out.strip())
        except Exception as e:
            samples.append("ERR %r" % e)
        time.sleep(0.3)
def loop(name, fn, flops):
    global stop, samples
    samples, stop = [], False
    th = threading.Thread(target=sampler); th.start()
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50): fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop = True; th.join()
    ms = e0.elapsed_time(e1) / n
    print("%s: %.1f us %.1f TF" % (name, ms * 1e3, flops / ms / 1e9))
    import json, re
    for s in samples[1:-1][:6]:
        try:
            j = json.loads(s); c = j[sorted(j)[0]]
            print("   ", {k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()})
        except Exception:
            print("   ", s[:200])
B = 128
def shape(offs, nt, Di, Do):
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B)
    K = len(offs)
    x = torch.randn(rows_in, Di, device="cuda"); W = torch.randn(Do, K * Di, device="cuda") / (K * Di) ** 0.5
    b = torch.randn(Do, device="cuda"); y = torch.zeros(N, Do, device="cuda")
    ix = abi.indexes(rho, ro); s = abi.stream()
    return (lambda: abi.check(lib.tdnnf_tdnn_propagate(C.byref(ix), abi.pmat(x), abi.ptr(W), K * Di, Do, Di, abi.ptr(b), None, 1, abi.pmat(y), s))), 2.0 * N * K * Di * Do
fn, fl = shape([-1, 0], 1564, 1536, 160); loop("linear fwd 128x160 K3072", fn, fl)
fn, fl = shape([0, 1], 1563, 160, 1536); loop("affine fwd 128x128 K320", fn, fl)
fn, fl = shape([-1, 0], 512, 1536, 1536); loop("probe128 longK", fn, fl)
