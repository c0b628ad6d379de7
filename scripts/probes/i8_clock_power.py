"""Shader clock and socket power while the int8 product kernel (or the register-only MFMA loop of i8_mfma_peak.hip) runs
back to back: rocm-smi sampled from a side thread.  Answers whether the 5-plane kernel is power/clock-limited.
    python3 scripts/probes/i8_clock_power.py [seconds]
"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402
from modegpt_amd import engine, ops  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
dev = torch.device("cuda:0")
shape = engine.SHAPES["llama-3-8b"]
x = engine.make_activation_batch(shape, 16 * 2048, seed=1, device=dev)["h"]
sigma = torch.zeros(x.shape[-1], x.shape[-1], dtype=torch.float64, device=dev)
samples, stop = [], False


def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "-d", "0", "-c", "-P", "--csv"], capture_output=True, text=True).stdout
        samples.append((time.time(), out.strip().replace("\n", " | ")))
        time.sleep(0.3)


x0 = torch.zeros_like(x)           # all digit planes zero: no operand toggling in LDS / MFMA
x1 = torch.ones_like(x)            # top digit 64 everywhere, lower planes zero
for label, fn in (("idle", lambda: time.sleep(0.05)), ("i8 route", lambda: ops.cov_accum_i8(sigma, x)),
                  ("i8 route, all-zero input", lambda: ops.cov_accum_i8(sigma, x0)),
                  ("i8 route, all-ones input", lambda: ops.cov_accum_i8(sigma, x1)),
                  ("f64 route", lambda: ops.cov_accum(sigma, x))):
    samples.clear()
    stop = False
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < (2.0 if label == "idle" else secs):
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.time() - t0
    stop = True
    th.join()
    print(f"== {label}: {n} calls in {dt:.2f} s ({1e3 * dt / n:.1f} ms per call)")
    for t, s in samples[:: max(1, len(samples) // 5)]:
        f = s.split("|")[-1].split(",")
        print(f"  t={t - t0:5.2f}s  sclk {f[5]}  power {f[-1]} W")
