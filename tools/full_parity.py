#!/usr/bin/env python3
"""One full-ensemble parity pass of a BASELINE configuration (GPU box): the device result of EVERY trajectory against the CPU
oracle's, not a sample.  configs[1] = 10 000 x 24 h is ~9 min of the box's host threads (the oracle is scalar C, one trajectory per
thread).  Writes gpurun_out/<out>.json: max / median / p99 |dr|, |dv|, the count of trajectories whose accepted-step / evaluation
counts differ, status words, an input digest and the kernel source stamp, so that the result can be tied to a tree.
usage: tools/full_parity.py <config 2|3|5> [n or 0] [hours or 0] [out name] [chunk]
TEST INFRASTRUCTURE (it drives the oracle): not part of the product path."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import nyx_amd as nx  # noqa: E402
import bench  # noqa: E402
import oracle_lib  # noqa: E402

cfg_id = int(sys.argv[1])
w = bench.workload(cfg_id)
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 0) or w["n"]
hours = (float(sys.argv[3]) if len(sys.argv) > 3 else 0) or w["hours"]
name = sys.argv[4] if len(sys.argv) > 4 else f"full_parity_cfg{cfg_id}"
chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
assert not w["stm"], "covariance-mapping configurations are compared by tests/test_gpu_predict.py"
compiled = w["prop"].compile(w["almanac"], w["central"], stm=False)
b = w["batch"](n, seed=0)
dur = int(round(hours * 3600)) * nx.NS_PER_S
h = hashlib.sha256()
h.update(np.ascontiguousarray(b.rv()).tobytes())
h.update(np.ascontiguousarray(b.epoch_ns).tobytes())
input_digest = h.hexdigest()[:16]

ctx = nx.GpuContext(compiled)
out, st = ctx.propagate(b, dur)
out, st = ctx.propagate(b, dur)
kernel_ms = ctx.last_kernel_ms()
helpers = ctx.last_coop_helpers()
print(f"device: config {cfg_id} n={n} hours={hours:g}: {kernel_ms:.3f} ms, helpers {helpers}, bad {(st.status != 0).sum()}", flush=True)

threads = os.cpu_count() or 1
ref_rv = np.zeros((n, 6))
ref_epoch = np.zeros(n, dtype=np.int64)
ref_acc = np.zeros(n, dtype=np.int64)
ref_rej = np.zeros(n, dtype=np.int64)
ref_ev = np.zeros(n, dtype=np.int64)
ref_status = np.zeros(n, dtype=np.int32)
t0 = time.time()
for lo in range(0, n, chunk):
    hi = min(n, lo + chunk)
    r, rs = oracle_lib.propagate(compiled, b.slice(lo, hi), dur, n_threads=threads)
    ref_rv[lo:hi] = r.rv()
    ref_epoch[lo:hi] = r.epoch_ns
    ref_acc[lo:hi] = rs.n_accepted
    ref_rej[lo:hi] = rs.n_rejected
    ref_ev[lo:hi] = rs.n_evals
    ref_status[lo:hi] = rs.status
    print(f"oracle: {hi} of {n} in {time.time() - t0:.0f} s", flush=True)
oracle_s = time.time() - t0

d = out.rv() - ref_rv
dr = np.linalg.norm(d[:, :3], axis=1) * 1e3      # km -> m
dv = np.linalg.norm(d[:, 3:], axis=1) * 1e3      # km/s -> m/s
try:
    stamp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_stamp.py")], capture_output=True, text=True).stdout.strip()
except Exception:
    stamp = None
res = {
    "what": "device (HIP, through the C-ABI) vs CPU oracle on EVERY trajectory of the configuration",
    "config": cfg_id, "workload": w["label"](n, hours), "n": n, "hours": hours,
    "input_digest": input_digest, "kernel_source_stamp": stamp,
    "device_kernel_ms": kernel_ms, "coop_helpers": helpers,
    "oracle_threads": threads, "oracle_wall_s": round(oracle_s, 1), "oracle_traj_per_s": round(n / oracle_s, 2),
    "status_bad_device": int((st.status != 0).sum()), "status_bad_oracle": int((ref_status != 0).sum()),
    "epoch_mismatch": int((out.epoch_ns != ref_epoch).sum()),
    "dr_m": {"max": float(dr.max()), "p99": float(np.percentile(dr, 99)), "median": float(np.median(dr)), "argmax": int(dr.argmax())},
    "dv_m_per_s": {"max": float(dv.max()), "p99": float(np.percentile(dv, 99)), "median": float(np.median(dv)), "argmax": int(dv.argmax())},
    "bar": {"dr_m": 1.0, "dv_m_per_s": 1e-3},
    "within_bar": int(((dr <= 1.0) & (dv <= 1e-3)).sum()),
    "n_accepted_differs": int((st.n_accepted != ref_acc).sum()), "n_rejected_differs": int((st.n_rejected != ref_rej).sum()),
    "n_evals_differs": int((st.n_evals != ref_ev).sum()),
    "max_abs_n_accepted_diff": int(np.abs(st.n_accepted - ref_acc).max()),
    "device_evals": int(st.n_evals.sum()), "oracle_evals": int(ref_ev.sum()),
}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", name + ".json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res), flush=True)
