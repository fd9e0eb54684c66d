#!/usr/bin/env python3
"""One full-ensemble parity pass of a BASELINE configuration (GPU box): the device result of EVERY trajectory against the CPU
oracle's, not a sample.  configs[1] = 10 000 x 24 h is ~9 min of the box's host threads (the oracle is scalar C, one trajectory per
thread).  Writes gpurun_out/<out>.json: max / median / p99 |dr|, |dv|, the count of trajectories whose accepted-step / evaluation
counts differ, status words, an input digest and the kernel source stamp, so that the result can be tied to a tree.
usage: tools/full_parity.py <config 2|3|5> [n or 0] [hours or 0] [out name] [chunk] [first] [count]
(first / count: the oracle runs over that range of the SAME seeded batch - a pass longer than one GPU call is made in pieces; the file is
rewritten after every chunk with the statistics of the trajectories compared so far, so that a call cut off by its limit leaves them)
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
first = int(sys.argv[6]) if len(sys.argv) > 6 else 0
count = int(sys.argv[7]) if len(sys.argv) > 7 else 0
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
try:
    stamp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_stamp.py")], capture_output=True, text=True).stdout.strip()
except Exception:
    stamp = None
last = min(n, first + count) if count > 0 else n
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
t0 = time.time()
res = None
for lo in range(first, last, chunk):
    hi = min(last, lo + chunk)
    r, rs = oracle_lib.propagate(compiled, b.slice(lo, hi), dur, n_threads=threads)
    ref_rv[lo:hi] = r.rv()
    ref_epoch[lo:hi] = r.epoch_ns
    ref_acc[lo:hi] = rs.n_accepted
    ref_rej[lo:hi] = rs.n_rejected
    ref_ev[lo:hi] = rs.n_evals
    ref_status[lo:hi] = rs.status
    oracle_s = time.time() - t0
    print(f"oracle: {hi - first} of {last - first} in {oracle_s:.0f} s", flush=True)
    sel = slice(first, hi)
    d = out.rv()[sel] - ref_rv[sel]
    dr = np.linalg.norm(d[:, :3], axis=1) * 1e3      # km -> m
    dv = np.linalg.norm(d[:, 3:], axis=1) * 1e3      # km/s -> m/s
    m = hi - first
    res = {
        "what": "device (HIP, through the C-ABI) vs CPU oracle on EVERY trajectory of the configuration" if m == n else
                f"device (HIP, through the C-ABI) vs CPU oracle on trajectories {first} .. {hi - 1} of the configuration's {n}",
        "config": cfg_id, "workload": w["label"](n, hours), "n": n, "hours": hours, "first": first, "n_compared": m,
        "input_digest": input_digest, "kernel_source_stamp": stamp,
        "device_kernel_ms": kernel_ms, "coop_helpers": helpers,
        "oracle_threads": threads, "oracle_wall_s": round(oracle_s, 1), "oracle_traj_per_s": round(m / oracle_s, 2),
        "status_bad_device": int((st.status[sel] != 0).sum()), "status_bad_oracle": int((ref_status[sel] != 0).sum()),
        "epoch_mismatch": int((out.epoch_ns[sel] != ref_epoch[sel]).sum()),
        "dr_m": {"max": float(dr.max()), "p99": float(np.percentile(dr, 99)), "median": float(np.median(dr)), "argmax": first + int(dr.argmax())},
        "dv_m_per_s": {"max": float(dv.max()), "p99": float(np.percentile(dv, 99)), "median": float(np.median(dv)), "argmax": first + int(dv.argmax())},
        "bar": {"dr_m": 1.0, "dv_m_per_s": 1e-3},
        "within_bar": int(((dr <= 1.0) & (dv <= 1e-3)).sum()),
        "n_accepted_differs": int((st.n_accepted[sel] != ref_acc[sel]).sum()), "n_rejected_differs": int((st.n_rejected[sel] != ref_rej[sel]).sum()),
        "n_evals_differs": int((st.n_evals[sel] != ref_ev[sel]).sum()),
        "max_abs_n_accepted_diff": int(np.abs(st.n_accepted[sel] - ref_acc[sel]).max()),
        "device_evals": int(st.n_evals[sel].sum()), "oracle_evals": int(ref_ev[sel].sum()),
    }
    with open(os.path.join(ROOT, "gpurun_out", name + ".json"), "w") as f:
        json.dump(res, f, indent=1)
print(json.dumps(res), flush=True)
