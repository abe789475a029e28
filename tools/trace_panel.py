#!/usr/bin/env python3
"""tools/trace_panel.py — where a chunk's time goes in the panel kernel (GPU box only).

Runs the diagnostic build of the gather-first pipeline (panel_trace = 1) on the C2 matrix and prints, per traced
wavefront, the mean length of the four phases of a chunk in microseconds:
    issue    chunk start -> gathers of this chunk and streamed loads of the next one issued
    gather   -> gathers back (s_waitcnt vmcnt(16))
    add      -> ds_add_f64 done (s_waitcnt lgkmcnt(0))
    stream   -> next chunk's entries back (s_waitcnt vmcnt(0))
    period   chunk start -> next chunk start (pace waits included)
"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--ncol", type=int, default=0)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--pace", type=int, default=-1, help="ns per chunk (-1: the trial's choice, 0: unthrottled)")
    ap.add_argument("--stagger", type=int, default=2)
    ap.add_argument("--sync", type=int, default=0, help="0 none, 1 barrier per chunk, 2 split barrier, 3 barrier between loads and adds")
    ap.add_argument("--legacy", type=int, default=0, help="1: the general kernel (run-time sync switch)")
    a = ap.parse_args()
    ctx = capi.Context(0)
    ncol = a.ncol or a.n
    A = ctx.gen_csr_uniform(0, a.n, ncol, a.k, seed=1)
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(a.n)
    y.fill(0.0)
    for name, v in (("panel_aos", 3), ("panel_unroll", 8), ("panel_pipe", 2), ("panel_stagger", a.stagger), ("panel_sync", a.sync), ("panel_legacy", a.legacy), ("panel_pace_ns", a.pace)):
        A.set_param(name, v)
    A.set_kernel(capi.CSR_PANEL)
    pace = A.get_param("panel_pace_ns")
    ms = ctx.apply_timed(A, x, y, 10)
    A.set_param("panel_trace", 1)
    ctx.apply(A, x, y)
    ms_tr = ctx.apply_timed(A, x, y, 5)
    WGS, CH, ST = 256, 32, 5
    t = np.array([A.get_param(f"panel_trace@{i}") for i in range(WGS * 2 * CH * ST)], dtype=np.int64).reshape(WGS, 2, CH, ST)
    A.set_param("panel_trace", 0)
    print(f"C2-shape n={a.n} ncol={ncol}: pace {pace} ns, stagger {a.stagger}, sync {a.sync}, legacy {a.legacy}: {ms:.4f} ms per product; traced build {ms_tr:.4f} ms")
    d = (np.diff(t, axis=3) & 0xFFFFFFFF) / 100.0          # [wg, wave, chunk, phase] in us
    busy = ((t[..., 4] - t[..., 0]) & 0xFFFFFFFF) / 100.0   # chunk start -> next chunk's entries back
    per = (np.diff(t[..., 0], axis=2) & 0xFFFFFFFF) / 100.0
    start = ((t[:, 1, :, 0] - t[0:1, 1, :, 0]) & 0xFFFFFFFF).astype(np.int64)
    start = np.where(start > 1 << 31, start - (1 << 32), start) / 100.0  # chunk start of wavefront 15 relative to workgroup 0
    for h, name in ((0, "wavefront 0"), (1, "wavefront 15")):
        m = d[:, h].mean(axis=(0, 1))
        print(f"{name}: mean over {WGS} workgroups x {CH} chunks: issue {m[0]:.2f}  gather {m[1]:.2f}  add {m[2]:.2f}  stream {m[3]:.2f}  "
              f"busy {busy[:, h].mean():.2f} us; period {per[:, h].mean():.2f} us")
    b15 = busy[:, 1]
    print(f"wavefront 15 busy time per chunk: mean {b15.mean():.2f}, p50 {np.percentile(b15, 50):.2f}, p90 {np.percentile(b15, 90):.2f}, "
          f"p99 {np.percentile(b15, 99):.2f}, max {b15.max():.2f} us")
    per_wg = b15.mean(axis=1)
    order = np.argsort(per_wg)
    print("slowest workgroups (mean busy us, workgroup, blockIdx % 8):", [(round(float(per_wg[i]), 2), int(i), int(i % 8)) for i in order[-8:]])
    print("fastest workgroups:", [(round(float(per_wg[i]), 2), int(i), int(i % 8)) for i in order[:8]])
    by_x = [round(float(per_wg[x::8].mean()), 2) for x in range(8)]
    print("mean busy by blockIdx % 8 (one XCD each):", by_x)
    print("per-chunk max over workgroups of the busy time:", np.round(b15.max(axis=0), 1).tolist())
    rel = start - np.median(start, axis=0, keepdims=True)  # chunk start relative to the median workgroup of that chunk
    print("chunk start relative to the median workgroup (us): p1 %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f min %.2f max %.2f"
          % tuple(np.percentile(rel, q) for q in (1, 10, 50, 90, 99, 0, 100)))
    lag = rel.mean(axis=1)
    o2 = np.argsort(lag)
    print("most behind (mean us, workgroup):", [(round(float(lag[i]), 1), int(i)) for i in o2[-10:]])
    print("most ahead:", [(round(float(lag[i]), 1), int(i)) for i in o2[:10]])
    print("mean lag by blockIdx % 8:", [round(float(lag[x::8].mean()), 2) for x in range(8)])
    big = np.argwhere(b15 > 12.0)
    print("chunks longer than 12 us: %d of %d; workgroups involved: %s" % (len(big), b15.size, sorted(set(int(i) for i, _ in big))[:40]))
    for i, c in big[:6]:
        print("   workgroup %d chunk %d: phases (issue, gather, add, stream) wave0 %s wave15 %s" % (i, c + 16, np.round(d[i, 0, c], 2).tolist(), np.round(d[i, 1, c], 2).tolist()))


if __name__ == "__main__":
    main()
