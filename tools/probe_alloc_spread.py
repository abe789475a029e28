#!/usr/bin/env python3
"""tools/probe_alloc_spread.py — do the other kernels' times depend on WHERE their matrix lies (the classes of physical memory
found for the two-phase product stream)?  Each workload is generated several times in one process, memory held
between the builds so that each lands elsewhere, and timed (x resident, 50 products, best of 3)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi


def main():
    ctx = capi.Context(0)
    builds = int(sys.argv[1]) if len(sys.argv) > 1 else 8

    def run(name, make, ncol, nrow, flags=0):
        times, held = [], []
        x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(nrow)
        y.fill(0.0)
        for b in range(builds):
            A = make()
            if flags:
                A.set_flags(flags)
            for _ in range(5):
                ctx.apply(A, x, y)
            times.append(min(ctx.apply_timed(A, x, y, 50) for _ in range(3)))
            del A
            held.append(ctx.vector((1 + b % 3) * (1 << 27)))  # 1-3 GB held: the next build starts elsewhere
        print(f"{name}: " + " ".join(f"{t:.4f}" for t in times) + f"   spread {max(times) / min(times):.3f}", flush=True)

    run("C3 ELL 4M x 64, diagonal slots", lambda: ctx.gen_ell_banded(4_000_000, 4_000_000, 64, seed=1), 4_000_000, 4_000_000)
    run("C3 ELL 4M x 64, columns read", lambda: ctx.gen_ell_banded(4_000_000, 4_000_000, 64, seed=1), 4_000_000, 4_000_000, flags=8)
    run("DIA 4M x 64", lambda: ctx.gen_dia_banded(4_000_000, 64, seed=1), 4_000_000, 4_000_000)
    run("C4 COO 2M power-law (panel path)", lambda: ctx.gen_coo_powerlaw(2_000_000, 2_000_000, 4096, seed=1), 2_000_000, 2_000_000)
    run("band 65536, 10M x 32 (panel)", lambda: ctx.gen_csr_uniform(0, 10_000_000, 10_000_000, 32, band=65536, seed=1), 10_000_000, 10_000_000)
    run("C2 10M x 32 uniform (panel)", lambda: ctx.gen_csr_uniform(0, 10_000_000, 10_000_000, 32, band=0, seed=1), 10_000_000, 10_000_000)


if __name__ == "__main__":
    main()
