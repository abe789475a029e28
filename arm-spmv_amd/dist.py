"""Row-range sharding across GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Mirrors the reference's NUMA driver (src/mat_vec.cpp:230-297): rank r plays thread/node r —
  * rows are split into `world` contiguous ranges, nrow // world each, the last takes the remainder (:233,245-246)
  * a shard keeps a rebased row_ptr and GLOBAL column indices (:250-265)
  * every rank holds a FULL replica of x (:257,266); here the replica is assembled from the ranks' own slices
    (rank r owns x[rows of r], the natural layout when y of one product feeds x of the next) by an all-gather
  * y stays sharded; concatenate_y() is the optional gather the reference only performs for DIA (:474-477)
The only exchange step is the x all-gather; there is no reduction (rows are independent).

Compute is NOT in this module: callers apply their shard with the HIP engine (capi.Context.apply).  That keeps
the collective logic testable on CPU with gloo, where the tests plug the oracle in as the per-shard product.
"""
from __future__ import annotations

import os

# One process per GPU shares device buffers through dmabuf handles on this platform; with the legacy IPC mode RCCL's
# cross-process registration fails (`hipIpcGetMemHandle: invalid argument`).  Harmless when already set; must be in the
# environment before the process first touches the GPU, so it is set when this module is imported.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

from . import capi


def shard_rows(nrow: int, world: int, rank: int) -> tuple[int, int]:
    """[begin, end) of rank's rows — the engine's own arithmetic (spmv_partition_rows)"""
    return capi.partition_rows(nrow, world, rank)


def all_bounds(nrow: int, world: int) -> list[tuple[int, int]]:
    return [shard_rows(nrow, world, r) for r in range(world)]


def balanced_bounds(row_ptr64, world: int) -> list[tuple[int, int]]:
    """[begin, end) per rank with about the same number of STORED ENTRIES each (spmv_partition_rows_balanced) — the option
    SURVEY.md 8e names for skewed matrices; the reference itself only splits by equal rows (src/mat_vec.cpp:233).  Every rank
    must compute it from the same offsets (they are small: 8 bytes per row).  Pass the result as `bounds=` below."""
    b = capi.partition_rows_balanced(row_ptr64, world)
    return [(int(b[r]), int(b[r + 1])) for r in range(world)]


def allgather_x(x_full: torch.Tensor, x_own: torch.Tensor, nrow: int, group=None, bounds=None) -> None:
    """x_full[rows of r] <- rank r's x_own, for every r.  One collective when the slices are equal
    (all_gather_into_tensor: ring/mesh over xGMI under RCCL), one broadcast per rank otherwise.
    bounds: the ranks' row ranges when they are not the equal-rows split (balanced_bounds)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    equal = bounds is None
    bounds = all_bounds(nrow, world) if bounds is None else [(int(b), int(e)) for b, e in bounds]
    if len(bounds) != world or bounds[0][0] != 0 or bounds[-1][1] != nrow or any(bounds[r][1] != bounds[r + 1][0] for r in range(world - 1)):
        raise ValueError(f"bounds {bounds} do not tile [0, {nrow}) over {world} ranks")
    b, e = bounds[rank]
    if x_own.numel() != e - b:
        raise ValueError(f"rank {rank} owns rows [{b},{e}) but passed a slice of {x_own.numel()} entries")
    if x_full.numel() != nrow:
        raise ValueError(f"x_full has {x_full.numel()} entries, expected {nrow}")
    if x_full.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (several ranks sharing one GPU, where RCCL refuses to run): stage through the host
        host = torch.empty(nrow, dtype=x_full.dtype)
        allgather_x(host, x_own.cpu(), nrow, group, None if equal else bounds)
        x_full.copy_(host)
        return
    if equal and nrow % world == 0:
        dist.all_gather_into_tensor(x_full, x_own.contiguous(), group=group)
        return
    for r, (rb, re) in enumerate(bounds):
        part = x_full[rb:re]
        if r == rank:
            part.copy_(x_own)
        if re > rb:
            dist.broadcast(part, src=dist.get_global_rank(group, r) if group is not None else r, group=group)


def concatenate_y(y_own: torch.Tensor, nrow: int, group=None, bounds=None) -> torch.Tensor:
    """full y on every rank from the row slices (the reference's DIA driver copy-back, generalised)"""
    y_full = torch.empty(nrow, dtype=y_own.dtype, device=y_own.device)
    allgather_x(y_full, y_own, nrow, group, bounds)
    return y_full


def max_over_ranks(value: float, device, group=None) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


# ---- plans across ranks (spmv_mat_get_plan / spmv_ctx_set_plan) -------------------------------------------------
def broadcast_plan(plan: bytes | None, device, src: int = 0, group=None) -> bytes:
    """rank `src`'s plan blob on every rank.  The reference builds all its shards the same way (src/mat_vec.cpp:240-268); with
    AUTO a measurement, every rank would otherwise draw its own kernel.  Typical use: rank `src` builds its shard with AUTO,
    takes `A.get_plan()`, every other rank passes None here, calls `ctx.set_plan(blob)` and builds its shard under it."""
    rank = dist.get_rank(group)
    g_src = dist.get_global_rank(group, src) if group is not None else src
    n = torch.tensor([len(plan) if (rank == src and plan is not None) else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=g_src, group=group)
    size = int(n.item())
    if size == 0:
        return b""
    if rank == src:
        buf = torch.frombuffer(bytearray(plan), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty(size, dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=g_src, group=group)
    return bytes(buf.cpu().numpy().tobytes())


def plans_equal(plan: bytes, device, group=None) -> bool:
    """True on every rank iff every rank holds the same plan blob (a 64-bit digest is compared: min == max over the ranks)"""
    import hashlib

    d = int.from_bytes(hashlib.sha256(plan).digest()[:7], "little")  # 56 bits: exact in int64
    lo = torch.tensor([d], dtype=torch.int64, device=device)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return int(lo.item()) == int(hi.item())


# ---- solver step across shards ----------------------------------------------------------
def sum_over_ranks(value: float, device, group=None) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())


class HipShardOps:
    """What one rank does locally in cg_sharded, on its GPU through the engine (no torch arithmetic):
    the shard product with the fused dot (spmv_apply_dot) and the reference's BLAS-1 (spmv_axpby, spmv_dot)."""

    def __init__(self, ctx: "capi.Context", A_shard: "capi.Matrix"):
        self.ctx, self.A = ctx, A_shard
        self._views: dict = {}

    def _v(self, t: torch.Tensor) -> "capi.Vector":
        key = (t.data_ptr(), t.numel())
        if key not in self._views:
            self._views[key] = self.ctx.wrap_vector(t)
        return self._views[key]

    def product_dot(self, p_full, w_own, q_own) -> float:  # q = A_shard p ; returns w_own . q
        return self.ctx.apply_dot(self.A, self._v(p_full), self._v(q_own), self._v(w_own), overwrite=True)

    # ---- the same product in two halves, for overlapping the exchange (enable_overlap) -------------------------
    def enable_overlap(self, col_begin: int, col_end: int) -> None:
        """split the shard by column range on the device: the part inside [col_begin, col_end) multiplies the rank's
        OWN slice of the direction and needs no exchange (spmv_csr_split_columns)"""
        self.A_in, self.A_out = self.ctx.csr_split_columns(self.A, col_begin, col_end)

    def begin_local(self, p_own, q_own) -> None:  # q = A_in p_own, queued on the engine's stream; returns at once
        self.ctx.apply(self.A_in, self._v(p_own), self._fill0(q_own))

    def finish_remote_dot(self, p_full, w_own, q_own) -> float:  # q += A_out p_full ; returns w_own . q
        return self.ctx.apply_dot(self.A_out, self._v(p_full), self._v(q_own), self._v(w_own), overwrite=False)

    def _fill0(self, t):
        v = self._v(t)
        v.fill(0.0)
        return v

    # ---- block-Jacobi preconditioner: symmetric Gauss-Seidel inside the rank's own diagonal block -------------
    def enable_symgs(self, col_begin: int, col_end: int, sweeps: int = 1) -> None:
        """z = M^-1 r with M = one symmetric Gauss-Seidel sweep (spmv_symgs, multicolour) on the square block of the shard
        whose columns are the rank's own rows [col_begin, col_end) - block Jacobi across the ranks, Gauss-Seidel inside,
        no exchange.  The block is the `inside` part of spmv_csr_split_columns (columns rebased to 0)."""
        if not hasattr(self, "A_in"):
            self.enable_overlap(col_begin, col_end)
        self._gs_sweeps = sweeps
        self.ctx.symgs_order(self.A_in)  # set-up now (colouring, split, levels), outside the iteration

    def precondition(self, r_own, z_own) -> None:
        self.ctx.symgs(self.A_in, self._v(r_own), self._fill0(z_own), self._gs_sweeps)

    def axpby(self, alpha, x, beta, y, w) -> None:  # w = alpha x + beta y (w may be x or y)
        self.ctx.axpby(alpha, self._v(x), beta, self._v(y), self._v(w))

    def dot(self, a, b) -> float:
        return self.ctx.dot(self._v(a), self._v(b))

    def sync(self) -> None:
        self.ctx.sync()


def cg_sharded(ops, b_own: torch.Tensor, x_own: torch.Tensor, nrow: int, max_iter: int = 1000, rel_tol: float = 1e-8,
               group=None) -> tuple[int, float]:
    """Conjugate gradients over row shards (A symmetric positive definite, x0 = x_own on entry, overwritten).

    Per iteration: ONE all-gather (the search direction p, which every shard needs in full — the x exchange of
    SURVEY.md 8e) and two scalar all-reduces (p.Ap and r.r).  When `ops` was split by column range
    (HipShardOps.enable_overlap), the product of the rank's own columns is queued on the engine's stream BEFORE the
    all-gather is issued on torch's, so the two run side by side; the remaining columns follow the exchange.  When `ops`
    has a `precondition(r_own, z_own)` (HipShardOps.enable_symgs: one symmetric Gauss-Seidel sweep on the rank's own diagonal
    block - block Jacobi across the ranks, so it needs no exchange) the iteration is preconditioned CG with one more scalar
    all-reduce (r.z).  Vectors stay sharded by rows and device-resident;
    `ops` does the local work (HipShardOps on a GPU; the CPU tests plug the oracle in).  Every rank returns the same
    (iterations, ||r|| / ||b||)."""
    dev = b_own.device
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = all_bounds(nrow, world)[rank]
    precond = hasattr(ops, "precondition") and getattr(ops, "_gs_sweeps", 0) > 0  # (enable_symgs was called)
    p_full = torch.empty(nrow, dtype=torch.float64, device=dev)
    q_own, r_own, p_own = torch.empty_like(b_own), torch.empty_like(b_own), torch.empty_like(b_own)
    z_own = torch.empty_like(b_own) if precond else r_own
    # r = b - A x0, z = M^-1 r, p = z
    allgather_x(p_full, x_own, nrow, group)
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    ops.product_dot(p_full, b_own, q_own)
    ops.axpby(1.0, b_own, -1.0, q_own, r_own)
    if precond:
        ops.precondition(r_own, z_own)
    ops.axpby(1.0, z_own, 0.0, z_own, p_own)
    bb = sum_over_ranks(ops.dot(b_own, b_own), dev, group)
    rr = sum_over_ranks(ops.dot(r_own, r_own), dev, group)
    rz = sum_over_ranks(ops.dot(r_own, z_own), dev, group) if precond else rr
    if bb == 0.0 or rr <= rel_tol * rel_tol * bb:
        return 0, (rr / bb) ** 0.5 if bb > 0.0 else 0.0
    k = 0
    overlap = (hasattr(ops, "A_in") or getattr(ops, "overlap", False)) and getattr(ops, "use_overlap", True)
    while k < max_iter:
        ops.sync()  # the collective runs on torch's stream: p_own must be complete
        if overlap:
            # the product of the rank's own column range runs on the engine's stream while the exchange is in flight
            ops.begin_local(p_own, q_own)
            allgather_x(p_full, p_own, nrow, group)
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()
            pq = sum_over_ranks(ops.finish_remote_dot(p_full, p_own, q_own), dev, group)
        else:
            allgather_x(p_full, p_own, nrow, group)
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()  # ... and p_full before the engine's stream reads it
            pq = sum_over_ranks(ops.product_dot(p_full, p_own, q_own), dev, group)
        if not pq > 0.0:
            raise ArithmeticError(f"cg_sharded: p.Ap = {pq} at iteration {k}: the matrix is not positive definite")
        alpha = rz / pq
        ops.axpby(alpha, p_own, 1.0, x_own, x_own)
        ops.axpby(-alpha, q_own, 1.0, r_own, r_own)
        rr = sum_over_ranks(ops.dot(r_own, r_own), dev, group)
        k += 1
        if rr <= rel_tol * rel_tol * bb:
            break
        if precond:
            ops.precondition(r_own, z_own)  # local: block Jacobi across the ranks, no exchange
            rz_new = sum_over_ranks(ops.dot(r_own, z_own), dev, group)
        else:
            rz_new = rr
        ops.axpby(1.0, z_own, rz_new / rz, p_own, p_own)
        rz = rz_new
    return k, (rr / bb) ** 0.5
