"""One-process-per-GPU evaluation of the GPRF objective: units (blocks and block pairs) are independent given
(X, Y, theta) — the reference already maps them over a process pool (gprf.py:218-233) and combines them by a
weighted sum (gprf.py:253-288) — so each rank evaluates its share of the units
(``gprf_set_shard`` / ``gprf_partition_units``) into a dense partial ``[ll | gradX | gradC]`` vector and ONE
``all_reduce(SUM)`` (RCCL over xGMI on GPUs; gloo in the CPU tests) combines them.

``torch`` is plumbing here: device buffers, the stream the kernels are enqueued on, and
``torch.distributed``.
"""
import numpy as np

from . import _capi


def pack_out(ll, gX, gC, n, dx, ncov):
    """[ll | gradX row-major | gradC] — the layout gprf_eval_device writes (include/gprf_hip.h)."""
    buf = np.zeros(1 + n * dx + ncov)
    buf[0] = ll
    if gX is not None and gX.size:
        buf[1:1 + n * dx] = np.asarray(gX).reshape(-1)
    if gC is not None and np.size(gC):
        buf[1 + n * dx:] = np.asarray(gC).reshape(-1)
    return buf


def unpack_out(buf, n, dx, ncov, grad_X, grad_cov):
    ll = float(buf[0])
    gX = np.array(buf[1:1 + n * dx]).reshape(n, dx) if grad_X else np.zeros((0, 0))
    gC = np.array(buf[1 + n * dx:1 + n * dx + ncov]).reshape(1, -1) if grad_cov else np.zeros((0, 0))
    return ll, gX, gC


def unit_sizes(block_idxs, neighbors):
    """Points per unit in the library's numbering: blocks first, then pairs in order."""
    bl = [len(b) for b in block_idxs]
    return np.array(bl + [bl[i] + bl[j] for (i, j) in neighbors], dtype=np.int32)


def local_units(block_idxs, neighbors, dy, rank, world):
    """Global ids of the units rank ``rank`` evaluates (same partition as gprf_set_shard)."""
    owner = _capi.partition_units(unit_sizes(block_idxs, neighbors), dy, world)
    return np.nonzero(owner == rank)[0]


def allreduce_sum_(t, group=None):
    """The one collective of an evaluation."""
    import os
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get("GPRF_FORCE_ALLREDUCE") == "1"):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_sum_async_(t, group=None):
    """The same collective, asynchronously: returns the work handle (``None`` when there is nothing to reduce).
    ``handle.wait()`` makes the CURRENT stream wait for the result, it does not block the host: consecutive,
    independent evaluations keep the compute stream busy while the previous partials are still on the wire."""
    import os
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get("GPRF_FORCE_ALLREDUCE") == "1"):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


def agree_first_bad(bad, group=None, device=None):
    """All ranks learn the lowest failing unit id (or -1): MIN over ids with -1 mapped to +inf."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bad
    big = 2 ** 62
    t = torch.tensor([bad if bad >= 0 else big], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    v = int(t.item())
    return -1 if v == big else v


class DeviceEvaluator(object):
    """Device-resident evaluation loop over one GPRF context: X and the output vector live in HBM (torch
    tensors), kernels are enqueued on torch's current stream, and with world > 1 the partial outputs are
    all-reduced in place.  This is the timed region of bench.py."""

    def __init__(self, gprf, group=None):
        import torch
        self.torch = torch
        self.g = gprf
        self.group = group
        ctx = gprf._ctx
        self.n, self.dx, self.ncov = ctx.n, ctx.dx, ctx.ncov
        dev = torch.device("cuda", torch.cuda.current_device())
        # a real (non-null) stream: the kernels, the RCCL all-reduce and torch's events all sit on it
        self.stream = torch.cuda.Stream(device=dev)
        self.d_X = torch.empty(self.n * self.dx, dtype=torch.float64, device=dev)
        self.d_out = torch.empty(1 + self.n * self.dx + self.ncov, dtype=torch.float64, device=dev)
        self._work = None              # the all-reduce still in flight on d_out, if any

    def set_X(self, X):
        with self.torch.cuda.stream(self.stream):
            self.d_X.copy_(self.torch.as_tensor(np.ascontiguousarray(X, dtype=np.float64).reshape(-1)))
        self.stream.synchronize()

    def enqueue(self, grad_X=True, grad_cov=False, stream=None):
        """Enqueue one evaluation (+ the all-reduce) on ``stream`` (default: this evaluator's own);
        returns immediately."""
        st = self.stream if stream is None else stream
        with self.torch.cuda.stream(st):
            if self._work is not None:
                self._work.wait()      # the stream (not the host) waits before d_out is overwritten
            self.g._ctx.eval_device(self.d_X.data_ptr(), grad_X, grad_cov, self.d_out.data_ptr(), st.cuda_stream)
            # asynchronous: RCCL runs on its own stream behind this evaluation's kernels; the next (independent)
            # evaluation enqueued on `st` does not wait for it
            self._work = allreduce_sum_async_(self.d_out, self.group)

    def wait(self):
        """Make the current stream wait for the pending all-reduce of this evaluator (no host block)."""
        if self._work is not None:
            self._work.wait()
            self._work = None

    def result(self, grad_X=True, grad_cov=False):
        self.wait()
        self.torch.cuda.synchronize()
        rc, bad = self.g._ctx.eval_status()
        if rc == _capi.GPRF_NOT_PD:
            raise _capi.NotPositiveDefinite("unit %d: kernel matrix not positive definite" % bad, bad)
        return unpack_out(self.d_out.cpu().numpy(), self.n, self.dx, self.ncov, grad_X, grad_cov)


def distributed_llgrad(gprf, grad_X=False, grad_cov=False, group=None, evaluator=None):
    """``GPRF.llgrad`` for a sharded GPRF (constructed with ``shard=(rank, world)``): local partial sums,
    then one all-reduce.  Every rank returns the full (ll, gradX, gradCov)."""
    if evaluator is None:
        evaluator = DeviceEvaluator(gprf, group)
    gprf._push_blocks()
    gprf._push_neighbors(gprf.neighbors)
    evaluator.set_X(gprf.X)
    evaluator.enqueue(grad_X, grad_cov)
    return evaluator.result(grad_X, grad_cov)
