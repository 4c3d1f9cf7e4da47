"""One-process-per-GPU evaluation of the GPRF objective: units (blocks and block pairs) are independent given
(X, Y, theta) — the reference already maps them over a process pool (gprf.py:218-233) and combines them by a
weighted sum (gprf.py:253-288) — so each rank evaluates its share of the units
(``gprf_set_shard`` / ``gprf_partition_units``) into a dense partial ``[ll | gradX | gradC | status]`` vector and ONE
``all_reduce(SUM)`` (RCCL over xGMI on GPUs; gloo in the CPU tests) combines them.  The two status words at the end of
the vector (workspace outgrown on some rank / units not positive definite on some rank) ride in the same collective,
so every rank takes the same decision — repeat, jitter, raise — without a second exchange on the common path.

``torch`` is plumbing here: device buffers, the stream the kernels are enqueued on, and ``torch.distributed``.
"""
import numpy as np

from . import _capi

N_STATUS = 2      # status words behind [ll | gradX | gradC] (include/gprf_hip.h, gprf_eval_device)
FATAL = 2.0 ** 40  # first status word of a rank whose library call failed (a count of ranks otherwise: far below)


def out_len(n, dx, ncov):
    return 1 + n * dx + ncov + N_STATUS


def pack_out(ll, gX, gC, n, dx, ncov):
    """[ll | gradX row-major | gradC | 0 0] — the layout gprf_eval_device writes (include/gprf_hip.h)."""
    buf = np.zeros(out_len(n, dx, ncov))
    buf[0] = ll
    if gX is not None and gX.size:
        buf[1:1 + n * dx] = np.asarray(gX).reshape(-1)
    if gC is not None and np.size(gC):
        buf[1 + n * dx:1 + n * dx + ncov] = np.asarray(gC).reshape(-1)
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


def _active(group=None):
    import os
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (
        dist.get_world_size(group) > 1 or os.environ.get("GPRF_FORCE_ALLREDUCE") == "1")


def allreduce_sum_(t, group=None):
    """The one collective of an evaluation."""
    import torch.distributed as dist
    if _active(group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_sum_async_(t, group=None):
    """The same collective, asynchronously: returns the work handle (``None`` when there is nothing to reduce).
    ``handle.wait()`` makes the CURRENT stream wait for the result, it does not block the host."""
    import torch.distributed as dist
    if _active(group):
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


def agree_any(flag, group=None, device=None):
    """True on every rank iff ``flag`` is true on some rank (one MAX all-reduce of one word; error paths only)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t.item())


def jitter_schedule(evaluate, first_bad, n_units, diag_mean, start=None):
    """jitchol's policy (gpy_linalg.py:81-97) per failing unit: require a positive diagonal, then retry on K + j I with
    j = mean(diag K) * 1e-6 * 10^k, k = 0..4, else LinAlgError.  ``evaluate(jitter) -> (result, bad)`` runs one
    evaluation with the per-unit jitter vector and returns the lowest failing unit id (-1 = none) — in a sharded run
    the id every rank agreed on, so that all ranks walk the same schedule."""
    from collections import defaultdict
    if not (diag_mean > 0.):
        raise _capi.NotPositiveDefinite("not pd: non-positive diagonal elements", first_bad)
    jitter = np.zeros(n_units) if start is None else np.array(start, dtype=np.float64)
    tries = defaultdict(int)
    bad = first_bad
    while True:
        k = tries[bad]
        if k >= 5:
            raise _capi.NotPositiveDefinite("not positive definite, even with jitter.", bad)
        jitter[bad] = diag_mean * 1e-6 * 10.0 ** k
        tries[bad] += 1
        result, bad = evaluate(jitter)
        if bad < 0:
            return result, jitter


class DeviceEvaluator(object):
    """Evaluation loop over one (sharded) GPRF context with X and the output vector in HBM (torch tensors): kernels on
    a torch stream, the partial outputs all-reduced in place when world > 1.

    ``evaluate`` is the optimiser-visible call — host X in, host (ll, gradX, gradC) out, the re-blocking of update_X
    and the all-reduce inside, nothing of the next evaluation overlapping it.  ``enqueue`` / ``result`` are the
    device-resident halves (bench.py's pipelined side figure, tests)."""

    def __init__(self, gprf, group=None):
        import torch
        self.torch = torch
        self.g = gprf
        self.group = group
        ctx = gprf._ctx
        self.n, self.dx, self.ncov = ctx.n, ctx.dx, ctx.ncov
        dev = torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        # a real (non-null) stream: the kernels, the RCCL all-reduce and torch's events all sit on it
        self.stream = torch.cuda.Stream(device=dev)
        nx, no = self.n * self.dx, out_len(self.n, self.dx, self.ncov)
        self.d_X = torch.empty(nx, dtype=torch.float64, device=dev)
        self.d_out = torch.empty(no, dtype=torch.float64, device=dev)
        self.h_X = torch.empty(nx, dtype=torch.float64, pin_memory=True)
        self.h_out = torch.empty(no, dtype=torch.float64, pin_memory=True)
        self._work = None              # the all-reduce still in flight on d_out, if any
        self._pipelines_on = False     # (set by the synchronous form, _round; enqueue() — back-to-back evaluations — leaves it off)

    def set_X(self, X):
        with self.torch.cuda.stream(self.stream):
            self.h_X.numpy()[:] = np.ascontiguousarray(X, dtype=np.float64).reshape(-1)
            self.d_X.copy_(self.h_X, non_blocking=True)
        self.stream.synchronize()

    def enqueue(self, grad_X=True, grad_cov=False, stream=None, reblock=False):
        """Enqueue one evaluation (+ the all-reduce) on ``stream`` (default: this evaluator's own);
        returns immediately."""
        st = self.stream if stream is None else stream
        with self.torch.cuda.stream(st):
            if self._work is not None:
                self._work.wait()      # the stream (not the host) waits before d_out is overwritten
            self.g._ctx.eval_device(self.d_X.data_ptr(), grad_X, grad_cov, self.d_out.data_ptr(), st.cuda_stream,
                                    reblock=reblock)
            # asynchronous: RCCL runs on its own stream behind this evaluation's kernels
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
        if rc == _capi.GPRF_RETRY:
            raise _capi.GprfHipError("the re-partition outgrew the workspace; enqueue the evaluation again")
        return unpack_out(self.d_out.cpu().numpy(), self.n, self.dx, self.ncov, grad_X, grad_cov)

    # ------------------------------------------------------------------ the optimiser-visible call
    def _round(self, grad_X, grad_cov, reblock, objective=False):
        """one evaluation + all-reduce + download, fully finished: -> (host vector, local status, local bad unit, error).

        Nothing raises between the collectives: a rank whose library call fails still takes part in the all-reduce, with
        a zero vector whose first status word is FATAL, and hands its error back — ``evaluate`` raises it on every rank
        together (a rank that left the loop alone would leave the others waiting in a collective)."""
        torch, st = self.torch, self.stream
        no = out_len(self.n, self.dx, self.ncov)
        err = None
        if not self._pipelines_on:
            # one evaluation at a time, waited for: the library may pipeline the size classes on this stream as it does on its own
            self.g._ctx.set_stream_pipelines(True)
            self._pipelines_on = True
        with torch.cuda.stream(st):
            enqueue = self.g._ctx.objective_device if objective else self.g._ctx.eval_device
            try:
                enqueue(self.h_X.data_ptr(), grad_X, grad_cov, self.d_out.data_ptr(), st.cuda_stream, reblock=reblock)
            except _capi.GprfHipError as e:
                err = e
                self.d_out.zero_()
                self.d_out[no - 2] = FATAL
            allreduce_sum_(self.d_out, self.group)
            self.h_out.copy_(self.d_out, non_blocking=True)
        st.synchronize()
        rc, bad = -1, -1
        if err is None:
            try:
                rc, bad = self.g._ctx.eval_status()
            except _capi.GprfHipError as e:      # e.g. the new partition has a unit beyond GPRF_MAX_UNIT points
                err = e
        return self.h_out.numpy(), rc, bad, err

    def evaluate(self, X, grad_X=False, grad_cov=False, reblock=False, objective=False):
        """-> (ll, gradX, gradC, reblocked) with jitchol's retry applied identically on every rank.
        ``objective``: the optimiser's form (gprf_objective_device): -(ll + location prior), -(gradX + its gradient),
        gradC unchanged."""
        torch, st = self.torch, self.stream
        # X goes into pinned host memory and the kernels read it from there (the partition kernel, which runs first when
        # re-blocking, leaves a copy in HBM for the others; without re-blocking only k_scatter_x reads it): no copy command
        self.h_X.numpy()[:] = np.ascontiguousarray(X, dtype=np.float64).reshape(-1)
        g = self.g
        no = out_len(self.n, self.dx, self.ncov)

        def run(reblock_now):
            # repeat while ANY rank's new partition outgrew its workspace (that rank has grown it by now)
            for _ in range(4):
                buf, rc, bad, err = self._round(grad_X, grad_cov, reblock_now, objective)
                if buf[no - 2] >= FATAL:                 # some rank's enqueue failed: everybody leaves here
                    raise err or _capi.GprfHipError("another rank's evaluation failed")
                if buf[no - 2] == 0.0:
                    if err is not None:
                        raise err                        # (a failed synchronisation: nothing left to agree on)
                    if reblock_now and g._ctx.last_reblocked():
                        reblocked[0] = True
                    return buf, bad
                # some rank's re-partition outgrew its workspace.  Growing it may have failed there (a unit past
                # GPRF_MAX_UNIT points, out of memory): agree before anybody enters the next round's collective
                if agree_any(err is not None, self.group, self.dev):
                    raise err or _capi.GprfHipError("another rank could not install the new partition")
                if reblock_now and g._ctx.last_reblocked():
                    reblocked[0] = True
                reblock_now = False
            raise _capi.GprfHipError("the unit tables did not fit the workspace after growing it three times")

        reblocked = [False]
        buf, bad = run(reblock)
        reblocked = reblocked[0]
        if buf[no - 1] != 0.0:
            # some unit on some rank is not positive definite: every rank learns the lowest such unit and walks the
            # same jitter schedule (no rank may return NaN while another raises)
            n_units = g.n_blocks + len(g._nbrs_pushed)
            diag_mean = g.cov.wfn_params[0] + g.noise_var

            def ev(jitter):
                g._ctx.set_unit_jitter(jitter)
                g._jitter = jitter
                b, bd = run(False)
                return b, (agree_first_bad(bd, self.group, self.dev) if b[no - 1] != 0.0 else -1)

            buf, _ = jitter_schedule(ev, agree_first_bad(bad, self.group, self.dev), n_units, diag_mean)
        ll, gX, gC = unpack_out(buf, self.n, self.dx, self.ncov, grad_X, grad_cov)
        return ll, gX, gC, reblocked


def distributed_llgrad(gprf, grad_X=False, grad_cov=False, group=None, evaluator=None):
    """``GPRF.llgrad`` for a sharded GPRF (constructed with ``shard=(rank, world)``): local partial sums,
    then one all-reduce.  Every rank returns the full (ll, gradX, gradCov).  (``GPRF.llgrad`` itself does this when
    torch.distributed is initialised with more than one rank; this form takes an explicit group / evaluator.)"""
    if evaluator is None:
        evaluator = DeviceEvaluator(gprf, group)
    gprf._push_blocks()
    gprf._push_neighbors(gprf.neighbors)
    ll, gX, gC, _ = evaluator.evaluate(gprf.X, grad_X, grad_cov, reblock=False)
    return ll, gX, gC
