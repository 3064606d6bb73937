"""View-sharded data parallelism for a training step (one process per GPU).

The reference trains one view per iteration on one GPU (train.py:172-188) and
has no distributed mode.  Views are independent given the parameters, so a
multi-view batch shards by view with no exchange inside the rasterizer: every
rank holds a full replica of the Gaussians, renders its slice of the batch and
accumulates gradients locally; a sum all-reduce of the parameter gradients (63
floats per Gaussian with the fork's 4 `reps` channels), issued range by range
of Gaussians while the per-Gaussian backward of the later ranges still runs,
then makes every replica see the gradient of the whole batch.  `torch.distributed` backend "nccl" is
RCCL on ROCm (xGMI between the GPUs of a node); "gloo" runs the same code on
CPU tensors for tests.

Densification statistics need the same treatment: the accumulated view-space
gradient norms and visibility counts are sums over views, the maximum screen
radius is a max (train.py:305-309, scene/gaussian_model.py:213-214).
"""
import torch
import torch.distributed as dist


def shard_views(n_views, rank, world_size):
    """Indices of the views rank `rank` renders: a contiguous, balanced split."""
    base, extra = divmod(n_views, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


class GradReducer:
    """Sums gradients over ranks with asynchronous all-reduces on the process
    group's communication stream (RCCL: the collective waits for what the
    current stream has enqueued when it is started, and runs beside what the
    caller enqueues afterwards).

    `start(tensors)` reduces whole tensors, one collective each, largest first.
    `start_rows(tensors, first, count)` reduces rows [first, first + count) of
    every tensor as ONE coalesced collective: with `ParamGradSink.flush(chunks,
    on_chunk=...)` the rows of a range of Gaussians go out as soon as the launch
    that completes them is enqueued, and their exchange overlaps the per-Gaussian
    backward of the remaining ranges.  `wait()` makes the current stream (gloo:
    the host) wait for everything started.

    Sparse exchange.  A view only gives gradients to the Gaussians its pixels
    reached, and in a large or heavily occluded scene the views of a step reach
    a small part of the model; all other rows are exactly zero on every rank.
    `plan_rows(touched)` takes this rank's uint8 row mask (`ParamGradSink.
    touched_rows()`), forms the union over the ranks (one MAX all-reduce of P
    bytes), and returns the row indices of the union when it is at most
    `sparse_below` of the rows, else None; `start_gathered(tensors, rows)` then
    exchanges only those rows, packed into one [n, row width] buffer, and
    `wait()` writes the sums back.  The result equals the dense all-reduce:
    the rows left out are zero everywhere.  `plan_rows` blocks the host (the
    row count sizes the collective, and every rank must take the same branch);
    `sparse`: "auto" probes every step while the union stays sparse and
    every `probe_every` steps after three dense steps in a row, "always" never
    falls back, "never" makes `plan_rows` return None without communication."""

    def __init__(self, group=None, average=False, sparse="auto", sparse_below=0.35, probe_every=16):
        if sparse not in ("auto", "always", "never"):
            raise ValueError("GradReducer: sparse must be 'auto', 'always' or 'never'")
        self.group = group
        self.average = average
        self.pending = []
        self.coalesce = hasattr(dist, "_coalescing_manager")
        self.calls = 0
        self.sparse = sparse
        self.sparse_below = float(sparse_below)
        self.probe_every = max(1, int(probe_every))
        self.dense_streak = 0
        self.plans = 0
        self.last_union = None      # (rows in the union, rows) of the last probe
        self.mask_as_bytes = True   # the row masks travel as uint8 with MAX (else as int32 with SUM)

    def _active(self):
        return dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _staged(self, t, op=dist.ReduceOp.SUM):
        """gloo rehearsal of the multi-rank path on GPU tensors: through host memory, synchronously."""
        h = t.cpu()
        dist.all_reduce(h, op=op, group=self.group)
        t.copy_(h)

    def start(self, tensors):
        if not self._active():
            return self
        staged = dist.get_backend(self.group) == "gloo"
        for t in sorted((t for t in tensors if t is not None), key=lambda t: -t.numel()):
            if staged and t.is_cuda:
                self._staged(t)
                continue
            self.pending.append(([t], dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        return self

    def start_rows(self, tensors, first, count):
        if not self._active() or count <= 0:
            return self
        rows = [t[first:first + count] for t in tensors if t is not None and t.numel() != 0]
        if dist.get_backend(self.group) == "gloo" and any(t.is_cuda for t in rows):
            for t in rows:
                self._staged(t)
            return self
        # one coalesced collective for the seven row blocks (a single grouped launch with RCCL); a backend or
        # PyTorch build without coalescing gets one asynchronous all-reduce per block instead
        if self.coalesce:
            try:
                with dist._coalescing_manager(group=self.group, async_ops=True) as cm:
                    for t in rows:
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                self.pending.append((rows, cm))
                self.calls += 1
                return self
            except (AttributeError, RuntimeError, ValueError, TypeError, NotImplementedError):
                # nothing was issued if entering or leaving the manager failed before the collective started;
                # a failure after a partial issue would double-count, so only the first call may fall back
                if self.pending or self.calls:
                    raise
                self.coalesce = False
        self.calls += 1
        for t in rows:
            self.pending.append(([t], dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        return self

    def plan_rows(self, touched):
        """touched: uint8 [P], non-zero where this rank's gradient rows may be non-zero, or a function
        that returns it (only called when the mask is needed); every rank calls this in every step, with
        its own mask.  Returns the int64 indices of the rows in the
        union over the ranks, or None when the exchange should be dense."""
        if not self._active() or self.sparse == "never":
            return None
        self.plans += 1
        if self.sparse == "auto" and self.dense_streak >= 3 and self.plans % self.probe_every != 0:
            return None
        if callable(touched):
            touched = touched()
        union = touched if touched.dtype == torch.uint8 else touched.ne(0).to(torch.uint8)
        if dist.get_backend(self.group) == "gloo" and union.is_cuda:
            self._staged(union, dist.ReduceOp.MAX)
        elif self.mask_as_bytes:
            try:
                dist.all_reduce(union, op=dist.ReduceOp.MAX, group=self.group)
            except (RuntimeError, ValueError, TypeError, NotImplementedError):
                # a backend without 8-bit MAX refuses the call on every rank alike, before anything is sent
                self.mask_as_bytes = False
        if not self.mask_as_bytes:
            wide = union.to(torch.int32)
            dist.all_reduce(wide, op=dist.ReduceOp.SUM, group=self.group)
            union = wide
        rows = torch.nonzero(union).view(-1)            # host-synchronous: the size of the exchange
        self.last_union = (int(rows.numel()), int(union.numel()))
        if self.sparse == "auto" and rows.numel() > self.sparse_below * union.numel():
            self.dense_streak += 1
            return None
        self.dense_streak = 0
        return rows

    def start_gathered(self, tensors, rows):
        """Sum over the ranks of the rows `rows` (from `plan_rows`: the same indices on every rank) of
        every tensor ([P, ...] each); the other rows must be zero on every rank and stay as they are."""
        if not self._active() or rows.numel() == 0:
            return self
        cols = [t.view(t.size(0), -1) for t in tensors if t is not None and t.numel() != 0]
        packed = torch.cat([c.index_select(0, rows) for c in cols], dim=1)

        def put_back():
            at = 0
            for c in cols:
                c.index_copy_(0, rows, packed[:, at:at + c.size(1)])
                at += c.size(1)

        if dist.get_backend(self.group) == "gloo" and packed.is_cuda:
            self._staged(packed)
            if self.average:
                packed.div_(dist.get_world_size(self.group))
            put_back()
            return self
        work = dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append(([packed], work, put_back))
        return self

    def wait(self):
        ws = dist.get_world_size(self.group) if dist.is_initialized() else 1
        for entry in self.pending:
            ts, work = entry[0], entry[1]
            work.wait()
            if self.average:
                for t in ts:
                    t.div_(ws)
            if len(entry) > 2:
                entry[2]()
        self.pending = []


def reduce_densify_stats(xyz_gradient_accum, denom, max_radii2D, group=None):
    """In-place reduction of the statistics densification reads."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    works = [dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM, group=group, async_op=True),
             dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group, async_op=True),
             dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group, async_op=True)]
    for w in works:
        w.wait()


def run_views(sink, settings_list, loss_of_view, streams=(), reducer=None, chunks=6):
    """The local views of one step through a `ParamGradSink` (raw-parameter rasterizer path):
    `begin_step`, the forward preprocess of all the views in one pass, `loss_of_view(i, settings)`
    (render view i with `forward_params(..., grad_sink=sink)` and return its scalar loss) and
    `backward()` per view, then `flush()`.  With `streams` (a list of `torch.cuda.Stream`) the
    views are dealt round-robin to the current stream and to them (three side streams make four
    lanes, the number of hardware queues a process has): the views are independent until the flush, and the
    kernels of a view leave much of the GPU idle (sort passes that wait on their look-back, the
    ramp-down of the blend kernels), so two or three views in flight finish sooner than one after
    the other.  Gradients are the same sums either way (float atomics: not bit for bit).
    The side streams are only used with a deferring sink: without deferral every view's backward
    writes the shared gradient buffers, which is only ordered on one stream.

    reducer: a `GradReducer`; the flush and the exchange of the gradients then go through
    `reduce_flush(sink, reducer, chunks)`.  Returns the detached losses."""
    sink.begin_step()
    sink.prepare_views(settings_list)
    main = torch.cuda.current_stream()
    streams = list(streams) if sink.defer else []
    lanes = [main] + streams
    if streams:
        ready = torch.cuda.Event()
        ready.record(main)
        for st in streams:
            st.wait_event(ready)
    losses = []
    for i, rs in enumerate(settings_list):
        with torch.cuda.stream(lanes[i % len(lanes)]):
            loss = loss_of_view(i, rs)
            loss.backward()
        losses.append(loss.detach())
    for st in streams:
        main.wait_stream(st)
    if reducer is None:
        sink.flush()
    else:
        reduce_flush(sink, reducer, chunks)
    return losses


def reduce_flush(sink, reducer, chunks=6, after_flush=None):
    """`sink.flush()` followed by the sum of the parameter gradients over the ranks; returns when the
    current stream has the sums.  When the views of this step (all ranks) reached few enough
    Gaussians, only the rows of those are exchanged (`GradReducer.plan_rows`); otherwise the flush
    runs over `chunks` ranges of Gaussians and each range's rows are exchanged while the later
    ranges are computed.  `after_flush()`, if given, is called when the per-Gaussian backward has been
    enqueued (a place to record an event).  Returns "sparse" or "dense"."""
    grads = sink.tensors()
    rows = reducer.plan_rows(sink.touched_rows)
    if rows is not None:
        sink.flush()
        if after_flush is not None:
            after_flush()
            after_flush = None
        reducer.start_gathered(grads, rows)
        mode = "sparse"
    else:
        sink.flush(chunks=chunks, on_chunk=lambda first, count: reducer.start_rows(grads, first, count))
        mode = "dense"
    if after_flush is not None:
        after_flush()
    reducer.wait()
    mark = getattr(sink, "mark_rows_dirty", None)
    active = getattr(reducer, "_active", None)   # a reducer that cannot say is taken to have exchanged
    if mark is not None and (active is None or active()):
        mark(rows)   # rows that other ranks reached are non-zero here now (dense: any row may be)
    return mode


def view_sharded_step(params, views, loss_of_view, group=None):
    """One multi-view step.  `params`: leaf tensors (requires_grad); `views`: the
    whole batch (every rank passes the same list); `loss_of_view(view)` renders
    one view from `params` and returns its scalar loss.  After the call each
    param.grad holds the sum over ALL views of the batch on every rank.
    Returns the local loss sum (a Python float is deliberately not produced:
    no host sync)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    for p in params:
        p.grad = None
    total = None
    for vi in shard_views(len(views), rank, world):
        loss = loss_of_view(views[vi])
        loss.backward()
        total = loss.detach() if total is None else total + loss.detach()
    for p in params:  # a rank with no view (more ranks than views) contributes zeros
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    GradReducer(group).start([p.grad for p in params]).wait()
    return total
