"""Batch-sharded data parallelism for the DLS_VAE step: one process per GPU, gradients exchanged with bucketed
RCCL all-reduces over xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

The reference is single-process / single-GPU (main_re.py:200), so this layer has no counterpart there.
Each rank runs the full model on its own 256 samples; the only exchange is the gradient reduction over the flat
gradient buffer of FlatAdam.  The 'seq'-averaged NLL divides by the LOCAL batch (criterions.py:64), so summed
gradients are divided by the world size; the clip norm / skip guards are then computed on the reduced gradient,
which is identical on every rank, so all ranks take the same decision without a further exchange.
xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU).  The process group runs its collectives one after another
on one stream, so splitting a stretch buys no concurrency and every call costs ~40 us of host time that the step cannot
hide: a stage travels as ONE all-reduce when its parameters are contiguous (64 MiB bucket limit).
"""
import os

import torch
import torch.distributed as dist


class LaunchError(RuntimeError):
    """The process layout cannot run the step (fewer GPUs than ranks, two ranks on one GPU, world size mismatch)."""


def device_identity(local):
    """What tells two GPUs of a node apart: PCI domain:bus:device (and the UUID when the runtime reports one)."""
    props = torch.cuda.get_device_properties(local)
    bus = tuple(getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    ident = "pci %04x:%02x:%02x" % bus if None not in bus else "device index %d" % local
    uuid = getattr(props, "uuid", None)
    return ident + (" uuid %s" % uuid if uuid is not None else "")


def check_one_rank_per_device(identity, group=None):
    """Every rank contributes the identity of the device it will use (an object all-gather: on a 'cpu:gloo,cuda:nccl'
    group it travels over gloo, BEFORE any RCCL communicator exists).  Raises LaunchError on every rank when two ranks
    name the same device: the persistent recurrence kernels need all CUs of a chip co-resident (two ranks on one GPU
    wait for each other's workgroups inside their bounded spins) and RCCL refuses or hangs on duplicate devices --
    gpurun_out/rehearse2.err of round 2 was exactly that layout.  Returns the identities in rank order."""
    world = dist.get_world_size(group)
    ids = [None] * world
    dist.all_gather_object(ids, identity, group=group)
    users = {}
    for r, d in enumerate(ids):
        users.setdefault(d, []).append(r)
    shared = sorted((rs, d) for d, rs in users.items() if len(rs) > 1)
    if shared:
        raise LaunchError("one rank per GPU is required, but " + "; ".join(
            "ranks %s share %s" % (",".join(map(str, rs)), d) for rs, d in shared))
    return ids


def init_from_env(backend=None, expect_world=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).  On GPUs the group
    is 'cpu:gloo,cuda:nccl': object collectives and barriers run over gloo, tensor collectives on device buffers over
    RCCL.  Fails loudly -- before any collective that could hang -- when the layout cannot work: WORLD_SIZE differs from
    `expect_world`, LOCAL_RANK names a GPU the node does not have, or two ranks would share a GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if expect_world is not None and world != expect_world:
        raise LaunchError("WORLD_SIZE = %d but %d ranks were asked for: start the ranks with torch.distributed.run "
                          "--nproc-per-node %d (bench.py --gpus N does that itself when WORLD_SIZE is unset)"
                          % (world, expect_world, expect_world))
    if world <= 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "cpu:gloo,cuda:nccl" if torch.cuda.is_available() else "gloo"
    on_gpu = "nccl" in backend
    if on_gpu:
        n_dev = torch.cuda.device_count()
        if local >= n_dev:
            raise LaunchError("rank %d (LOCAL_RANK %d) has no GPU of its own: this node shows %d GPU(s) for %d ranks; "
                              "one rank per GPU is required" % (rank, local, n_dev, world))
        torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if on_gpu:
        check_one_rank_per_device(device_identity(local))
    return rank, world, local


def rccl_rank_count(device):
    """All-reduce of ones on `device`: the number of ranks the collective backend really joined."""
    one = torch.ones(1, device=device)
    dist.all_reduce(one)
    return int(one.item())


def self_launch_command(n_gpus, script, argv, env, device_count, port=None):
    """What `python bench.py --gpus N` does when it was not started by torchrun (WORLD_SIZE unset).  Returns None when
    this process is itself a rank (or N = 1), else the command line that starts N fresh ranks -- decided before anything
    touches a GPU.  Raises LaunchError when the node has fewer than N GPUs: never a silent one-GPU run."""
    if n_gpus <= 1 or "WORLD_SIZE" in env:
        return None
    if device_count < n_gpus:
        raise LaunchError("--gpus %d asked for, but this node shows %d GPU(s): one rank per GPU is required"
                          % (n_gpus, device_count))
    if port is None:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    import sys
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)


def _host_staged(x, group):
    """A gloo group handed device tensors (the two-rank rehearsal on one card): the exchange goes through host memory."""
    return x.is_cuda and dist.get_backend(group) == "gloo"


class GatherRows(torch.autograd.Function):
    """[B_local, d] rows of every rank -> [world * B_local, d] in rank order, differentiable (config.global_negatives).

    The batch-coupled losses of the latent stacks -- DLS_Auto.cons_loss (dialog_models.py:3266-3272: a B x B score matrix,
    soft-max over rows and columns) and the Gumbel hard negatives of DeN_Auto (:2676-2685, :2722-2726: a row of ANOTHER
    sample as the negative) -- see only the local shard under batch sharding.  Gathering their inputs makes them see the
    global batch, which is what one process at batch world * B_local computes.
    Backward.  The step's parameter gradient is (1 / world) * sum over ranks of what each rank back-propagates
    (FlatGradReducer with average=True), while the loss meant is (1 / world) * sum_r L_r.  A rank's rows reach the
    losses of ALL ranks through the gathered copies, so its rows' gradient is sum_s dL_s / d rows_r: an all-reduce (sum)
    of the gathered gradient, of which the rank keeps its own slice.  replicated=True says every rank computes the SAME
    loss from the gathered rows (the global cons_loss): then all ranks hold the same gathered gradient and the sum is
    `world` times the local slice -- no collective in backward.
    All ranks must hold the same number of rows (a ragged last batch cannot be gathered; drop it or pad it)."""

    @staticmethod
    def forward(ctx, x, group=None, replicated=False):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        src = x.detach().contiguous()
        staged = _host_staged(src, group)
        send = src.cpu() if staged else src
        out = send.new_empty((world * send.shape[0],) + tuple(send.shape[1:]))
        dist.all_gather(list(out.chunk(world, 0)), send, group=group)
        ctx.meta = (group, world, rank, x.shape[0], bool(replicated))
        return out.to(x.device) if staged else out

    @staticmethod
    def backward(ctx, g):
        group, world, rank, rows, replicated = ctx.meta
        if replicated:
            return g[rank * rows:(rank + 1) * rows] * float(world), None, None
        g = g.contiguous().clone()
        if _host_staged(g, group):
            h = g.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            g = h.to(g.device)
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
        return g[rank * rows:(rank + 1) * rows].clone(), None, None


def global_rows(config, group=None):
    """-> (gather(x, replicated=False), row offset of this rank) when config.global_negatives is set and more than one
    rank runs, else (None, 0): the callers then keep the reference's per-batch form (the default; a shard IS a batch)."""
    if not getattr(config, "global_negatives", False) or not dist.is_available() or not dist.is_initialized():
        return None, 0
    world = dist.get_world_size(group)
    if world <= 1:
        return None, 0
    rank = dist.get_rank(group)

    def gather(x, replicated=False):
        return GatherRows.apply(x, group, replicated)

    gather.rank, gather.world = rank, world
    return gather, rank


def bucket_bounds(n, bucket_elems):
    """[(start, end)] covering [0, n) in order."""
    return [(s, min(s + bucket_elems, n)) for s in range(0, n, bucket_elems)]


def pack_sparse_rows(table_grad, ids):
    """What one rank contributes for a table whose gradient is non-zero only in the rows it looked up: (ids [n],
    rows [n, D]) where every distinct id carries its (already locally summed) gradient row once -- later
    occurrences of the same id carry zeros, so the merge may simply add all of them."""
    ids = ids.reshape(-1)
    n = ids.numel()
    pos = torch.arange(n, device=ids.device)
    first = torch.full((table_grad.shape[0],), n, dtype=torch.long, device=ids.device)
    first.scatter_reduce_(0, ids, pos, reduce="amin")
    keep = (first.index_select(0, ids) == pos).to(table_grad.dtype).unsqueeze(1)
    return ids, table_grad.index_select(0, ids) * keep


def merge_sparse_rows(table_grad, local_ids, ids_all, rows_all):
    """table_grad <- sum over ranks, given every rank's packed contribution (this rank's included): the locally touched
    rows are cleared and all contributions added in a fixed order, so every rank ends with the same bits.  On the GPU
    the add is the library's sorted segmented sum (pv_embedding_bwd, no atomics); CPU tensors (gloo rehearsal of the
    exchange logic) use index_add_, which is sequential there."""
    table_grad.index_fill_(0, local_ids.reshape(-1), 0.0)
    ids_all, rows_all = ids_all.reshape(-1).contiguous(), rows_all.reshape(-1, table_grad.shape[1]).contiguous()
    if table_grad.is_cuda:
        from . import _lib, ops
        n, (rows, D) = ids_all.numel(), table_grad.shape
        nbytes = _lib.load().pv_embedding_bwd_ws_bytes(n, D)
        ws = ops.workspace("dp_sparse_merge", nbytes, torch.uint8, table_grad.device)
        _lib.call("pv_embedding_bwd", ops._p(rows_all), D, ops._p(ids_all), n, n, n, 0, 1, rows, D, -1, 1,
                  ops._p(table_grad), ops._p(ws), nbytes, ops._stream())
    else:
        table_grad.index_add_(0, ids_all, rows_all)
    return table_grad


class FlatGradReducer(object):
    """Averages a flat fp32 gradient buffer across ranks: fixed-size buckets of all-reduce (async, then one wait) for
    the dense part, and for `sparse_tables` -- [(offset, rows, dim)] ranges of embedding tables whose local gradient is
    non-zero only in the rows looked up this step (user / item tables: 256 of 30000 / 20000 rows) -- one all-gather of
    (ids, rows) per step instead of all-reducing the whole table (0.26 MB instead of 2 x 25 MB per rank at
    the Yelp shape).  reduce(ids) takes the looked-up ids of each sparse table, in the same order.

    Overlap with backward: `early_ranges` -- [(start, end)] stretches whose gradient is final before the rest of the
    backward pass has run (the vocabulary projection: a third of the dense bytes, written by the first backward
    kernels).  start() launches their all-reduces and returns at once, so they travel over xGMI while the caller runs
    the remaining backward; finish(ids) exchanges everything else and waits for all of it.  reduce() = start + finish.
    `mid_ranges`: a second such stage (start_mid), for stretches that are final part-way through the remaining backward."""

    def __init__(self, flat_grad, world_size, bucket_bytes=64 << 20, group=None, sparse_tables=(), average=True,
                 early_ranges=(), mid_ranges=(), rehearse=False):
        # average=False leaves the SUM in the buffer: FlatAdam(grad_scale=1/world) applies the 1/world inside its guard
        # and Adam kernels, which saves a read-modify-write pass over the whole flat gradient (124 MB at the Yelp shape)
        # rehearse: issue the collectives even on a one-rank group (tools/rccl_selftest.py walks the N > 1 code path on one GPU)
        self.flat, self.world, self.group, self.average = flat_grad, world_size, group, average
        self.active = world_size > 1 or bool(rehearse)
        self.tables = sorted(tuple(t) for t in sparse_tables)
        be = max(1, bucket_bytes // 4)
        stages = [sorted((int(a), int(b)) for a, b in early_ranges), sorted((int(a), int(b)) for a, b in mid_ranges)]
        self.bounds, self.early_bounds, self.mid_bounds, at = [], [], [], 0
        staged = [self.early_bounds, self.mid_bounds]
        for off, rows, dim in self.tables + [(flat_grad.numel(), 0, 0)]:       # dense stretches between the tables
            if off < at:
                raise ValueError("FlatGradReducer: sparse tables overlap")
            cuts = sorted(set([at, off] + [x for st in stages for a, b in st for x in (a, b) if at < x < off]))
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                which = [k for k, st in enumerate(stages) if any(a <= lo and hi <= b for a, b in st)]
                if len(which) > 1:
                    raise ValueError("FlatGradReducer: early and mid ranges overlap")
                dst = staged[which[0]] if which else self.bounds
                dst.extend((lo + s, lo + e) for s, e in bucket_bounds(hi - lo, be))
            at = off + rows * dim
        if at > flat_grad.numel():
            raise ValueError("FlatGradReducer: sparse table beyond the buffer")
        for st, got in zip(stages, staged):
            for a, b in st:
                if sum(e - s for s, e in got if a <= s and e <= b) != b - a:
                    raise ValueError("FlatGradReducer: an early / mid range must lie inside one dense stretch")
        self._works, self._gathered, self._aux, self._sparse_buf = [], None, None, None
        self.external_sparse = False          # the caller runs pack_sparse() / merge_sparse() itself (inside its graphs)

    def table_view(self, i):
        off, rows, dim = self.tables[i]
        return self.flat[off:off + rows * dim].view(rows, dim)

    def _all_reduce(self, bounds):
        return [dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for s, e in bounds]

    def device_sparse(self, sparse_ids):
        """True when the sparse tables travel through pv_sparse_pack / pv_sparse_merge (device tensors, ids exact in fp32)."""
        return bool(self.flat.is_cuda and self.tables and all(rows <= (1 << 24) for _, rows, _ in self.tables)
                    and self.world * max(ids.numel() for ids in sparse_ids) <= 8192)

    def pack_sparse(self, sparse_ids):
        """The pack launches alone (device path), on the current stream -- for a caller that sets `external_sparse` and
        makes them (and merge_sparse) part of its own captured graphs: the table gradients must be final, and
        start_mid / finish then only issue the all-gather.  The buffers are allocated on the first call."""
        if not self.active:
            return
        from . import _lib, ops
        lay, at = [], 0
        for (off, rows, dim), ids in zip(self.tables, sparse_ids):
            n = ids.numel()
            lay.append((at, n, dim, rows))
            at += n * (dim + 1)
        if self._sparse_buf is None or self._sparse_buf[0] != lay:
            dev = self.flat.device
            self._sparse_buf = (lay, torch.empty(at, dtype=torch.float32, device=dev),
                                torch.empty(self.world * at, dtype=torch.float32, device=dev))
        _, send, recv = self._sparse_buf
        for i, ((o, n, dim, rows), ids) in enumerate(zip(lay, sparse_ids)):
            ids_c = ids.reshape(-1).contiguous()
            _lib.call("pv_sparse_pack", ops._p(self.table_view(i)), rows, dim, ops._p(ids_c), n, ops._p(send[o:o + n]),
                      ops._p(send[o + n:o + n * (dim + 1)]), ops._stream())

    def _gather_sparse_device(self, sparse_ids):
        """On the device: one pack launch per table into ONE send buffer ([ids as fp32 | rows] per table), one all-gather,
        and (finish) one merge launch per table -- pv_sparse_pack / pv_sparse_merge.  The buffers are allocated once: the
        host side of an exchange is a handful of calls, all of which the step cannot hide."""
        if not self.external_sparse:
            self.pack_sparse(sparse_ids)
        _, send, recv = self._sparse_buf
        self._works.append(dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True))
        self._gathered = "device"

    def merge_sparse(self):
        """The merge launches alone (see pack_sparse): after finish() on the same stream."""
        if self.active:
            self._merge_sparse_device()

    def _merge_sparse_device(self):
        from . import _lib, ops
        lay, send, recv = self._sparse_buf
        for i, (o, n, dim, rows) in enumerate(lay):
            _lib.call("pv_sparse_merge", ops._p(recv[o:]), self.world, send.numel(), n, dim, rows, ops._p(self.table_view(i)),
                      ops._stream())

    def _gather_sparse(self, sparse_ids):
        if len(sparse_ids) != len(self.tables):
            raise ValueError("FlatGradReducer: %d sparse tables need %d id tensors" % (len(self.tables), len(sparse_ids)))
        if self.device_sparse(sparse_ids):
            return self._gather_sparse_device(sparse_ids)
        packed = [pack_sparse_rows(self.table_view(i), ids) for i, ids in enumerate(sparse_ids)]
        gathered = []
        for ids, rows in packed:                         # every rank looks up the same number of ids (its batch)
            ids_all = torch.empty(self.world * ids.shape[0], dtype=ids.dtype, device=ids.device)
            rows_all = torch.empty((self.world * rows.shape[0], rows.shape[1]), dtype=rows.dtype, device=rows.device)
            self._works.append(dist.all_gather_into_tensor(ids_all, ids.contiguous(), group=self.group, async_op=True))
            self._works.append(dist.all_gather_into_tensor(rows_all, rows.contiguous(), group=self.group, async_op=True))
            gathered.append((ids, ids_all, rows_all))
        self._gathered = gathered

    def _aux_stream(self, after):
        """The reducer's own stream, ordered after `after` (an event; the caller's current stream when None).  Collectives
        issued under it are ordered after that point only -- RCCL's stream follows the stream that is current at the call
        -- so the caller may already have queued later work (its next backward graph) on its own stream: the host then
        never stands between two graphs with a dozen launches of its own."""
        if self._aux is None:
            self._aux = torch.cuda.Stream(device=self.flat.device)
        if after is None:
            self._aux.wait_stream(torch.cuda.current_stream(self.flat.device))
        else:
            self._aux.wait_event(after)
        return torch.cuda.stream(self._aux)

    def start(self, after=None):
        """Launch the all-reduces of the early ranges.  Their gradient must be complete at `after` (an event recorded on
        the caller's stream), or on the current stream when no event is given."""
        if self.active:
            if self.flat.is_cuda and after is not None:
                with self._aux_stream(after):
                    self._works += self._all_reduce(self.early_bounds)
            else:
                self._works += self._all_reduce(self.early_bounds)

    def start_mid(self, sparse_ids=None, after=None):
        """Second stage of the overlap schedule: the mid ranges -- and, with sparse_ids, the sparse tables -- are final at
        `after` / on the current stream although the backward pass has not finished (DLS_VAE: everything but the
        utterance encoder and the word table, once the latent stack has run backward).  Returns at once."""
        if self.active:
            if self.flat.is_cuda:
                # also keeps the packing launches (a dozen small kernels) off the caller's stream
                with self._aux_stream(after):
                    self._works += self._all_reduce(self.mid_bounds)
                    if sparse_ids is not None:
                        self._gather_sparse(sparse_ids)
            else:
                self._works += self._all_reduce(self.mid_bounds)
                if sparse_ids is not None:
                    self._gather_sparse(sparse_ids)

    def finish(self, sparse_ids=()):
        if not self.active:
            return
        self._works += self._all_reduce(self.bounds)
        if self._gathered is None:
            self._gather_sparse(sparse_ids)
        works, self._works = self._works, []
        gathered, self._gathered = self._gathered, None
        for w in works:
            w.wait()
        if self._aux is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._aux)
        if gathered == "device":
            if not self.external_sparse:
                self._merge_sparse_device()
        else:
            for i, (ids, ids_all, rows_all) in enumerate(gathered):
                merge_sparse_rows(self.table_view(i), ids, ids_all, rows_all)
        if self.average:
            self.flat.mul_(1.0 / self.world)

    def reduce(self, sparse_ids=()):
        self.start()
        self.finish(sparse_ids)


def broadcast_parameters(flat_params, src=0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
