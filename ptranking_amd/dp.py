"""Data-parallel training of the hot path: one process per GPU, queries sharded across ranks, replicated scorer.

The reference is single-device (no torch.distributed call site anywhere); queries are independent
units in every in-scope loss, and every loss but one is a SUM over queries (lambdarank.py:56, listnet.py:39, ...; RankMSE is
a batch MEAN — its ranker re-scales around the collective, rankers.py), so the
only exchange step is ONE `all_reduce(SUM)` of the flattened parameter gradient per iteration —
136 KB for the 136-feature scorer: latency-bound on xGMI, so everything is kept in a single bucket.

`FlatGradBucket` makes every `p.grad` a view into one contiguous buffer, so the collective runs in place with no
flatten/unflatten copies; `extra` trailing floats carry scalars that must be reduced with the gradients (ApproxNDCG's
batch coupling: local sum(1/IDCG) and local sum(DCG)).
Backend: 'nccl' (= RCCL on ROCm) on GPUs, 'gloo' in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


# Optional profiling hook (bench.py): TIMING = [] makes every gradient all-reduce record a (start, end) HIP event pair on the
# current stream, so that a multi-GPU bench line can report what the exchange step costs.
TIMING = None


def _timed_all_reduce(t):
    if TIMING is not None and t.is_cuda:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ev1.record()
        TIMING.append((ev0, ev1))
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


# A process group of ONE rank normally takes the single-device path (no collective, optimiser step fused into the backward).  SINGLE_RANK_COLLECTIVES
# = True keeps the data-parallel path — gradient all-reduce over the group's backend, then the optimiser step — so that the RCCL code path can be
# executed on a one-GPU box (tests/test_dp_gpu.py, `bench.py --force-collectives`); the result is the single-device one (a sum over one rank).
SINGLE_RANK_COLLECTIVES = False


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or SINGLE_RANK_COLLECTIVES)


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun) and bind this process
    to its GPU (LOCAL_RANK).  Returns (rank, world_size, local_rank).  No-op for single-process runs."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
    if (ws > 1 or os.environ.get("PTR_DP_INIT_SINGLE") == "1") and not dist.is_initialized():     # PTR_DP_INIT_SINGLE=1: a group of one rank too
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:   # PTR_DP_BACKEND=gloo lets several ranks share one GPU (tests); production = nccl (RCCL over xGMI)
            backend = os.environ.get("PTR_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rk, world_size=ws, **kw)
    return rk, ws, local


ROW_KEY = 0x9E3779B1        # the per-row multiplier of the counter-based dropout generator (csrc/ptr_dropout.h drop_bits)
ROW_OFFSET = None           # explicit global index of this rank's first row; None: derived from QUERY_SHARD, else rank * local rows
QUERY_SHARD = None          # (first global query, number of local queries) of this rank's slice, recorded by shard_queries()


def fold_row_offset(seed, row0):
    """Dropout seed of a replica whose local row r is global row row0 + r: the generator keys an element by
    lowbias32(row * ROW_KEY + column/site terms + seed_lo), so adding row0 * ROW_KEY to the low seed word shifts the row index."""
    lo = (seed + row0 * ROW_KEY) & 0xFFFFFFFF
    return ((seed >> 32) << 32) | lo


def local_dropout_seed(seed, local_rows, local_queries=None):
    """The seed this rank passes to the dropout kernels for a batch of `local_rows` rows (documents for the pointwise scorers).  With
    every rank drawing the same base seed (same torch.manual_seed, as the reference's single process would) this makes rank r's row i
    use the mask of global row (rows of ranks < r) + i: replicas never share masks and N ranks x B/N reproduce
    one rank x B.  Equal shards: rank * local_rows; a slice taken with shard_queries() is placed by its recorded query offset (uneven
    splits included); ROW_OFFSET overrides both.  Single process: the seed unchanged."""
    if ROW_OFFSET is not None:
        return fold_row_offset(seed, int(ROW_OFFSET))
    if not is_distributed() or world_size() == 1:
        return seed
    local_rows = int(local_rows)
    if QUERY_SHARD is not None and QUERY_SHARD[1] > 0 and local_rows % QUERY_SHARD[1] == 0 and \
            (local_queries is None or int(local_queries) == QUERY_SHARD[1]):
        # (local_queries, where the caller knows it, must BE the recorded slice: a later batch whose row count merely happens to divide
        # by a stale slice's query count is placed by rank * rows like any equal split — ADVICE r4; end_step() drops the record)
        return fold_row_offset(seed, QUERY_SHARD[0] * (local_rows // QUERY_SHARD[1]))     # rows per query x queries in front of this rank
    return fold_row_offset(seed, rank() * local_rows)


def end_step():
    """Called by the train step once its forward passes are done (rankers.FusedStepMixin, DeviceTrainLoop): the slice recorded by
    shard_queries() described THIS step's batch only."""
    global QUERY_SHARD
    QUERY_SHARD = None


def seed_replica(seed):
    """Seed a data-parallel replica: the CPU generator IDENTICALLY on every rank (model initialisation; the base dropout seeds of our
    kernels, which local_dropout_seed then shifts to the rank's own global rows) and the CUDA generator DIFFERENTLY per rank, so that the
    torch.nn.Dropout modules left in a scoring function (listsf sublayers) do not repeat their masks across replicas either."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed + 7919 * rank())


def shard_queries(num_queries, rank_=None, world_=None):
    """Contiguous slice [lo, hi) of the query dimension owned by this rank (remainder spread over the first ranks)."""
    global QUERY_SHARD
    r = rank() if rank_ is None else rank_
    w = world_size() if world_ is None else world_
    base, rem = divmod(num_queries, w)
    lo = r * base + min(r, rem)
    hi = lo + base + (1 if r < rem else 0)
    if rank_ is None and world_ is None:
        # this rank's own slice: remembered so that local_dropout_seed can place the replica's rows in the global batch when the split is
        # uneven (rank * local rows would overlap the neighbours' mask windows) — no collective needed
        QUERY_SHARD = (lo, hi - lo)
    return lo, hi


class ViewGradBucket:
    """FlatGradBucket's interface over a gradient buffer that ALREADY is flat and owned by someone else (scorer.FlatViewAdam: the
    module parameters' .grad are views of it and the fused stack writes its gradients straight into it): `gbuf` = [numel gradient
    floats | spare tail], of which `extra` tail floats travel with the gradients in the one all-reduce.  Nothing is re-pointed."""

    def __init__(self, gbuf, numel, extra, on_zero, before_reduce=None):
        if gbuf.numel() < numel + extra:
            raise ValueError("gradient buffer has no room for the extra scalars")
        self.numel, self.extra = numel, extra
        self.flat = gbuf[:numel + extra]
        self._on_zero = on_zero
        self._before_reduce = before_reduce      # owner hook: gradients something re-pointed go back into the flat buffer first

    @property
    def extras(self):
        return self.flat[self.numel:]

    def zero(self):
        self._on_zero()                 # the owner's zero_grad(): memset + "next backward may write in place"
        if self.extra:
            self.extras.zero_()

    def all_reduce(self):
        if self._before_reduce is not None:
            self._before_reduce()
        if is_distributed():
            _timed_all_reduce(self.flat)


class FlatGradBucket:
    """All gradients of `params` as views of one flat fp32 buffer (+ `extra` trailing scalars)."""

    def __init__(self, params, extra=0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.extra = extra
        self.flat = torch.zeros(self.numel + extra, device=dev, dtype=dt)
        self.attach()

    def attach(self):
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                p.grad = view
            off += n

    @property
    def extras(self):
        return self.flat[self.numel:]

    def zero(self):
        self.flat.zero_()
        self.attach()     # re-alias in case something set a grad to None

    def all_reduce(self):
        if is_distributed():
            _timed_all_reduce(self.flat)


def all_reduce_sum(t):
    """In-place SUM all-reduce of one contiguous tensor (no-op when not distributed)."""
    if is_distributed():
        _timed_all_reduce(t)
    return t


def broadcast_parameters(params, src=0):
    """Make every replica start from rank `src`'s weights."""
    if is_distributed():
        for p in params:
            dist.broadcast(p.data, src=src)
