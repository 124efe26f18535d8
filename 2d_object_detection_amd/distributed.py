"""Image-level data parallelism for the post-processing path.

Images are independent (the reference itself loops over them, detect_net.py:125), so a
batch is split into contiguous slices, one per rank (one process per GPU); decode, filter
and NMS run with no communication.  The only exchange is a final variable-length
all-gather of detection rows: first one count per rank, then one padded all-gather sized to
the largest count (both as all_gather_into_tensor).  Payloads are KB..MB, i.e. latency-bound on xGMI, hence two
collectives in total rather than one per image.  With the ``nccl`` backend (RCCL on ROCm)
the tensors stay on the GPU; the ``gloo`` backend is used by the CPU tests of this
index arithmetic.
"""
import torch
import torch.distributed as dist


def shard_bounds(batch, world_size, rank):
    """Contiguous slice [lo, hi) of a batch owned by ``rank``; the first ``batch %
    world_size`` ranks get one extra image."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    q, r = divmod(int(batch), int(world_size))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_rows(rows, count=None, group=None):
    """Variable-length all-gather along dim 0.

    ``rows`` is this rank's (n_r, W) tensor (or a (cap, W) buffer of which the first
    ``count`` rows are valid; ``count`` may be an int or a one-element tensor -- a tensor on
    ``rows``' device is used where it is, without a host round trip).  Returns
    (gathered (sum n_r, W) ordered by rank, counts list).  Every rank gets the same
    result.  Cost: two collectives and one host synchronisation (the counts, which size the
    second collective); a buffer with spare capacity is sent from where it is."""
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = rows.device
    cap = rows.shape[0]
    if torch.is_tensor(count) and count.device == dev:
        cnt = count.reshape(1).to(torch.int64).clamp(max=cap)
    else:
        n = cap if count is None else min(int(count), cap)
        cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    all_cnt = torch.empty((world,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_cnt, cnt, group=group)
    counts = [int(c) for c in all_cnt.tolist()]
    n = counts[rank]
    width = tuple(rows.shape[1:])
    m = max(counts)
    if m == 0:
        return rows.new_zeros((0,) + width), counts
    if cap >= m and rows.is_contiguous():
        send = rows[:m]            # rows past n are padding, stripped below
    else:
        send = rows.new_zeros((m,) + width)
        send[:n] = rows[:n]
    recv = rows.new_empty((world * m,) + width)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view((world, m) + width)
    return torch.cat([recv[r, :c] for r, c in enumerate(counts)], dim=0), counts


class PaddedRowGather:
    """All-gather of detection rows with a fixed per-rank capacity: no allocation and no
    host synchronisation per call, so it can follow the kernels on the stream.

    ``send`` is a (cap + 1, W) buffer: row 0 is a header whose first element carries the
    rank's row count, rows 1.. are the payload (``rows`` is that view; hand it to the engine
    as its output so nothing is copied).  ``gather`` enqueues ONE all_gather_into_tensor of
    the whole buffer; ``unpack`` (host side, after the work completed) strips the padding.
    Counts are carried as float32, exact below 2**24 rows per rank."""

    def __init__(self, cap_rows, width, device, group=None, dtype=torch.float32):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if not 0 < cap_rows < (1 << 24):
            raise ValueError("cap_rows must be in (0, 2**24)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.cap, self.width = int(cap_rows), int(width)
        self.send = torch.zeros((self.cap + 1, self.width), dtype=dtype, device=device)
        self.recv = torch.zeros((self.world * (self.cap + 1), self.width), dtype=dtype, device=device)

    @property
    def rows(self):
        return self.send[1:]

    def gather(self, count, async_op=False):
        """count: one-element tensor (any numeric dtype) on the buffer's device."""
        self.send[0, 0:1].copy_(count.reshape(1))
        return dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=async_op)

    def unpack(self):
        """(rows ordered by rank, true counts).  Rows a rank produced beyond ``cap`` were
        never sent; compare the counts with ``cap`` to detect that."""
        r = self.recv.view(self.world, self.cap + 1, self.width)
        counts = [int(c) for c in r[:, 0, 0].tolist()]
        return torch.cat([r[k, 1:1 + min(c, self.cap)] for k, c in enumerate(counts)], 0), counts


class PaddedSetGather:
    """``PaddedRowGather`` for the multi-set engine: the detection rows of ``n_sets`` batches --
    one launch of ``YoloV3PostMulti``, or several launches whose rows are exchanged together --
    leave in ONE all_gather_into_tensor.

    ``send`` is (H + n_sets * cap, W): the first H = ceil(n_sets / W) rows are a header that
    carries the per-set row counts (float32, exact below 2**24), then ``n_sets`` slabs of ``cap``
    rows (``rows`` is that (n_sets, cap, W) view: hand it, or a run of its slabs, to the engine as
    its output).  ``set_counts`` stages the counts of one launch, ``gather`` enqueues the
    collective behind whatever the current stream holds and returns without a host
    synchronisation when the backend is nccl (RCCL); nothing is allocated per call."""

    def __init__(self, n_sets, cap_rows, width, device, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if n_sets <= 0 or width <= 0:
            raise ValueError("n_sets and width must be positive")
        if not 0 < cap_rows < (1 << 24):
            raise ValueError("cap_rows must be in (0, 2**24)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.n_sets, self.cap, self.width = int(n_sets), int(cap_rows), int(width)
        self.header_rows = -(-self.n_sets // self.width)
        n = self.header_rows + self.n_sets * self.cap
        self.send = torch.zeros((n, self.width), dtype=torch.float32, device=device)
        self.recv = torch.zeros((self.world * n, self.width), dtype=torch.float32, device=device)

    @property
    def rows(self):
        return self.send[self.header_rows:].view(self.n_sets, self.cap, self.width)

    @property
    def nbytes(self):
        """Bytes one rank sends per collective."""
        return self.send.numel() * 4

    def set_counts(self, counts, first=0, n_used=None, n_slots=None):
        """Stages the counts of one launch: ``counts`` (numeric tensor on the buffer's device)
        holds one count per set of the launch, of which the first ``n_used`` (default all) were
        produced; they go to the header slots ``first ..``, and the launch's remaining slots (it
        owns ``n_slots``, default len(counts)) read as zero."""
        counts = counts.reshape(-1)
        slots = counts.numel() if n_slots is None else int(n_slots)
        n = counts.numel() if n_used is None else int(n_used)
        if first < 0 or first + slots > self.n_sets or n > slots:
            raise ValueError("counts do not fit the header")
        hdr = self.send[:self.header_rows].view(-1)
        if n < slots:
            hdr[first + n:first + slots].zero_()
        hdr[first:first + n].copy_(counts[:n])

    def clear_counts(self, first=0, n_slots=None):
        """Marks the header slots ``first .. first + n_slots`` (default: to the end) as unused."""
        last = self.n_sets if n_slots is None else first + int(n_slots)
        self.send[:self.header_rows].view(-1)[first:last].zero_()

    def gather(self, counts=None, n_used=None):
        """Enqueues the collective.  With ``counts`` (one launch that fills the whole buffer, of
        which the first ``n_used`` sets were produced) they are staged first."""
        if counts is not None:
            self.set_counts(counts, 0, n_used, self.n_sets)
        dist.all_gather_into_tensor(self.recv, self.send, group=self.group)

    def unpack(self):
        """[[rows of set s of rank r for s] for r] and the (world, n_sets) true counts (rows a
        rank produced beyond ``cap`` were never sent)."""
        r = self.recv.view(self.world, self.header_rows + self.n_sets * self.cap, self.width)
        counts = r[:, :self.header_rows].reshape(self.world, -1)[:, :self.n_sets].to(torch.int64).tolist()
        body = r[:, self.header_rows:].view(self.world, self.n_sets, self.cap, self.width)
        rows = [[body[k, s, :min(int(c), self.cap)] for s, c in enumerate(counts[k])]
                for k in range(self.world)]
        return rows, [[int(c) for c in row] for row in counts]


class ShardedYoloV3Post:
    """One rank's share of a global batch: a ``YoloV3Post`` over the rank's slice whose
    image indices are offset to global numbering, plus the final all-gather.  The gathered
    (D, 8) tensor is identical on every rank and identical to a single-GPU run over the
    whole batch (order: image, class id, objectness descending)."""

    def __init__(self, global_batch, inp_dim, num_classes=5, confidence=0.5, nms_conf=0.4,
                 device=None, group=None, **kw):
        from .util import YoloV3Post
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.lo, self.hi = shard_bounds(global_batch, self.world, self.rank)
        if self.hi <= self.lo:
            raise ValueError("global batch smaller than the number of ranks")
        self.engine = YoloV3Post(self.hi - self.lo, inp_dim, num_classes, confidence, nms_conf,
                                 device=device, batch_offset=self.lo, **kw)

    def run_local(self, local_heads):
        return self.engine.run(local_heads)

    def detect(self, local_heads):
        out, count = self.engine.run(local_heads)
        if self.world == 1:
            return out[:int(count.item())]
        gathered, _ = all_gather_rows(out, count, self.group)
        return gathered
