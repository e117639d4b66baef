"""Multi-GPU exchange for the operator path: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

What it replaces in the reference: the shuffle = hash-partitioned Arrow-IPC files served over Flight
(ballista/core/src/execution_plans/shuffle_writer.rs:328-392 -> shuffle_reader.rs:226-298).  On one
node the same exchange is (1) an all-to-all of per-destination row COUNTS, then (2) one variable-size
all-to-all per column buffer: every GPU drives all of its xGMI links at once (no ring), and nothing
is compressed or written to disk.  Aggregation states of low-cardinality group-bys (q1: 4 groups)
are merged with a tiny all-gather instead.

Only torch tensors move here; partition ids / permutations come from libgpuq (gpuq_partition_run).
"""
from .table import DeviceColumn, DeviceTable, type_width


def _dist():
    import torch.distributed as dist
    return dist


def world():
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def allgather_table(table, cap, group=None):
    """All-gather a small materialised fixed-width table (e.g. partial-aggregate states): every rank
    receives the concatenation of all ranks' rows.  `cap` = per-rank row capacity (>= max rows)."""
    import torch
    dist = _dist()
    rank, ws = world()
    if ws == 1:
        return table
    dev = table.columns[0].data.device
    n = table.num_rows
    if n > cap:
        raise ValueError("allgather_table: %d rows exceed cap %d" % (n, cap))
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    total = sum(counts)
    cols = []
    for c in table.columns:
        if c.offsets is not None:
            raise ValueError("allgather_table: materialise Utf8 as PACKED15 first")
        w = type_width(c.type)
        send = torch.zeros(cap * w, dtype=torch.uint8, device=dev)
        send[: n * w] = c.data[: n * w]
        recv = [torch.empty(cap * w, dtype=torch.uint8, device=dev) for _ in range(ws)]
        dist.all_gather(recv, send, group=group)
        data = torch.cat([r[: k * w] for r, k in zip(recv, counts)] + [torch.zeros(16, dtype=torch.uint8, device=dev)])
        validity = None
        if c.validity is not None:
            # validity bitmaps are re-packed from per-rank bit arrays (rows are not byte aligned across ranks)
            vb = (cap + 7) // 8
            vsend = torch.zeros(vb, dtype=torch.uint8, device=dev)
            vsend[: (n + 7) // 8] = c.validity[: (n + 7) // 8]
            vrecv = [torch.empty(vb, dtype=torch.uint8, device=dev) for _ in range(ws)]
            dist.all_gather(vrecv, vsend, group=group)
            bits = []
            shifts = torch.arange(8, dtype=torch.uint8, device=dev)
            for r, k in zip(vrecv, counts):
                b = ((r[:, None] >> shifts[None, :]) & 1).reshape(-1)[:k]
                bits.append(b)
            allb = torch.cat(bits)
            pad = (-allb.numel()) % 64
            allb = torch.cat([allb, torch.zeros(pad + 64, dtype=torch.uint8, device=dev)])
            validity = (allb.reshape(-1, 8) << shifts[None, :]).sum(dim=1).to(torch.uint8)
        cols.append(DeviceColumn(c.name, c.type, data, total, validity=validity, nullable=c.nullable, repr=c.repr))
    return DeviceTable(cols, total)


def exchange_partitions(parts, group=None):
    """Hash-repartition exchange.  parts[d] = materialised fixed-width DeviceTable destined for rank d
    (len(parts) == world size, same schema).  Returns the concatenation of what every rank sent here.
    Step 1: all-to-all of row counts.  Step 2: one all_to_all_single per column with split sizes."""
    import torch
    dist = _dist()
    rank, ws = world()
    if ws == 1:
        return parts[0]
    if len(parts) != ws:
        raise ValueError("need one partition per rank")
    dev = parts[0].columns[0].data.device
    send_counts = torch.tensor([p.num_rows for p in parts], dtype=torch.int64, device=dev)
    recv_counts = torch.empty(ws, dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc = [int(x) for x in send_counts.tolist()]
    rc = [int(x) for x in recv_counts.tolist()]
    total = sum(rc)
    cols = []
    for ci, c0 in enumerate(parts[0].columns):
        if c0.offsets is not None or c0.validity is not None:
            raise ValueError("exchange_partitions: fixed-width non-null columns only (materialise / PACKED15 first)")
        w = type_width(c0.type)
        send = torch.cat([p.columns[ci].data[: p.num_rows * w] for p in parts]) if sum(sc) else torch.zeros(0, dtype=torch.uint8, device=dev)
        recv = torch.empty(total * w + 16, dtype=torch.uint8, device=dev)
        dist.all_to_all_single(recv[: total * w], send, output_split_sizes=[k * w for k in rc], input_split_sizes=[k * w for k in sc], group=group)
        cols.append(DeviceColumn(c0.name, c0.type, recv, total, nullable=c0.nullable, repr=c0.repr))
    return DeviceTable(cols, total)
