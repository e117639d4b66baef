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


class Comm:
    """gpuq_comm (include/gpuq.h, "exchange"): the ranks of the node as the native library sees them.  With the "nccl" backend
    it is an RCCL communicator created inside libgpuq (rank 0 makes the id, torch.distributed's store only carries the 128
    bytes); otherwise (gloo: CPU tests, several ranks on one GPU) a host-staged transport whose all-to-all is
    torch.distributed's -- the exchange logic in csrc/exchange.cpp is the same either way."""

    def __init__(self, tc, group=None):
        import ctypes as C
        import torch
        self.tc, self.group = tc, group
        L = tc.ctx.L
        dist = _dist()
        self.rank, self.world = world()
        h = C.c_void_p()
        backend = dist.get_backend(group) if self.world > 1 else "none"
        self.backend = backend
        if backend == "nccl" or (self.world == 1 and _rccl_wanted()):
            ident = (C.c_uint8 * 128)()
            if self.rank == 0:
                self._xcheck(L.gpuq_comm_unique_id(ident))
            if self.world > 1:
                box = [bytes(ident)]
                dist.broadcast_object_list(box, src=0, group=group)
                ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
            self._xcheck(L.gpuq_comm_create(tc.ctx.h, ident, self.rank, self.world, C.byref(h)))
            self.transport = "rccl"
        else:
            from .binding import gpuq_column  # noqa: F401  (ctypes structs are defined there)
            CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.c_int)

            def a2av(_user, send, scnt, recv, rcnt, w):
                try:
                    sc = [int(scnt[i]) for i in range(w)]
                    rc = [int(rcnt[i]) for i in range(w)]
                    st = torch.frombuffer((C.c_uint8 * max(1, sum(sc))).from_address(send), dtype=torch.uint8)[: sum(sc)] if sum(sc) else torch.zeros(0, dtype=torch.uint8)
                    rt = torch.frombuffer((C.c_uint8 * max(1, sum(rc))).from_address(recv), dtype=torch.uint8)[: sum(rc)] if sum(rc) else torch.zeros(0, dtype=torch.uint8)
                    if w == 1:
                        rt.copy_(st)
                    else:
                        dist.all_to_all_single(rt, st.contiguous(), output_split_sizes=rc, input_split_sizes=sc, group=group)
                    return 0
                except Exception:      # noqa: BLE001 -- must not unwind into C
                    import traceback
                    traceback.print_exc()
                    return 1

            class Transport(C.Structure):
                _fields_ = [("user", C.c_void_p), ("all_to_all_v", CB)]
            self._cb = CB(a2av)
            self._tr = Transport(None, self._cb)
            self._xcheck(L.gpuq_comm_create_host(tc.ctx.h, C.byref(self._tr), self.rank, self.world, C.byref(h)))
            self.transport = "host"
        self.h = h

    def _xcheck(self, rc):
        if rc != 0:
            from .binding import GpuqError
            raise GpuqError(rc, (self.tc.ctx.L.gpuq_exchange_last_error() or b"").decode())

    def _table_arrays(self, table):
        import ctypes as C
        from . import binding as B
        from .table import type_id
        n = len(table.columns)
        cols, fields, keep = (B.gpuq_column * max(1, n))(), (B.gpuq_field_info * max(1, n))(), []
        for i, c in enumerate(table.columns):
            cc = c.to_c()
            cols[i] = cc
            keep.append(cc)
            tid, p, s_ = type_id(c.type)
            fields[i].name = c.name.encode()[:255]
            fields[i].type, fields[i].precision, fields[i].scale, fields[i].nullable, fields[i].repr = tid, p, s_, 1 if c.nullable else 0, c.repr
        return cols, fields, keep

    def _wrap(self, h, like):
        """gpuq_table -> DeviceTable whose columns alias the library's buffers (freed with the table)."""
        import ctypes as C
        import torch
        from . import binding as B
        L = self.tc.ctx.L
        n = int(L.gpuq_table_num_rows(h))
        owner = _OwnedTable(L, h)
        cols = []
        for i, c in enumerate(like.columns):
            cc = B.gpuq_column()
            L.gpuq_table_column(h, i, C.byref(cc), None)
            w = type_width(c.type) if c.offsets is None else 0

            def alias(ptr, nb):
                class _A:
                    pass
                a = _A()
                a.__cuda_array_interface__ = {"shape": (int(max(nb, 1)),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
                a.owner = owner
                return torch.as_tensor(a, device=self.tc.device)
            bm = ((n + 63) // 64) * 8 + 8
            if cc.offsets:
                offs = alias(cc.offsets, (n + 1) * 4).view(torch.int32)
                nbytes = int(offs[n].item()) if n else 0
                data = alias(cc.data, nbytes + 16)
            else:
                offs = None
                data = alias(cc.data, bm if w == 0 else n * w + 16)
            validity = alias(cc.validity, bm) if cc.validity else None
            cols.append(DeviceColumn(c.name, c.type, data, n, offsets=offs, validity=validity, nullable=c.nullable, repr=c.repr))
        t = DeviceTable(cols, n)
        t._keep = owner
        return t

    def exchange_partitions(self, grouped, dest_offsets):
        """grouped: materialised DeviceTable whose rows [dest_offsets[d], dest_offsets[d+1]) go to rank d."""
        import ctypes as C
        cols, fields, keep = self._table_arrays(grouped)
        off = (C.c_int64 * (self.world + 1))(*[int(x) for x in dest_offsets])
        h = C.c_void_p()
        self._xcheck(self.tc.ctx.L.gpuq_exchange_partitions(self.h, self.tc.stream_ptr(), cols, fields, len(grouped.columns), off, C.byref(h)))
        return self._wrap(h, grouped)

    def allgather(self, table):
        import ctypes as C
        cols, fields, keep = self._table_arrays(table)
        h = C.c_void_p()
        self._xcheck(self.tc.ctx.L.gpuq_allgather_table(self.h, self.tc.stream_ptr(), cols, fields, len(table.columns), table.num_rows, C.byref(h)))
        return self._wrap(h, table)

    def close(self):
        if getattr(self, "h", None):
            self.tc.ctx.L.gpuq_comm_free(self.h)
            self.h = None


def _rccl_wanted():
    import os
    return os.environ.get("GPUQ_COMM_TRANSPORT", "rccl") == "rccl"


class _OwnedTable:
    def __init__(self, L, h):
        self.L, self.h = L, h

    def __del__(self):
        try:
            if self.h:
                self.L.gpuq_table_free(self.h)
                self.h = None
        except Exception:      # noqa: BLE001
            pass


def allgather_table(table, cap, group=None):
    """All-gather a small materialised fixed-width table (e.g. partial-aggregate states): every rank receives the
    concatenation, in rank order, of all ranks' rows.  `cap` = per-rank row capacity (>= max rows on any rank, the SAME
    value on every rank).

    ONE collective: a rank's row count and all of its column buffers travel as a single fixed-layout record
    (table.record_layout, a pure function of schema and cap).  An aggregate result produced with output capacity `cap`
    already IS such a record and is shipped as it lies; anything else is packed into one first.  The received records are
    not unpacked: the result is a late-materialised view whose columns point into the receive buffer and whose index
    vectors (one per value width) enumerate the live rows, so q1's merge of partial states costs one all-gather, one
    count read-back and the upload of the index vectors."""
    import torch
    from .table import record_layout, RECORD_HEADER
    dist = _dist()
    rank, ws = world()
    if ws == 1:
        return table
    dev = table.columns[0].data.device
    n = table.num_rows
    if n > cap:
        raise ValueError("allgather_table: %d rows exceed cap %d" % (n, cap))
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        # gloo has no device all-gather: stage through host memory (CPU tests / several ranks sharing one GPU)
        host = DeviceTable([DeviceColumn(c.name, c.type, c.data.cpu(), c.length, offsets=c.offsets, validity=c.validity.cpu() if c.validity is not None else None,
                                         nullable=c.nullable, repr=c.repr) for c in table.columns], n)
        out = _materialize_gathered(allgather_table(host, cap, group=group))
        return DeviceTable([DeviceColumn(c.name, c.type, c.data.to(dev), c.length, validity=c.validity.to(dev) if c.validity is not None else None,
                                         nullable=c.nullable, repr=c.repr) for c in out.columns], out.num_rows)
    if any(c.offsets is not None for c in table.columns):
        raise ValueError("allgather_table: materialise Utf8 as PACKED15 first")
    widths = [type_width(c.type) for c in table.columns]
    B, lay = record_layout([(w, c.nullable) for w, c in zip(widths, table.columns)], cap)
    rec = getattr(table, "_record", None)
    if rec is not None and rec[1] == cap and rec[0].numel() >= B and not table.is_view():
        send = rec[0][:B]        # the allocation may be larger than the record (pooled block)
    else:
        send = torch.zeros(B, dtype=torch.uint8, device=dev)
        for c, w, (doff, dbytes, voff, vbytes) in zip(table.columns, widths, lay):
            k = n * w if w else (n + 7) // 8
            send[doff: doff + k] = c.data[:k]
            if voff >= 0:
                if c.validity is not None:
                    send[voff: voff + (n + 7) // 8] = c.validity[: (n + 7) // 8]
                else:
                    send[voff: voff + (n + 7) // 8] = 255
    send[:8] = torch.tensor([n], dtype=torch.int64).view(torch.uint8).to(dev, non_blocking=True)
    recv = torch.empty(ws * B, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    counts = [int(x) for x in recv.view(ws, B)[:, :8].contiguous().view(torch.int64).flatten().tolist()]     # the one host round trip
    return unpack_records(table.columns, recv, counts, cap)


def unpack_records(columns, recv, counts, cap):
    """Records of len(counts) ranks, back to back in `recv`, as one table of sum(counts) rows in rank order.  `columns`
    supplies names / types / nullability (any rank's table).  Non-nullable 4/8/16-byte columns are not copied: the result
    is a view into `recv`."""
    import torch
    from .table import record_layout
    dev = recv.device
    ws = len(counts)
    widths = [type_width(c.type) for c in columns]
    B, lay = record_layout([(w, c.nullable) for w, c in zip(widths, columns)], cap)
    recs = recv.view(ws, B)
    total = sum(counts)
    simple = all(w in (4, 8, 16) for w in widths) and not any(c.nullable for c in columns)
    if simple:
        # view: element (rank r, row i) of a w-byte column sits at index r * (B / w) + i from the column's first record
        classes = sorted(set(widths))
        vias = []
        for w in classes:
            pieces = [torch.arange(k, dtype=torch.int32) + r * (B // w) for r, k in enumerate(counts) if k > 0]      # no per-row Python work
            idx = torch.cat(pieces) if pieces else torch.zeros(1, dtype=torch.int32)
            vias.append(idx.to(dev, non_blocking=True))
        cols = [DeviceColumn(c.name, c.type, recv[doff:], total, nullable=False, repr=c.repr) for c, (doff, _, _, _) in zip(columns, lay)]
        out = DeviceTable(cols, total, via=vias, sides=[classes.index(w) + 1 for w in widths], dense=True)
        out._keep = recv
        return out
    pad = torch.zeros(16, dtype=torch.uint8, device=dev)
    shifts = torch.arange(8, dtype=torch.uint8, device=dev)
    vb = (cap + 7) // 8

    def join_bits(o):
        # pieces are not byte aligned across ranks: unpack, concatenate, re-pack (tiny arrays)
        bits = [((recs[r, o: o + vb][:, None] >> shifts[None, :]) & 1).reshape(-1)[:k] for r, k in enumerate(counts)]
        allb = torch.cat(bits)
        allb = torch.cat([allb, torch.zeros((-allb.numel()) % 64 + 64, dtype=torch.uint8, device=dev)])
        return (allb.reshape(-1, 8) << shifts[None, :]).sum(dim=1).to(torch.uint8)
    cols = []
    for c, w, (doff, dbytes, voff, vbytes) in zip(columns, widths, lay):
        data = torch.cat([recs[r, doff: doff + k * w] for r, k in enumerate(counts)] + [pad]) if w else join_bits(doff)
        validity = join_bits(voff) if voff >= 0 else None
        cols.append(DeviceColumn(c.name, c.type, data, total, validity=validity, nullable=c.nullable, repr=c.repr))
    return DeviceTable(cols, total)


def _materialize_gathered(t):
    """Host-side (CPU tensors) flattening of the view allgather_table returns, for the gloo staging path."""
    import torch
    if not t.is_view():
        return t
    cols = []
    for c, sd in zip(t.columns, t.sides):
        w = type_width(c.type)
        idx = t.via[sd - 1].to(torch.int64)
        flat = c.data[: (c.data.numel() // w) * w].view(-1, w)[idx].reshape(-1)
        cols.append(DeviceColumn(c.name, c.type, torch.cat([flat, torch.zeros(16, dtype=torch.uint8)]), t.num_rows, nullable=False, repr=c.repr))
    return DeviceTable(cols, t.num_rows)


def _a2a(send, send_splits, recv_splits, dev, group=None):
    """Variable-size all-to-all of a uint8 tensor.  RCCL moves device memory directly; under gloo (CPU tests, or several
    ranks sharing one GPU in the 2-process GPU test) the bytes are staged through host memory."""
    import torch
    dist = _dist()
    total = int(sum(recv_splits))
    staged = send.is_cuda and dist.get_backend(group) == "gloo"
    if staged:
        send = send.cpu()
    recv = torch.empty(total, dtype=torch.uint8, device=send.device)
    dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=[int(x) for x in recv_splits], input_split_sizes=[int(x) for x in send_splits], group=group)
    return recv.to(dev) if staged else recv


def exchange_partitions(parts, group=None, tc=None):
    """Hash-repartition exchange.  parts[d] = materialised fixed-width DeviceTable destined for rank d
    (len(parts) == world size, same schema).  Returns the concatenation (in rank order) of what every rank sent here.
    Step 1: all-to-all of row counts.  Step 2: one variable-size all-to-all per column buffer.  Validity bitmaps travel as
    whole bytes per piece and are re-joined at bit granularity with gpuq_concat_bitmap (needs `tc`)."""
    import torch
    dist = _dist()
    rank, ws = world()
    if ws == 1:
        return parts[0]
    if len(parts) != ws:
        raise ValueError("need one partition per rank")
    dev = parts[0].columns[0].data.device
    staged = dev.type == "cuda" and dist.get_backend(group) == "gloo"
    send_counts = torch.tensor([p.num_rows for p in parts], dtype=torch.int64, device="cpu" if staged else dev)
    recv_counts = torch.empty(ws, dtype=torch.int64, device=send_counts.device)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc = [int(x) for x in send_counts.tolist()]
    rc = [int(x) for x in recv_counts.tolist()]
    total = sum(rc)
    pad = torch.zeros(16, dtype=torch.uint8, device=dev)
    vb = lambda k: (k + 7) // 8

    def exchange_bitmap(pieces):
        """pieces[d]: bit-packed uint8 tensor of parts[d].num_rows bits, or None (all ones).  Bitmaps travel as whole bytes
        per piece and are re-joined at bit granularity on the device."""
        if tc is None:
            raise ValueError("exchange_partitions: nullable / Boolean columns need a TaskContext (bitmap re-join runs on the device)")
        send = [pc[: vb(p.num_rows)] if pc is not None else torch.full((vb(p.num_rows),), 255, dtype=torch.uint8, device=dev) for pc, p in zip(pieces, parts)]
        vrecv = _a2a(torch.cat(send) if send else torch.zeros(0, dtype=torch.uint8, device=dev), [vb(k) for k in sc], [vb(k) for k in rc], dev, group)
        out = torch.zeros(((total + 63) // 64) * 8 + 8, dtype=torch.uint8, device=dev)
        boff, bit = 0, 0
        for k in rc:
            if k > 0:
                piece = torch.cat([vrecv[boff: boff + vb(k)], pad])           # own storage: 8-byte readable tail
                tc.ctx.check(tc.ctx.L.gpuq_concat_bitmap(tc.ctx.h, tc.stream_ptr(), out.data_ptr(), bit, piece.data_ptr(), k))
                tc.sync()                                                       # `piece` is released after this iteration
            boff += vb(k); bit += k
        return out
    cols = []
    for ci, c0 in enumerate(parts[0].columns):
        pcs = [p.columns[ci] for p in parts]
        if any(c.offsets is not None for c in pcs):
            raise ValueError("exchange_partitions: fixed-width columns only (materialise Utf8 as PACKED15 first)")
        w = type_width(c0.type)
        if w == 0:       # Boolean: bit-packed values
            data = exchange_bitmap([c.data for c in pcs])
        else:
            send = torch.cat([c.data[: p.num_rows * w] for c, p in zip(pcs, parts)]) if sum(sc) else torch.zeros(0, dtype=torch.uint8, device=dev)
            data = torch.cat([_a2a(send, [k * w for k in sc], [k * w for k in rc], dev, group), pad])
        # every rank must take the same branch: nullability is a property of the schema, not of the data
        nullable = any(c.nullable for c in pcs)
        validity = exchange_bitmap([c.validity for c in pcs]) if nullable else None
        cols.append(DeviceColumn(c0.name, c0.type, data, total, validity=validity, nullable=nullable, repr=c0.repr))
    out = DeviceTable(cols, total)
    out.recv_counts = rc          # rows received from every rank, in rank order (the runs of an ordered fan-in)
    return out


def repartition_exchange(tc, table, hash_expr, group=None, comm=None):
    """RepartitionExec(Hash(hash_expr, world)) followed by the exchange: returns the rows of ALL ranks whose key hashes to
    this rank (planner.rs:137-151 splits a stage here; the reference then writes/reads shuffle files over Flight).
    comm (a Comm): the exchange runs inside libgpuq (gpuq_exchange_partitions: RCCL, or the host-staged transport); without
    it the torch.distributed path below moves the tensors (CPU tests)."""
    from . import plan as PL
    rank, ws = world()
    if comm is not None:
        perm, offs = PL.partition_perm(tc, table, hash_expr, comm.world)
        grouped = PL.materialize(tc, PL._select_view(tc, table, perm[: table.num_rows], table.num_rows))
        return comm.exchange_partitions(grouped, offs)
    if ws == 1:
        return table
    views = PL.partition_table(tc, table, hash_expr, ws)
    parts = [PL.materialize(tc, v, force=True) for v in views]
    return exchange_partitions(parts, group=group, tc=tc)


def broadcast_table(tc, table, cap=None, group=None, comm=None):
    """Every rank receives all ranks' rows (CollectLeft build sides below the broadcast threshold, config.rs:198-200)."""
    from . import plan as PL
    rank, ws = world()
    if comm is not None:
        return comm.allgather(PL.materialize(tc, table))
    if ws == 1:
        return table
    t = PL.materialize(tc, table, force=True)
    if cap is None:
        import torch
        dist = _dist()
        dev = t.columns[0].data.device
        staged = dev.type == "cuda" and dist.get_backend(group) == "gloo"
        m = torch.tensor([t.num_rows], dtype=torch.int64, device="cpu" if staged else dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
        cap = max(1, int(m.item()))
    return allgather_table(t, cap, group=group)


def partitioned_hash_join(tc, left, right, on, join_type="Inner", filter=None, group=None, comm=None):
    """HashJoinExec(PartitionMode::Partitioned) across ranks: both inputs are repartitioned on their join keys with the
    SAME partition function, exchanged, and joined locally.  `on` = [(left_expr, right_expr)].  Returns this rank's share
    of the join result (a view)."""
    from . import plan as PL
    lt = repartition_exchange(tc, left, [l for l, _ in on], group=group, comm=comm)
    rt = repartition_exchange(tc, right, [r for _, r in on], group=group, comm=comm)
    L, R = PL.MemoryExec([lt]), PL.MemoryExec([rt])
    ls, rs = L.schema(), R.schema()
    from . import expr as E
    on2 = [(E.rebind(l, ls), E.rebind(r, rs)) for l, r in on]
    return PL.HashJoinExec(L, R, on2, filter, join_type, "Partitioned", False).execute(0, tc)


def distributed_sort(tc, table, sort_expr, samples_per_rank=1024, group=None):
    """SortExec + SortPreservingMergeExec across ranks: range partitioning.
      1. every rank contributes `samples_per_rank` strided sample rows; all ranks sort the gathered samples and take the
         same world-1 splitters;
      2. the local rows and the splitters are sorted TOGETHER (splitters tagged to sort after equal rows): the positions
         of the splitters in the result are the range boundaries of the locally sorted run;
      3. range r of every rank goes to rank r (one exchange); the received sorted runs are merged.
    The result is this rank's range, sorted; ranks hold ascending, non-overlapping ranges (rank order = global order)."""
    import torch
    from . import plan as PL
    from . import expr as E
    rank, ws = world()
    if ws == 1:
        return PL.sort_table(tc, table, sort_expr)
    t = PL.materialize(tc, table, force=True)
    schema = t.plain_schema()
    n = t.num_rows
    dev = tc.device
    # 1. samples -> splitters (identical on every rank: same gathered rows in rank order, same stable sort)
    k = min(n, samples_per_rank)
    idx = (torch.arange(k, dtype=torch.int64, device=dev) * max(1, n // max(k, 1))).to(torch.int32) if k > 0 else torch.zeros(1, dtype=torch.int32, device=dev)
    samp = PL.materialize(tc, PL._select_view(tc, t, idx[:k], k), force=True)
    allsamp = allgather_table(samp, cap=samples_per_rank, group=group)
    ssorted = PL.materialize(tc, PL.sort_table(tc, allsamp, sort_expr), force=True)
    m = ssorted.num_rows
    cut = [min(m - 1, max(0, (j * m) // ws)) for j in range(1, ws)] if m > 0 else []
    # 2. sort rows + splitters together; tag column: 0 = row, 1 = splitter
    tag_name = "__gpuq_splitter"
    def with_tag(tab, val):
        tagcol = DeviceColumn(tag_name, "Int32", torch.full((max(1, tab.num_rows) + 4,), val, dtype=torch.int32, device=dev).view(torch.uint8), tab.num_rows, nullable=False)
        return DeviceTable(list(tab.columns) + [tagcol], tab.num_rows)
    if cut:
        ci = torch.tensor(cut, dtype=torch.int32, device=dev)
        spl = PL.materialize(tc, PL._select_view(tc, ssorted, ci, len(cut)), force=True)
        both = PL.concat_tables(tc, [with_tag(t, 0), with_tag(spl, 1)])
    else:
        both = with_tag(t, 0)
    bs = both.plain_schema()
    keys = [dict(s, expr=E.rebind(s["expr"], bs)) for s in sort_expr] + [{"expr": E.col(tag_name, bs), "asc": True, "nulls_first": False}]
    run = PL.sort_table(tc, both, keys)
    tags = PL.materialize(tc, PL._select_view(tc, DeviceTable([both.columns[-1]], both.num_rows), run.via[0], run.num_rows), force=True)
    is_spl = PL.filter_table(tc, tags, E.binary(E.col(tag_name, tags.plain_schema()), E.Operator.Eq, E.lit(1, "Int32")))
    pos = is_spl.via[0][: is_spl.num_rows].tolist() if is_spl.num_rows else []      # positions of the splitters in the sorted run
    bounds = [0] + [p - j for j, p in enumerate(pos)] + [n]    # boundaries in the run with the splitters removed
    rows_only = PL.filter_table(tc, run, E.binary(E.col(tag_name, run.plain_schema()), E.Operator.Eq, E.lit(0, "Int32")))
    rows_only = DeviceTable(rows_only.columns[:-1], rows_only.num_rows, via=rows_only.via, sides=rows_only.sides[:-1])
    # 3. ranges -> owners, merge the received runs
    parts = [PL.materialize(tc, PL.slice_table(tc, rows_only, bounds[r], bounds[r + 1] - bounds[r]), force=True) for r in range(ws)]
    mine = exchange_partitions(parts, group=group, tc=tc)
    # every received piece is a sorted run (a slice of its sender's sorted rows): ordered fan-in, not a second sort
    rc = [int(k) for k in getattr(mine, "recv_counts", [mine.num_rows])]
    return PL.merge_tables(tc, rc, [dict(s, expr=E.rebind(s["expr"], schema)) for s in sort_expr], pre_concatenated=mine)
