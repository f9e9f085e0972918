"""Row-sharded multi-GPU converge: one process per GPU, torch.distributed (RCCL) for the exchange.

Every rank holds the rows (destination peers) of its segment.  Per iteration a rank runs the step
kernel on its rows, which writes its own block of the new vector: its rows' entries and, in the block's
tail, its partial sums of the dangling-peer score mass.  One exchange per iteration then completes the
vector on every rank, and it is the only collective on the path:

  * integer opinions (the production case): a rank's vectors are gather vectors -- its own block, then
    only the entries of the other ranks' rows that its own rows cite.  After a step the rank packs what
    each other rank cites (et_graph_pack_*) and ONE all-to-all delivers it (RCCL: all_to_all_single);
  * field-valued opinions, a chunked exchange, or full_vector=True: the vectors hold every rank's block
    in position order and the exchange is one in-place all-gather of the blocks.

Segments are dealt 64 rows at a time round-robin in degree order, so every rank gets the same number
of rows and a near-identical degree mix.  The Fr variant is exact integer arithmetic: its result does
not depend on the number of ranks, bit for bit.  The f64 variant adds a row's terms in whatever order
the hardware serves its LDS atomics, so its last bits vary from run to run and with the layout (the
difference stays within the usual summation error, far below the 1e-9 the variant is held to).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _ffi
from .graph import DistributedPlan, Graph, R_MODULUS


def _all_reduce_np(arr, signed_view, op, group, world, comm_device, always=False):
    """All-reduce of a numpy array of unsigned integers (summed or maximised as the signed type of the same width:
    per-peer degree sums stay far below 2^31, row sums below 2^63)."""
    if world == 1 and not always:
        return arr
    t = torch.from_numpy(arr.view(signed_view)).to(comm_device)
    dist.all_reduce(t, op=op, group=group)
    return t.cpu().numpy().view(arr.dtype)


def distributed_plan(n, frm, to, value_u32, member=None, group=None, world=1, rank=0, comm_device=None,
                     always_collective=False, **opts):
    """The share of `rank` laid out from rank-local attestations (every attestation of an attester on one rank), ready for
    DistributedPlan.finish() / finish_host().  Collectives: four all-reduces of per-peer figures (8 + 4 + 4 + 4 bytes per
    peer), one of a scalar, one all-gather of the record counts and three all-to-alls of the edge records (12 bytes per
    surviving edge in all).  `comm_device`: where tensors handed to a collective live (the GPU for RCCL, the host for
    gloo).  `always_collective`: run the collectives with a single rank too (each is then the identity): the way to
    take this function's RCCL calls on a one-GPU box."""
    comm_device = comm_device or torch.device("cpu")
    plan = DistributedPlan(n, frm, to, value_u32, member=member)
    row_sum, in_deg, out_deg, attested, vmax = plan.local_stats()
    ar = lambda a, view, op: _all_reduce_np(a, view, op, group, world, comm_device, always_collective)
    row_sum = ar(row_sum, np.int64, dist.ReduceOp.SUM)
    in_deg = ar(in_deg, np.int32, dist.ReduceOp.SUM)
    out_deg = ar(out_deg, np.int32, dist.ReduceOp.SUM)
    attested = ar(attested, np.int32, dist.ReduceOp.SUM)
    # the largest opinion as a 64-bit integer: a 32-bit opinion of 2^31 or more must not turn negative in the maximum
    vmax = int(ar(np.array([vmax], dtype=np.int64), np.int64, dist.ReduceOp.MAX)[0])
    plan.set_global_stats(row_sum, in_deg, out_deg, attested, vmax, world, rank, **opts)
    del row_sum, in_deg, out_deg, attested
    outgoing = [plan.outgoing(s) for s in range(world)]
    if world == 1 and not always_collective:
        plan.add_incoming(*outgoing[0])
        return plan                                          # one share: no gather vector, nobody to tell
    send = [len(o[0]) for o in outgoing]
    counts = torch.tensor(send, dtype=torch.int64, device=comm_device)
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    recv = [int(c[rank]) for c in all_counts]                 # what every rank holds for this one
    got = []
    for field in range(3):                                    # source codes, destination rows, opinions
        buf = np.concatenate([o[field] for o in outgoing])
        t_in = torch.from_numpy(buf.view(np.int32)).to(comm_device)
        t_out = torch.empty(sum(recv), dtype=torch.int32, device=comm_device)
        dist.all_to_all_single(t_out, t_in, recv, send, group=group)
        got.append(t_out.cpu().numpy().view(np.uint32))
        del buf, t_in, t_out
    plan.add_incoming(*got)
    del got
    if plan.info().gather:
        # gather vectors: every rank tells the others which of their rows it cites (they pack those after every step)
        # (what it cites of its OWN rows stays here: the block of the share's own cited entries is filled by its own pack)
        needs = [plan.needs(s) if s != rank else np.zeros(0, dtype=np.uint32) for s in range(world)]
        send = [len(x) for x in needs]
        counts = torch.tensor(send, dtype=torch.int64, device=comm_device)
        all_counts = [torch.empty_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts, group=group)
        recv = [int(c[rank]) for c in all_counts]
        t_in = torch.from_numpy(np.concatenate(needs).view(np.int32)).to(comm_device)
        t_out = torch.empty(sum(recv), dtype=torch.int32, device=comm_device)
        dist.all_to_all_single(t_out, t_in, recv, send, group=group)
        wanted = t_out.cpu().numpy().view(np.uint32)
        at = 0
        for s in range(world):
            if s != rank:
                plan.add_wanted(s, wanted[at:at + recv[s]])
            at += recv[s]
    return plan


class ShardedGraph:
    def __init__(self, n, frm, to, value_u32=None, value_fr=None, member=None, group=None,
                 device=None, variants=_ffi.ET_VARIANT_FR | _ffi.ET_VARIANT_F64, always_exchange=False, n_chunks=1,
                 local_attestations=False, **build_kw):
        # always_exchange: take the stepping + all-gather path even with a single rank (the collective is then a
        # copy of the vector onto itself): the way to run the RCCL call of this driver on a one-GPU box
        # local_attestations: frm / to / value_u32 hold only the attestations of THIS rank's attesters (every
        # attestation of an attester on one rank); the ranks agree on the layout through three all-reduces and send
        # each edge to the rank that owns its destination row, so that no rank ever holds the whole list
        self.always_exchange = always_exchange
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        dev_index = self.device.index if self.device.index is not None else -1
        chunks = n_chunks if (self.world > 1 or always_exchange) else 1
        if local_attestations:
            if value_fr is not None or value_u32 is None:
                raise ValueError("attestations ingested rank by rank carry integer opinions")
            self.graph = self._build_from_local(n, frm, to, value_u32, member, variants, dev_index, chunks, **build_kw)
        else:
            self.graph = Graph(n, frm, to, value_u32=value_u32, value_fr=value_fr, member=member,
                               n_seg=self.world, seg=self.rank, variants=variants, device=dev_index, n_chunks=chunks, **build_kw)
        info = self.graph.info
        # chunked exchange: a segment travels in n_chunks pieces while the iteration's n_chunks launches run, launch q
        # waiting for piece q only (1 where the graph does not take the blocked sweeps)
        self.n_chunks = int(info.n_chunks)
        self.chunk_rows = int(info.chunk_rows)
        self.n = n
        self.vec_len = int(info.vec_len)
        self.full_len = int(info.full_len)
        self.seg_stride = int(info.rows_cap + info.nd_cap)
        ex = self.graph.exchange()
        self.gather = bool(ex.gather)
        self.own_off = int(ex.own_off)
        self.send_total = int(ex.send_total)
        # entries sent to / received from every other rank per exchange (nothing to itself)
        self.send_counts = [int(ex.send_count[c]) if c != self.rank else 0 for c in range(self.world)]
        self.recv_counts = [int(ex.recv_count[c]) if c != self.rank else 0 for c in range(self.world)]
        # gather vector: own block, then the block of this rank's own cited entries (filled by the pack), then what arrives
        self.self_count = int(ex.self_count)
        if self.gather:
            assert self.own_off == 0 and int(ex.self_off) == self.seg_stride
            assert self.vec_len == self.seg_stride + self.self_count + sum(self.recv_counts)
        else:
            assert self.vec_len == self.seg_stride * self.world
        self.nnz = int(info.nnz)
        self.n_valid = int(info.n_valid)
        self._vec = {}

    def _build_from_local(self, n, frm, to, value_u32, member, variants, dev_index, n_chunks, **build_kw):
        comm = self.device if dist.is_initialized() and dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        plan = distributed_plan(n, frm, to, value_u32, member=member, group=self.group, world=self.world, rank=self.rank,
                                comm_device=comm, always_collective=self.always_exchange, variants=variants, device=dev_index,
                                n_chunks=n_chunks, **build_kw)
        graph = plan.finish()
        plan.close()
        return graph

    # ---- buffers ---------------------------------------------------------------------------------
    def _buffers(self, kind):
        if kind not in self._vec:
            if kind == "f64":
                mk = lambda: torch.zeros(self.vec_len, dtype=torch.float64, device=self.device)
                out = torch.zeros(self.n, dtype=torch.float64, device=self.device)
            else:  # field elements: 4 x int64 per entry
                mk = lambda: torch.zeros((self.vec_len, 4), dtype=torch.int64, device=self.device)
                out = torch.zeros((self.n, 4), dtype=torch.int64, device=self.device)
            self._vec[kind] = (mk(), mk(), out)
        return self._vec[kind]

    def _segment(self, t):
        """This rank's own block of a vector."""
        return t[self.own_off:self.own_off + self.seg_stride]

    def exchange_bytes(self, kind):
        """(sent, received) bytes per rank and iteration."""
        elem = 8 if kind == "f64" else 32
        if self.world == 1:
            return 0, 0
        if self.gather:
            return self.send_total * elem, sum(self.recv_counts) * elem
        return self.seg_stride * elem * (self.world - 1), self.seg_stride * elem * (self.world - 1)

    def _exchange(self, t, kind):
        """Completes the vector `t` whose own block this rank has just written."""
        if not self.gather:
            return self._all_gather(t)
        key = ("send", kind)
        if key not in self._vec:
            self._vec[key] = torch.empty((max(1, self.send_total),) + tuple(t.shape[1:]), dtype=t.dtype, device=self.device)
        send = self._vec[key][:self.send_total]
        st = torch.cuda.current_stream(self.device).cuda_stream
        (self.graph.pack_f64 if kind == "f64" else self.graph.pack_fr)(t.data_ptr(), send.data_ptr(), st)
        others = t[self.seg_stride + self.self_count:]    # the other ranks' blocks, in rank order
        if dist.get_backend(self.group) == "nccl":
            dist.all_to_all_single(others, send, self.recv_counts, self.send_counts, group=self.group)
        else:  # gloo (CPU rendezvous, used by tests): stage through host memory
            host_in = send.cpu().contiguous()
            host_out = torch.empty((sum(self.recv_counts),) + tuple(t.shape[1:]), dtype=t.dtype)
            dist.all_to_all_single(host_out, host_in, self.recv_counts, self.send_counts, group=self.group)
            others.copy_(host_out)

    def _full_vector(self, t, kind):
        """The vector in position order (every rank's own block) from the last step's output."""
        if not self.gather:
            self._all_gather(t)
            return t
        key = ("full", kind)
        if key not in self._vec:
            self._vec[key] = torch.empty((self.full_len,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.device)
        full = self._vec[key]
        mine = self._segment(t)
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(full, mine, group=self.group)
        else:
            host = mine.cpu().contiguous()
            parts = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(parts, host, group=self.group)
            for r, p in enumerate(parts):
                full[r * self.seg_stride:(r + 1) * self.seg_stride].copy_(p)
        return full

    def _all_gather(self, t):
        """In-place all-gather of the segments of `t`."""
        if self.world == 1 and not self.always_exchange:
            return
        backend = dist.get_backend(self.group)
        if backend == "nccl":
            dist.all_gather_into_tensor(t, self._segment(t), group=self.group)
        else:  # gloo (CPU rendezvous, used by tests): stage through host memory
            mine = self._segment(t).cpu().contiguous()
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine, group=self.group)
            for r, p in enumerate(parts):
                if r != self.rank:
                    t[r * self.seg_stride:(r + 1) * self.seg_stride].copy_(p)

    # ---- chunked exchange ------------------------------------------------------------------------
    def _pieces(self):
        k = self.n_chunks
        return [(q * self.chunk_rows, (self.chunk_rows if q + 1 < k else self.seg_stride - q * self.chunk_rows)) for q in range(k)]

    def _start_exchange(self, t, kind):
        """Starts the exchange of every piece of the vector `t` (this rank's segment just written); returns what
        _finish_piece needs.  With RCCL every piece is an asynchronous all-gather into a staging buffer: the
        collectives run in order on the communicator's stream while the caller's stream goes on launching."""
        pend = []
        nccl = dist.get_backend(self.group) == "nccl"
        for q, (lo, ln) in enumerate(self._pieces()):
            mine = self._segment(t)[lo:lo + ln]
            if nccl:
                key = ("stage", kind, q)
                if key not in self._vec:
                    self._vec[key] = torch.empty((self.world * ln,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.device)
                stage = self._vec[key]
                work = dist.all_gather_into_tensor(stage, mine, group=self.group, async_op=True)
                pend.append((work, stage, lo, ln))
            else:  # gloo (tests): through host memory, synchronously
                host = mine.cpu().contiguous()
                parts = [torch.empty_like(host) for _ in range(self.world)]
                dist.all_gather(parts, host, group=self.group)
                pend.append((None, parts, lo, ln))
        return pend

    def _finish_piece(self, t, pend, q):
        """Makes piece q of every segment of `t` current on the caller's stream."""
        work, got, lo, ln = pend[q]
        if work is not None:
            work.wait()                       # the current stream waits for the collective; the host does not
            full = t.view((self.world, self.seg_stride) + tuple(t.shape[1:]))
            full[:, lo:lo + ln].copy_(got.view((self.world, ln) + tuple(t.shape[1:])))
        else:
            for r, p in enumerate(got):
                if r != self.rank:
                    t[r * self.seg_stride + lo:r * self.seg_stride + lo + ln].copy_(p)

    def _run_chunked(self, kind, num_iterations, initial, alpha, record):
        st = torch.cuda.current_stream(self.device).cuda_stream
        g = self.graph
        x, y, out = self._buffers(kind)
        pre = None
        if kind == "f64":
            if alpha != 0.0:
                if "pre" not in self._vec:
                    self._vec["pre"] = torch.zeros_like(x)
                g.init_f64(self._vec["pre"].data_ptr(), initial, st, plain=True)
                pre = self._vec["pre"].data_ptr()
            g.init_f64(x.data_ptr(), initial, st, plain=num_iterations == 0)
        else:
            g.init_fr(x.data_ptr(), initial, st, plain=num_iterations == 0)
        pend = self._start_exchange(x, kind)
        bufs = (x, y)
        for it in range(num_iterations):
            a, b = bufs[it & 1], bufs[(it + 1) & 1]
            final = it + 1 == num_iterations
            for q in range(self.n_chunks):
                self._finish_piece(a, pend, q)
                if record:
                    g.record_begin(st)
                if kind == "f64":
                    g.step_f64(a.data_ptr(), b.data_ptr(), it, alpha, pre, st, final=final, phase=q, stats=final)
                else:
                    g.step_fr(a.data_ptr(), b.data_ptr(), it, st, final=final, phase=q, stats=final)
                if record:
                    g.record_end(st)
            pend = self._start_exchange(b, kind)
        last = bufs[num_iterations & 1]
        for q in range(self.n_chunks):
            self._finish_piece(last, pend, q)
        if kind == "f64":
            g.unpermute_f64(last.data_ptr(), out.data_ptr(), st)
        else:
            g.unpermute_fr(last.data_ptr(), out.data_ptr(), st)

    # ---- converge --------------------------------------------------------------------------------
    def run_f64(self, num_iterations=20, initial=1000.0, alpha=0.0, record=False):
        """Enqueues a whole converge; the slot-order result stays on the device (result_f64)."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        g = self.graph
        if self.world == 1 and not self.always_exchange:  # no exchange step: the library's own back-to-back loop
            g.run_f64(num_iterations, initial, alpha, st, record)
            return
        if self.n_chunks > 1:
            return self._run_chunked("f64", num_iterations, initial, alpha, record)
        x, y, out = self._buffers("f64")
        pre = None
        if alpha != 0.0:  # default pre-trust: the initial scores (only this share's segment is read)
            if "pre" not in self._vec:
                self._vec["pre"] = torch.zeros_like(x)
            g.init_f64(self._vec["pre"].data_ptr(), initial, st, plain=True)
            pre = self._vec["pre"].data_ptr()
        g.init_f64(x.data_ptr(), initial, st, plain=num_iterations == 0)
        if num_iterations:
            self._exchange(x, "f64")
        bufs = (x, y)
        for it in range(num_iterations):
            a, b = bufs[it & 1], bufs[(it + 1) & 1]
            if record:
                g.record_begin(st)
            g.step_f64(a.data_ptr(), b.data_ptr(), it, alpha, pre, st, final=it + 1 == num_iterations, stats=it + 1 == num_iterations)
            if record:
                g.record_end(st)
            if it + 1 < num_iterations:
                self._exchange(b, "f64")
        g.unpermute_f64(self._full_vector(bufs[num_iterations & 1], "f64").data_ptr(), out.data_ptr(), st)

    def run_fr(self, num_iterations=20, initial=1000, record=False):
        st = torch.cuda.current_stream(self.device).cuda_stream
        g = self.graph
        if self.world == 1 and not self.always_exchange:
            g.run_fr(num_iterations, initial, st, record)
            return
        if self.n_chunks > 1:
            return self._run_chunked("fr", num_iterations, initial, 0.0, record)
        x, y, out = self._buffers("fr")
        g.init_fr(x.data_ptr(), initial, st, plain=num_iterations == 0)
        if num_iterations:
            self._exchange(x, "fr")
        bufs = (x, y)
        for it in range(num_iterations):
            a, b = bufs[it & 1], bufs[(it + 1) & 1]
            if record:
                g.record_begin(st)
            g.step_fr(a.data_ptr(), b.data_ptr(), it, st, final=it + 1 == num_iterations, stats=it + 1 == num_iterations)
            if record:
                g.record_end(st)
            if it + 1 < num_iterations:
                self._exchange(b, "fr")
        g.unpermute_fr(self._full_vector(bufs[num_iterations & 1], "fr").data_ptr(), out.data_ptr(), st)

    def result_f64(self):
        torch.cuda.synchronize(self.device)
        if self.world == 1 and not self.always_exchange:
            return self.graph.result_f64(torch.cuda.current_stream(self.device).cuda_stream)
        return self._buffers("f64")[2].cpu().numpy()

    def result_fr(self):
        torch.cuda.synchronize(self.device)
        if self.world == 1 and not self.always_exchange:
            return self.graph.result_fr(torch.cuda.current_stream(self.device).cuda_stream)
        return self._buffers("fr")[2].cpu().numpy().view(np.uint8).reshape(self.n, 32)

    def converge_fr(self, num_iterations=20, initial=1000):
        """Scores of all n slots as a uint8 array [n, 32]; checks conservation across ranks
        (dynamic_sets/native.rs:331-334)."""
        self.run_fr(num_iterations, initial)
        res = self.result_fr()
        if num_iterations > 0:
            s, _changed = self.graph.stats_fr(num_iterations - 1, torch.cuda.current_stream(self.device).cuda_stream)
            total = self._sum_mod_r(s)
            if total != initial * self.n_valid % R_MODULUS:
                raise _ffi.EigenTrustError(_ffi.ET_E_SUM_MISMATCH, "field sum of scores changed")
        return res

    def converge_f64(self, num_iterations=20, initial=1000.0, alpha=0.0):
        self.run_f64(num_iterations, initial, alpha)
        return self.result_f64()

    def _sum_mod_r(self, mine):
        """Field sum over the ranks' partial sums: one gather of 32 bytes per rank."""
        if self.world == 1 and not self.always_exchange:
            return mine % R_MODULUS
        dev = self.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        words = np.frombuffer(int(mine % R_MODULUS).to_bytes(32, "little"), dtype=np.int64).copy()
        t = torch.from_numpy(words).to(dev)
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t, group=self.group)
        total = 0
        for p in parts:
            total += int.from_bytes(p.cpu().numpy().tobytes(), "little")
        return total % R_MODULUS
