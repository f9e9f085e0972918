"""Python handle on the C-ABI graph engine (et_graph_* in include/eigentrust_hip.h)."""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import check, lib

R_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def fr_bytes_from_ints(vals):
    """list of python ints -> uint8 array [len, 32], canonical little-endian."""
    out = np.zeros((len(vals), 32), dtype=np.uint8)
    for i, v in enumerate(vals):
        out[i] = np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)
    return out


def fr_ints_from_bytes(arr):
    b = np.ascontiguousarray(arr, dtype=np.uint8).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def synth_graph(kind, n, target_edges, seed=42, value_max=255):
    """Deterministic synthetic attestation list (et_synth_graph): returns from, to, value."""
    ne = C.c_uint64(0)
    check(lib().et_synth_graph(kind, n, target_edges, seed, value_max, C.byref(ne), None, None, None))
    frm = np.empty(ne.value, dtype=np.uint32)
    to = np.empty(ne.value, dtype=np.uint32)
    val = np.empty(ne.value, dtype=np.uint32)
    check(lib().et_synth_graph(kind, n, target_edges, seed, value_max, C.byref(ne), _ptr(frm), _ptr(to), _ptr(val)))
    return frm, to, val


def synth_graph_range(kind, n, target_edges, attester_begin, attester_end, seed=42, value_max=255):
    """The part of synth_graph's list that the attesters [attester_begin, attester_end) wrote (et_synth_graph_range)."""
    ne = C.c_uint64(0)
    check(lib().et_synth_graph_range(kind, n, target_edges, seed, value_max, attester_begin, attester_end, C.byref(ne), None, None, None))
    frm = np.empty(ne.value, dtype=np.uint32)
    to = np.empty(ne.value, dtype=np.uint32)
    val = np.empty(ne.value, dtype=np.uint32)
    check(lib().et_synth_graph_range(kind, n, target_edges, seed, value_max, attester_begin, attester_end, C.byref(ne),
                                     _ptr(frm), _ptr(to), _ptr(val)))
    return frm, to, val


class DistributedPlan:
    """One share laid out from the attestations of this process's own attesters (et_dplan_*).  The caller moves the
    statistics and the edge records between the processes; see ShardedGraph for the torch.distributed way."""

    def __init__(self, n, frm, to, value_u32, member=None):
        frm = np.ascontiguousarray(frm, dtype=np.uint32)
        to = np.ascontiguousarray(to, dtype=np.uint32)
        value_u32 = np.ascontiguousarray(value_u32, dtype=np.uint32)
        if not (len(frm) == len(to) == len(value_u32)):
            raise ValueError("from/to/value length mismatch")
        if member is not None:
            member = np.ascontiguousarray(member, dtype=np.uint8)
            if len(member) != n:
                raise ValueError("member flags must have n entries")
        h = C.c_void_p()
        check(lib().et_dplan_create(n, _ptr(member), len(frm), _ptr(frm), _ptr(to), _ptr(value_u32), C.byref(h)))
        self._h = h
        self.n = n
        self.n_seg = None

    def close(self):
        if getattr(self, "_h", None):
            lib().et_dplan_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def local_stats(self):
        """(row_sum uint64[n], in_degree uint32[n], out_degree uint32[n], attested uint32[n], value_max)."""
        row_sum = np.empty(self.n, dtype=np.uint64)
        in_deg = np.empty(self.n, dtype=np.uint32)
        out_deg = np.empty(self.n, dtype=np.uint32)
        attested = np.empty(self.n, dtype=np.uint32)
        vmax = C.c_uint32(0)
        check(lib().et_dplan_local_stats(self._h, _ptr(row_sum), _ptr(in_deg), _ptr(out_deg), _ptr(attested), C.byref(vmax)))
        return row_sum, in_deg, out_deg, attested, vmax.value

    def set_global_stats(self, row_sum, in_deg, out_deg, attested, value_max, n_seg, seg, variants=_ffi.ET_VARIANT_FR | _ffi.ET_VARIANT_F64,
                         device=-1, n_chunks=1, long_row=0, target_work=0, full_vector=False):
        row_sum = np.ascontiguousarray(row_sum, dtype=np.uint64)
        in_deg = np.ascontiguousarray(in_deg, dtype=np.uint32)
        out_deg = np.ascontiguousarray(out_deg, dtype=np.uint32)
        attested = np.ascontiguousarray(attested, dtype=np.uint32)
        if not 0 <= int(value_max) < 2**32:
            raise ValueError("value_max must be the largest 32-bit opinion of any process")
        opts = _ffi.EtBuildOpts(n_seg, seg, long_row, target_work, variants, device, n_chunks,
                                _ffi.ET_BUILD_FULL_VECTOR if full_vector else 0)
        check(lib().et_dplan_set_global_stats(self._h, _ptr(row_sum), _ptr(in_deg), _ptr(out_deg), _ptr(attested), int(value_max), C.byref(opts)))
        self.n_seg = n_seg

    def outgoing(self, share):
        """(source codes, destination local rows, opinions) for `share`: views into the plan's memory."""
        cnt = C.c_uint64(0)
        ps, pr, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().et_dplan_outgoing(self._h, share, C.byref(cnt), C.byref(ps), C.byref(pr), C.byref(pv)))
        k = cnt.value
        if k == 0:
            z = np.zeros(0, dtype=np.uint32)
            return z, z, z
        mk = lambda p: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), (k,))
        return mk(ps), mk(pr), mk(pv)

    def add_incoming(self, src, row, val):
        src = np.ascontiguousarray(src, dtype=np.uint32)
        row = np.ascontiguousarray(row, dtype=np.uint32)
        val = np.ascontiguousarray(val, dtype=np.uint32)
        if not (len(src) == len(row) == len(val)):
            raise ValueError("record arrays differ in length")
        check(lib().et_dplan_add_incoming(self._h, len(src), _ptr(src), _ptr(row), _ptr(val)))

    def needs(self, share):
        """The local rows of `share` this share's rows cite (ascending; empty without a gather vector and for the share
        itself): a view into the plan's memory.  The first call ends the intake of edge records."""
        cnt = C.c_uint64(0)
        pr = C.c_void_p()
        check(lib().et_dplan_needs(self._h, share, C.byref(cnt), C.byref(pr)))
        if cnt.value == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_uint32)), (cnt.value,))

    def add_wanted(self, share, rows):
        """What `share` cites of this share's rows (the list its needs() returned for this share)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        check(lib().et_dplan_add_wanted(self._h, share, len(rows), _ptr(rows)))

    def info(self):
        info = _ffi.EtGraphInfo()
        check(lib().et_dplan_info_get(self._h, C.byref(info)))
        return info

    def finish(self):
        h = C.c_void_p()
        check(lib().et_dplan_finish(self._h, C.byref(h)))
        return Graph._from_handle(h, self.n)

    def finish_host(self):
        """The share's host-side plan (no GPU needed)."""
        h = C.c_void_p()
        check(lib().et_dplan_finish_host(self._h, C.byref(h)))
        return Plan._from_handle(h, self.n)


class Graph:
    """A filtered, normalised attestation graph resident on one GPU."""

    def __init__(self, n, frm, to, value_u32=None, value_fr=None, member=None, n_seg=1, seg=0,
                 variants=_ffi.ET_VARIANT_FR | _ffi.ET_VARIANT_F64, device=-1, long_row=0,
                 target_work=0, n_chunks=1, full_vector=False, reused=False):
        # reused: the graph will be converged thousands of times before it changes (ET_BUILD_REUSED): a slower plan, faster steps
        frm = np.ascontiguousarray(frm, dtype=np.uint32)
        to = np.ascontiguousarray(to, dtype=np.uint32)
        if len(frm) != len(to):
            raise ValueError("from/to length mismatch")
        if value_u32 is not None:
            value_u32 = np.ascontiguousarray(value_u32, dtype=np.uint32)
        if value_fr is not None:
            value_fr = np.ascontiguousarray(value_fr, dtype=np.uint8)
        if member is not None:
            member = np.ascontiguousarray(member, dtype=np.uint8)
            if len(member) != n:
                raise ValueError("member flags must have n entries")
        opts = _ffi.EtBuildOpts(n_seg, seg, long_row, target_work, variants, device, n_chunks,
                                (_ffi.ET_BUILD_FULL_VECTOR if full_vector else 0) | (_ffi.ET_BUILD_REUSED if reused else 0))
        h = C.c_void_p()
        check(lib().et_graph_build(n, _ptr(member), len(frm), _ptr(frm), _ptr(to), _ptr(value_u32),
                                   _ptr(value_fr), C.byref(opts), C.byref(h)))
        self._h = h
        self.n = n
        info = _ffi.EtGraphInfo()
        check(lib().et_graph_info_get(self._h, C.byref(info)))
        self.info = info

    @classmethod
    def _from_handle(cls, h, n):
        self = cls.__new__(cls)
        self._h = h
        self.n = n
        info = _ffi.EtGraphInfo()
        check(lib().et_graph_info_get(self._h, C.byref(info)))
        self.info = info
        return self

    def close(self):
        if getattr(self, "_h", None):
            lib().et_graph_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def exact_kernel_name(self):
        """Which kernel iterates the exact variant on this graph (for reports)."""
        return "k_step_fr_blocked" if int(self.info.exact_sweep_blocks) != 0 else "k_step_fr_raw"

    # whole converge, host in / host out ------------------------------------------------------
    def converge_fr(self, num_iterations=20, initial=1000):
        out = np.zeros((self.n, 32), dtype=np.uint8)
        check(lib().et_graph_converge_fr(self._h, num_iterations, initial & (2**64 - 1), initial >> 64, _ptr(out)))
        return out

    def converge_f64(self, num_iterations=20, initial=1000.0, alpha=0.0, pretrust=None):
        out = np.zeros(self.n, dtype=np.float64)
        res = C.c_double(0)
        if pretrust is not None:
            pretrust = np.ascontiguousarray(pretrust, dtype=np.float64)
        check(lib().et_graph_converge_f64(self._h, num_iterations, initial, alpha, _ptr(pretrust), _ptr(out), C.byref(res)))
        return out, res.value

    # device-resident runs ------------------------------------------------------------------------
    def run_fr(self, num_iterations=20, initial=1000, stream=None, record=False):
        check(lib().et_graph_run_fr(self._h, num_iterations, initial & (2**64 - 1), initial >> 64, stream, int(record)))

    def run_f64(self, num_iterations=20, initial=1000.0, alpha=0.0, stream=None, record=False):
        check(lib().et_graph_run_f64(self._h, num_iterations, initial, alpha, stream, int(record)))

    def record_begin(self, stream=None):
        check(lib().et_graph_record_begin(self._h, stream))

    def record_end(self, stream=None):
        check(lib().et_graph_record_end(self._h, stream))

    def event_times(self):
        """(total ms between the recorded event pairs, number of pairs); empties the ring."""
        ms = C.c_float(0)
        n = C.c_uint32(0)
        check(lib().et_graph_event_times(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def result_fr(self, stream=None):
        out = np.zeros((self.n, 32), dtype=np.uint8)
        check(lib().et_graph_result_fr(self._h, _ptr(out), stream))
        return out

    def result_f64(self, stream=None):
        out = np.zeros(self.n, dtype=np.float64)
        check(lib().et_graph_result_f64(self._h, _ptr(out), stream))
        return out

    def stats_fr(self, iteration, stream=None):
        st = _ffi.EtIterStats()
        check(lib().et_graph_stats_fr(self._h, iteration, stream, C.byref(st)))
        return int.from_bytes(bytes(st.sum.b), "little"), st.changed

    def stats_f64(self, iteration, stream=None):
        st = _ffi.EtIterStats()
        check(lib().et_graph_stats_f64(self._h, iteration, stream, C.byref(st)))
        return st.residual_l1, st.sum_f64

    # stepping on caller-owned device vectors (raw device pointers as ints) ---------------------
    def init_fr(self, x_ptr, initial=1000, stream=None, plain=False):
        """Initial vector of this share; `plain=True` for a vector that is read back instead of iterated."""
        check(lib().et_graph_init_fr_async(self._h, initial & (2**64 - 1), initial >> 64, x_ptr, 1 if plain else 0, stream))

    def init_f64(self, x_ptr, initial=1000.0, stream=None, plain=False):
        check(lib().et_graph_init_f64_async(self._h, initial, x_ptr, 1 if plain else 0, stream))

    def step_fr(self, x_ptr, y_ptr, iteration, stream=None, final=False, phase=None, stats=True):
        """One iteration; `final=True` on the last step of a converge (its output holds plain scores).
        `phase`: one of the info.n_chunks launches of the iteration (graphs built with n_chunks > 1), None = all.
        `stats=False`: the step does not read the old entries, stats_fr reports 0 changed rows for it."""
        flags = (_ffi.ET_STEP_FINAL if final else 0) | (0 if stats else _ffi.ET_STEP_NO_STATS)
        if phase is None:
            check(lib().et_graph_step_fr_async(self._h, x_ptr, y_ptr, iteration, flags, stream))
        else:
            check(lib().et_graph_step_fr_phase_async(self._h, x_ptr, y_ptr, iteration, flags, phase, stream))

    def step_f64(self, x_ptr, y_ptr, iteration, alpha=0.0, pretrust_ptr=None, stream=None, final=False, phase=None, stats=True):
        """`stats=False`: neither residual nor score sum for this iteration (stats_f64 reports zeros for it)."""
        flags = (_ffi.ET_STEP_FINAL if final else 0) | (0 if stats else _ffi.ET_STEP_NO_STATS)
        if phase is None:
            check(lib().et_graph_step_f64_async(self._h, x_ptr, y_ptr, iteration, alpha, pretrust_ptr, flags, stream))
        else:
            check(lib().et_graph_step_f64_phase_async(self._h, x_ptr, y_ptr, iteration, alpha, pretrust_ptr, flags, phase, stream))

    def exchange(self):
        """What this share sends to and receives from the others between two steps (et_exchange_info)."""
        ex = _ffi.EtExchangeInfo()
        check(lib().et_graph_exchange_get(self._h, C.byref(ex)))
        return ex

    def pack_fr(self, y_ptr, send_ptr, stream=None):
        """Copies the entries of the own block of y that the other shares cite into the send buffer."""
        check(lib().et_graph_pack_fr_async(self._h, y_ptr, send_ptr, stream))

    def pack_f64(self, y_ptr, send_ptr, stream=None):
        check(lib().et_graph_pack_f64_async(self._h, y_ptr, send_ptr, stream))

    def unpermute_fr(self, x_ptr, out_ptr, stream=None):
        """x: a vector in position order (info.full_len entries)."""
        check(lib().et_graph_unpermute_fr_async(self._h, x_ptr, out_ptr, stream))

    def unpermute_f64(self, x_ptr, out_ptr, stream=None):
        check(lib().et_graph_unpermute_f64_async(self._h, x_ptr, out_ptr, stream))


class MultiGpuGraph:
    """One graph row-sharded over several GPUs of this node, driven from this one process (et_sharded_*).
    `devices` lists one HIP ordinal per share; the same ordinal twice rehearses the layout on fewer GPUs."""

    def __init__(self, n, frm, to, value_u32=None, value_fr=None, member=None, devices=(0,),
                 variants=_ffi.ET_VARIANT_FR | _ffi.ET_VARIANT_F64, exchange=_ffi.ET_EXCHANGE_AUTO, n_chunks=1):
        frm = np.ascontiguousarray(frm, dtype=np.uint32)
        to = np.ascontiguousarray(to, dtype=np.uint32)
        if len(frm) != len(to):
            raise ValueError("from/to length mismatch")
        if value_u32 is not None:
            value_u32 = np.ascontiguousarray(value_u32, dtype=np.uint32)
        if value_fr is not None:
            value_fr = np.ascontiguousarray(value_fr, dtype=np.uint8)
        if member is not None:
            member = np.ascontiguousarray(member, dtype=np.uint8)
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        check(lib().et_sharded_create(n, _ptr(member), len(frm), _ptr(frm), _ptr(to), _ptr(value_u32), _ptr(value_fr),
                                      len(devs), _ptr(devs), variants, exchange, n_chunks, C.byref(h)))
        self._h = h
        self.n = n
        self.world = len(devs)
        self.exchange = lib().et_sharded_exchange_kind(self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib().et_sharded_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self, share=0):
        info = _ffi.EtGraphInfo()
        check(lib().et_sharded_info_get(self._h, share, C.byref(info)))
        return info

    def converge_fr(self, num_iterations=20, initial=1000):
        out = np.zeros((self.n, 32), dtype=np.uint8)
        check(lib().et_sharded_converge_fr(self._h, num_iterations, initial & (2**64 - 1), initial >> 64, _ptr(out)))
        return out

    def converge_f64(self, num_iterations=20, initial=1000.0, alpha=0.0, pretrust=None):
        out = np.zeros(self.n, dtype=np.float64)
        res = C.c_double(0)
        if pretrust is not None:
            pretrust = np.ascontiguousarray(pretrust, dtype=np.float64)
        check(lib().et_sharded_converge_f64(self._h, num_iterations, initial, alpha, _ptr(pretrust), _ptr(out), C.byref(res)))
        return out, res.value


class Plan:
    """Host-side layout of an attestation list (et_plan_*): what Graph uploads, without a GPU.
    Arrays are copied into numpy so they outlive the native object."""

    def __init__(self, n, frm, to, value_u32=None, value_fr=None, member=None, n_seg=1, seg=0,
                 long_row=0, target_work=0, full_vector=False, reused=False):
        frm = np.ascontiguousarray(frm, dtype=np.uint32)
        to = np.ascontiguousarray(to, dtype=np.uint32)
        if value_u32 is not None:
            value_u32 = np.ascontiguousarray(value_u32, dtype=np.uint32)
        if value_fr is not None:
            value_fr = np.ascontiguousarray(value_fr, dtype=np.uint8)
        if member is not None:
            member = np.ascontiguousarray(member, dtype=np.uint8)
        opts = _ffi.EtBuildOpts(n_seg, seg, long_row, target_work, 0, -1, 1,
                                (_ffi.ET_BUILD_FULL_VECTOR if full_vector else 0) | (_ffi.ET_BUILD_REUSED if reused else 0))
        h = C.c_void_p()
        check(lib().et_plan_build(n, _ptr(member), len(frm), _ptr(frm), _ptr(to), _ptr(value_u32),
                                  _ptr(value_fr), C.byref(opts), C.byref(h)))
        self._load(h, n)

    @classmethod
    def _from_handle(cls, h, n):
        self = cls.__new__(cls)
        self._load(h, n)
        return self

    def _load(self, h, n):
        try:
            info = _ffi.EtGraphInfo()
            check(lib().et_plan_info_get(h, C.byref(info)))
            arr = _ffi.EtPlanArrays()
            check(lib().et_plan_arrays_get(h, C.byref(arr)))
            self.info = info
            self.n = n
            # layout of the blocked sweeps: workgroups, edges and 128-byte lines of the vector a launch pulls in
            self.sweep = {}
            for name, variant in (("fr", _ffi.ET_VARIANT_FR), ("f64", _ffi.ET_VARIANT_F64)):
                ss = _ffi.EtSweepStats()
                check(lib().et_plan_sweep_stats(h, variant, C.byref(ss)))
                self.sweep[name] = {f: int(getattr(ss, f)) for f, _ in _ffi.EtSweepStats._fields_}
            # the exchange tables and, with a gather vector, the rows cited in either direction
            ex = _ffi.EtExchangeInfo()
            need = (C.c_void_p * 8)()
            send = (C.c_void_p * 8)()
            check(lib().et_plan_exchange_get(h, C.byref(ex), need, send))
            self.exchange = ex
            nd = int(info.nd_cap)
            grab = lambda ptr, k: (np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), (k,)).copy()
                                   if ptr and k else np.zeros(0, dtype=np.uint32))
            self.need = [grab(need[c], int(ex.recv_count[c]) - nd) if ex.gather and ex.recv_count[c] else np.zeros(0, np.uint32) for c in range(8)]
            if ex.gather:     # the share's own cited rows: a block of the gather vector that nothing travels for
                self.need[int(ex.seg)] = grab(need[int(ex.seg)], int(ex.self_count))
            self.send = [grab(send[c], int(ex.send_count[c]) - nd) if ex.gather and ex.send_count[c] else np.zeros(0, np.uint32) for c in range(8)]
            slots = int(info.padded_local)
            vec_len = int(info.full_len)
            self.pos_of_slot = np.ctypeslib.as_array(arr.pos_of_slot, (n,)).copy()
            self.slices = [(s.off, s.len, s.row0, s.nrows) for s in arr.slices[:info.n_slices]]
            self.work = [(b.slice_begin, b.slice_end, b.flags, b.dslot) for b in arr.work[:info.n_work]]
            self.col = np.ctypeslib.as_array(arr.col, (slots,)).copy() if slots else np.zeros(0, np.uint32)
            if arr.value_u32:
                self.value_u32 = np.ctypeslib.as_array(arr.value_u32, (slots,)).copy() if slots else np.zeros(0, np.uint32)
                self.value_fr = None
            else:
                self.value_u32 = None
                raw = C.string_at(arr.value_fr, slots * 32) if slots else b""
                self.value_fr = [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]
            raw = C.string_at(arr.scale_fr, vec_len * 32)
            self.scale_fr = [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]
            self.scale_f64 = np.ctypeslib.as_array(arr.scale_f64, (vec_len,)).copy()
            self.inv_nm1_w = int.from_bytes(bytes(arr.inv_nm1_w.b), "little")
            self.n_active, self.n_active_pad = arr.n_active, arr.n_active_pad
            self.n_dangling_pad, self.n_dangling_local = arr.n_dangling_pad, arr.n_dangling_local
        finally:
            lib().et_plan_free(h)


class Batch:
    """Many small independent sets resident on one GPU (et_batch_*); one workgroup per set."""

    def __init__(self, values, member=None, device=-1):
        values = np.ascontiguousarray(values, dtype=np.uint32)
        if values.ndim != 3 or values.shape[1] != values.shape[2]:
            raise ValueError("values must have shape [n_sets, n, n]")
        self.n_sets, self.n = int(values.shape[0]), int(values.shape[1])
        if member is not None:
            member = np.ascontiguousarray(member, dtype=np.uint8)
            if member.shape != (self.n_sets, self.n):
                raise ValueError("member must have shape [n_sets, n]")
        h = C.c_void_p()
        check(lib().et_batch_create(self.n_sets, self.n, _ptr(member), _ptr(values), device, C.byref(h)))
        self._h = h
        e = C.c_uint64(0)
        check(lib().et_batch_edges(self._h, C.byref(e)))
        self.edges = e.value

    def close(self):
        if getattr(self, "_h", None):
            lib().et_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_fr(self, num_iterations=20, initial=1000, stream=None):
        check(lib().et_batch_run_fr_async(self._h, num_iterations, initial & (2**64 - 1), initial >> 64, stream))

    def result_fr(self, stream=None):
        """(scores uint8 [n_sets, n, 32], status int32 [n_sets])."""
        out = np.zeros((self.n_sets, self.n, 32), dtype=np.uint8)
        status = np.zeros(self.n_sets, dtype=np.int32)
        check(lib().et_batch_result_fr(self._h, _ptr(out), _ptr(status), stream))
        return out, status

    def converge_fr(self, num_iterations=20, initial=1000):
        self.run_fr(num_iterations, initial)
        return self.result_fr()
