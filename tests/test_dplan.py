"""A share laid out from rank-local attestations (et_dplan_*): every process ingests the attestations of its own
attesters only, and the share it lays out must be the one et_plan_build lays out from the whole list -- same geometry,
same positions, same per-row edge stream, same factors on its own rows.  Host only; the GPU side of the same path is in
tests/test_gpu_sharded.py."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for pth in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if pth not in sys.path:
        sys.path.insert(0, pth)

import protocol_amd as pa  # noqa: E402
from protocol_amd import _ffi  # noqa: E402


def messy_case(n, m, seed):
    """Power-law list with what the filter has to deal with: repeated pairs (a later one replaces an earlier one),
    zero opinions, attesters whose opinions sum to zero, empty slots."""
    rng = np.random.default_rng(seed)
    frm, to, val = pa.synth_graph(3, n, m, seed=seed)
    k = len(frm) // 20
    idx = rng.integers(0, len(frm), k)
    frm = np.concatenate([frm, frm[idx]])
    to = np.concatenate([to, to[idx]])
    val = np.concatenate([val, rng.integers(0, 256, k, dtype=np.uint32)])
    val[rng.random(len(val)) < 0.02] = 0
    val[np.isin(frm, rng.integers(0, n, n // 50))] = 0
    member = (rng.random(n) > 0.03).astype(np.uint8)
    return frm, to, val, member


def assert_same_share(got, ref, rank):
    for f in ("n", "n_valid", "n_dangling", "nnz", "nnz_local", "padded_local", "n_slices", "n_work", "rows_cap", "nd_cap", "vec_len",
              "full_len", "gather"):
        assert getattr(got.info, f) == getattr(ref.info, f), f
    for f in ("gather", "own_off", "own_len", "vec_len", "send_total", "self_off", "self_count"):
        assert getattr(got.exchange, f) == getattr(ref.exchange, f), f
    for f in ("recv_off", "recv_count", "send_off", "send_count"):
        assert list(getattr(got.exchange, f)) == list(getattr(ref.exchange, f)), f
    for c in range(8):
        assert np.array_equal(got.need[c], ref.need[c]) and np.array_equal(got.send[c], ref.send[c]), c
    assert np.array_equal(got.pos_of_slot, ref.pos_of_slot)
    assert got.slices == ref.slices and got.work == ref.work
    assert np.array_equal(got.col, ref.col) and np.array_equal(got.value_u32, ref.value_u32)
    stride = int(ref.info.rows_cap + ref.info.nd_cap)
    lo, hi = rank * stride, (rank + 1) * stride
    assert got.scale_fr[lo:hi] == ref.scale_fr[lo:hi]
    assert np.array_equal(got.scale_f64[lo:hi], ref.scale_f64[lo:hi])
    assert got.inv_nm1_w == ref.inv_nm1_w
    assert (got.n_active, got.n_active_pad, got.n_dangling_pad, got.n_dangling_local) == \
           (ref.n_active, ref.n_active_pad, ref.n_dangling_pad, ref.n_dangling_local)


@pytest.mark.parametrize("n,m,world,seed,assign,full_vector", [
    (3000, 30000, 1, 1, "modulo", False),
    (3000, 30000, 2, 2, "modulo", False),
    (5000, 60000, 3, 3, "ranges", False),
    (4000, 20000, 8, 4, "scattered", False),
    (5000, 60000, 3, 5, "ranges", True),
])
def test_shares_from_local_attestations_equal_shares_from_the_whole_list(n, m, world, seed, assign, full_vector):
    frm, to, val, member = messy_case(n, m, seed)
    ids = np.arange(n)
    owner = {"modulo": ids % world, "ranges": ids * world // n, "scattered": ids * 7919 % world}[assign]
    plans = []
    for r in range(world):
        sel = owner[frm] == r
        plans.append(pa.DistributedPlan(n, frm[sel], to[sel], val[sel], member=member))
    stats = [p.local_stats() for p in plans]
    row_sum = sum(s[0] for s in stats)
    in_deg = sum(s[1] for s in stats)
    out_deg = sum(s[2] for s in stats)
    attested = sum(s[3] for s in stats)
    vmax = max(s[4] for s in stats)
    for r, p in enumerate(plans):
        p.set_global_stats(row_sum, in_deg, out_deg, attested, vmax, world, r, full_vector=full_vector)
    sent = 0
    for r, p in enumerate(plans):
        for q in plans:
            rec = q.outgoing(r)
            sent += len(rec[0])
            p.add_incoming(*rec)
    assert sent == int(plans[0].info().nnz)
    # every share tells the others which of their rows it cites (nothing to tell in position-order layouts)
    for r, p in enumerate(plans):
        assert int(p.info().gather) == int(world > 1 and not full_vector)
        for q, other in enumerate(plans):
            if q != r:
                other.add_wanted(r, p.needs(q).copy())
    for r, p in enumerate(plans):
        assert_same_share(p.finish_host(), pa.Plan(n, frm, to, value_u32=val, member=member, n_seg=world, seg=r,
                                                   full_vector=full_vector), r)


def test_an_attester_split_over_two_processes_is_refused():
    n = 200
    frm, to, val, member = messy_case(n, 1500, 9)
    half = len(frm) // 2                                    # cut by list position: attesters straddle the cut
    a = pa.DistributedPlan(n, frm[:half], to[:half], val[:half], member=member)
    b = pa.DistributedPlan(n, frm[half:], to[half:], val[half:], member=member)
    sa, sb = a.local_stats(), b.local_stats()
    with pytest.raises(_ffi.EigenTrustError) as e:
        a.set_global_stats(sa[0] + sb[0], sa[1] + sb[1], sa[2] + sb[2], sa[3] + sb[3], max(sa[4], sb[4]), 2, 0)
    assert e.value.code == _ffi.ET_E_INVALID_ARG


def test_an_attester_whose_zero_opinions_went_elsewhere_is_refused():
    """Attester 3 gives its opinions with a value to process a and its zero-valued ones to process b: b sees a peer
    without opinions, a's own figures equal the sums, and only the count of attestations gives the split away."""
    n = 40
    frm = np.array([3, 3, 5, 7, 3, 3], dtype=np.uint32)
    to = np.array([1, 2, 6, 8, 9, 10], dtype=np.uint32)
    val = np.array([4, 5, 6, 7, 0, 0], dtype=np.uint32)
    a = pa.DistributedPlan(n, frm[:4], to[:4], val[:4])
    b = pa.DistributedPlan(n, frm[4:], to[4:], val[4:])
    sa, sb = a.local_stats(), b.local_stats()
    assert sb[2][3] == 0 and sb[3][3] == 2                   # dangling in b, but it attested there
    for plan, rank in ((a, 0), (b, 1)):
        with pytest.raises(_ffi.EigenTrustError) as e:
            plan.set_global_stats(sa[0] + sb[0], sa[1] + sb[1], sa[2] + sb[2], sa[3] + sb[3], max(sa[4], sb[4]), 2, rank)
        assert e.value.code == _ffi.ET_E_INVALID_ARG


def test_a_global_maximum_below_the_local_one_is_refused():
    n = 30
    frm = np.array([1, 2], dtype=np.uint32)
    to = np.array([2, 1], dtype=np.uint32)
    val = np.array([7, 0x90000000], dtype=np.uint32)
    p = pa.DistributedPlan(n, frm, to, val)
    st = p.local_stats()
    assert st[4] == 0x90000000
    with pytest.raises(_ffi.EigenTrustError) as e:
        p.set_global_stats(st[0], st[1], st[2], st[3], 255, 1, 0)
    assert e.value.code == _ffi.ET_E_INVALID_ARG


def test_stages_are_enforced_and_too_few_peers_are_reported():
    n = 50
    frm, to, val, member = messy_case(n, 300, 5)
    p = pa.DistributedPlan(n, frm, to, val, member=member)
    with pytest.raises(_ffi.EigenTrustError):
        p.outgoing(0)                                        # no geometry yet
    st = p.local_stats()
    p.set_global_stats(st[0], st[1], st[2], st[3], st[4], 1, 0)
    with pytest.raises(_ffi.EigenTrustError):
        p.local_stats()
    with pytest.raises(_ffi.EigenTrustError):
        p.add_incoming(np.array([1 << 31], dtype=np.uint32), np.array([0], dtype=np.uint32), np.array([1], dtype=np.uint32)) or p.finish_host()
    lonely = np.zeros(n, dtype=np.uint8)
    lonely[3] = 1
    with pytest.raises(_ffi.EigenTrustError) as e:          # native.rs:288-289
        pa.DistributedPlan(n, frm, to, val, member=lonely)
    assert e.value.code == _ffi.ET_E_INSUFFICIENT_PEERS


def _worker(rank, world, port, case, ret):
    import torch
    from protocol_amd.sharded import distributed_plan
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, frm, to, val, member = case
    mine = frm % world == rank                               # this rank's attesters
    plan = distributed_plan(n, frm[mine], to[mine], val[mine], member=member, world=world, rank=rank,
                            comm_device=torch.device("cpu"))
    try:
        ref = pa.Plan(n, frm, to, value_u32=val, member=member, n_seg=world, seg=rank)
        assert ref.info.gather == 1
        assert_same_share(plan.finish_host(), ref, rank)
        ret[rank] = "ok"
    except AssertionError as e:
        ret[rank] = "differs: %r" % (e,)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_agree_on_the_layout_over_gloo():
    """The collectives themselves: all-reduce of the per-peer figures, all-to-all of the edge records, world size 2."""
    n = 2500
    frm, to, val, member = messy_case(n, 20000, 17)
    ret = mp.Manager().dict()
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, (n, frm, to, val, member), ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_an_opinion_of_2_to_the_31_on_one_rank_reaches_every_rank():
    """The largest opinion is agreed on as a 64-bit maximum: one rank holds byte opinions, the other an opinion with the
    top bit set.  Both must lay out 32-bit edge records (the per-row stream; no byte-wide blocked sweep), equal to the
    plan made from the whole list."""
    n = 2500
    frm, to, val, member = messy_case(n, 20000, 23)
    val[:] = val % 200
    big = np.flatnonzero((frm % 2 == 1) & (val > 0))[0]      # an attestation that rank 1 ingests
    val[big] = 0x80000001
    ret = mp.Manager().dict()
    port = 33500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, (n, frm, to, val, member), ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}
    assert int(pa.Plan(n, frm, to, value_u32=val, member=member, n_seg=2, seg=0).value_u32.max()) == 0x80000001
