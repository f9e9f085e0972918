"""The multi-GPU exchange protocol on CPU: world_size 2, gloo.

Each rank plans its own segment of the graph (protocol_amd.Plan, host only), executes its rows with
the plan emulator (exact integers / doubles) and exchanges segments with one all-gather per
iteration, exactly as protocol_amd.sharded.ShardedGraph does with the HIP kernels and RCCL.
The result must equal the CPU checker's, bit for bit in the field and to 1e-12 in f64.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for pth in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if pth not in sys.path:
        sys.path.insert(0, pth)

LIMB = (1 << 62) - 1   # field elements travel as 5 x 62-bit limbs in int64 tensors


def _to_limbs(vals):
    return torch.tensor([[(v >> (62 * i)) & LIMB for i in range(5)] for v in vals], dtype=torch.int64)


def _from_limbs(t):
    return [sum(int(x) << (62 * i) for i, x in enumerate(row)) for row in t.tolist()]


def _worker(rank, world, port, case, ret):
    import c_oracle  # noqa: F401  (path check)
    import protocol_amd as pa
    from plan_emulator import SegmentEmulator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, member, frm, to, val, iters, full_vector = case
    em = SegmentEmulator(pa.Plan(n, frm, to, value_u32=val, member=member, n_seg=world, seg=rank,
                                 long_row=16, target_work=6, full_vector=full_vector))
    assert em.gather == (not full_vector)
    own = em.own_slice()

    def as_tensor(vals, field):
        return _to_limbs(vals) if field else torch.tensor(np.asarray(vals, dtype=np.float64))

    def all_gather_own(x, field):
        """every share's own block, in share order"""
        mine = as_tensor(x[own], field)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        return [_from_limbs(p) if field else p.numpy() for p in parts]

    def exchange(x, field):
        """Completes x after init or a step, the way ShardedGraph does: an all-gather of own blocks in position order,
        or pack + all-to-all of the cited entries into the gather vector."""
        if not em.gather:
            for r, vals in enumerate(all_gather_own(x, field)):
                x[r * em.stride:(r + 1) * em.stride] = vals
            return
        em.pack_self(x)                                  # the pack also fills the block of the share's own cited entries
        send = [em.pack(x, c) if c != rank else [] for c in range(world)]
        t_in = as_tensor([v for part in send for v in part], field)
        in_splits = [len(part) for part in send]
        out_splits = [int(em.p.exchange.recv_count[p]) if p != rank else 0 for p in range(world)]
        assert [int(em.p.exchange.send_count[c]) if c != rank else 0 for c in range(world)] == in_splits
        t_out = torch.empty((sum(out_splits),) + tuple(t_in.shape[1:]), dtype=t_in.dtype)
        dist.all_to_all_single(t_out, t_in, out_splits, in_splits)
        got = _from_limbs(t_out) if field else t_out.numpy()
        at = 0
        for p in range(world):
            if p != rank:
                x[em.recv_slice(p)] = got[at:at + out_splits[p]]
                at += out_splits[p]

    out = {}
    for field in (True, False):
        x = em.init(1000 if field else 1000.0, field)
        exchange(x, field)
        for it in range(iters):
            y = [0] * len(x) if field else np.zeros(len(x))
            em.step(x, y, field, final=it + 1 == iters)
            if it + 1 < iters:
                exchange(y, field)
            x = y
        full = [v for part in all_gather_own(x, field) for v in part]      # the result: own blocks in position order
        out["fr" if field else "f64"] = em.unpermute(full, field)
    if rank == 0:
        ret["fr"] = out["fr"]
        ret["f64"] = list(out["f64"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("full_vector", [False, True])
def test_two_rank_exchange_matches_oracle(full_vector):
    import c_oracle
    rng = np.random.default_rng(12)
    n = 300
    member = (rng.random(n) > 0.05).astype(np.uint8)
    deg = rng.poisson(5, n)
    deg[rng.random(n) < 0.1] = 0
    frm = np.repeat(np.arange(n, dtype=np.uint32), deg)
    to = rng.integers(0, n, len(frm), dtype=np.uint32)
    sel = rng.random(len(frm)) < 0.25
    to[sel] = rng.integers(0, 2, sel.sum())
    val = rng.integers(0, 256, len(frm), dtype=np.uint32)
    iters = 3
    rc, exp, _nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000)
    assert rc == 0
    rc, e64, _ld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000.0)
    assert rc == 0
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port + int(full_vector), (n, member, frm, to, val, iters, full_vector), ret), nprocs=2, join=True)
    assert ret["fr"] == c_oracle.fr_bytes_to_ints(exp)
    got = np.array(ret["f64"])
    assert np.abs(got - e64).sum() / np.abs(e64).sum() <= 1e-12


def _case(seed, n):
    rng = np.random.default_rng(seed)
    member = (rng.random(n) > 0.05).astype(np.uint8)
    deg = rng.poisson(5, n)
    deg[rng.random(n) < 0.1] = 0
    frm = np.repeat(np.arange(n, dtype=np.uint32), deg)
    to = rng.integers(0, n, len(frm), dtype=np.uint32)
    sel = rng.random(len(frm)) < 0.25
    to[sel] = rng.integers(0, 2, sel.sum())
    val = rng.integers(0, 256, len(frm), dtype=np.uint32)
    return member, frm, to, val


def test_four_and_eight_segment_layouts_in_one_process():
    """The layouts of 4 and 8 shares (8 is the most the source encoding holds), all shares emulated in one
    process with the exchange as plain copies: whole own blocks in position order, or the cited entries into the
    gather vectors."""
    import c_oracle
    import protocol_amd as pa
    from plan_emulator import SegmentEmulator
    n, iters = 1500, 3
    member, frm, to, val = _case(21, n)
    rc, exp, _nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000)
    assert rc == 0
    rc, e64, _ld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000.0)
    assert rc == 0
    for world, full_vector in ((4, False), (8, False), (4, True)):
        ems = [SegmentEmulator(pa.Plan(n, frm, to, value_u32=val, member=member, n_seg=world, seg=r, long_row=16, target_work=6,
                                       full_vector=full_vector))
               for r in range(world)]
        # what share c says it needs of share p is what share p says it sends to c
        for c, ec in enumerate(ems):
            for p, ep in enumerate(ems):
                if c != p and not full_vector:
                    assert np.array_equal(ec.p.need[p], ep.p.send[c])
        for field in (True, False):
            xs = [em.init(1000 if field else 1000.0, field) for em in ems]

            def exchange(vecs):
                for r, em in enumerate(ems):
                    em.pack_self(vecs[r])
                    for q, eq in enumerate(ems):
                        if q == r:
                            continue
                        if full_vector:
                            vecs[q][em.own_slice()] = vecs[r][em.own_slice()]
                        else:
                            vecs[q][eq.recv_slice(r)] = em.pack(vecs[r], q)
            exchange(xs)
            for it in range(iters):
                ys = [[0] * len(x) if field else np.zeros(len(x)) for x in xs]
                for em, x, y in zip(ems, xs, ys):
                    em.step(x, y, field, final=it + 1 == iters)
                if it + 1 < iters:
                    exchange(ys)
                xs = ys
            full = [v for em, x in zip(ems, xs) for v in x[em.own_slice()]]
            got = ems[0].unpermute(full, field)
            if field:
                assert got == c_oracle.fr_bytes_to_ints(exp), world
            else:
                assert np.abs(np.array(got) - e64).sum() / np.abs(e64).sum() <= 1e-12, world
    with pytest.raises(pa.EigenTrustError):
        pa.Plan(n, frm, to, value_u32=val, member=member, n_seg=9, seg=0)
