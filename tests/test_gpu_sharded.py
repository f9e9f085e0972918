"""Row-sharded converge with the real HIP kernels: two ranks (both on cuda:0) exchanging segments
through torch.distributed.  gloo stands in for RCCL here because two ranks of one communicator cannot
share a GPU under RCCL; the kernels, buffers and the protocol are the ones bench.py uses with nccl."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for pth in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if pth not in sys.path:
        sys.path.insert(0, pth)

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, case, ret):
    import torch
    import torch.distributed as dist
    import protocol_amd as pa
    from protocol_amd.sharded import ShardedGraph
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, member, frm, to, val, iters = case[:6]
    chunks = case[6] if len(case) > 6 else 1
    if len(case) > 7 and case[7]:
        # this rank ingests the attestations of its own attesters only; the ranks agree on the layout and route the edges
        mine = (frm * 2654435761 >> 7) % world == rank
        sg = ShardedGraph(n, frm[mine], to[mine], value_u32=val[mine], member=member, device=torch.device("cuda", 0),
                          n_chunks=chunks, local_attestations=True)
    else:
        sg = ShardedGraph(n, frm, to, value_u32=val, member=member, device=torch.device("cuda", 0), n_chunks=chunks)
    assert sg.n_chunks == chunks
    assert int(sg.graph.info.exact_sweep_blocks) != 0 and int(sg.graph.info.f64_sweep_blocks) != 0
    fr = sg.converge_fr(iters, 1000)          # raises on a conservation failure
    f64 = sg.converge_f64(iters, 1000.0)
    damped = sg.converge_f64(iters, 1000.0, alpha=0.15)     # default pre-trust: the initial scores
    start = sg.converge_fr(0, 1000)                         # zero iterations: the initial vector
    if rank == 0:
        ret["fr"] = fr.tobytes()
        ret["f64"] = f64.tolist()
        ret["damped"] = damped.tolist()
        ret["start"] = start.tobytes()
        ret["nnz"] = sg.nnz
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks,local", [(2, 1, False), (3, 1, False), (2, 3, False), (2, 1, True), (3, 2, True)])
def test_sharded_converge_matches_oracle(world, chunks, local):
    import c_oracle
    import protocol_amd as pa
    n = 50000
    frm, to, val = pa.synth_graph(3, n, 16 * n, seed=5, value_max=255)
    if local:   # repeated pairs: the later one counts, whichever rank ingests the attester
        rng = np.random.default_rng(8)
        idx = rng.integers(0, len(frm), len(frm) // 50)
        frm, to = np.concatenate([frm, frm[idx]]), np.concatenate([to, to[idx]])
        val = np.concatenate([val, rng.integers(0, 256, len(idx), dtype=np.uint32)])
    member = np.ones(n, dtype=np.uint8)
    member[::97] = 0
    iters = 20
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000)
    assert rc == 0
    rc, _e64, eld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters,
                                                             initial=1000.0, want_ld=True)
    assert rc == 0
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() + world + 10 * chunks + 100 * local) % 2000
    mp.spawn(_worker, args=(world, port, (n, member, frm, to, val, iters, chunks, local), ret), nprocs=world, join=True)
    assert ret["nnz"] == nnz
    assert ret["fr"] == exp.tobytes()
    got = np.array(ret["f64"], dtype=np.longdouble)
    assert float(np.abs(got - eld).sum() / np.abs(eld).sum()) <= 1e-9
    rc, _e, dld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000.0,
                                                           alpha=0.15, pretrust=1000.0 * member, want_ld=True)
    assert rc == 0
    got = np.array(ret["damped"], dtype=np.longdouble)
    assert float(np.abs(got - dld).sum() / np.abs(dld).sum()) <= 1e-9
    assert ret["start"] == pa.fr_bytes_from_ints([1000 if m else 0 for m in member]).tobytes()


def test_two_ranks_with_cited_entries_ordered_by_workgroup():
    """Shares of half a million rows laid out from rank-local attestations: each rank asks the other for the entries it
    cites in the order its workgroups want them (et_dplan_needs), and packs what the other asked for."""
    import c_oracle
    import protocol_amd as pa
    n = 1 << 20
    frm, to, val = pa.synth_graph(3, n, 12 * n, seed=31, value_max=255)
    member = np.ones(n, dtype=np.uint8)
    member[::997] = 0
    iters = 5
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0
    rc, _e64, eld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000.0,
                                                             n_threads=c_oracle.max_threads(), want_ld=True)
    assert rc == 0
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 27500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, (n, member, frm, to, val, iters, 1, True), ret), nprocs=2, join=True)
    assert ret["nnz"] == nnz
    assert ret["fr"] == exp.tobytes()
    got = np.array(ret["f64"], dtype=np.longdouble)
    assert float(np.abs(got - eld).sum() / np.abs(eld).sum()) <= 1e-9


@pytest.mark.parametrize("world,full_vector,value_max", [(4, False, 255), (8, False, 255), (4, True, 255), (4, False, 2**32 - 1)])
def test_all_shares_in_one_process(world, full_vector, value_max):
    """The kernels on the layouts of 4 and 8 shares (what the 4- and 8-GPU runs execute), every share held by
    this one process and the exchange done as device copies: the cited entries packed by the producer into the
    consumers' gather vectors, or whole segments in position order (full_vector).  Byte opinions and 32-bit ones (the
    exact sweep's ten-limb rows on gather vectors)."""
    import torch
    import c_oracle
    import protocol_amd as pa
    n = 60000
    frm, to, val = pa.synth_graph(3, n, 16 * n, seed=9, value_max=value_max)
    member = np.ones(n, dtype=np.uint8)
    member[::89] = 0
    iters = 20
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000)
    assert rc == 0
    rc, _e64, eld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters,
                                                             initial=1000.0, want_ld=True)
    assert rc == 0
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    st = torch.cuda.current_stream(dev).cuda_stream
    graphs = [pa.Graph(n, frm, to, value_u32=val, member=member, n_seg=world, seg=r, device=0, full_vector=full_vector)
              for r in range(world)]
    exs = [g.exchange() for g in graphs]
    stride = int(exs[0].own_len)
    assert all(bool(ex.gather) == (not full_vector) for ex in exs)
    assert int(graphs[0].info.nnz) == nnz and sum(int(g.info.nnz_local) for g in graphs) == nnz
    if not full_vector:   # own block, the share's own cited entries, the other shares' cited entries and partial sums
        assert all(int(ex.own_off) == 0 and int(ex.self_off) == stride and int(ex.self_count) <= stride and
                   int(ex.vec_len) == stride + int(ex.self_count) + sum(int(c) for c in ex.recv_count) for ex in exs)
        assert all(int(ex.vec_len) <= int(ex.full_len) + int(ex.self_count) for ex in exs)

    def exchange(vecs, kind):
        for r, (g, ex) in enumerate(zip(graphs, exs)):
            own = vecs[r][int(ex.own_off):int(ex.own_off) + stride]
            if full_vector:
                for q in range(world):
                    if q != r:
                        vecs[q][r * stride:(r + 1) * stride].copy_(own)
                continue
            send = torch.zeros((int(ex.send_total),) + tuple(vecs[r].shape[1:]), dtype=vecs[r].dtype, device=dev)
            (g.pack_fr if kind == "fr" else g.pack_f64)(vecs[r].data_ptr(), send.data_ptr(), st)
            for q in range(world):
                if q == r:
                    continue
                cnt = int(ex.send_count[q])
                assert cnt == int(exs[q].recv_count[r])
                vecs[q][int(exs[q].recv_off[r]):int(exs[q].recv_off[r]) + cnt].copy_(send[int(ex.send_off[q]):int(ex.send_off[q]) + cnt])

    for kind in ("fr", "f64"):
        mk = lambda ln: torch.zeros((ln, 4), dtype=torch.int64, device=dev) if kind == "fr" else torch.zeros(ln, dtype=torch.float64, device=dev)
        xs = [mk(int(ex.vec_len)) for ex in exs]
        ys = [mk(int(ex.vec_len)) for ex in exs]
        for g, x in zip(graphs, xs):
            (g.init_fr if kind == "fr" else g.init_f64)(x.data_ptr(), 1000, st)
        exchange(xs, kind)
        for it in range(iters):
            for g, x, y in zip(graphs, xs, ys):
                if kind == "fr":
                    g.step_fr(x.data_ptr(), y.data_ptr(), it, st, final=it + 1 == iters)
                else:
                    g.step_f64(x.data_ptr(), y.data_ptr(), it, 0.0, None, st, final=it + 1 == iters)
            if it + 1 < iters:
                exchange(ys, kind)
            xs, ys = ys, xs
        # the result: every share's own block, in position order
        full = torch.cat([x[int(ex.own_off):int(ex.own_off) + stride] for x, ex in zip(xs, exs)])
        assert full.shape[0] == int(exs[0].full_len)
        out = torch.zeros((n, 4), dtype=torch.int64, device=dev) if kind == "fr" else torch.zeros(n, dtype=torch.float64, device=dev)
        (graphs[0].unpermute_fr if kind == "fr" else graphs[0].unpermute_f64)(full.data_ptr(), out.data_ptr(), st)
        torch.cuda.synchronize(dev)
        if kind == "fr":
            assert out.cpu().numpy().view(np.uint8).reshape(n, 32).tobytes() == exp.tobytes()
        else:
            got = out.cpu().numpy().astype(np.longdouble)
            assert float(np.abs(got - eld).sum() / np.abs(eld).sum()) <= 1e-9


def _nccl_worker(rank, world, port, case, ret):
    import torch
    import torch.distributed as dist
    from protocol_amd.sharded import ShardedGraph
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n, member, frm, to, val, iters = case[:6]
    chunks = case[6] if len(case) > 6 else 1
    local = len(case) > 7 and case[7]
    # local: the layout through the rank-local path, whose all-reduces and all-to-all then run on device tensors over RCCL
    sg = ShardedGraph(n, frm, to, value_u32=val, member=member, device=dev, always_exchange=True, n_chunks=chunks,
                      local_attestations=local)
    assert sg.n_chunks == chunks
    ret["fr"] = sg.converge_fr(iters, 1000).tobytes()      # conservation checked through the 32-byte tensor gather
    ret["f64"] = sg.converge_f64(iters, 1000.0).tolist()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("chunks,local", [(1, False), (3, False), (1, True)])
def test_rccl_backend_with_the_one_rank_a_single_gpu_allows(chunks, local):
    """The driver's RCCL calls themselves (in-place all_gather_into_tensor of the segments, the 32-byte gather of the
    conservation check) with backend nccl: one rank is all a single GPU allows, so the collective copies the vector
    onto itself, but it is the same call, on the same tensors, that N ranks make."""
    import c_oracle
    import protocol_amd as pa
    n = 40000
    frm, to, val = pa.synth_graph(3, n, 16 * n, seed=11, value_max=255)
    member = np.ones(n, dtype=np.uint8)
    member[::53] = 0
    iters = 6
    rc, exp, _nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000)
    assert rc == 0
    rc, _e64, eld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters,
                                                             initial=1000.0, want_ld=True)
    assert rc == 0
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() + 7) % 2000
    mp.spawn(_nccl_worker, args=(1, port, (n, member, frm, to, val, iters, chunks, local), ret), nprocs=1, join=True)
    assert ret["fr"] == exp.tobytes()
    got = np.array(ret["f64"], dtype=np.longdouble)
    assert float(np.abs(got - eld).sum() / np.abs(eld).sum()) <= 1e-9
