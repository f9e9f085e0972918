"""The benchmark configurations at their full sizes, through the C ABI, against the CPU checker:
64K peers / 1M attestations, 16M peers / 256M attestations (one GPU, and the layout eight GPUs run),
4096 sets of 256 slots.  Bit-exact for the field variant, <= 1e-9 relative L1 for f64."""
import hashlib

import numpy as np
import pytest

import c_oracle
import protocol_amd as pa

pytestmark = pytest.mark.gpu
F64_TOL = 1e-9


def rel_l1(a, b):
    a = np.asarray(a, dtype=np.longdouble)
    b = np.asarray(b, dtype=np.longdouble)
    return float(np.abs(a - b).sum() / np.abs(b).sum())


def test_64k_peers_1m_attestations_full_result():
    n = 1 << 16
    frm, to, val = pa.synth_graph(2, n, 1 << 20, seed=42, value_max=255)
    member = np.ones(n, dtype=np.uint8)
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    assert g.info.nnz == 1 << 20
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=20, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0 and nnz == g.info.nnz
    assert np.array_equal(g.converge_fr(20, 1000), exp)
    rc, _e, eld, _r, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=20, initial=1000.0,
                                                         n_threads=c_oracle.max_threads(), want_ld=True)
    assert rc == 0
    o64, _ = g.converge_f64(20, 1000.0)
    assert rel_l1(o64, eld) <= F64_TOL


def test_1m_peers_16m_attestations_with_32_bit_opinions():
    """The benchmark graph with opinions drawn from 1..2^32-1 (the stress set for opinions that do not fit a byte): whole
    result on one GPU, and the same graph cut into two shares with gather vectors."""
    n = 1 << 20
    frm, to, val = pa.synth_graph(3, n, 1 << 24, seed=42, value_max=0xFFFFFFFF)
    assert int(val.max()) > 0xF0000000
    member = np.ones(n, dtype=np.uint8)
    iters = 20
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0
    rc, _e, eld, _r, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000.0,
                                                         n_threads=c_oracle.max_threads(), want_ld=True)
    assert rc == 0
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    assert int(g.info.nnz) == nnz and int(g.info.exact_sweep_blocks) == 256     # 38-byte rows: one round of 256 workgroups
    assert np.array_equal(g.converge_fr(iters, 1000), exp)
    o64, _ = g.converge_f64(iters, 1000.0)
    assert rel_l1(o64, eld) <= F64_TOL
    g.close()
    mg = pa.MultiGpuGraph(n, frm, to, value_u32=val, member=member, devices=[0, 0])
    assert int(mg.info(0).gather) == 1 and int(mg.info(1).exact_sweep_blocks) != 0
    assert np.array_equal(mg.converge_fr(iters, 1000), exp)
    o64, _ = mg.converge_f64(iters, 1000.0)
    assert rel_l1(o64, eld) <= F64_TOL
    mg.close()


@pytest.fixture(scope="module")
def big_graph():
    n = 1 << 24
    frm, to, val = pa.synth_graph(3, n, 1 << 28, seed=42, value_max=255)
    return n, frm, to, val, np.ones(n, dtype=np.uint8)


@pytest.fixture(scope="module")
def big_device_graph(big_graph):
    import time
    n, frm, to, val, member = big_graph
    t0 = time.time()
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    print("16M-peer graph laid out and uploaded in %.1f s" % (time.time() - t0), flush=True)
    yield g
    g.close()


def test_16m_peers_256m_attestations_on_one_gpu_exact(big_graph, big_device_graph):
    """Whole result of 20 iterations of the exact variant, by sha256 against the C restatement."""
    import time
    n, frm, to, val, member = big_graph
    g = big_device_graph
    t0 = time.time()
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=20, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    print("CPU checker: %.1f s on %d threads" % (time.time() - t0, c_oracle.max_threads()), flush=True)
    assert rc == 0 and nnz == g.info.nnz
    t0 = time.time()
    out = g.converge_fr(20, 1000)
    print("GPU converge and read-back: %.1f s" % (time.time() - t0), flush=True)
    assert hashlib.sha256(out.tobytes()).hexdigest() == hashlib.sha256(exp.tobytes()).hexdigest()


def test_16m_peers_256m_attestations_on_one_gpu_f64(big_graph, big_device_graph):
    """Whole result of 20 iterations of the f64 variant against the long-double restatement."""
    n, frm, to, val, member = big_graph
    rc, _e, eld, _r, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=20, initial=1000.0,
                                                         n_threads=c_oracle.max_threads(), want_ld=True)
    assert rc == 0
    o64, _ = big_device_graph.converge_f64(20, 1000.0)
    assert rel_l1(o64, eld) <= F64_TOL


def test_16m_peers_in_the_eight_gpu_layout(big_graph):
    """The same graph cut into the eight shares an 8-GPU run holds, every share resident on this one GPU and driven by
    the one-thread C driver (the exchange as device copies of the segments); three iterations of the exact variant."""
    n, frm, to, val, member = big_graph
    world, iters = 8, 3
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0
    mg = pa.MultiGpuGraph(n, frm, to, value_u32=val, member=member, devices=[0] * world, variants=1)
    assert sum(int(mg.info(r).nnz_local) for r in range(world)) == nnz
    assert all(int(mg.info(r).exact_sweep_blocks) == 512 for r in range(world))
    got = mg.converge_fr(iters, 1000)          # ET_E_SUM_MISMATCH if the score sum over the shares changed (native.rs:331-334)
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(exp.tobytes()).hexdigest()
    mg.close()


def test_4096_sets_of_256_slots():
    """The whole batch: every set reports success (conservation is checked per set on the device), and 32 sets
    spread over the batch are compared with the dense restatement bit for bit."""
    sets, slots = 4096, 256
    values = np.empty((sets, slots, slots), dtype=np.uint32)
    member = np.ones((sets, slots), dtype=np.uint8)
    for s in range(sets):
        rng = np.random.default_rng([47, s])
        v = rng.integers(1, 256, (slots, slots), dtype=np.uint32)
        v[rng.random(slots) < 0.05] = 0
        values[s] = v
        member[s, int(rng.integers(slots - slots // 16, slots + 1)):] = 0
    out, status = pa.Batch(values, member).converge_fr(20, 1000)
    assert (status == 0).all()
    for s in range(0, sets, sets // 32):
        rows = [[int(v) for v in values[s, i]] for i in range(slots)]
        rc, exp, _ = c_oracle.dense_converge_fr(slots, 20, 1000, member[s], member[s], rows)
        assert rc == 0 and pa.fr_ints_from_bytes(out[s]) == exp, s


@pytest.mark.parametrize("case", ["bytes", "field", "blocked"])
def test_steps_own_their_whole_segment_tail(case):
    """Vectors handed to the stepping calls need no zero fill: buffers that start as all-ones bytes give the same
    scores.  (The partial sums of the dangling peers' score mass live in each segment's tail; a step writes all of it.)"""
    import random
    import torch
    rng = np.random.default_rng(3)
    n = 600000 if case == "blocked" else 5000
    deg = rng.poisson(6, n)
    deg[rng.random(n) < 0.1] = 0                       # peers without opinions
    frm = np.repeat(np.arange(n, dtype=np.uint32), deg)
    to = rng.integers(0, n, len(frm), dtype=np.uint32)
    member = (rng.random(n) > 0.03).astype(np.uint8)
    iters = 4
    if case == "field":
        r = random.Random(5)
        vb = pa.fr_bytes_from_ints([r.randrange(pa.R_MODULUS) for _ in range(len(frm))])
        kw = dict(value_fr=vb)
    else:
        kw = dict(value_u32=rng.integers(0, 256, len(frm), dtype=np.uint32))
    rc, exp, _nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, num_iter=iters, initial=1000,
                                                   n_threads=c_oracle.max_threads(), **kw)
    assert rc == 0
    g = pa.Graph(n, frm, to, member=member, **kw)
    assert (int(g.info.exact_sweep_blocks) != 0) == (case == "blocked")
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    vec_len = int(g.info.vec_len)
    x = torch.full((vec_len, 4), -1, dtype=torch.int64, device=dev)
    y = torch.full((vec_len, 4), -1, dtype=torch.int64, device=dev)
    g.init_fr(x.data_ptr(), 1000, st)
    for it in range(iters):
        g.step_fr(x.data_ptr(), y.data_ptr(), it, st, final=it + 1 == iters)
        x, y = y, x
    out = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    g.unpermute_fr(x.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize(dev)
    assert np.array_equal(out.cpu().numpy().view(np.uint8).reshape(n, 32), exp)
    if case == "field":
        return                                          # the f64 variant is for integer opinions
    rc, _e, eld, _r, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, num_iter=iters, initial=1000.0,
                                                         n_threads=c_oracle.max_threads(), want_ld=True, **kw)
    assert rc == 0
    nan = float("nan")
    x = torch.full((vec_len,), nan, dtype=torch.float64, device=dev)
    y = torch.full((vec_len,), nan, dtype=torch.float64, device=dev)
    g.init_f64(x.data_ptr(), 1000.0, st)
    for it in range(iters):
        g.step_f64(x.data_ptr(), y.data_ptr(), it, 0.0, None, st, final=it + 1 == iters)
        x, y = y, x
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    g.unpermute_f64(x.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize(dev)
    assert rel_l1(out.cpu().numpy(), eld) <= F64_TOL
