"""Host logic behind the C ABI that needs no GPU: symbols, set API errors, converge_rational, planner."""
import ctypes as C
import os
import re
from fractions import Fraction

import numpy as np
import pytest

import c_oracle
import protocol_amd as pa
from plan_emulator import SegmentEmulator
from protocol_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "eigentrust_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(et_[a-z0-9_]+)\s*\(", header))
    assert len(declared) > 30
    handle = C.CDLL(pa.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), "missing export: " + name
    assert declared == set(_ffi.PROTOTYPES), declared ^ set(_ffi.PROTOTYPES)
    assert pa.lib().et_abi_version() >= 1
    assert pa.lib().et_strerror(_ffi.ET_E_INSUFFICIENT_PEERS) == b"Insufficient peers for calculation!"


def test_set_api_error_codes_mirror_reference_panics():
    s = pa.EigenTrustSet(42, 2, 10, 1000)
    s.add_member(11)
    with pytest.raises(pa.EigenTrustError) as e:
        s.add_member(11)                       # native.rs:178
    assert e.value.code == _ffi.ET_E_DUP_MEMBER
    with pytest.raises(pa.EigenTrustError) as e:
        s.converge()                           # native.rs:289 (checked before any GPU work)
    assert e.value.code == _ffi.ET_E_INSUFFICIENT_PEERS
    s.add_member(12)
    with pytest.raises(pa.EigenTrustError) as e:
        s.add_member(13)                       # native.rs:181
    assert e.value.code == _ffi.ET_E_SET_FULL
    with pytest.raises(pa.EigenTrustError) as e:
        s.remove_member(99)                    # native.rs:192
    assert e.value.code == _ffi.ET_E_NOT_MEMBER
    with pytest.raises(pa.EigenTrustError) as e:
        s.update_op(99, [0, 0])                # opinion/native.rs:66-67
    assert e.value.code == _ffi.ET_E_NOT_MEMBER
    with pytest.raises(pa.EigenTrustError) as e:
        s.update_op(11, [pa.R_MODULUS, 0])     # not a canonical field element
    assert e.value.code == _ffi.ET_E_NONCANONICAL_FR
    with pytest.raises(pa.EigenTrustError) as e:
        s.add_member(0)                        # zero marks an empty slot
    assert e.value.code == _ffi.ET_E_INVALID_ARG
    s.remove_member(12)
    assert s.members()[0] == [11, 0] and s.members()[1] == [1000, 0]
    s.add_member(77)                           # re-uses the freed slot (native.rs:180)
    assert s.members()[0] == [11, 77]


def replay(sc):
    s = pa.EigenTrustSet(sc["domain"], sc["num_neighbours"], sc["num_iterations"], sc["initial_score"])
    k = 0
    for st in sc["steps"]:
        if st[0] == "add":
            s.add_member(int(st[1], 16))
        elif st[0] == "remove":
            s.remove_member(int(st[1], 16))
        elif st[0] == "op":
            s.update_op(int(st[1], 16), [int(v, 16) for v in st[2]])
        else:
            yield s, sc["checkpoints"][k]
            k += 1


def test_converge_rational_matches_golden(scenarios):
    for sc in scenarios:
        for s, exp in replay(sc):
            q = s.converge_rational()
            want = [Fraction(int(a.split("/")[0], 16), int(a.split("/")[1], 16)) for a in exp["scores_rational"]]
            assert q == want, sc["name"]
            assert [hex(a) for a in s.members()[0]] == exp["slots"]


def test_converge_rational_buffer_protocol():
    s = pa.EigenTrustSet(42, 4, 20, 1000)
    for a in (1, 2, 3):
        s.add_member(a)
    s.update_op(1, [0, 3, 4, 0])
    need = C.c_size_t(0)
    rc = pa.lib().et_set_converge_rational(s._h, None, None, 0, C.byref(need))
    assert rc == _ffi.ET_E_BUFFER_TOO_SMALL and need.value >= 2
    small = (C.c_uint8 * 4)()
    assert pa.lib().et_set_converge_rational(s._h, small, small, 1, None) == _ffi.ET_E_BUFFER_TOO_SMALL


def test_compute_entry_points_fail_loudly_without_a_gpu():
    if pa.lib().et_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pa.EigenTrustError) as e:
        pa.Graph(4, [0, 1], [1, 0], value_u32=[1, 1])
    assert e.value.code in (_ffi.ET_E_NO_DEVICE, _ffi.ET_E_HIP)
    s = pa.EigenTrustSet(42, 4, 20, 1000)
    s.add_member(1)
    s.add_member(2)
    with pytest.raises(pa.EigenTrustError) as e:
        s.converge()
    assert e.value.code in (_ffi.ET_E_NO_DEVICE, _ffi.ET_E_HIP)


def test_synthetic_generator_is_deterministic_and_well_formed():
    f1, t1, v1 = pa.synth_graph(3, 20000, 320000, seed=42)
    f2, t2, v2 = pa.synth_graph(3, 20000, 320000, seed=42)
    assert np.array_equal(f1, f2) and np.array_equal(t1, t2) and np.array_equal(v1, v2)
    assert len(f1) == 320000 and (f1 != t1).all() and v1.min() >= 1 and v1.max() <= 255
    key = f1.astype(np.uint64) << 32 | t1
    assert len(np.unique(key)) == len(key)                      # no duplicate (from, to)
    outd = np.bincount(f1, minlength=20000)
    assert 100 <= (outd == 0).sum() <= 320                      # about 1 % without opinions
    f3, _, _ = pa.synth_graph(3, 20000, 320000, seed=43)
    assert not np.array_equal(f1, f3)
    f4, t4, _ = pa.synth_graph(2, 4096, 65536, seed=42)
    assert (np.bincount(f4, minlength=4096) == 16).all()


def random_case(seed, n=400):
    rng = np.random.default_rng(seed)
    member = (rng.random(n) > 0.05).astype(np.uint8)
    deg = rng.poisson(6, n)
    deg[rng.random(n) < 0.1] = 0
    frm = np.repeat(np.arange(n, dtype=np.uint32), deg)
    to = rng.integers(0, n, len(frm), dtype=np.uint32)
    sel = rng.random(len(frm)) < 0.3
    to[sel] = rng.integers(0, 2, sel.sum())      # two very popular peers -> wavefront-reduced rows
    val = rng.integers(0, 256, len(frm), dtype=np.uint32)
    return n, member, frm, to, val


def test_plan_layout_is_a_faithful_rearrangement():
    n, member, frm, to, val = random_case(1)
    p = pa.Plan(n, frm, to, value_u32=val, member=member, long_row=16, target_work=9)
    pos = p.pos_of_slot
    assert ((pos == 0xFFFFFFFF) == (member == 0)).all()
    assert len(np.unique(pos[member == 1])) == member.sum()     # positions are distinct
    # every slice belongs to exactly one block, blocks are contiguous and ordered
    covered = []
    for sb, se, flags, dslot in p.work:
        covered.extend(range(sb, se))
        if flags & 1:
            assert dslot < p.info.nd_cap
    assert covered == list(range(p.info.n_slices))
    # non-zero stream entries == surviving attestations of non-dangling members (last write wins)
    assert int((p.value_u32 != 0).sum()) <= p.info.nnz_local == p.info.nnz
    assert p.info.n_valid == member.sum()


@pytest.mark.parametrize("seed", [1, 2])
def test_plan_executed_on_cpu_matches_oracle(seed):
    n, member, frm, to, val = random_case(seed, n=250)
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=4, initial=1000)
    assert rc == 0
    em = SegmentEmulator(pa.Plan(n, frm, to, value_u32=val, member=member, long_row=16, target_work=5))
    assert em.p.info.nnz == nnz
    x = em.init(1000)
    for it in range(4):
        y = [0] * len(x)
        em.step(x, y, final=it == 3)
        x = y
    assert em.unpermute(x) == c_oracle.fr_bytes_to_ints(exp)
