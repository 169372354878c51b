"""GPU parity: Hamming matrix kernel + guided greedy matchers vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from vo_slam_test_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1], ids=["mfma", "valu"])
def ham_kernel(vo, request):
    """both forms of K6 (vo_set_option(VO_OPT_HAMMING_KERNEL)): int8 matrix-core dot products (default) / xor + popcount"""
    vo.set_option("hamming_kernel", request.param)
    yield request.param
    vo.set_option("hamming_kernel", 0)


@pytest.mark.parametrize("na,nb", [(1, 1), (7, 13), (1000, 1000), (1003, 999), (33, 2049), (0, 5), (129, 8), (31, 1024),
                                   (128, 136), (257, 127)])
def test_hamming_matrix(vo, orc, ham_kernel, na, nb):
    a, b = synth.random_descriptors(na, 1), synth.random_descriptors(nb, 2)
    d = vo.hamming_matrix(a, b)
    assert d.shape == (na, nb)
    if na and nb:
        assert np.array_equal(d, orc.hamming_matrix(a, b))
        bits = np.unpackbits(a[:5, None, :] ^ b[None, :7, :], axis=2).sum(2)
        assert np.array_equal(d[:5, :7], bits[:, :min(7, nb)])


def test_hamming_extremes(vo, ham_kernel):
    """all-zero / all-one descriptors (popcounts 0 and 256: the byte-sized pieces of the matrix-core form's ninth K-step) and
    sparse / dense ones in between"""
    z, o = np.zeros((3, 32), np.uint8), np.full((2, 32), 255, np.uint8)
    assert (vo.hamming_matrix(z, o) == 256).all() and (vo.hamming_matrix(z, z) == 0).all()
    assert (vo.hamming_matrix(o, z) == 256).all() and (vo.hamming_matrix(o, o) == 0).all()
    rng = np.random.default_rng(5)
    dens = rng.random((40, 1, 1)) ** 3
    a = np.packbits(rng.random((40, 32, 8)) < dens, axis=2).reshape(40, 32)
    b = np.packbits(rng.random((40, 32, 8)) < 1 - dens, axis=2).reshape(40, 32)
    want = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
    assert np.array_equal(vo.hamming_matrix(a, b), want)


def test_hamming_batch_dev(vo, orc, ham_kernel):
    import torch
    P, n = 5, 1000
    a = np.stack([synth.random_descriptors(n, 10 + p) for p in range(P)])
    b = np.stack([synth.random_descriptors(n, 50 + p) for p in range(P)])
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    td = torch.zeros((P, n, n), dtype=torch.int16, device="cuda")
    vo.hamming_matrix_batch_dev(ta, tb, td, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = td.cpu().numpy().view(np.uint16)
    for p in range(P):
        assert np.array_equal(got[p], orc.hamming_matrix(a[p], b[p]))


def test_hamming_random_shapes(vo, ham_kernel):
    """40 seeded random shapes (1 .. 1500 rows / columns: row blocks, column chunks and super-chunks of the matrix-core form
    cut at every kind of remainder) against numpy, with descriptors of every density"""
    rng = np.random.default_rng(2026)
    for k in range(40):
        na, nb = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))
        if k % 5 == 0:
            nb = int(rng.integers(1, 180)) * 8   # the 16-byte store path
        if k % 7 == 0:
            nb = 1024 + int(rng.integers(1, 400))  # a second super-chunk of column tiles
        da, db = rng.random((na, 1, 1)), rng.random((nb, 1, 1))
        a = np.packbits(rng.random((na, 32, 8)) < da, axis=2).reshape(na, 32)
        b = np.packbits(rng.random((nb, 32, 8)) < db, axis=2).reshape(nb, 32)
        got = vo.hamming_matrix(a, b)
        ca, cb = np.unpackbits(a, axis=1).astype(np.int32), np.unpackbits(b, axis=1).astype(np.int32)
        want = ca.sum(1)[:, None] + cb.sum(1)[None, :] - 2 * (ca @ cb.T)
        assert np.array_equal(got, want.astype(np.uint16)), (k, na, nb)


def test_hamming_unaligned_views(vo, orc, ham_kernel):
    """descriptor arrays at 4-byte (not 16-byte) aligned addresses, an odd column count and a matrix at a 2-byte aligned
    address: the dword-load / u16-store paths of both kernels"""
    import torch
    na, nb = 77, 203
    a, b = synth.random_descriptors(na, 3), synth.random_descriptors(nb, 4)
    ta = torch.zeros(na * 32 + 4, dtype=torch.uint8, device="cuda")
    tb = torch.zeros(nb * 32 + 12, dtype=torch.uint8, device="cuda")
    ta[4:].copy_(torch.from_numpy(a).reshape(-1))
    tb[12:].copy_(torch.from_numpy(b).reshape(-1))
    td = torch.full((na * nb + 1,), -1, dtype=torch.int16, device="cuda")
    vo.hamming_matrix_dev(ta[4:].view(na, 32), tb[12:].view(nb, 32), td[1:], stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = td.cpu().numpy().view(np.uint16)
    assert got[0] == 0xFFFF
    assert np.array_equal(got[1:].reshape(na, nb), orc.hamming_matrix(a, b))


def _frame_pair(orc, idx):
    p = orc.orb_params()
    f0 = synth.make_frame(idx)
    f1, dx, dy = synth.make_shifted(f0, idx)
    k0, d0, _ = orc.extract(p, f0)
    k1, d1, _ = orc.extract(p, f1)
    return k0, d0, k1, d1, dx, dy


def _shifted_frame(orc, idx, j):
    """neighbour j of frame idx: the frame shifted by a neighbour-specific integer offset with fresh noise"""
    f0 = synth.make_frame(idx)
    f1, dx, dy = synth.make_shifted(f0, 1000 * (idx + 1) + j)
    k1, d1, _ = orc.extract(orc.orb_params(), f1)
    return k1, d1, dx, dy


def _uright(k, seed):
    rng = np.random.default_rng(seed)
    z = rng.uniform(0.8, 4.5, len(k)).astype(np.float32)
    ur = (k["x"] - np.float32(40.0) / z).astype(np.float32)
    ur[rng.random(len(k)) < 0.1] = -1.0
    return ur, z


@pytest.mark.parametrize("idx,direction,check_rot", [(0, 0, 1), (1, 1, 1), (2, 2, 0), (3, 0, 1)])
def test_search_by_projection_frame(vo, orc, idx, direction, check_rot):
    import ctypes as C
    k0, d0, k1, d1, dx, dy = _frame_pair(orc, idx)
    ur1, _ = _uright(k1, idx)
    ur0, z0 = _uright(k0, idx + 100)
    sf = np.array(list(orc.orb_params().scale)[:8], np.float32)
    rng = np.random.default_rng(idx)
    q = dict(flags=(1 | (rng.random(len(k0)) < 0.7).astype(np.uint8) << 1).astype(np.uint8),
             u=(k0["x"] + dx + rng.normal(0, 1.0, len(k0))).astype(np.float32),
             v=(k0["y"] + dy + rng.normal(0, 1.0, len(k0))).astype(np.float32),
             invz=(1.0 / z0).astype(np.float32), octave=k0["octave"].astype(np.int32),
             angle=k0["angle"].astype(np.float32), desc=np.ascontiguousarray(d0))
    q["flags"][rng.random(len(k0)) < 0.05] = 0
    blocked = (rng.random(len(k1)) < 0.03).astype(np.uint8)
    cur = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    n, assigned = vo.Matcher(0.8).searchByProjection_frame(cur, q, 15.0, 40.0, direction, check_rot, sf, blocked)
    of = orc.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    oa = np.full(len(k1), -1, np.int32)
    on = orc.lib().orc_match_frame_projection(C.byref(of.c), len(k0), q["flags"], q["u"], q["v"], q["invz"],
                                              q["octave"], q["angle"], q["desc"], 15.0, 40.0, direction, check_rot,
                                              8, sf, blocked, oa)
    assert n == on and n > 200
    assert np.array_equal(assigned, oa)


@pytest.mark.parametrize("idx", [0, 4])
def test_search_local_map(vo, orc, idx):
    import ctypes as C
    k0, d0, k1, d1, dx, dy = _frame_pair(orc, idx)
    ur1, _ = _uright(k1, idx)
    sf = np.array(list(orc.orb_params().scale)[:8], np.float32)
    rng = np.random.default_rng(idx + 7)
    nq = len(k0)
    q = dict(flags=np.full(nq, 3, np.uint8), u=(k0["x"] + dx).astype(np.float32), v=(k0["y"] + dy).astype(np.float32),
             ur=(k0["x"] + dx - 12.0).astype(np.float32),
             level=np.clip(k0["octave"] + rng.integers(0, 2, nq), 0, 7).astype(np.int32),
             viewcos=rng.uniform(0.99, 1.0, nq).astype(np.float32), desc=np.ascontiguousarray(d0))
    cur = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    n, assigned = vo.Matcher(0.8).searchByProjection_localmap(cur, q, 3.0, sf)
    of = orc.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    oa = np.full(len(k1), -1, np.int32)
    on = orc.lib().orc_match_local_map(C.byref(of.c), nq, q["flags"], q["u"], q["v"], q["ur"], q["level"],
                                       q["viewcos"], q["desc"], 3.0, 0.8, sf, np.zeros(len(k1), np.uint8), oa)
    assert n == on and n > 100
    assert np.array_equal(assigned, oa)


# ------------------------------------------------------------------ M2 / M5 / M6 / M8 / M9

def _nodes(k, dx=0.0, dy=0.0):
    """synthetic vocabulary node of a feature: coarse cell of its (shift-compensated) position"""
    return (np.floor((k["x"] - dx) / 64.0).astype(np.int64) + 16 * np.floor((k["y"] - dy) / 64.0).astype(np.int64)
            + 1000).astype(np.uint32)


@pytest.mark.parametrize("idx,check_rot", [(0, 1), (1, 0), (2, 1)])
def test_search_by_projection_keyframe(vo, orc, idx, check_rot):
    import ctypes as C
    k0, d0, k1, d1, dx, dy = _frame_pair(orc, idx)
    ur1, _ = _uright(k1, idx)
    sf = np.array(list(orc.orb_params().scale)[:8], np.float32)
    rng = np.random.default_rng(idx + 7)
    flags = (rng.random(len(k0)) > 0.1).astype(np.uint8)
    q = dict(flags=flags, u=(k0["x"] + dx + rng.normal(0, 1.5, len(k0))).astype(np.float32),
             v=(k0["y"] + dy + rng.normal(0, 1.5, len(k0))).astype(np.float32),
             level=np.clip(k0["octave"] + rng.integers(-1, 2, len(k0)), 0, 7).astype(np.int32),
             angle=k0["angle"].astype(np.float32), desc=np.ascontiguousarray(d0))
    has = (rng.random(len(k1)) < 0.2).astype(np.uint8)
    cur = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    n, assigned = vo.Matcher(0.8).searchByProjection_keyframe(cur, q, 10.0, 64.0, check_rot, sf, has)
    of = orc.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    oa = np.full(len(k1), -1, np.int32)
    on = orc.lib().orc_match_frame_keyframe(C.byref(of.c), len(k0), q["flags"], q["u"], q["v"], q["level"], q["angle"],
                                            q["desc"], 10.0, 64.0, check_rot, sf, has, oa)
    assert n == on and n > 150
    assert np.array_equal(assigned, oa)
    assert not (assigned[has == 1] >= 0).any()


@pytest.mark.parametrize("idx,mode,check_rot", [(0, 0, 1), (1, 1, 1), (2, 0, 0), (3, 1, 0)])
def test_search_by_bow(vo, orc, idx, mode, check_rot):
    import ctypes as C
    k0, d0, k1, d1, dx, dy = _frame_pair(orc, idx)
    ur0, _ = _uright(k0, idx)
    ur1, _ = _uright(k1, idx + 1)
    rng = np.random.default_rng(idx + 11)
    va, vb = (rng.random(len(k0)) > 0.15).astype(np.uint8), (rng.random(len(k1)) > 0.15).astype(np.uint8)
    na, nb = _nodes(k0), _nodes(k1, dx, dy)
    A = vo.FrameArrays(k0["x"], k0["y"], k0["octave"], k0["angle"], ur0, d0)
    B = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    n, match = vo.Matcher(0.75).searchByBoW(A, va, vo.BowNodes(na), B, vb, vo.BowNodes(nb), bool(mode), bool(check_rot))
    oA = orc.FrameData(k0["x"], k0["y"], k0["octave"], k0["angle"], ur0, d0)
    oB = orc.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    ba, bb = orc.BowData(na), orc.BowData(nb)
    om = np.full(len(k0) if mode else len(k1), -1, np.int32)
    on = orc.lib().orc_match_bow(C.byref(oA.c), va, C.byref(ba.c), C.byref(oB.c), vb, C.byref(bb.c), mode, 0.75,
                                 check_rot, om)
    assert n == on and n > 100
    assert np.array_equal(match, om)


@pytest.mark.parametrize("idx,check_rot,epi_inside", [(0, 1, 0), (1, 0, 1), (2, 1, 1)])
def test_search_for_triangulation(vo, orc, idx, check_rot, epi_inside):
    import ctypes as C
    k0, d0, k1, d1, dx, dy = _frame_pair(orc, idx)
    rng = np.random.default_rng(idx + 13)
    ur0, _ = _uright(k0, idx)
    ur1, _ = _uright(k1, idx + 1)
    ur0[rng.random(len(k0)) < 0.5] = -1.0   # monocular features exercise the epipole gate
    ur1[rng.random(len(k1)) < 0.5] = -1.0
    ha, hb = (rng.random(len(k0)) < 0.3).astype(np.uint8), (rng.random(len(k1)) < 0.3).astype(np.uint8)
    # pure image shift e = (dx, dy, 0): p1^T [e]x p2 = 0 for p2 = p1 + e
    F = np.array([[0, 0, dy], [0, 0, -dx], [-dy, dx, 0]], np.float64) * 1e-3
    ex, ey = (320.0, 240.0) if epi_inside else (1e6, 1e6)
    sf = np.array(list(orc.orb_params().scale)[:8], np.float32)
    na, nb = _nodes(k0), _nodes(k1, dx, dy)
    A = vo.FrameArrays(k0["x"], k0["y"], k0["octave"], k0["angle"], ur0, d0)
    B = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    n, match = vo.Matcher(0.6).searchForTriangulation(A, ha, vo.BowNodes(na), B, hb, vo.BowNodes(nb), F, ex, ey, sf,
                                                      bool(check_rot))
    oA = orc.FrameData(k0["x"], k0["y"], k0["octave"], k0["angle"], ur0, d0)
    oB = orc.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    ba, bb = orc.BowData(na), orc.BowData(nb)
    om = np.full(len(k0), -1, np.int32)
    on = orc.lib().orc_match_triangulation(C.byref(oA.c), ha, C.byref(ba.c), C.byref(oB.c), hb, C.byref(bb.c),
                                           np.ascontiguousarray(F.reshape(-1)), ex, ey, sf, check_rot, om)
    assert n == on and n > 50
    assert np.array_equal(match, om)
    m = match >= 0
    assert not ha[m].any() and not hb[match[m]].any()
    assert len(np.unique(match[m])) == m.sum()


def test_triangulation_and_bow_searches_batched(vo, orc):
    """createNewMapPoints searches the current key-frame against <= 10 neighbours (localMapping.cpp:160-190): ten
    searchForTriangulation calls as ONE launch (a workgroup per neighbour) equal ten sequential oracle calls bit for bit;
    the same for searchByBoW over several candidates (KF-KF and KF-frame)."""
    import ctypes as C
    sf = np.array(list(orc.orb_params().scale)[:8], np.float32)
    k0, d0, _, _, _, _ = _frame_pair(orc, 0)
    rng = np.random.default_rng(77)
    ur0, _ = _uright(k0, 0)
    ur0[rng.random(len(k0)) < 0.5] = -1.0
    ha = (rng.random(len(k0)) < 0.3).astype(np.uint8)
    na = _nodes(k0)
    A = vo.FrameArrays(k0["x"], k0["y"], k0["octave"], k0["angle"], ur0, d0)
    oA, ba = orc.FrameData(k0["x"], k0["y"], k0["octave"], k0["angle"], ur0, d0), orc.BowData(na)
    pairs, want = [], []
    keep = []
    for j in range(10):
        kj, dj, dx, dy = _shifted_frame(orc, 0, j)
        urj, _ = _uright(kj, 20 + j)
        urj[rng.random(len(kj)) < 0.5] = -1.0
        hb = (rng.random(len(kj)) < 0.3).astype(np.uint8)
        F = np.array([[0, 0, dy], [0, 0, -dx], [-dy, dx, 0]], np.float64) * 1e-3
        ex, ey = (320.0, 240.0) if j % 2 else (1e6, 1e6)
        nb = _nodes(kj, dx, dy)
        B, bn = vo.FrameArrays(kj["x"], kj["y"], kj["octave"], kj["angle"], urj, dj), vo.BowNodes(nb)
        keep.append((B, bn))
        pairs.append((B, hb, bn, F, ex, ey))
        oB, bb = orc.FrameData(kj["x"], kj["y"], kj["octave"], kj["angle"], urj, dj), orc.BowData(nb)
        om = np.full(len(k0), -1, np.int32)
        on = orc.lib().orc_match_triangulation(C.byref(oA.c), ha, C.byref(ba.c), C.byref(oB.c), hb, C.byref(bb.c),
                                               np.ascontiguousarray(F.reshape(-1)), ex, ey, sf, 1, om)
        want.append((on, om))
    counts, match = vo.Matcher(0.6).searchForTriangulation_batch(A, ha, vo.BowNodes(na), pairs, sf, True)
    for j in range(10):
        assert counts[j] == want[j][0] and np.array_equal(match[j], want[j][1]), j
    assert counts.sum() > 300 and len({int(c) for c in counts}) > 3
    # searchByBoW, both modes, over the same candidates
    va = (rng.random(len(k0)) < 0.8).astype(np.uint8)
    for k2k in (False, True):
        bp, bw = [], []
        for j in range(6):
            B, bn = keep[j]
            vb = (rng.random(B.view.n) < 0.8).astype(np.uint8)
            bp.append((A, va, vo.BowNodes(na), B, vb, bn))
            n1, m1 = vo.Matcher(0.7).searchByBoW(A, va, vo.BowNodes(na), B, vb, bn, k2k, True)
            bw.append((n1, m1))
        c2, m2 = vo.Matcher(0.7).searchByBoW_batch(bp, k2k, True)
        for j in range(6):
            assert c2[j] == bw[j][0] and np.array_equal(m2[j], bw[j][1]), (k2k, j)
        assert c2.sum() > 200


@pytest.mark.parametrize("idx,threshold", [(0, 3.0), (1, 2.5)])
def test_fuse_matching(vo, orc, idx, threshold):
    import ctypes as C
    k0, d0, k1, d1, dx, dy = _frame_pair(orc, idx)
    ur1, _ = _uright(k1, idx)
    rng = np.random.default_rng(idx + 17)
    sf = np.array(list(orc.orb_params().scale)[:8], np.float32)
    u = (k0["x"] + dx + rng.normal(0, 0.7, len(k0))).astype(np.float32)
    q = dict(flags=(rng.random(len(k0)) > 0.1).astype(np.uint8), u=u,
             v=(k0["y"] + dy + rng.normal(0, 0.7, len(k0))).astype(np.float32),
             ur=(u - rng.uniform(5, 40, len(k0))).astype(np.float32),
             level=np.clip(k0["octave"] + rng.integers(0, 2, len(k0)), 0, 7).astype(np.int32),
             desc=np.ascontiguousarray(d0))
    kf = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    n, best = vo.Matcher(0.8).fuseMapPoints_match(kf, q, threshold, sf)
    of = orc.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    ob = np.full(len(k0), -1, np.int32)
    on = orc.lib().orc_match_fuse(C.byref(of.c), len(k0), q["flags"], q["u"], q["v"], q["ur"], q["level"], q["desc"],
                                  threshold, sf, ob)
    assert n == on and n > 50
    assert np.array_equal(best, ob)


def test_new_matchers_empty_inputs(vo):
    e = np.zeros(0, np.float32)
    empty = vo.FrameArrays(e, e, np.zeros(0, np.int32), e, e, np.zeros((0, 32), np.uint8))
    nodes = vo.BowNodes(np.zeros(0, np.uint32))
    z8 = np.zeros(0, np.uint8)
    m = vo.Matcher(0.8)
    assert m.searchByBoW(empty, z8, nodes, empty, z8, nodes, False)[0] == 0
    assert m.searchByBoW(empty, z8, nodes, empty, z8, nodes, True)[0] == 0
    assert m.searchForTriangulation(empty, z8, nodes, empty, z8, nodes, np.eye(3), 0.0, 0.0, np.ones(8, np.float32))[0] == 0
    q = dict(flags=z8, u=e, v=e, ur=e, level=np.zeros(0, np.int32), angle=e, desc=np.zeros((0, 32), np.uint8))
    assert m.fuseMapPoints_match(empty, q, 3.0, np.ones(8, np.float32))[0] == 0
    assert m.searchByProjection_keyframe(empty, q, 10.0, 64.0, True, np.ones(8, np.float32))[0] == 0


# ------------------------------------------------------------------ M4 / M7 / M10 (loop closure)

def _loop_queries(k_from, d_from, dx, dy, seed, drop=0.1, jitter=1.0):
    rng = np.random.default_rng(seed)
    n = len(k_from)
    return dict(flags=(rng.random(n) > drop).astype(np.uint8),
                u=(k_from["x"] + dx + rng.normal(0, jitter, n)).astype(np.float32),
                v=(k_from["y"] + dy + rng.normal(0, jitter, n)).astype(np.float32),
                level=np.clip(k_from["octave"] + rng.integers(0, 2, n), 0, 7).astype(np.int32),
                desc=np.ascontiguousarray(d_from))


@pytest.mark.parametrize("idx,max_dist,th", [(0, 100, 7.5), (1, 50, 4.0), (2, 50, 3.0)])
def test_area_best(vo, orc, idx, max_dist, th):
    import ctypes as C
    k0, d0, k1, d1, dx, dy = _frame_pair(orc, idx)
    ur1, _ = _uright(k1, idx)
    sf = np.array(list(orc.orb_params().scale)[:8], np.float32)
    q = _loop_queries(k0, d0, dx, dy, idx + 21)
    kf = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    n, best = vo.Matcher(0.8).areaBest(kf, q, th, sf, max_dist)
    of = orc.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    ob = np.full(len(k0), -1, np.int32)
    on = orc.lib().orc_match_area_best(C.byref(of.c), len(k0), q["flags"], q["u"], q["v"], q["level"], q["desc"], th, sf,
                                       max_dist, ob)
    assert n == on and n > 100
    assert np.array_equal(best, ob)


@pytest.mark.parametrize("idx,th", [(0, 10), (1, 3), (3, 5)])
def test_search_by_projection_sim3(vo, orc, idx, th):
    import ctypes as C
    k0, d0, k1, d1, dx, dy = _frame_pair(orc, idx)
    ur1, _ = _uright(k1, idx)
    sf = np.array(list(orc.orb_params().scale)[:8], np.float32)
    q = _loop_queries(k0, d0, dx, dy, idx + 23)
    occ = (np.random.default_rng(idx + 29).random(len(k1)) < 0.15).astype(np.uint8)   # exercises the :422 quirk
    kf = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    n, assigned = vo.Matcher(0.8).searchByProjection_sim3(kf, q, th, sf, occ)
    of = orc.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    oa = np.full(len(k1), -1, np.int32)
    on = orc.lib().orc_match_sim3_projection(C.byref(of.c), len(k0), q["flags"], q["u"], q["v"], q["level"], q["desc"],
                                             th, sf, occ, oa)
    assert n == on and n > 100
    assert np.array_equal(assigned, oa)


@pytest.mark.parametrize("idx", [0, 2])
def test_search_by_sim3(vo, orc, idx):
    import ctypes as C
    k0, d0, k1, d1, dx, dy = _frame_pair(orc, idx)
    ur0, _ = _uright(k0, idx)
    ur1, _ = _uright(k1, idx + 1)
    sf = np.array(list(orc.orb_params().scale)[:8], np.float32)
    q1 = _loop_queries(k0, d0, dx, dy, idx + 31)       # points of key-frame 1 seen from key-frame 2
    q2 = _loop_queries(k1, d1, -dx, -dy, idx + 37)     # and the reverse
    A = vo.FrameArrays(k0["x"], k0["y"], k0["octave"], k0["angle"], ur0, d0)
    B = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    n, m12 = vo.Matcher(0.8).searchBySim3(A, B, q1, q2, 7.5, sf, sf)
    oA = orc.FrameData(k0["x"], k0["y"], k0["octave"], k0["angle"], ur0, d0)
    oB = orc.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    om = np.full(len(k0), -1, np.int32)
    on = orc.lib().orc_match_sim3_mutual(C.byref(oA.c), C.byref(oB.c), q1["flags"], q1["u"], q1["v"], q1["level"], q1["desc"],
                                         q2["flags"], q2["u"], q2["v"], q2["level"], q2["desc"], 7.5, sf, sf, om)
    assert n == on and n > 100
    assert np.array_equal(m12, om)
    sel = np.nonzero(m12 >= 0)[0]
    assert len(np.unique(m12[sel])) == len(sel)      # mutual agreement makes the map injective


# ------------------------------------------------------------------ N2: BoW transform

@pytest.mark.parametrize("seed,k,L,levelsup", [(0, 10, 4, 3), (1, 8, 3, 1), (2, 10, 3, 5), (3, 4, 5, 2)])
def test_bow_transform(vo, orc, seed, k, L, levelsup):
    from vo_slam_test_amd import synth
    voc = synth.make_vocabulary(seed, k=k, L=L)
    # features: real ORB descriptors plus exact copies of some node descriptors (distance-0 ties)
    _, d0, _, d1, _, _ = _frame_pair(orc, seed)
    desc = np.ascontiguousarray(np.concatenate([d0[:700], voc["node_desc"][1:200]]))
    V = vo.Vocabulary(voc["L"], voc["child_start"], voc["children"], voc["node_desc"], voc["node_weight"], voc["word_id"])
    word, weight, node = V.transform(desc, levelsup)
    V.close()
    n = len(desc)
    ow, owt, on = np.zeros(n, np.int32), np.zeros(n, np.float64), np.zeros(n, np.int32)
    orc.lib().orc_bow_transform(voc["L"], voc["child_start"], voc["children"], np.ascontiguousarray(voc["node_desc"]),
                                voc["node_weight"], voc["word_id"], n, desc, levelsup, ow, owt, on)
    assert np.array_equal(word, ow) and np.array_equal(weight, owt) and np.array_equal(node, on)
    assert (word >= 0).all() and (weight > 0).all()
    if levelsup >= L:
        assert (node == 0).all()            # nid_level <= 0: the root
    else:
        lvl_first = sum(k ** i for i in range(L - levelsup))      # breadth-first numbering
        assert ((node >= lvl_first) & (node < lvl_first + k ** (L - levelsup))).all()


def test_bow_feature_vector_feeds_search_by_bow(vo, orc):
    """end to end: the node ids of the transform are the FeatureVector keys searchByBoW walks"""
    import ctypes as C
    from vo_slam_test_amd import synth
    voc = synth.make_vocabulary(5, k=6, L=3)
    k0, d0, k1, d1, dx, dy = _frame_pair(orc, 1)
    V = vo.Vocabulary(voc["L"], voc["child_start"], voc["children"], voc["node_desc"], voc["node_weight"], voc["word_id"])
    _, _, na = V.transform(d0, 2)
    _, _, nb = V.transform(d1, 2)
    V.close()
    ur0, _ = _uright(k0, 1)
    ur1, _ = _uright(k1, 2)
    A = vo.FrameArrays(k0["x"], k0["y"], k0["octave"], k0["angle"], ur0, d0)
    B = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    ones0, ones1 = np.ones(len(k0), np.uint8), np.ones(len(k1), np.uint8)
    n, match = vo.Matcher(0.75).searchByBoW(A, ones0, vo.BowNodes(na), B, ones1, vo.BowNodes(nb), True, True)
    oA = orc.FrameData(k0["x"], k0["y"], k0["octave"], k0["angle"], ur0, d0)
    oB = orc.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    ba, bb = orc.BowData(na), orc.BowData(nb)
    om = np.full(len(k0), -1, np.int32)
    on = orc.lib().orc_match_bow(C.byref(oA.c), ones0, C.byref(ba.c), C.byref(oB.c), ones1, C.byref(bb.c), 1, 0.75, 1, om)
    assert n == on and np.array_equal(match, om) and n > 50


@pytest.mark.parametrize("mode", [0, 1])
def test_search_by_bow_under_heavy_conflicts(vo, orc, mode):
    """Frame A holds every feature three times (two of the copies one descriptor bit off) in coarse vocabulary nodes of
    ~50 features: most lanes of a 64-query step want a feature an earlier lane takes, so the device replay runs many
    conflict rounds per step -- and must still give the node-by-node sequential result."""
    import ctypes as C
    k0, d0, k1, d1, dx, dy = _frame_pair(orc, 3)
    rng = np.random.default_rng(23)
    rep = np.concatenate([np.arange(len(k0))] * 3)
    rng.shuffle(rep)
    kA, dA = k0[rep], d0[rep].copy()
    off = rng.random(len(rep)) < 0.66
    dA[off, rng.integers(0, 32, off.sum())] ^= (1 << rng.integers(0, 8, off.sum())).astype(np.uint8)
    urA, _ = _uright(kA, 1)
    ur1, _ = _uright(k1, 2)
    va, vb = np.ones(len(kA), np.uint8), (rng.random(len(k1)) > 0.1).astype(np.uint8)
    coarse = lambda k, sx=0.0, sy=0.0: (np.floor((k["x"] - sx) / 160.0).astype(np.int64) + 8 * np.floor((k["y"] - sy) / 160.0).astype(np.int64) + 7).astype(np.uint32)
    na, nb = coarse(kA), coarse(k1, dx, dy)
    A = vo.FrameArrays(kA["x"], kA["y"], kA["octave"], kA["angle"], urA, dA)
    B = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    n, match = vo.Matcher(0.9).searchByBoW(A, va, vo.BowNodes(na), B, vb, vo.BowNodes(nb), bool(mode), True)
    oA = orc.FrameData(kA["x"], kA["y"], kA["octave"], kA["angle"], urA, dA)
    oB = orc.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    ba, bb = orc.BowData(na), orc.BowData(nb)
    om = np.full(len(kA) if mode else len(k1), -1, np.int32)
    on = orc.lib().orc_match_bow(C.byref(oA.c), va, C.byref(ba.c), C.byref(oB.c), vb, C.byref(bb.c), mode, 0.9, 1, om)
    assert n == on and n > 50
    assert np.array_equal(match, om)
