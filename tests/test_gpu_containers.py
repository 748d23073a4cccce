"""GPU: the host-side mirror of the reference's plugin surface.

Mirrors custom_invlist_cpp/test_compressed_ivfs.py (per-list set equality + search equality for the five
containers, deferred decode == direct search, returned codes, 1-by-1 decode) and alt-graph-index/test_altid.py
(identical neighbour sets after swapping in each compressed graph), on the in-repo IVF harness since Faiss is
not installed.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dataset(d, nt, nb, nq, seed=0):
    rng = np.random.default_rng(seed)
    c = rng.normal(size=(16, d)).astype(np.float32) * 3
    def draw(n):
        return (c[rng.integers(0, 16, n)] + rng.normal(size=(n, d))).astype(np.float32)
    return draw(nt), draw(nb), draw(nq)


def _make_wt1(il):
    from vector_db_id_compression_amd import custom_invlists

    return custom_invlists.CompressedIDInvertedListsWaveletTree(il, 1)


def _classes():
    from vector_db_id_compression_amd import custom_invlists as ci

    return [ci.CompressedIDInvertedListsPackedBits, ci.CompressedIDInvertedListsFenwickTree,
            ci.CompressedIDInvertedListsEliasFano, ci.CompressedIDInvertedListsWaveletTree, _make_wt1]


@pytest.mark.parametrize("which", range(5))
def test_compressed_ivf_lists_and_search(which):
    """test_compressed_ivfs.py:43-91 on IVF8,Flat with nb = 100."""
    from vector_db_id_compression_amd.ivf import IVFIndex

    CompressedIVF = _classes()[which]
    xt, xb, xq = _dataset(4, 1000, 100, 1)
    index = IVFIndex(4, 8, "Flat")
    index.train(xt)
    index.add(xb)
    ref_lists = [index.invlists.get_ids(c).copy() for c in range(8)]
    _, Iref = index.search(xq, 5)
    comp = CompressedIVF(index.invlists)
    index.replace_invlists(comp, False)
    for c in range(8):
        n = comp.list_size(c)
        assert n == ref_lists[c].size
        ids = comp.get_ids(c)
        if n == 0:
            assert ids is None
            continue
        assert np.all(np.sort(ids) == ref_lists[c])  # :74-79
    _, Icomp = index.search(xq, 5)
    np.testing.assert_array_equal(Iref, Icomp)  # :84-86
    assert comp.compressed_ids_size_in_bytes > 0


def test_size_attributes_match_reference_formulas(oracle):
    from vector_db_id_compression_amd import custom_invlists as ci
    from vector_db_id_compression_amd.invlists import ArrayInvertedLists

    rng = np.random.default_rng(1)
    assign = rng.integers(0, 32, 5000)
    il = ArrayInvertedLists.from_assignment(assign, 32, code_size=4)
    roc = ci.CompressedIDInvertedListsFenwickTree(il)
    ef = ci.CompressedIDInvertedListsEliasFano(il)
    pk = ci.CompressedIDInvertedListsPackedBits(il)
    tot_roc, tot_ef_bits = 0, 0
    for l in range(32):
        ids = il.get_ids(l).astype(np.uint64)
        e = oracle.roc_encode(ids, oracle.list_precision(ids))
        tot_roc += 8 + 4 * e["words"].size
        f = oracle.ef_build(np.sort(ids))
        tot_ef_bits += f["low_nbits"] + f["high_nbits"]
        assert int(roc.id_symbol_precision[l]) == oracle.list_precision(ids)
    assert roc.compressed_ids_size_in_bytes == tot_roc      # custom_invlists_impl.cpp:196-206
    assert ef.compressed_ids_size_in_bytes == tot_ef_bits // 8  # :272-282
    assert pk.bits == 13 and pk.compressed_ids_size_in_bytes == sum((il.list_size(l) * 13 + 7) // 8 for l in range(32))
    assert roc.overhead_in_bytes == 0 and ef.overhead_in_bytes == 0
    assert roc.codes_size_in_bytes == 32 * 5000 * 4  # the reference's accidental nlist x total accounting (:203-205)


def test_deferred_decoding_matches_direct_search_and_returns_codes():
    """test_compressed_ivfs.py:95-126 on IVF32,PQ4 with nb = 10000, nprobe = 4."""
    from vector_db_id_compression_amd.ivf import IVFIndex

    xt, xb, xq = _dataset(32, 10000, 10000, 10)
    index = IVFIndex(32, 32, ("PQ", 4))
    index.train(xt)
    index.add(xb)
    index.nprobe = 4
    Dref, Iref = index.search(xq, 10)
    with pytest.raises(RuntimeError):  # FAISS_THROW_IF_NOT_MSG(index.parallel_mode == 3, ...), :420-422
        index.search_defer_id_decoding(xq, 10)
    index.parallel_mode = 3
    D, I = index.search_defer_id_decoding(xq, 10)
    np.testing.assert_array_equal(I, Iref)
    np.testing.assert_array_equal(D, Dref)
    D, I, codes = index.search_defer_id_decoding(xq, 10, return_codes=2)
    assert codes.shape == (10, 10, 5)
    for q in range(10):
        for ki in range(10):
            if I[q, ki] < 0:
                continue
            list_no = int(codes[q, ki, 0])
            il_ids, il_codes = index.invlists.get_ids(list_no), index.invlists.get_codes(list_no)
            offset = np.where(il_ids == I[q, ki])[0][0]
            assert np.all(codes[q, ki, 1:] == il_codes[offset])


@pytest.mark.parametrize("which", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("one_by_one", [True, False])
def test_deferred_decoding_with_compressed_lists(which, one_by_one):
    """test_compressed_ivfs.py:128-156 (1-by-1) plus the per-touched-list path for every container."""
    from vector_db_id_compression_amd.ivf import IVFIndex

    xt, xb, xq = _dataset(32, 10000, 10000, 10)
    index = IVFIndex(32, 32, ("PQ", 4))
    index.train(xt)
    index.add(xb)
    index.nprobe = 4
    Dref, Iref = index.search(xq, 10)
    index.replace_invlists(_classes()[which](index.invlists), False)
    index.parallel_mode = 3
    D, I = index.search_defer_id_decoding(xq, 10, decode_1by1=one_by_one)
    np.testing.assert_array_equal(I, Iref)
    np.testing.assert_array_equal(D, Dref)


def test_wavelet_tree_select_and_asserts(oracle):
    from vector_db_id_compression_amd import VidcError
    from vector_db_id_compression_amd.codecs import WaveletTreeLists

    rng = np.random.default_rng(3)
    for nlist, ntotal in [(1, 10), (2, 100), (5, 1000), (37, 5000), (256, 20000), (1000, 3000), (100, 150000)]:
        assign = rng.integers(0, nlist, ntotal)
        order = np.argsort(assign, kind="stable")
        counts = np.bincount(assign, minlength=nlist)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        ids = order.astype(np.uint64)
        wt = WaveletTreeLists.build(off, ids)
        assert wt.levels == max(1, int(nlist - 1).bit_length())
        assert np.array_equal(wt.decode_all().cpu().numpy().view(np.uint64), ids)
        ql = rng.integers(0, nlist, 200)
        ql = ql[counts[ql] > 0]
        qo = (rng.random(ql.size) * counts[ql]).astype(np.int64)
        got = wt.select(ql, qo)
        list_nos = assign.astype(np.uint32)
        for l, o, g in list(zip(ql, qo, got))[:40]:
            assert g == oracle.wt_select(list_nos, int(l), int(o))  # wt.select(offset + 1, list_no), :377-379
        assert np.array_equal(got, ids[off[ql].astype(np.int64) + qo].astype(np.int64))
        # wt_type 1: the levels are RRR-63 coded (no plain bits kept): same answers from every entry point, and the
        # size is the size of that coding -- recomputed here from the level bit vectors
        wt1 = WaveletTreeLists.build(off, ids, wt_type=1)
        assert np.array_equal(wt1.select(ql, qo), got)
        assert np.array_equal(wt1.decode_all().cpu().numpy().view(np.uint64), ids)
        sel = np.unique(ql)[:17].astype(np.uint64)
        d1, o1 = wt1.decode_lists(sel)
        d0, o0 = wt.decode_lists(sel)
        assert np.array_equal(o1, o0) and np.array_equal(d1.cpu().numpy(), d0.cpu().numpy())
        assert wt1.size_in_bytes == _rrr_wt_size(list_nos, nlist)
    with pytest.raises(VidcError):  # assert(ids_data[i] > prev_id) / < ntotal, :359-360
        WaveletTreeLists.build(np.array([0, 3], dtype=np.uint64), np.array([2, 1, 0], dtype=np.uint64))
    with pytest.raises(VidcError):
        WaveletTreeLists.build(np.array([0, 3], dtype=np.uint64), np.array([0, 1, 7], dtype=np.uint64))


@pytest.mark.parametrize("nlist,ntotal", [(700, 300_001), (65536, 1 << 20), (3, 262_144)])
def test_wavelet_tree_built_through_the_partitioned_scatter(nlist, ntotal, monkeypatch):
    """From 2^18 ids on list_nos[id] is built in two streaming passes (partition by id >> 14, place per bucket: wt.hip) instead of
    one random scatter: same tree as the direct scatter (VIDC_WT_SCATTER=1) down to the last word, and the same refusals --
    an id twice (so another one missing), an id >= ntotal, a list that does not ascend (custom_invlists_impl.cpp:359-360)."""
    from vector_db_id_compression_amd import VidcError
    from vector_db_id_compression_amd.codecs import WaveletTreeLists

    rng = np.random.default_rng(nlist)
    assign = rng.integers(0, nlist, ntotal)
    order = np.argsort(assign, kind="stable")
    counts = np.bincount(assign, minlength=nlist)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    ids = order.astype(np.uint64)
    wt = WaveletTreeLists.build(off, ids)
    assert np.array_equal(wt.decode_all().cpu().numpy().view(np.uint64), ids)
    ql = rng.integers(0, nlist, 500)
    ql = ql[counts[ql] > 0]
    qo = (rng.random(ql.size) * counts[ql]).astype(np.int64)
    got = wt.select(ql, qo)
    assert np.array_equal(got, ids[off[ql].astype(np.int64) + qo].astype(np.int64))
    monkeypatch.setenv("VIDC_WT_SCATTER", "1")
    ref = WaveletTreeLists.build(off, ids)
    monkeypatch.delenv("VIDC_WT_SCATTER")
    assert ref.size_in_bytes == wt.size_in_bytes and np.array_equal(ref.select(ql, qo), got)
    assert np.array_equal(ref.decode_all().cpu().numpy(), wt.decode_all().cpu().numpy())
    big = int(np.argmax(counts))
    a, b = int(off[big]), int(off[big + 1])
    assert b - a >= 3
    for what in ("twice", "too large", "descending"):
        bad = ids.copy()
        if what == "twice":
            bad[a + 1] = bad[a]  # (also breaks the order inside the list; the missing id is what the placement pass sees)
            bad[b - 1] = ids[a + 1] if ids[a + 1] > bad[b - 2] else bad[b - 1]
        elif what == "too large":
            bad[b - 1] = ntotal + 5
        else:
            bad[a], bad[a + 1] = ids[a + 1], ids[a]
        with pytest.raises(VidcError):
            WaveletTreeLists.build(off, bad)


def _rrr_wt_size(list_nos, nlist):
    """Bytes of a levelwise wavelet tree whose levels are RRR coded with 63-bit blocks and one sample per 32 blocks:
    6-bit class + ceil(log2 C(63, class)) offset bits per block, (32-bit stream pointer + 32-bit rank) per sample
    (+ the final one), plus the table of symbol start positions (csrc/wt.hip)."""
    import math

    nt = list_nos.size
    L = max(1, int(nlist - 1).bit_length())
    ow = [0 if c in (0, 63) else math.ceil(math.log2(math.comb(63, c))) for c in range(64)]
    nblk = (nt + 62) // 63
    nsamp = (nblk + 31) // 32
    order = np.arange(nt)
    off_bits = 0
    syms = list_nos.astype(np.int64)
    for level in range(L):
        bits = (syms[order] >> (L - 1 - level)) & 1
        padded = np.zeros(nblk * 63, dtype=np.int64)
        padded[:nt] = bits
        cls = padded.reshape(nblk, 63).sum(1)
        off_bits += int(sum(ow[int(c)] for c in cls))
        # next level: stable sort by the top (level + 1) bits of the symbol
        order = order[np.argsort(syms[order] >> (L - 1 - level), kind="stable")]
    return (off_bits + 7) // 8 + L * ((6 * nblk + 7) // 8) + L * (nsamp + 1) * 8 + (nlist + 1) * 8


def _random_graph(rng, N, K):
    rows = np.full((N, K), -1, dtype=np.int32)
    for i in range(N):
        d = int(rng.integers(1, K + 1))
        rows[i, :d] = rng.choice(N, size=d, replace=False)
    return rows


def test_compressed_graphs_return_the_same_neighbours(oracle):
    """test_altid.py:17-44: every compressed graph answers get_neighbors with the original edge set."""
    from vector_db_id_compression_amd import altid

    rng = np.random.default_rng(5)
    N, K = 1000, 32
    rows = _random_graph(rng, N, K)
    rows[17, :] = -1  # a node without edges
    rows[18, :] = rng.choice(N, size=K, replace=False)  # a full row (no terminator)
    graphs = {name: cls(rows.copy()) for name, cls in altid.AVAILABLE_COMPRESSED_GRAPHS.items() if cls}
    nodes = np.arange(N)
    for name, g in graphs.items():
        out, cnt = g.get_neighbors_batch(nodes)
        for i in range(N):
            d = int((rows[i] >= 0).sum())
            assert cnt[i] == d, (name, i)
            got = out[i, :d]
            want = rows[i, :d].astype(np.uint64)
            if name == "roc" and d:
                # bit-compatible with the reference INCLUDING its precision quirk: a row whose largest id is a
                # power of two decodes lossily -- expect what the reference codec would return
                P = oracle.list_precision(want)
                e = oracle.roc_encode(want, P)
                want = oracle.roc_decode(e["head"], e["words"], d, P, e["mt_draws"])[0]
                assert got.astype(np.uint64).tolist() == want.tolist(), (name, i)
            assert sorted(got.tolist()) == sorted(want.astype(np.int64).tolist()), (name, i)
            assert np.all(out[i, d:] == -1)
            if name == "compact":
                assert got.tolist() == rows[i, :d].tolist()  # order preserved (:28-37)
            if name == "elias-fano":
                assert got.tolist() == sorted(rows[i, :d].tolist())  # std::sort of the row (:76)
        assert g.get_neighbors(3).tolist() == out[3, : int(cnt[3])].tolist()
    comp = graphs["compact"]
    assert comp.bits == oracle.packed_bits_for(N) and comp.stride == (K * comp.bits + 7) // 8  # :22-24
    # byte image of a row: neighbours then the sentinel N (:28-37)
    d = int((rows[3] >= 0).sum())
    vals = np.concatenate([rows[3, :d], [N] if d < K else []]).astype(np.uint64)
    want = oracle.packed_encode(vals, comp.bits)
    got = comp._c.export_row(3)
    assert np.array_equal(got[: want.size], want) and not got[want.size:].any()
    # sizes: EF = sum of per-node low+high bits / 8 (:86,88); ROC = sum of ANSState::size() over ALL nodes (:148)
    ef_bits, roc_bytes = 0, 0
    for i in range(N):
        d = int((rows[i] >= 0).sum())
        if d:
            li = rows[i, :d].astype(np.uint64)
            f = oracle.ef_build(np.sort(li))
            ef_bits += f["low_nbits"] + f["high_nbits"]
            roc_bytes += 4 * oracle.roc_encode(li, oracle.list_precision(li))["words"].size
        roc_bytes += 8
    assert graphs["elias-fano"].compressed_ids_size_in_bytes == ef_bits // 8
    assert graphs["roc"].compressed_ids_size_in_bytes == roc_bytes
    assert np.array_equal(graphs["roc"].num_outgoing_edges, (rows >= 0).sum(1))


def test_graph_search_identical_after_swapping_compressed_graphs():
    """test_altid.py:17-44: search results (I and D) are identical with each compressed graph swapped in."""
    from vector_db_id_compression_amd import altid
    from vector_db_id_compression_amd.graph_search import RawGraph, knn_graph, search

    rng = np.random.default_rng(11)
    x = rng.normal(size=(1000, 16)).astype(np.float32)
    xq = rng.normal(size=(8, 16)).astype(np.float32)
    rows = knn_graph(x, 32, seed=1)
    # avoid the reference's power-of-two precision quirk in this equality test (it is covered elsewhere):
    # make sure no row's largest id is a power of two
    for i in range(rows.shape[0]):
        d = int((rows[i] >= 0).sum())
        m = int(rows[i, :d].max())
        if m & (m - 1) == 0:
            rows[i, int(np.argmax(rows[i, :d]))] = m - 1 if (m - 1) not in rows[i, :d] and m - 1 != i else rows[i, 0]
    Dref, Iref = search(RawGraph(rows), x, xq, k=10)
    for name, cls in altid.AVAILABLE_COMPRESSED_GRAPHS.items():
        if cls is None:
            continue
        g = cls(rows.copy())
        D, I = search(g, x, xq, k=10)
        np.testing.assert_array_equal(I, Iref, err_msg=name)
        np.testing.assert_array_equal(D, Dref, err_msg=name)


def test_sharded_lists_single_rank_gpu():
    """sharding.ShardedInvLists with the real ROC codec (world size 1; the 2-rank path runs under gloo on CPU)."""
    from vector_db_id_compression_amd import synth
    from vector_db_id_compression_amd.codecs import RocLists
    from vector_db_id_compression_amd.sharding import ShardedInvLists

    off, ids = synth.make_lists_numpy(20000, 64, 0.75, seed=3)
    sh = ShardedInvLists(off, ids, 0, 1, lambda o, i: RocLists.encode(o, i))
    req = np.array([5, 0, 63, 5], dtype=np.int64)
    out, roff = sh.gather_ids(req, dst=0)
    out = out.cpu().numpy()
    full = sh.codec.decode_all().cpu().numpy()
    for i, l in enumerate(req):
        assert np.array_equal(out[int(roff[i]):int(roff[i + 1])], full[int(off[l]):int(off[l + 1])])


@pytest.mark.parametrize("N,K", [(4096, 64), (8193, 32), (5000, 50)])
def test_graph_codecs_device_side_offsets(N, K, monkeypatch):
    """Edge counts -> CSR offsets are scanned on the device for the ROC and Elias-Fano graph objects; rows decoded by the
    lane-per-row kernels (forced) and by the wave-per-row kernels must both give back the neighbour SETS."""
    from vector_db_id_compression_amd import synth
    from vector_db_id_compression_amd.codecs import EfLists, RocLists

    rows = synth.make_graph_rows(N, K, seed=N, dmin=1)
    deg = (rows >= 0).sum(1)
    nodes = np.arange(N, dtype=np.uint64)
    want = np.sort(np.where(rows >= 0, rows, np.iinfo(np.int32).max), axis=1)
    for force in ("1", "0"):
        monkeypatch.setenv("VIDC_FORCE_LANE", force)
        monkeypatch.setenv("VIDC_NO_LANE", "0" if force == "1" else "1")
        for cls in (RocLists, EfLists):
            g = cls.encode_rows(rows)
            dec, cnt = g.decode_rows(nodes, K)
            assert np.array_equal(cnt, deg)
            got = dec.cpu().numpy()
            assert ((got >= 0).sum(1) == deg).all()
            if cls is EfLists or rows.max() & (rows.max() - 1):  # ROC is lossy for a pow-2 max id
                assert np.array_equal(np.sort(np.where(got >= 0, got, np.iinfo(np.int32).max), axis=1), want)
            assert np.array_equal(np.diff(g.offsets.astype(np.int64)), deg)
            sub, c2 = g.decode_rows(np.array([N - 1, 0, N // 2], dtype=np.uint64), K, want_counts=False)
            assert c2 is None and np.array_equal(sub.cpu().numpy(), got[[N - 1, 0, N // 2]])
            every, c3 = g.decode_rows(None, K)  # nodes == NULL: all rows in order, no index array
            assert np.array_equal(every.cpu().numpy(), got) and np.array_equal(c3, deg)


@pytest.mark.parametrize("K", [1, 3, 32, 33, 64])
def test_graph_rows_edge_shapes(K, monkeypatch, oracle):
    """Empty rows, full rows, odd row widths, duplicate-free tiny universes: ROC / Elias-Fano / compact-bit graph objects
    through the lane-per-row kernels (forced) give back every neighbour set."""
    from vector_db_id_compression_amd.codecs import CompactRows, EfLists, RocLists

    monkeypatch.setenv("VIDC_FORCE_LANE", "1")
    rng = np.random.default_rng(K)
    N = 700
    rows = np.full((N, K), -1, dtype=np.int32)
    for i in range(N):
        d = int(rng.integers(0, K + 1))
        if i % 97 == 0:
            d = 0  # empty row
        if i % 89 == 0:
            d = K  # full row (no -1 terminator)
        rows[i, :d] = rng.choice(N, size=d, replace=False)
    deg = (rows >= 0).sum(1)
    big = np.iinfo(np.int32).max
    want = np.sort(np.where(rows >= 0, rows, big), axis=1)
    for cls in (EfLists, RocLists, CompactRows):
        g = cls.encode_rows(rows)
        res = g.decode_rows(None, K)
        got, cnt = res[0].cpu().numpy(), np.asarray(res[1])
        assert np.array_equal(cnt, deg), cls.__name__
        assert ((got >= 0).sum(1) == deg).all(), cls.__name__
        if cls is not RocLists:  # (ROC: rows whose max id is a power of two are lossy in the reference)
            assert np.array_equal(np.sort(np.where(got >= 0, got, big), axis=1), want), cls.__name__
        else:
            # bit-compatible with the reference including its precision quirk: expectation = what the reference
            # decoder makes of the reference stream of that row (sampling order)
            for i in range(N):
                d = int(deg[i])
                ids = np.sort(rows[i, :d]).astype(np.uint64)
                if d == 0:
                    continue
                P = oracle.list_precision(ids)
                e = oracle.roc_encode(ids, P)
                ref = oracle.roc_decode(e["head"], e["words"], d, P, e["mt_draws"])[0]
                assert np.array_equal(got[i, :d].astype(np.uint64), ref), (cls.__name__, i)


@pytest.mark.parametrize("K", [32, 64])
def test_graph_tile_kernels_off_their_fast_paths(K, oracle):
    """The 64-row tile kernels (rows_tile.h) away from the shape the bench exercises: a row array that is only 4-byte aligned
    (no 16-byte block loads), a node count that is not a multiple of 64, neighbour ids far beyond the node count (the Elias-Fano
    arena is sized for ids < N first and must size itself again from the largest id it met), requests by node list, and the
    Elias-Fano words of sampled rows against the oracle's streams."""
    import torch
    from vector_db_id_compression_amd.codecs import CompactRows, EfLists, RocLists

    rng = np.random.default_rng(100 + K)
    N = 70000 + 37  # (>= 65 536: the whole-graph ROC decode takes the rows ordered by edge count)
    deg = rng.integers(0, K + 1, size=N)
    deg[::1000] = K
    deg[1::1000] = 0
    deg[1280:1408] = K  # two whole tiles of full rows with 30-bit ids: more stream words than the compaction requests in one pass
    flat = torch.full((N * K + 1,), -1, dtype=torch.int32)
    rows_np = np.full((N, K), -1, dtype=np.int32)
    big_ids = rng.integers(0, 1 << 30, size=(N, K)).astype(np.int64)
    for i in range(0, N, 1):
        d = int(deg[i])
        if d:
            # distinct ids; every 7th row from a universe of 2^30, the others below N
            src = np.unique(big_ids[i]) if (i % 7 == 0 or 1280 <= i < 1408) else np.unique(rng.integers(0, N, size=2 * K))
            if src.size < d:
                d = deg[i] = src.size
            rows_np[i, :d] = rng.permutation(src)[:d].astype(np.int32)
    flat[1:] = torch.from_numpy(rows_np.reshape(-1))
    rows = flat.cuda()[1:].view(N, K)  # 4 bytes off a 16-byte boundary
    assert rows.data_ptr() % 16 != 0
    big = np.iinfo(np.int32).max
    want = np.sort(np.where(rows_np >= 0, rows_np, big), axis=1)
    nodes = rng.integers(0, N, size=3000).astype(np.uint64)
    for cls in (EfLists, RocLists):  # (compact bits stores ids < N only: further down)
        g = cls.encode_rows(rows)
        every, cnt = g.decode_rows(None, K)
        got = every.cpu().numpy()
        assert np.array_equal(cnt, deg), cls.__name__
        if cls is EfLists:
            assert np.array_equal(np.sort(np.where(got >= 0, got, big), axis=1), want)
            assert np.array_equal(got[:, :1][deg > 0, 0], want[deg > 0, 0])  # Elias-Fano decodes ascending
        else:
            ok = (rows_np.max(axis=1) & (rows_np.max(axis=1) - 1)) != 0  # (ROC is lossy for a power-of-two maximum)
            assert np.array_equal(np.sort(np.where(got >= 0, got, big), axis=1)[ok], want[ok])
        sub, c2 = g.decode_rows(nodes, K)
        assert np.array_equal(sub.cpu().numpy(), got[nodes.astype(np.int64)]) and np.array_equal(c2, deg[nodes.astype(np.int64)])
        if cls is EfLists:
            for i in (0, 7, 14, 1000, 1001, N - 1, N - 2):
                d = int(deg[i])
                if not d:
                    continue
                e = oracle.ef_build(np.sort(rows_np[i, :d]).astype(np.uint64))
                low, high, lb, hb = g.export(i)
                assert (lb, hb) == (e["low_nbits"], e["high_nbits"]), i
                assert np.array_equal(low, e["low"]) and np.array_equal(high, e["high"]), i
    small = np.where(rows_np >= N, rows_np % N, rows_np)  # compact bits: ids below N (duplicates inside a row are fine for it)
    flat[1:] = torch.from_numpy(small.reshape(-1))
    rows_c = flat.cuda()[1:].view(N, K)
    c = CompactRows.encode_rows(rows_c)
    dec, cnt = c.decode_rows(None, K)
    assert np.array_equal(dec.cpu().numpy(), small) and np.array_equal(cnt, deg)
    sub, c2 = c.decode_rows(nodes, K)
    assert np.array_equal(sub.cpu().numpy(), small[nodes.astype(np.int64)])


@pytest.mark.parametrize("N", [300, 70000])
def test_graph_encoders_refuse_a_negative_id_that_is_not_the_terminator(N):
    """A row ends at its first -1 (altid_impl.cpp:61-68, 110-117); any other negative entry in front of it is not an id.  The three
    graph encoders report the row instead of encoding garbage -- through the wave-per-row kernels (N = 300) and the 64-row tile /
    lane kernels with their device-side totals (N = 70 000); entries BEHIND the terminator are never looked at."""
    from vector_db_id_compression_amd._lib import VidcError
    from vector_db_id_compression_amd.codecs import CompactRows, EfLists, RocLists

    rng = np.random.default_rng(N)
    K = 32
    rows = np.full((N, K), -1, dtype=np.int32)
    for i in range(N):
        d = int(rng.integers(1, K + 1))
        rows[i, :d] = rng.choice(N, size=d, replace=False)
    ok = rows.copy()
    ok[N // 2, K - 1] = -1
    ok[N // 2, K - 2] = -1
    ok[N // 2, K - 1] = -9  # behind a terminator: ignored
    bad = rows.copy()
    victim = N - 3
    bad[victim, 0] = -9
    for cls in (RocLists, EfLists, CompactRows):
        g = cls.encode_rows(ok)
        dec = g.decode_rows(None, K)[0].cpu().numpy()
        assert ((dec >= 0).sum(1) == ((ok >= 0).cumprod(1)).sum(1)).all(), cls.__name__
        with pytest.raises(VidcError) as ei:
            cls.encode_rows(bad)
        assert "status -4" in str(ei.value), (cls.__name__, str(ei.value))  # VIDC_ERR_DOMAIN
        if cls is RocLists:
            assert f"list {victim} " in str(ei.value)  # (the first offending row, from the device-side summary)


def test_empty_graph_and_list_objects():
    """Zero nodes / zero lists / all-empty rows through every codec (scan kernels with n = 0, zero-sized streams)."""
    from vector_db_id_compression_amd.codecs import CompactRows, EfLists, PackedLists, RocLists

    for K in (4, 64):
        rows0 = np.zeros((0, K), dtype=np.int32)
        rows_e = np.full((5, K), -1, dtype=np.int32)
        for cls in (RocLists, EfLists, CompactRows):
            g = cls.encode_rows(rows_e)  # five nodes without edges
            got, cnt = g.decode_rows(None, K)
            assert np.array_equal(cnt, np.zeros(5, np.uint32)) and (got.cpu().numpy() == -1).all(), cls.__name__
            got, cnt = g.decode_rows(np.array([4, 0], dtype=np.uint64), K)
            assert (got.cpu().numpy() == -1).all() and got.shape == (2, K)
            if cls is not CompactRows:
                assert g.compressed_bytes == 0  # empty lists have no bitstream object (:199-201, :239-241)
            if cls is RocLists or cls is EfLists:
                g0 = cls.encode_rows(rows0)
                got0, cnt0 = g0.decode_rows(None, K)
                assert got0.shape[0] == 0 and cnt0.size == 0
    off0 = np.array([0], dtype=np.uint64)
    for cls in (EfLists, PackedLists):
        o = cls.encode(off0, np.zeros(0, np.uint64))
        assert o.compressed_bytes == 0 and o.decode_all().numel() == 0
        o = cls.encode(np.array([0, 0, 0, 0], dtype=np.uint64), np.zeros(0, np.uint64))
        assert o.compressed_bytes == 0 and o.decode_all().numel() == 0


def test_ctx_trim_releases_cached_blocks():
    """vidc_ctx_trim: idle scratch goes back to the driver, live objects keep working, later calls re-allocate."""
    import torch
    from vector_db_id_compression_amd import _lib
    from vector_db_id_compression_amd.codecs import RocLists
    ctx = _lib.Context(0)
    rng = np.random.default_rng(123)
    sizes = rng.integers(0, 3000, 400)
    lists = [np.sort(rng.choice(1 << 22, size=int(s), replace=False)).astype(np.uint64) for s in sizes]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ids = np.concatenate(lists)
    r = RocLists.encode(off, ids, ctx=ctx)
    dec = r.decode_all().cpu().numpy().copy()
    free0 = torch.cuda.mem_get_info()[0]
    freed = ctx.trim()
    assert freed > 0
    assert torch.cuda.mem_get_info()[0] >= free0  # nothing new was taken; cached scratch went back
    assert ctx.trim() == 0
    dec2 = r.decode_all().cpu().numpy()  # the object's own buffers were not touched
    assert np.array_equal(dec, dec2)
    assert np.array_equal(np.sort(dec2.view(np.uint64)), np.sort(ids))
    del r
    ctx.close()


@pytest.mark.parametrize("K", [65, 128, 200])
def test_wide_graph_rows_take_the_list_kernels(K, oracle):
    """NSG128 / NSG256-style graphs (max_degree is a CLI argument of graph_dynamic_bench_invlists.py): rows wider than
    the lane-per-row kernels become CSR lists; every container answers get_neighbors like the reference."""
    from vector_db_id_compression_amd import altid

    rng = np.random.default_rng(K)
    N = 3000
    rows = np.full((N, K), -1, dtype=np.int32)
    for i in range(N):
        d = int(rng.integers(0, K + 1)) if i > 2 else (0, K, 1)[i]
        v = rng.choice(N, size=d, replace=False)
        rows[i, :d] = v
    for name, cls in (("compact", altid.CompactBitNSGGraph), ("elias-fano", altid.EliasFanoNSGGraph), ("roc", altid.ROCNSGGraph)):
        g = cls(rows.copy())
        out, cnt = g.get_neighbors_batch(np.arange(N))
        for i in range(0, N, 7):
            d = int((rows[i] >= 0).sum())
            assert cnt[i] == d, (name, i)
            li = rows[i, :d].astype(np.uint64)
            if name == "roc" and d:
                P = oracle.list_precision(li)
                e = oracle.roc_encode(li, P)
                want = oracle.roc_decode(e["head"], e["words"], d, P, e["mt_draws"])[0]
                assert out[i, :d].astype(np.uint64).tolist() == want.tolist(), (name, i)
            elif name == "elias-fano":
                assert out[i, :d].tolist() == sorted(rows[i, :d].tolist()), (name, i)
            else:
                assert out[i, :d].tolist() == rows[i, :d].tolist(), (name, i)
            assert np.all(out[i, d:] == -1), (name, i)
        if name == "compact":
            assert g.stride == (K * g.bits + 7) // 8
            d = int((rows[3] >= 0).sum())
            vals = np.concatenate([rows[3, :d], [N] if d < K else []]).astype(np.uint64)
            want = oracle.packed_encode(vals, g.bits)
            got = g._c.export_row(3)
            assert np.array_equal(got[: want.size], want) and not got[want.size:].any()
        assert g.get_neighbors(1).size == K


@pytest.mark.parametrize("K", [600, 5000])
def test_very_wide_roc_graph_rows_decode_through_kernels_that_write_rows(K, oracle):
    """Rows with more than 256 (K = 600) and more than 4096 (K = 5000) edges: as lists they would qualify for the chain / bucket
    decoders, which only write u64 list output -- the wide-row path must plan the decoders that honour int32 rows (a NULL output
    pointer otherwise).  Decoded rows against the oracle, -1 padding, node subsets in request order."""
    from vector_db_id_compression_amd import altid

    rng = np.random.default_rng(K)
    N = 40
    rows = np.full((N, K), -1, dtype=np.int32)
    deg = [K, K - 1, 300, 0, 1, 257, min(K, 4500), 64, 65] + [int(v) for v in rng.integers(0, K + 1, N - 9)]
    for i, d in enumerate(deg):
        rows[i, :d] = rng.choice(1 << 22, size=d, replace=False)
    g = altid.ROCNSGGraph(rows.copy())
    out, cnt = g.get_neighbors_batch(np.arange(N))
    for i, d in enumerate(deg):
        assert cnt[i] == d
        if d:
            li = rows[i, :d].astype(np.uint64)
            P = oracle.list_precision(li)
            e = oracle.roc_encode(li, P)
            want = oracle.roc_decode(e["head"], e["words"], d, P, e["mt_draws"])[0]
            assert out[i, :d].astype(np.uint64).tolist() == want.tolist(), i
        assert np.all(out[i, d:] == -1), i
    sub = np.array([6, 0, 3, 6, 2])
    out2, cnt2 = g.get_neighbors_batch(sub)
    for k, i in enumerate(sub):
        assert cnt2[k] == cnt[i] and np.array_equal(out2[k], out[i])


def test_batched_graph_search_identical_with_compressed_graphs():
    """graph_dynamic_bench_invlists.py:103-146 in miniature: the frontier search of a query batch returns the same ids and
    distances with every compressed graph swapped in (one get_neighbors launch per round)."""
    from vector_db_id_compression_amd import altid
    from vector_db_id_compression_amd.graph_search import RawGraph, knn_graph, search, search_batched

    rng = np.random.default_rng(12)
    x = rng.normal(size=(4000, 16)).astype(np.float32)
    xq = rng.normal(size=(12, 16)).astype(np.float32)
    rows = knn_graph(x, 24, seed=3)
    Dref, Iref = search_batched(RawGraph(rows), x, xq, 10, L=32)
    assert (Iref >= 0).all()
    D1, I1 = search(RawGraph(rows), x, xq[:3], 10, L=32)  # the one-query-at-a-time search visits the same pool
    np.testing.assert_array_equal(I1, Iref[:3])
    for name, cls in altid.AVAILABLE_COMPRESSED_GRAPHS.items():
        if cls is None:
            continue
        D, I = search_batched(cls(rows.copy()), x, xq, 10, L=32)
        np.testing.assert_array_equal(I, Iref, err_msg=name)
        np.testing.assert_allclose(D, Dref, rtol=1e-6, err_msg=name)


@pytest.mark.parametrize("name", ["packed-bits", "elias-fano", "roc", "wavelet-tree", "wavelet-tree-1"])
def test_id_compression_switch(name):
    """search_ivf_qinco.py:502-523: every `--id_compression` choice installs its container and leaves the search unchanged."""
    from vector_db_id_compression_amd import custom_invlists as ci
    from vector_db_id_compression_amd.ivf import IVFIndex

    xt, xb, xq = _dataset(16, 4000, 4000, 6)
    index = IVFIndex(16, 16, ("PQ", 4))
    index.train(xt)
    index.add(xb)
    index.nprobe = 4
    index.parallel_mode = 3
    Dref, Iref = index.search(xq, 5)
    assert ci.apply_id_compression(index, "none")[1] == {}
    il, st = ci.apply_id_compression(index, name)
    assert index.invlists is il and st["compressed_ids_size_in_bytes"] == il.compressed_ids_size_in_bytes > 0
    assert st["id_compression_time"] >= 0 and getattr(il, "wt_type", 0) == (1 if name == "wavelet-tree-1" else 0)
    D, I = index.search_defer_id_decoding(xq, 5)
    np.testing.assert_array_equal(I, Iref)
    with pytest.raises(ValueError):
        ci.apply_id_compression(index, "zstd")


def test_sharded_bench_over_rccl_when_two_gpus_are_visible(tmp_path):
    """`bench.py --sharded` over the nccl (= RCCL) backend: one index partitioned over 2 ranks, per-rank encode + decode,
    search-shaped gather on rank 0 (verified inside the bench against the index).  Needs two GPUs; the driver's multi-GPU
    tier and this test are the only places the RCCL path runs (the gloo test covers the logic on CPU)."""
    import json
    import os
    import socket
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--sharded", "--workload", "c5",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["gather_verified"] is True
    assert sum(d["per_rank"]["ids"]) == 10_000_000 and abs(d["per_rank"]["ids"][0] - d["per_rank"]["ids"][1]) <= 65536


def test_bench_gpus_flag_launches_the_ranks_itself(tmp_path):
    """`python bench.py --gpus N` (no torchrun): the bench starts its N ranks, one per GPU over RCCL, and rank 0 prints
    n_gpus == N.  With fewer GPUs than asked for it must fail loudly instead of printing a 1-GPU line as N GPUs."""
    import json
    import os
    import subprocess
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    if have < 2:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode != 0 and "only 1 GPU" in (r.stdout + r.stderr)
        return
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extra",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["verified_roundtrip"] is True
    assert d["config"]["ids_per_gpu"] == 1_000_000


@pytest.mark.parametrize("which", range(5))
def test_decode_gather_is_decode_lists_plus_indexing_and_copies_only_the_results(which):
    """SURVEY 8(f)-1 / custom_invlists_impl.cpp:508-525: `ids = get_ids(list)` per touched list + `labels[r] = ids[offset]` as ONE
    library call with the scatter on the device.  Same ids as decode_lists + host indexing; 8 bytes per result cross PCIe;
    items outside their list, slots outside the request and list numbers outside the object are errors."""
    from vector_db_id_compression_amd import VidcError
    from vector_db_id_compression_amd.ivf import IVFIndex

    xt, xb, _ = _dataset(8, 2000, 30000, 1, seed=3)
    index = IVFIndex(8, 64, "Flat")
    index.train(xt)
    index.add(xb)
    comp = _classes()[which](index.invlists)
    rng = np.random.default_rng(which)
    sizes = np.array([comp.list_size(l) for l in range(64)])
    touched = rng.permutation(np.nonzero(sizes)[0])[:23].astype(np.uint64)  # an arbitrary order, not ascending
    n_items = 5000
    slot = rng.integers(0, touched.size, n_items).astype(np.uint64)
    off = (rng.random(n_items) * sizes[touched.astype(np.int64)][slot.astype(np.int64)]).astype(np.uint64)
    ids, out_off = comp.decode_lists(touched)
    want = ids.cpu().numpy()[out_off[slot.astype(np.int64)].astype(np.int64) + off.astype(np.int64)]
    ctx = comp._c.ctx
    before = ctx.d2h_bytes()
    got = comp.decode_gather(touched, slot, off)
    assert ctx.d2h_bytes() - before == 8 * n_items
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(comp.get_single_ids(touched[slot.astype(np.int64)], off), want)
    assert comp.decode_gather(touched, [], []).size == 0
    bad_off = off.copy()
    bad_off[17] = sizes[int(touched[int(slot[17])])]
    with pytest.raises(VidcError):
        comp.decode_gather(touched, slot, bad_off)
    bad_slot = slot.copy()
    bad_slot[3] = touched.size
    with pytest.raises(VidcError):
        comp.decode_gather(touched, bad_slot, off)
    with pytest.raises(VidcError):
        comp.decode_gather(np.array([64], np.uint64), [0], [0])
