"""GPU parity: packed-bits and Elias-Fano kernels vs the CPU oracle (layouts, sizes, decode, random access)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lists(rng, sizes, nbits=20, sort=True):
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    lists = []
    for s in sizes:
        li = rng.choice(1 << nbits, size=int(s), replace=False).astype(np.uint64)
        lists.append(np.sort(li) if sort else li)
    return off, (np.concatenate(lists) if lists else np.zeros(0, np.uint64)), lists


def test_packed_bits_layout_and_roundtrip(oracle):
    from vector_db_id_compression_amd.codecs import PackedLists

    rng = np.random.default_rng(0)
    sizes = [0, 1, 2, 3, 7, 64, 65, 1000, 0, 4097]
    ntotal = int(sum(sizes))
    perm = rng.permutation(ntotal).astype(np.uint64)  # ids < ntotal as the reference requires (:87)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    pk = PackedLists.encode(off, perm)
    bits = oracle.packed_bits_for(ntotal)
    assert pk.bits == bits == PackedLists.bits_for(ntotal)
    assert pk.compressed_bytes == sum((s * bits + 7) // 8 for s in sizes)  # custom_invlists_impl.cpp:80,85
    for l, s in enumerate(sizes):
        li = perm[int(off[l]):int(off[l + 1])]
        assert np.array_equal(pk.export_bytes(l), oracle.packed_encode(li, bits)), f"list {l}"
    assert np.array_equal(pk.decode_all().cpu().numpy().view(np.uint64), perm)
    qs_l = np.array([3, 7, 7, 9, 1], dtype=np.uint64)
    qs_o = np.array([2, 0, 999, 4096, 0], dtype=np.uint64)
    got = pk.get(qs_l, qs_o)  # get_single_id, :108-113
    want = [perm[int(off[int(l)]) + int(o)] for l, o in zip(qs_l, qs_o)]
    assert got.tolist() == [int(x) for x in want]


@pytest.mark.parametrize("bits", [1, 5, 13, 31, 32, 33, 47, 63, 64])
def test_packed_bits_widths(oracle, bits):
    from vector_db_id_compression_amd.codecs import PackedLists

    rng = np.random.default_rng(bits)
    n = 777
    hi = (1 << bits) - 1
    ids = (rng.integers(0, 1 << 62, size=n, dtype=np.uint64) * np.uint64(3) + np.uint64(1)) & np.uint64(hi)
    off = np.array([0, 300, 300, n], dtype=np.uint64)
    pk = PackedLists.encode(off, ids, bits=bits)
    assert np.array_equal(pk.decode_all().cpu().numpy().view(np.uint64), ids)
    assert np.array_equal(pk.export_bytes(0), oracle.packed_encode(ids[:300], bits))


@pytest.mark.parametrize("nlist", [1, 2, 63, 255, 256, 257, 511, 512, 1024, 2047, 2303, 4095, 4096, 4097, 8192, 12289, 70000])
def test_packed_geometry_kernel_list_counts(oracle, nlist):
    """One launch builds chunk table, word offsets and padding words by a chained scan over 4096-list tiles (k_packed_table):
    list counts on and around the tile size, empty lists, lists of several chunks; per-list byte image against the oracle.
    (An object of one tile spreads its nlist + 1 indices over the 256 threads, 1 .. 16 per thread: the counts below 4096.)"""
    from vector_db_id_compression_amd.codecs import PackedLists

    rng = np.random.default_rng(1000 + nlist)
    sizes = rng.integers(0, 40, size=nlist)
    sizes[rng.integers(0, nlist, size=max(1, nlist // 50))] = 0
    for k in rng.integers(0, nlist, size=min(nlist, 6)):
        sizes[k] = int(rng.integers(500, 3000))  # several 512-id chunks
    ntotal = int(sizes.sum())
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ids = rng.permutation(max(ntotal, 1))[:ntotal].astype(np.uint64)
    pk = PackedLists.encode(off, ids)
    bits = pk.bits
    assert pk.compressed_bytes == int(sum((int(s) * bits + 7) // 8 for s in sizes))
    assert np.array_equal(pk.decode_all().cpu().numpy().view(np.uint64), ids)
    for l in list(rng.integers(0, nlist, size=min(nlist, 40))) + [0, nlist - 1, int(np.argmax(sizes))]:
        li = ids[int(off[l]):int(off[l + 1])]
        assert np.array_equal(pk.export_bytes(int(l)), oracle.packed_encode(li, bits)), f"list {l}"
    if ntotal:
        ql = rng.integers(0, nlist, size=64).astype(np.uint64)
        ql = ql[sizes[ql.astype(np.int64)] > 0]
        qo = np.array([rng.integers(0, sizes[int(l)]) for l in ql], dtype=np.uint64)
        want = [int(ids[int(off[int(l)]) + int(o)]) for l, o in zip(ql, qo)]
        assert pk.get(ql, qo).tolist() == want


def test_packed_domain_error():
    from vector_db_id_compression_amd import VidcError
    from vector_db_id_compression_amd.codecs import PackedLists

    with pytest.raises(VidcError):
        PackedLists.encode(np.array([0, 3], dtype=np.uint64), np.array([1, 2, 9], dtype=np.uint64), bits=3)


def test_elias_fano_vs_oracle(oracle):
    from vector_db_id_compression_amd.codecs import EfLists

    rng = np.random.default_rng(1)
    sizes = [0, 1, 2, 5, 63, 64, 65, 300, 5000, 0, 20000]
    off, ids, lists = _lists(rng, sizes)
    ef = EfLists.encode(off, ids)
    info = ef.info()
    total_bits = 0
    for l, li in enumerate(lists):
        if li.size == 0:
            continue
        e = oracle.ef_build(li)
        low, high, lb, hb = ef.export(l)
        assert int(info["low_bits"][l]) == e["l"] and int(info["universe"][l]) == int(li.max())
        assert lb == e["low_nbits"] and hb == e["high_nbits"]  # elias_fano.hpp:28-29
        assert np.array_equal(low, e["low"]), f"list {l}: low stream words"
        assert np.array_equal(high, e["high"]), f"list {l}: high stream words"
        total_bits += lb + hb
    assert ef.compressed_bytes == total_bits // 8  # custom_invlists_impl.cpp:272-282
    assert np.array_equal(ef.decode_all().cpu().numpy().view(np.uint64), ids)  # ascending (:305-308)
    ql = np.array([10, 10, 10, 8, 1, 5, 7], dtype=np.uint64)
    qo = np.array([0, 19999, 7777, 4999, 0, 63, 123], dtype=np.uint64)
    got = ef.get(ql, qo)  # ef->select(offset), :314-318
    want = [int(lists[int(l)][int(o)]) for l, o in zip(ql, qo)]
    assert got.tolist() == want


def _check_ef_lists(oracle, ef, off, ids, sample):
    info = ef.info()
    for l in sample:
        li = ids[int(off[l]):int(off[l + 1])]
        if li.size == 0:
            continue
        e = oracle.ef_build(li)
        low, high, lb, hb = ef.export(int(l))
        assert int(info["low_bits"][l]) == e["l"] and lb == e["low_nbits"] and hb == e["high_nbits"]
        assert np.array_equal(low, e["low"]) and np.array_equal(high, e["high"]), f"list {l}"
    assert np.array_equal(ef.decode_all().cpu().numpy().view(np.uint64), ids)


@pytest.mark.parametrize("nlist", [1, 1023, 1024, 1025, 4096, 262144, 262145, 300000])
def test_elias_fano_single_pass_encoder_list_counts(oracle, nlist):
    """The single-pass encoder computes offsets per tile of lists: one launch up to 1024 lists, tiles of 256 lists up
    to 2^18 lists, tiles of 1024 beyond (csrc/ef.hip, k_ef_meta / k_ef_offsets): list counts on both sides of every
    switch, lists of 0..3 chunks mixed, streams of sampled lists against the oracle and the decode of everything."""
    from vector_db_id_compression_amd.codecs import EfLists

    rng = np.random.default_rng(nlist)
    sizes = rng.integers(0, 12, size=nlist)
    big = rng.integers(0, nlist, size=min(nlist, 40))
    sizes[big] = rng.integers(500, 1600, size=big.size)  # one to four chunks of 512 ids
    sizes[-1] = 700 if nlist > 1 else 5  # (the last list: its end is the entry behind all lists)
    off = np.zeros(nlist + 1, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)
    ids = rng.integers(0, 1 << 22, size=int(off[-1]), dtype=np.uint64)
    ids = ids[np.lexsort((ids, np.repeat(np.arange(nlist), sizes)))]  # every list ascending: the single-pass path
    ef = EfLists.encode(off, ids)
    sample = np.unique(np.concatenate([big, [0, nlist - 1], np.arange(min(nlist, 50)), rng.integers(0, nlist, size=100)]))
    _check_ef_lists(oracle, ef, off, ids, sample)


def test_elias_fano_decode_records_from_the_encoder_and_from_the_lazy_build(monkeypatch):
    """Objects below 2^18 batches get their bulk-decode records from the encoder's chunk wavefronts (ef_store_rec); the others --
    and every object under VIDC_EF_LAZY_RECS=1 -- build them on the first decode_all (k_ef_build_recs).  Both must decode alike:
    dense lists (many batches per list), sparse lists (a chunk spanning many batches), empty lists, 64-bit ids."""
    import torch
    from vector_db_id_compression_amd.codecs import EfLists

    rng = np.random.default_rng(77)
    for wide in (False, True):
        sizes = np.concatenate([rng.integers(0, 300, size=3000), [0, 0, 70000, 1, 513, 512, 511, 20000]])
        rng.shuffle(sizes)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        parts = []
        for s in sizes:
            s = int(s)
            if s == 0:
                continue
            if s >= 20000:  # dense: consecutive ids with a few gaps
                li = np.cumsum(rng.integers(1, 3, size=s)).astype(np.uint64)
            else:  # sparse
                li = np.sort(rng.choice(1 << 28, size=s, replace=False)).astype(np.uint64)
            parts.append(li + (np.uint64(1) << np.uint64(40) if wide else np.uint64(0)))
        ids = torch.from_numpy(np.concatenate(parts).view(np.int64)).cuda()
        outs = []
        for lazy in (False, True):
            if lazy:
                monkeypatch.setenv("VIDC_EF_LAZY_RECS", "1")
            else:
                monkeypatch.delenv("VIDC_EF_LAZY_RECS", raising=False)
            e = EfLists.encode(off, ids)
            a = e.decode_all().clone()
            b = e.decode_all().clone()  # (second call: the records are there either way)
            assert torch.equal(a, ids) and torch.equal(b, ids), (wide, lazy)
            outs.append(e.compressed_bytes)
        assert outs[0] == outs[1]


def test_elias_fano_sparse_chunk_owns_many_directory_entries(oracle):
    """A chunk whose ids are far apart in the high stream owns many batches of the select directory (the lane-per-batch
    path of ef_chunk_directory), a dense one owns none: lists that are dense first and sparse at the end, and the
    other way round; `get` walks the directory."""
    from vector_db_id_compression_amd.codecs import EfLists

    rng = np.random.default_rng(77)
    dense = np.arange(100000, dtype=np.uint64)
    sparse = np.uint64(100000) + np.sort(rng.integers(0, 1 << 31, size=600, dtype=np.uint64))
    a = np.concatenate([dense, sparse])                       # sparse tail
    b = np.concatenate([np.sort(rng.integers(0, 1 << 20, size=300, dtype=np.uint64)),
                        np.uint64(1 << 30) + np.arange(50000, dtype=np.uint64)])  # sparse head, one huge gap
    c = np.array([3, 1 << 31], dtype=np.uint64)               # two ids, l = 30
    lists = [a, b, c, np.zeros(0, np.uint64), np.sort(rng.integers(0, 1 << 40, size=5000, dtype=np.uint64))]  # (wide ids)
    off = np.zeros(len(lists) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([x.size for x in lists])
    ids = np.concatenate(lists)
    for sub in ([0, 1, 2, 3], [0, 1, 2, 3, 4]):  # 32-bit and 64-bit encoder kernels
        o = np.zeros(len(sub) + 1, dtype=np.uint64)
        o[1:] = np.cumsum([lists[i].size for i in sub])
        x = np.concatenate([lists[i] for i in sub])
        ef = EfLists.encode(o, x)
        _check_ef_lists(oracle, ef, o, x, range(len(sub)))
        ql = np.array([0, 0, 0, 1, 1, 1, 2], dtype=np.uint64)
        qo = np.array([0, 99999, 100599, 0, 299, 50299, 1], dtype=np.uint64)
        assert ef.get(ql, qo).tolist() == [int(lists[int(l)][int(q)]) for l, q in zip(ql, qo)]


def test_elias_fano_unsorted_input_and_perm(oracle):
    """canonicalize_order_inplace (custom_invlists_impl.cpp:324-339): ids are sorted, codes follow via perm."""
    from vector_db_id_compression_amd.codecs import EfLists

    rng = np.random.default_rng(2)
    sizes = [0, 3, 100, 64, 2500, 1]
    off, ids, lists = _lists(rng, sizes, sort=False)
    ef = EfLists.encode(off, ids, want_perm=True)
    dec = ef.decode_all().cpu().numpy().view(np.uint64)
    perm = ef.perm()
    for l, li in enumerate(lists):
        a, b = int(off[l]), int(off[l + 1])
        assert np.array_equal(dec[a:b], np.sort(li))
        assert np.array_equal(li[perm[a:b]], dec[a:b])


def test_elias_fano_edge_universes(oracle):
    from vector_db_id_compression_amd.codecs import EfLists

    cases = [np.array([0], np.uint64), np.array([0, 1, 2, 3], np.uint64), np.array([1 << 40], np.uint64),
             np.array([5, 5, 5, 9], np.uint64), np.arange(1000, dtype=np.uint64) * np.uint64(1000003)]
    off = np.concatenate([[0], np.cumsum([c.size for c in cases])]).astype(np.uint64)
    ids = np.concatenate(cases)
    ef = EfLists.encode(off, ids)
    assert np.array_equal(ef.decode_all().cpu().numpy().view(np.uint64), ids)
    for l, c in enumerate(cases):
        e = oracle.ef_build(c)
        low, high, lb, hb = ef.export(l)
        assert (lb, hb) == (e["low_nbits"], e["high_nbits"])
        assert np.array_equal(low, e["low"]) and np.array_equal(high, e["high"])


def test_elias_fano_narrow_and_wide_encoders_agree(oracle):
    """Objects whose ids all fit 32 bits take the 32-bit single-pass encoder; one id beyond 2^32 anywhere switches
    the object to the 64-bit kernel.  Both must write the same streams for the lists they share."""
    from vector_db_id_compression_amd.codecs import EfLists

    rng = np.random.default_rng(321)
    sizes = [1, 2, 63, 64, 65, 511, 512, 513, 1024, 1500, 5000, 70000]
    lists = [np.sort(rng.choice(1 << 31, size=n, replace=False).astype(np.uint64) * np.uint64(2) + np.uint64(1))
             for n in sizes]  # odd ids up to 2^32 - 1
    lists.append(np.arange(3000, dtype=np.uint64))              # dense: l = 0, no low stream
    lists.append(np.array([0xffffffff], dtype=np.uint64))       # the largest narrow id
    wide = lists + [np.array([7, 1 << 32, (1 << 45) + 3], dtype=np.uint64)]
    got = []
    for ls in (lists, wide):
        off = np.concatenate([[0], np.cumsum([c.size for c in ls])]).astype(np.uint64)
        ef = EfLists.encode(off, np.concatenate(ls))
        assert np.array_equal(ef.decode_all().cpu().numpy().view(np.uint64), np.concatenate(ls))
        got.append([ef.export(l) for l in range(len(lists))])
    for l, c in enumerate(lists):
        (lo_a, hi_a, lb_a, hb_a), (lo_b, hi_b, lb_b, hb_b) = got[0][l], got[1][l]
        assert (lb_a, hb_a) == (lb_b, hb_b) and np.array_equal(lo_a, lo_b) and np.array_equal(hi_a, hi_b), f"list {l}"
        if c.size <= 5000:
            e = oracle.ef_build(c)
            assert (lb_a, hb_a) == (e["low_nbits"], e["high_nbits"])
            assert np.array_equal(lo_a, e["low"]) and np.array_equal(hi_a, e["high"]), f"list {l}"


def test_full_size_properties():
    """Config-2 shape at full size: EF and packed round-trip every id, sizes match SURVEY 6."""
    from vector_db_id_compression_amd import synth
    from vector_db_id_compression_amd.codecs import EfLists, PackedLists

    w = synth.workload("s1")
    off, ids = w["offsets"], w["ids"]
    ef = EfLists.encode(off, ids)
    assert np.array_equal(ef.decode_all().cpu().numpy().view(np.uint64), ids)
    assert 10.7 < 8.0 * ef.compressed_bytes / ids.size < 11.0  # 10.844 bit/id
    pk = PackedLists.encode(off, ids)
    assert pk.bits == 20 and np.array_equal(pk.decode_all().cpu().numpy().view(np.uint64), ids)


def test_save_load_roundtrip(tmp_path):
    """Flat images of Elias-Fano / packed-bits objects (lists and graph rows): load rebuilds an object that decodes,
    selects and reports sizes exactly like the original (the select directory is rebuilt from the high stream)."""
    from vector_db_id_compression_amd import synth
    from vector_db_id_compression_amd.codecs import EfLists, PackedLists

    rng = np.random.default_rng(5)
    sizes = [0, 1, 3, 64, 65, 700, 0, 9000, 4097, 2]
    off, ids, lists = _lists(rng, sizes, nbits=26)
    ef = EfLists.encode(off, ids)
    ef.save(tmp_path / "ef.npz")
    ef2 = EfLists.load(tmp_path / "ef.npz")
    assert ef2.compressed_bytes == ef.compressed_bytes
    assert np.array_equal(ef2.decode_all().cpu().numpy(), ef.decode_all().cpu().numpy())
    ql = np.array([3, 7, 7, 8, 1, 9], dtype=np.uint64)
    qo = np.array([63, 0, 8999, 4096, 0, 1], dtype=np.uint64)
    assert np.array_equal(ef2.get(ql, qo), ef.get(ql, qo))
    d, o = ef2.decode_lists(np.array([7, 4, 7], dtype=np.uint64))
    assert np.array_equal(d.cpu().numpy().view(np.uint64)[: sizes[7]], np.sort(lists[7]))
    for l in (3, 7):
        for a, b in zip(ef.export(l), ef2.export(l)):
            assert np.array_equal(a, b)

    rows = synth.make_graph_rows(3000, 32, seed=9, dmin=0)
    g = EfLists.encode_rows(rows)
    g.save(tmp_path / "efg.npz")
    g2 = EfLists.load(tmp_path / "efg.npz")
    a, ca = g.decode_rows(None, 32)
    b, cb = g2.decode_rows(None, 32)
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy()) and np.array_equal(ca, cb)

    perm = rng.permutation(int(off[-1])).astype(np.uint64)
    pk = PackedLists.encode(off, perm)
    pk.save(tmp_path / "pk.npz")
    pk2 = PackedLists.load(tmp_path / "pk.npz")
    assert pk2.bits == pk.bits and pk2.compressed_bytes == pk.compressed_bytes
    assert np.array_equal(pk2.decode_all().cpu().numpy().view(np.uint64), perm)
    assert np.array_equal(pk2.get(ql, qo), pk.get(ql, qo))
    for l in (3, 7, 9):
        assert np.array_equal(pk2.export_bytes(l), pk.export_bytes(l))


def test_streams_from_dirty_pool_blocks_equal_streams_from_fresh_ones(monkeypatch, tmp_path):
    """The high stream of an Elias-Fano object up to 64 MB is not cleared before the chunk kernels write it (every high word has ONE
    owning chunk, empty words included), the encoder writes the decode records, packed bits writes its own padding words, ROC its
    arenas: all into blocks from the context's pool, which hold whatever an earlier call left there.  the context's pool-poison switch (vidc_ctx_debug_pool_poison; VIDC_POOL_POISON=1 at start-up) fills every
    block handed out with 0xFF; the word images (Elias-Fano low / high, packed words, ROC stack words) and the decoded ids must be
    the ones a cleared stream (VIDC_EF_MEMSET=1) / a fresh block gives.  Zipf lists, empty lists, lists whose high stream spans
    more than a chunk's LDS window, a sparse tail, 64-bit ids."""
    import torch
    from vector_db_id_compression_amd import synth
    from vector_db_id_compression_amd.codecs import EfLists, PackedLists, RocLists

    rng = np.random.default_rng(99)
    cases = []
    off, ids = synth.make_lists_numpy(300_000, 700, 0.75, seed=3)
    cases.append(("zipf", off, ids))
    sizes = np.concatenate([rng.integers(0, 40, size=5000), [0, 0, 0, 2000, 513, 512, 1]])
    rng.shuffle(sizes)
    off2 = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    # few ids over a universe of 2^31 (long runs of empty high words) and a dense run with a far outlier behind it
    parts = [np.sort(rng.choice(1 << 31, size=int(s), replace=False)).astype(np.uint64) if s != 2000
             else np.concatenate([np.arange(1999, dtype=np.uint64), [np.uint64((1 << 31) - 5)]]) for s in sizes if s]
    cases.append(("sparse + empty lists", off2, np.concatenate(parts)))
    sz3 = np.array([30000, 0, 700, 9], dtype=np.int64)
    off3 = np.concatenate([[0], np.cumsum(sz3)]).astype(np.uint64)
    ids3 = np.concatenate([np.sort(rng.choice(1 << 40, size=int(s), replace=False)).astype(np.uint64) for s in sz3 if s])
    cases.append(("64-bit ids", off3, ids3))

    def images(tag):
        out = {}
        for name, o, x in cases:
            d = torch.from_numpy(x.view(np.int64)).cuda()
            e = EfLists.encode(o, d)
            e.save(tmp_path / f"ef_{tag}.npz")
            z = np.load(tmp_path / f"ef_{tag}.npz")
            out[name, "ef"] = (z["low"].copy(), z["high"].copy(), e.decode_all().cpu().numpy().copy(), e.compressed_bytes)
            if int(x.max()) < (1 << 31):
                p = PackedLists.encode(o, d, bits=32)
                p.save(tmp_path / f"pk_{tag}.npz")
                out[name, "packed"] = (np.load(tmp_path / f"pk_{tag}.npz")["words"].copy(), p.decode_all().cpu().numpy().copy())
                r = RocLists.encode(o, d, want_perm=True)
                out[name, "roc"] = (r.all_words().copy(), r.info()["heads"].copy(), r.perm().copy(), r.decode_all().cpu().numpy().copy())
        return out

    from vector_db_id_compression_amd import _lib

    ctx = _lib.default_context()
    ctx.set_pool_poison(False)
    monkeypatch.setenv("VIDC_EF_MEMSET", "1")
    clean = images("clean")
    monkeypatch.delenv("VIDC_EF_MEMSET", raising=False)
    ctx.set_pool_poison(True)
    try:
        dirty = images("dirty")
        dirty2 = images("dirty2")  # (blocks released by the first poisoned pass, poisoned again)
    finally:
        ctx.set_pool_poison(False)
    assert clean.keys() == dirty.keys()
    for key in clean:
        for i, (a, b, c) in enumerate(zip(clean[key], dirty[key], dirty2[key])):
            assert np.array_equal(a, b) and np.array_equal(a, c), (key, i)


def test_elias_fano_decode_lists_shares_long_lists_between_workgroups():
    """decode_lists of a few long lists (what a search with nprobe 16 touches) spreads every list's 64-word batches over several
    workgroups (k_ef_decode, nsplit > 1): lists of 1 .. 200 000 ids, dense and sparse, repeated and empty lists in the request,
    against slices of decode_all."""
    import torch
    from vector_db_id_compression_amd.codecs import EfLists

    rng = np.random.default_rng(5)
    sizes = np.array([200000, 0, 1, 52114, 4097, 63, 70000, 2048, 9], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    parts = []
    for i, s in enumerate(sizes):
        if s == 0:
            continue
        if i % 2 == 0:  # dense: consecutive ids with small gaps (many ids per high word)
            parts.append(np.cumsum(rng.integers(1, 3, size=int(s))).astype(np.uint64))
        else:           # sparse over 2^31
            parts.append(np.sort(rng.choice(1 << 31, size=int(s), replace=False)).astype(np.uint64))
    ids = np.concatenate(parts)
    ef = EfLists.encode(off, ids)
    full = ef.decode_all().cpu().numpy().view(np.uint64)
    assert np.array_equal(full, ids)
    for req in ([0], [3, 0, 6], [6, 6, 1, 2, 0, 8, 5, 4, 7, 3], list(range(9))):
        got, goff = ef.decode_lists(np.array(req, dtype=np.uint64))
        got = got.cpu().numpy().view(np.uint64)
        assert int(goff[-1]) == int(sizes[req].sum())
        for k, l in enumerate(req):
            assert np.array_equal(got[int(goff[k]):int(goff[k + 1])], ids[int(off[l]):int(off[l + 1])]), (req, k)
