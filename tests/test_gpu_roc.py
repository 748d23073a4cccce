"""GPU parity: HIP ROC kernels (through the C-ABI) vs the reference-generated golden vectors and the CPU oracle."""
import numpy as np
import pytest

from golden_cases import CASES, fnv_stream, fnv_u64, make_ids

pytestmark = pytest.mark.gpu

CASE_BY_NAME = {c["name"]: c for c in CASES}


@pytest.fixture(scope="module")
def roc():
    from vector_db_id_compression_amd.codecs import RocLists

    return RocLists


@pytest.mark.parametrize("name", [c["name"] for c in CASES])
def test_golden_case(roc, golden, name):
    """Encoder stream (head + words), sampling permutation and decoded order are bit-identical to the reference."""
    g = {c["name"]: c for c in golden}[name]
    case = CASE_BY_NAME[name]
    ids = make_ids(case)
    off = np.array([0, ids.size], dtype=np.uint64)
    mode = case.get("precision")
    r = roc.encode(off, ids, precision_mode=-1 if mode is None else mode, want_perm=True)
    info = r.info()
    words = r.words(0)
    assert int(info["precision"][0]) == g["precision"]
    assert int(info["heads"][0]) == g["head"]
    assert int(info["nwords"][0]) == g["nwords"]
    assert fnv_stream(int(info["heads"][0]), words) == g["stream_fnv"]
    assert r.compressed_bytes == 8 + 4 * g["nwords"]  # ANSState::size(), codec.h:42-44
    assert fnv_u64(r.perm().astype(np.uint64)) == g["perm_fnv"]
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    assert fnv_u64(dec) == g["decoded_fnv"]
    if "decoded" in g:
        assert [int(x) for x in dec] == g["decoded"]
        assert [int(x) for x in words] == g["words"]
    # SURVEY appendix A self-check: valid streams end in the initial ANS state
    assert (r.last_decode_nonclean == 0) == g["roundtrip_set_ok"] or not g["roundtrip_set_ok"]
    if g["roundtrip_set_ok"]:
        assert r.last_decode_nonclean == 0
        assert np.array_equal(np.sort(dec), np.sort(ids))


def _random_lists(rng, sizes, nbits=20, sort=True):
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    lists = []
    for s in sizes:
        li = rng.choice(1 << nbits, size=int(s), replace=False).astype(np.uint64)
        lists.append(np.sort(li) if sort else li)
    ids = np.concatenate(lists) if lists else np.zeros(0, np.uint64)
    return off, ids, lists


def _check_against_oracle(o, r, off, lists, perm, dec):
    info = r.info()
    for l, li in enumerate(lists):
        if li.size == 0:
            assert info["nwords"][l] == 0
            continue
        P = o.list_precision(li)
        e = o.roc_encode(li, P)
        a, b = int(off[l]), int(off[l + 1])
        assert int(info["precision"][l]) == P
        assert int(info["heads"][l]) == e["head"], f"list {l} (n={li.size})"
        assert np.array_equal(r.words(l, int(info["nwords"][l])), e["words"]), f"list {l}"
        assert int(info["mt_draws"][l]) == e["mt_draws"]
        if perm is not None:
            assert np.array_equal(perm[a:b], e["perm"]), f"list {l}"
        if dec is not None:
            # what the reference decoder makes of this stream (== the sampling order unless the precision quirk
            # Q3 makes the list lossy)
            want = o.roc_decode(e["head"], e["words"], li.size, P, e["mt_draws"])[0]
            assert np.array_equal(dec[a:b], want), f"list {l}"


def test_batch_mixed_sizes_vs_oracle(roc, oracle):
    """Many lists of ragged sizes (empty, tiny, every kernel class) in one call."""
    rng = np.random.default_rng(1)
    sizes = np.concatenate([rng.integers(0, 70, 200), rng.integers(60, 3000, 60), [5000, 9000, 0, 1, 2, 40000, 64, 65]])
    rng.shuffle(sizes)
    off, ids, lists = _random_lists(rng, sizes)
    r = roc.encode(off, ids, want_perm=True)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    _check_against_oracle(oracle, r, off, lists, r.perm(), dec)
    assert r.last_decode_nonclean == 0
    total = sum(8 + 4 * int(w) for w, s in zip(r.info()["nwords"], sizes) if s)
    assert r.compressed_bytes == total  # custom_invlists_impl.cpp:196-206


def test_unsorted_lists_and_decode_lists(roc, oracle):
    """Input order must not matter for the stream; perm maps back to input positions."""
    rng = np.random.default_rng(2)
    sizes = np.concatenate([rng.integers(0, 70, 40), rng.integers(60, 3000, 30), [7000, 0, 33000]])
    off, ids, lists = _random_lists(rng, sizes, sort=False)
    r = roc.encode(off, ids, want_perm=True)
    perm = r.perm()
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    _check_against_oracle(oracle, r, off, lists, perm, dec)
    for l, li in enumerate(lists):  # decoded order == input re-ordered by perm (code re-ordering contract, :188-193)
        a, b = int(off[l]), int(off[l + 1])
        assert np.array_equal(li[perm[a:b]], dec[a:b])
    sel = np.array([5, 0, len(sizes) - 1, 17, 5, 71], dtype=np.uint64)
    d, doff = r.decode_lists(sel)
    d = d.cpu().numpy().view(np.uint64)
    for i, l in enumerate(sel):
        l = int(l)
        assert np.array_equal(d[int(doff[i]):int(doff[i + 1])], dec[int(off[l]):int(off[l + 1])])


def test_lane_encoders_check_what_the_light_prepass_assumes(roc, oracle, monkeypatch):
    """With lane-per-list classes in the call the prepass also looks at the last id of each list only: the lane encoders
    compare every sampled id with its left neighbour and hand anything not strictly ascending (or outside [0, 2^31)) back to
    the wave-per-list kernel.  Streams, precisions and permutation must equal the full prepass's and the oracle's."""
    from vector_db_id_compression_amd import VidcError

    monkeypatch.setenv("VIDC_FORCE_LANE", "1")
    rng = np.random.default_rng(78)
    lists = []
    for k, n in enumerate([66, 100, 256, 257, 700, 1024, 1025, 3000, 4096, 150, 512, 2048]):
        li = np.sort(rng.choice(1 << 22, size=n, replace=False)).astype(np.uint64)
        if k % 3 == 0:  # unsorted, with a small last id: the precision taken from it is too small for the list
            li = rng.permutation(li)
            li[[-1, int(np.argmin(li))]] = li[[int(np.argmin(li)), -1]]
        elif k % 3 == 1:  # one duplicate somewhere in the middle
            li[n // 2] = li[n // 2 - 1]
        lists.append(li)
    for n in (80, 300, 1500):  # and clean ones next to them
        lists.append(np.sort(rng.choice(1 << 22, size=n, replace=False)).astype(np.uint64))
    off = np.concatenate([[0], np.cumsum([li.size for li in lists])]).astype(np.uint64)
    ids = np.concatenate(lists)
    for mode in (-1, 22):
        for want_perm in (False, True):
            monkeypatch.delenv("VIDC_FULL_PREPASS", raising=False)
            r = roc.encode(off, ids, precision_mode=mode, want_perm=want_perm)
            monkeypatch.setenv("VIDC_FULL_PREPASS", "1")
            r2 = roc.encode(off, ids, precision_mode=mode, want_perm=want_perm)
            i1, i2 = r.info(), r2.info()
            for k in ("heads", "nwords", "precision", "mt_draws"):
                assert np.array_equal(i1[k], i2[k]), (mode, want_perm, k)
            assert np.array_equal(r.all_words(), r2.all_words())
            dec = r.decode_all().cpu().numpy().view(np.uint64)
            assert np.array_equal(dec, r2.decode_all().cpu().numpy().view(np.uint64))
            if want_perm:
                assert np.array_equal(r.perm(), r2.perm())
            if mode == -1:
                _check_against_oracle(oracle, r, off, lists, r.perm() if want_perm else None, dec)
    monkeypatch.delenv("VIDC_FULL_PREPASS", raising=False)
    # an id outside [0, 2^31) in the middle of an ascending lane-class list (the last id is fine)
    for n in (200, 900, 3000):
        li = np.sort(rng.choice(1 << 20, size=n, replace=False)).astype(np.uint64)
        for badv in (1 << 31, (1 << 40) + 5):
            bad = li.copy()
            bad[n // 3] = badv
            both = np.concatenate([lists[-1], bad])
            with pytest.raises(VidcError):
                roc.encode(np.array([0, lists[-1].size, both.size], dtype=np.uint64), both)


def test_empty_inputs(roc):
    r = roc.encode(np.array([0], dtype=np.uint64), np.zeros(0, np.uint64))
    assert r.nlist == 0 and r.ntotal == 0 and r.compressed_bytes == 0
    r = roc.encode(np.array([0, 0, 0], dtype=np.uint64), np.zeros(0, np.uint64))
    assert r.nlist == 2 and r.compressed_bytes == 0
    assert r.decode_all().numel() == 0


def test_domain_errors(roc):
    from vector_db_id_compression_amd import VidcError

    with pytest.raises(VidcError):  # reference: int max_id (custom_invlists_impl.cpp:163) -> ids must be < 2^31
        roc.encode(np.array([0, 3], dtype=np.uint64), np.array([1, 2, 1 << 31], dtype=np.uint64))
    with pytest.raises(VidcError):
        roc.encode(np.array([0, 200], dtype=np.uint64), np.arange(200, dtype=np.uint64) + (1 << 40))


def test_light_prepass_assumptions_are_checked_by_the_kernels(roc, oracle, monkeypatch):
    """A call without lane-per-list classes classifies each list by its LAST id only; everything that id does not
    prove is verified by the encode kernels (tools/fuzz_families.py seed 3 found the second case)."""
    from vector_db_id_compression_amd import VidcError

    monkeypatch.setenv("VIDC_NO_LANE", "1")
    rng = np.random.default_rng(77)
    # (a) unsorted long lists whose last id is small: the bitmap class chosen from it cannot hold the list
    lists = []
    for nbits, n in ((19, 4253), (20, 5000), (24, 4500), (17, 4100)):
        li = rng.choice(1 << nbits, size=n, replace=False).astype(np.uint64)
        li[-1] = li.min()  # classification sees a tiny maximum
        lists.append(li)
    lists.append(np.sort(rng.choice(1 << 19, size=6000, replace=False)).astype(np.uint64))
    off = np.concatenate([[0], np.cumsum([li.size for li in lists])]).astype(np.uint64)
    ids = np.concatenate(lists)
    for mode in (-1, 19, 24):
        for want_perm in (False, True):
            monkeypatch.delenv("VIDC_FULL_PREPASS", raising=False)
            r = roc.encode(off, ids, precision_mode=mode, want_perm=want_perm)
            monkeypatch.setenv("VIDC_FULL_PREPASS", "1")
            r2 = roc.encode(off, ids, precision_mode=mode, want_perm=want_perm)
            i1, i2 = r.info(), r2.info()
            for k in ("heads", "nwords", "precision", "mt_draws"):
                assert np.array_equal(i1[k], i2[k]), (mode, want_perm, k)
            assert np.array_equal(r.all_words(), r2.all_words())
            dec = r.decode_all().cpu().numpy().view(np.uint64)
            assert np.array_equal(dec, r2.decode_all().cpu().numpy().view(np.uint64))
            if want_perm:
                assert np.array_equal(r.perm(), r2.perm())
            if mode == -1:
                _check_against_oracle(oracle, r, off, lists, r.perm() if want_perm else None, dec)
    monkeypatch.delenv("VIDC_FULL_PREPASS", raising=False)
    # (b) an id outside [0, 2^31) in the middle of a long ascending list (the last id is fine)
    li = np.sort(rng.choice(1 << 20, size=5000, replace=False)).astype(np.uint64)
    for badv in (1 << 31, (1 << 40) + 5):
        bad = li.copy()
        bad[1234] = badv
        with pytest.raises(VidcError):
            roc.encode(np.array([0, bad.size], dtype=np.uint64), bad)
    # (c) duplicates in a bitmap-class list, with and without the permutation
    dup = li.copy()
    dup[2000] = dup[1999]
    offd = np.array([0, dup.size], dtype=np.uint64)
    for want_perm in (False, True):
        r = roc.encode(offd, dup, want_perm=want_perm)
        dec = r.decode_all().cpu().numpy().view(np.uint64)
        _check_against_oracle(oracle, r, offd, [dup], r.perm() if want_perm else None, dec)


def test_wide_precision_chain_kernels_match_general_kernels_and_oracle(roc, oracle, monkeypatch):
    """Long lists whose ids need more than 20 bits take k_roc_encode_r2 (bitmap over list positions, id by scalar load)
    and k_roc_decode_b2 (4096 value buckets, member rows in memory) when the call holds few of them; VIDC_NO_R2=1 keeps
    them on the general kernels.  Streams, permutations and decoded order must not depend on it; a clustered list
    overflows a 64-member bucket row and comes back through the retry pass."""
    rng = np.random.default_rng(2024)
    lists = [
        np.sort(rng.choice(1 << 27, size=30000, replace=False)),
        np.sort(rng.choice((1 << 31) - 1, size=9000, replace=False)),
        np.sort(np.unique(np.concatenate([rng.choice(1 << 26, size=9000, replace=False),
                                          (5 << 20) + rng.choice(1 << 10, size=600, replace=False)]))),  # 600 ids in one bucket
        rng.choice(1 << 24, size=8000, replace=False),                                                  # unsorted
        np.sort(rng.integers(0, 1 << 23, size=7000)),                                                   # duplicates
        np.sort(rng.choice(1 << 21, size=4100, replace=False)),
        np.sort(rng.choice(1 << 30, size=65536, replace=False)),
    ]
    lists = [li.astype(np.uint64) for li in lists]
    off = np.concatenate([[0], np.cumsum([li.size for li in lists])]).astype(np.uint64)
    ids = np.concatenate(lists)
    for want_perm in (True, False):
        monkeypatch.delenv("VIDC_NO_R2", raising=False)
        r = roc.encode(off, ids, want_perm=want_perm)
        dec = r.decode_all().cpu().numpy().view(np.uint64)
        monkeypatch.setenv("VIDC_NO_R2", "1")
        r2 = roc.encode(off, ids, want_perm=want_perm)
        dec_g = r.decode_all().cpu().numpy().view(np.uint64)   # the chain kernels' streams through the general decoder
        dec2 = r2.decode_all().cpu().numpy().view(np.uint64)
        monkeypatch.delenv("VIDC_NO_R2", raising=False)
        i1, i2 = r.info(), r2.info()
        for k in ("heads", "nwords", "precision", "mt_draws"):
            assert np.array_equal(i1[k], i2[k]), (want_perm, k)
        assert np.array_equal(r.all_words(), r2.all_words())
        assert np.array_equal(dec, dec2) and np.array_equal(dec, dec_g)
        if want_perm:
            assert np.array_equal(r.perm(), r2.perm())
        _check_against_oracle(oracle, r, off, lists, r.perm() if want_perm else None, dec)
        # (round 5: k_roc_decode_b2 launches of 512 chains or more load a bucket's row under exec = lanes below its member count,
        # smaller launches the whole row; both forms forced here, incl. the clustered list's full row and its retry)
        for form in ("0", "1"):
            monkeypatch.setenv("VIDC_B2_MASK", form)
            assert np.array_equal(r.decode_all().cpu().numpy().view(np.uint64), dec), ("VIDC_B2_MASK", form)
        monkeypatch.delenv("VIDC_B2_MASK", raising=False)
        sel = np.array([6, 2, 0], dtype=np.uint64)  # a search's decode_lists takes the same kernels
        got, goff = r.decode_lists(sel)
        got = got.cpu().numpy().view(np.uint64)
        for k, l in enumerate(sel):
            assert np.array_equal(got[int(goff[k]):int(goff[k + 1])], dec[int(off[int(l)]):int(off[int(l) + 1])])


def test_chain_kernels_at_the_largest_list_sizes(roc, monkeypatch):
    """k_roc_encode_r2 takes lists up to 262 144 ids (the position bitmap's 18 bits), k_roc_decode_b2 up to 98 304; the
    reference's codec is lossy beyond 65 536 ids, so the check is against the general kernels' streams and
    their decode of them, not against the input."""
    rng = np.random.default_rng(4242)
    lists = [np.sort(rng.choice(1 << 28, size=n, replace=False)).astype(np.uint64) for n in (262144, 200000, 98304, 98305, 65537)]
    off = np.concatenate([[0], np.cumsum([li.size for li in lists])]).astype(np.uint64)
    ids = np.concatenate(lists)
    r = roc.encode(off, ids, want_perm=True)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    monkeypatch.setenv("VIDC_NO_R2", "1")
    r2 = roc.encode(off, ids, want_perm=True)
    dec2 = r2.decode_all().cpu().numpy().view(np.uint64)
    monkeypatch.delenv("VIDC_NO_R2", raising=False)
    i1, i2 = r.info(), r2.info()
    for k in ("heads", "nwords", "precision", "mt_draws"):
        assert np.array_equal(i1[k], i2[k]), k
    assert np.array_equal(r.all_words(), r2.all_words()) and np.array_equal(r.perm(), r2.perm())
    assert np.array_equal(dec, dec2)


def test_short_list_chain_kernels_of_small_calls(roc, oracle, monkeypatch):
    """Calls with a few hundred lists of 257..4096 ids: encode through k_roc_encode_r2, decode through the bucket loop with
    its rows in LDS (128 / 256 buckets by expected load).  One list packs 700 ids into a single bucket (row overflow ->
    VIDC_ST_RETRY -> general kernel), one is a multiset, one is unsorted; imported streams (no maximum known) take the
    worst-case bucket count.  Everything must equal the general kernels' output and the oracle."""
    rng = np.random.default_rng(515)
    lists = []
    for nbits in (13, 20, 24, 31):
        for n in (257, 300, 1000, 2048, 2049, 3900, 4096):
            n = min(n, (1 << nbits) - 1)
            lists.append(np.sort(rng.choice(1 << nbits, size=n, replace=False)))
    lists.append(np.sort(np.unique(np.concatenate([rng.choice(1 << 24, size=1500, replace=False),
                                                   (3 << 18) + rng.choice(1 << 12, size=700, replace=False)]))))
    lists.append(np.sort(rng.integers(0, 1 << 22, size=3000)))
    lists.append(rng.choice(1 << 20, size=2500, replace=False))
    lists = [li.astype(np.uint64) for li in lists]
    off = np.concatenate([[0], np.cumsum([li.size for li in lists])]).astype(np.uint64)
    ids = np.concatenate(lists)
    r = roc.encode(off, ids, want_perm=True)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    monkeypatch.setenv("VIDC_NO_R2", "1")
    r2 = roc.encode(off, ids, want_perm=True)
    dec2 = r.decode_all().cpu().numpy().view(np.uint64)
    monkeypatch.delenv("VIDC_NO_R2", raising=False)
    assert np.array_equal(r.all_words(), r2.all_words()) and np.array_equal(r.perm(), r2.perm())
    assert np.array_equal(dec, dec2)
    _check_against_oracle(oracle, r, off, lists, r.perm(), dec)
    # the same streams imported (the decoder then knows precisions only)
    info = r.info()
    from vector_db_id_compression_amd.codecs import RocLists

    imp = RocLists.from_streams(off, info["precision"], info["heads"], info["nwords"], r.all_words(), mt_draws=info["mt_draws"])
    assert np.array_equal(imp.decode_all().cpu().numpy().view(np.uint64), dec)


def test_exact_precision_mode_is_lossless_for_pow2_max(roc):
    """VIDC_PREC_EXACT fixes the reference's pow-2 precision quirk (Q3); reference mode reproduces it."""
    ids = np.array([3, 1024, 7, 100], dtype=np.uint64)
    off = np.array([0, 4], dtype=np.uint64)
    r = roc.encode(off, ids, precision_mode=-2)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    assert sorted(dec.tolist()) == [3, 7, 100, 1024]
    r = roc.encode(off, ids, precision_mode=-1)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    assert sorted(dec.tolist()) == [0, 4, 7, 100]


def test_import_streams_then_decode(roc, oracle):
    """Decode-only path: streams produced by the CPU oracle decode to the oracle's order on the GPU."""
    rng = np.random.default_rng(3)
    sizes = [0, 1, 64, 65, 500, 2000, 10000]
    off, ids, lists = _random_lists(rng, sizes, nbits=24)
    encs = [oracle.roc_encode(li, oracle.list_precision(li)) if li.size else None for li in lists]
    prec = [oracle.list_precision(li) if li.size else 0 for li in lists]
    heads = [e["head"] if e else 1 << 31 for e in encs]
    nw = [e["words"].size if e else 0 for e in encs]
    words = np.concatenate([e["words"] for e in encs if e] + [np.zeros(0, np.uint32)])
    r = roc.from_streams(off, prec, heads, nw, words)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    for l, e in enumerate(encs):
        if e:
            assert np.array_equal(dec[int(off[l]):int(off[l + 1])], e["order"])
    assert r.last_decode_nonclean == 0


def test_graph_rows_vs_oracle(roc, oracle):
    """ROCNSGGraph flavour (altid_impl.cpp:103-165): int32 rows, -1 terminated, unsorted neighbours."""
    rng = np.random.default_rng(4)
    N, K = 700, 64
    rows = np.full((N, K), -1, dtype=np.int32)
    for i in range(N):
        d = int(rng.integers(0, K + 1))
        rows[i, :d] = rng.choice(100000, size=d, replace=False)
    rows[5, :] = -1
    rg = roc.encode_rows(rows)
    info = rg.info()
    out, cnt = rg.decode_rows(np.arange(N))
    out = out.cpu().numpy()
    total = 0
    for i in range(N):
        d = int((rows[i] >= 0).sum())
        assert cnt[i] == d and info["sizes"][i] == d
        if d == 0:
            continue
        li = rows[i, :d].astype(np.uint64)
        e = oracle.roc_encode(li, oracle.list_precision(li))
        assert int(info["heads"][i]) == e["head"]
        assert np.array_equal(rg.words(i, int(info["nwords"][i])), e["words"])
        assert np.array_equal(out[i, :d].astype(np.uint64), e["order"])  # first n slots (Q7)
        assert np.all(out[i, d:] == -1)
        total += 8 + 4 * e["words"].size  # altid_impl.cpp:148 sums ans_states[list_no].size() for every node
    assert rg.compressed_bytes == total
    sub, c2 = rg.decode_rows([3, 699, 3])
    assert np.array_equal(sub.cpu().numpy()[0], out[3]) and np.array_equal(sub.cpu().numpy()[2], out[3])


def test_full_size_config2_properties(roc):
    """BASELINE configs[1] at full size: round trip as sets, clean end states, size accounting."""
    from vector_db_id_compression_amd import synth

    w = synth.workload("s1")
    off, ids = w["offsets"], w["ids"]
    r = roc.encode(off, ids, want_perm=True)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    perm = r.perm()
    assert r.last_decode_nonclean == 0
    for l in range(off.size - 1):
        a, b = int(off[l]), int(off[l + 1])
        assert np.array_equal(ids[a:b][perm[a:b]], dec[a:b])  # decode == input re-ordered by the sampling perm
    assert np.array_equal(np.sort(dec), np.arange(ids.size, dtype=np.uint64))  # every id exactly once
    bits = 8.0 * r.compressed_bytes / ids.size
    assert 10.3 < bits < 10.6  # 10.455 bit/id on this shape
    # idempotence: encoding the same input twice gives identical streams
    r2 = roc.encode(off, ids)
    assert np.array_equal(r2.info()["heads"], r.info()["heads"]) and r2.total_words == r.total_words


def test_save_load_roundtrip(roc, tmp_path):
    """Flat image of the compressed lists: reload and decode without re-encoding."""
    rng = np.random.default_rng(9)
    sizes = [0, 5, 64, 300, 5000]
    off, ids, lists = _random_lists(rng, sizes)
    r = roc.encode(off, ids)
    want = r.decode_all().cpu().numpy()
    p = str(tmp_path / "roc.npz")
    r.save(p)
    r2 = roc.load(p)
    assert r2.compressed_bytes == r.compressed_bytes
    assert np.array_equal(r2.decode_all().cpu().numpy(), want)


# ---------------------------------------------------------------------------------------------------------
# lane-per-list kernels (roc_lane.h): lists of 65..4096 strictly ascending ids, tiny lists, graph rows.  The library
# only picks them for calls with thousands of lists; VIDC_FORCE_LANE=1 selects them regardless of the batch size.
@pytest.fixture
def force_lane(monkeypatch):
    monkeypatch.setenv("VIDC_FORCE_LANE", "1")


def test_lane_kernels_boundaries_vs_oracle(roc, oracle, force_lane):
    """Sizes around every class boundary of the lane-per-list kernels, several universes, one call."""
    rng = np.random.default_rng(77)
    sizes = [65, 66, 80, 81, 96, 97, 127, 128, 129, 191, 192, 193, 194, 208, 209, 255, 256, 257, 300, 511, 512, 513, 767, 768, 769, 1000, 1023, 1024, 1025, 1100,
             1279, 1280, 1281, 2047, 2048, 2049, 3071, 3072, 3073, 3840, 4000, 4095, 4096, 4097, 4200]
    for nbits in (11, 16, 20, 24, 31):
        sz = [s for s in sizes if s <= (1 << nbits)]
        off, ids, lists = _random_lists(rng, sz, nbits=nbits)
        r = roc.encode(off, ids, want_perm=True)
        dec = r.decode_all().cpu().numpy().view(np.uint64)
        _check_against_oracle(oracle, r, off, lists, r.perm(), dec)


def test_lane_register_decoder_matches_bucket_decoder(roc, oracle, force_lane, monkeypatch):
    """Lists of 65..256 ids decode with the ids in registers (k_roc_decode_lane_reg: 192 register slots + an LDS strip);
    VIDC_NO_LANE_REG=1 sends them to the bucket-row decoder instead.  Ragged sizes inside one wavefront, several
    universes (dense 9-bit lists pop the stream window fastest), the decoded order must be the same and the oracle's."""
    rng = np.random.default_rng(81)
    for nbits in (9, 13, 24, 31):
        sizes = rng.integers(65, 257, 700)
        sizes[:8] = [256, 256, 193, 192, 191, 65, 255, 129]
        off, ids, lists = _random_lists(rng, sizes, nbits=nbits)
        r = roc.encode(off, ids)
        monkeypatch.delenv("VIDC_NO_LANE_REG", raising=False)
        dec = r.decode_all().cpu().numpy().view(np.uint64)
        assert r.last_decode_nonclean == 0
        monkeypatch.setenv("VIDC_NO_LANE_REG", "1")
        dec2 = r.decode_all().cpu().numpy().view(np.uint64)
        monkeypatch.delenv("VIDC_NO_LANE_REG", raising=False)
        assert np.array_equal(dec, dec2)
        sub = list(range(12)) + [int(v) for v in rng.integers(0, len(lists), 20)]
        for l in sub:
            li = lists[l]
            e = oracle.roc_encode(li, oracle.list_precision(li))
            want = oracle.roc_decode(e["head"], e["words"], li.size, oracle.list_precision(li), e["mt_draws"])[0]
            assert np.array_equal(dec[int(off[l]):int(off[l + 1])], want), (nbits, l)
        got, goff = r.decode_lists(np.array(sub, dtype=np.uint64))
        got = got.cpu().numpy().view(np.uint64)
        for k, l in enumerate(sub):
            assert np.array_equal(got[int(goff[k]):int(goff[k + 1])], dec[int(off[l]):int(off[l + 1])])


def test_lane_kernels_dense_and_small_precision(roc, oracle, force_lane):
    """Dense lists (n close to the universe: tiny precision, many renormalisation pops) and fixed precisions."""
    rng = np.random.default_rng(78)
    lists = [np.sort(rng.choice(m, size=n, replace=False)).astype(np.uint64)
             for n, m in ((65, 66), (100, 101), (128, 200), (300, 301), (1024, 1025), (1000, 1 << 10), (70, 1 << 7),
                          (1025, 1026), (2500, 2501), (4096, 4097), (4096, 5000), (3000, 1 << 12))]
    off = np.concatenate([[0], np.cumsum([li.size for li in lists])]).astype(np.uint64)
    ids = np.concatenate(lists)
    r = roc.encode(off, ids, want_perm=True)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    _check_against_oracle(oracle, r, off, lists, r.perm(), dec)
    for P in (12, 17, 32):  # explicit precision (codec.cpp compress(), precision argument)
        r = roc.encode(off, ids, precision_mode=P, want_perm=True)
        info = r.info()
        dec = r.decode_all().cpu().numpy().view(np.uint64)
        for l, li in enumerate(lists):
            e = oracle.roc_encode(li, P)
            assert int(info["heads"][l]) == e["head"] and np.array_equal(r.words(l), e["words"])
            # the reference decoder's view of the stream (P = 12 is one bit short for the lists holding id 4096: Q3)
            want = oracle.roc_decode(e["head"], e["words"], li.size, P, e["mt_draws"])[0]
            assert np.array_equal(dec[int(off[l]):int(off[l + 1])], want)


def test_lane_decoder_hands_back_skewed_lists(roc, oracle, force_lane):
    """All ids of a list in one 1/64 (1/256) slice of the universe: the lane decoder's bucket row overflows and the
    list is redone by the wave-per-list kernel (VIDC_ST_RETRY) -- same output."""
    rng = np.random.default_rng(79)
    lists = []
    for n in (70, 300, 900, 1500, 3900):
        top = (1 << 20) - 1
        body = np.sort(rng.choice(4000, size=n - 1, replace=False)).astype(np.uint64) + 5
        lists.append(np.concatenate([body, [top]]).astype(np.uint64))  # max id sets the precision, the rest is clustered
    lists.append(np.sort(rng.choice(1 << 20, size=500, replace=False)).astype(np.uint64))  # a clean neighbour
    lists.append(np.sort(rng.choice(1 << 20, size=2000, replace=False)).astype(np.uint64))  # a clean 256-bucket list
    off = np.concatenate([[0], np.cumsum([li.size for li in lists])]).astype(np.uint64)
    ids = np.concatenate(lists)
    r = roc.encode(off, ids, want_perm=True)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    _check_against_oracle(oracle, r, off, lists, r.perm(), dec)
    assert r.last_decode_nonclean == 0
    req = [2, 5, 0, 2, 4, 6, 3]
    sub, sub_off = r.decode_lists(np.array(req, dtype=np.uint64))  # repeated + mixed retry / clean requests
    sub = sub.cpu().numpy().view(np.uint64)
    for i, l in enumerate(req):
        assert np.array_equal(sub[int(sub_off[i]):int(sub_off[i + 1])], dec[int(off[l]):int(off[l + 1])])


def test_lane_decoder_hands_back_more_lists_than_the_summary_lists(roc, force_lane):
    """The decode summary names up to 504 handed-back lists; beyond that the host fetches the statuses: 700 clustered lists
    next to 300 clean ones, every list identical to the decode of the wave-per-list kernels' path."""
    rng = np.random.default_rng(81)
    lists = []
    for k in range(1000):
        if k % 10 < 7:
            body = np.sort(rng.choice(12000, size=599, replace=False)).astype(np.uint64) + 3  # (bucket-row class: one full bucket)
            lists.append(np.concatenate([body, [(1 << 20) - 1 - k]]).astype(np.uint64))
        else:
            lists.append(np.sort(rng.choice(1 << 20, size=620, replace=False)).astype(np.uint64))
    off = np.concatenate([[0], np.cumsum([li.size for li in lists])]).astype(np.uint64)
    ids = np.concatenate(lists)
    r = roc.encode(off, ids)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    assert r.last_decode_nonclean == 0
    for l, li in enumerate(lists):
        assert np.array_equal(np.sort(dec[int(off[l]):int(off[l + 1])]), li), l


def test_lane_kernels_match_wave_kernels(roc, monkeypatch):
    """Same streams from the two kernel families (VIDC_FORCE_LANE / VIDC_NO_LANE test hooks) on 3000 ragged lists,
    and from the automatic choice on a batch large enough to take the lane kernels by itself."""
    rng = np.random.default_rng(80)
    sizes = np.concatenate([rng.integers(0, 1200, 3000), rng.integers(1025, 4200, 300)])
    off, ids, _ = _random_lists(rng, sizes, nbits=22)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VIDC_NO_LANE", mode)
        monkeypatch.setenv("VIDC_FORCE_LANE", "1" if mode == "0" else "0")
        r = roc.encode(off, ids, want_perm=True)
        info = r.info()
        got[mode] = (info["heads"], info["nwords"], info["mt_draws"], r.all_words(), r.perm(),
                     r.decode_all().cpu().numpy().copy())
    for a, b in zip(got["0"], got["1"]):
        assert np.array_equal(a, b)
    # automatic policy: 12000 lists of 65..200 ids and 3000 tiny ones -> lane kernels without any hook
    monkeypatch.delenv("VIDC_FORCE_LANE")
    sizes = np.concatenate([rng.integers(65, 200, 12000), rng.integers(0, 64, 3000)])
    off, ids, _ = _random_lists(rng, sizes, nbits=24)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VIDC_NO_LANE", mode)
        r = roc.encode(off, ids)
        info = r.info()
        got[mode] = (info["heads"], info["nwords"], r.all_words(), r.decode_all().cpu().numpy().copy())
    for a, b in zip(got["0"], got["1"]):
        assert np.array_equal(a, b)
    assert np.array_equal(np.sort(got["0"][3]), np.sort(ids.view(np.int64)))


def test_threaded_host_planning_matches_single_thread(roc, monkeypatch):
    """Calls with >= 131072 lists classify and sort their work lists on several host threads (VIDC_HOST_THREADS=1
    forces one): same streams, same decode, and a sample of lists against the oracle."""
    rng = np.random.default_rng(91)
    sizes = np.concatenate([rng.integers(0, 40, 120000), rng.integers(65, 300, 20000), rng.integers(1025, 2300, 300),
                            [5000, 9000, 40000]])
    rng.shuffle(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    gaps = rng.integers(1, 400, int(off[-1])).astype(np.uint64)  # strictly ascending ids per list: running gap sums
    cs = np.cumsum(gaps)
    first = np.repeat(np.concatenate([[0], cs])[off[:-1].astype(np.int64)], sizes)
    ids = (cs - first).astype(np.uint64)
    got = []
    for threads in ("1", "8"):
        monkeypatch.setenv("VIDC_HOST_THREADS", threads)
        r = roc.encode(off, ids, precision_mode=-2)  # bit_width(max id): lossless also when a max id is a power of two
        info = r.info()
        got.append((info["heads"], info["nwords"], info["precision"], r.all_words(), r.decode_all().cpu().numpy().copy()))
    for a, b in zip(got[0], got[1]):
        assert np.array_equal(a, b)
    assert np.array_equal(np.sort(got[1][4].view(np.uint64)), np.sort(ids))


@pytest.mark.parametrize("nlist", [4096, 8192, 12289])
def test_device_metadata_matches_host_view(roc, nlist):
    """Word offsets, sizes and totals are computed on the device (exclusive scans over a multiple of the scan tile,
    a multiple plus one, ...): they must agree with what the per-list metadata says."""
    rng = np.random.default_rng(nlist)
    sizes = rng.integers(0, 40, nlist)
    sizes[rng.integers(0, nlist, 8)] = rng.integers(65, 300, 8)
    off, ids, lists = _random_lists(rng, sizes, nbits=21)
    r = roc.encode(off, ids)
    info = r.info()
    nw = info["nwords"].astype(np.uint64)
    assert r.total_words == int(nw.sum())
    assert r.compressed_bytes == int((8 + 4 * nw[sizes > 0]).sum())  # custom_invlists_impl.cpp:196-206
    words = r.all_words()
    woff = np.concatenate([[0], np.cumsum(nw)]).astype(np.int64)
    for l in (0, 1, nlist // 2, nlist - 2, nlist - 1):
        assert np.array_equal(r.words(l), words[woff[l]:woff[l + 1]])
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    for l in (0, nlist // 3, nlist - 1):
        assert np.array_equal(np.sort(dec[int(off[l]):int(off[l + 1])]), lists[l])
    assert r.last_decode_nonclean == 0


def test_concurrent_decode_from_two_contexts(roc):
    """One compressed object decoded by two host threads, each with its own context (the reference calls get_ids
    from OpenMP threads, custom_invlists_impl.cpp:467,508): lazy host mirrors and the cached decode plan are shared."""
    import threading

    import torch

    from vector_db_id_compression_amd import _lib

    rng = np.random.default_rng(91)
    sizes = rng.integers(0, 900, 400)
    off, ids, lists = _random_lists(rng, sizes, nbits=21)
    r = roc.encode(off, ids)  # encoded through the default context
    ctxs = [_lib.Context(0), _lib.Context(0)]
    outs = [torch.empty(int(off[-1]), dtype=torch.int64, device="cuda") for _ in ctxs]
    errs = []

    def work(i):
        try:
            for _ in range(5):
                _lib.check(_lib.lib().vidc_roc_decode_all(ctxs[i].h, r.h, _lib.ptr(outs[i])))
                sel = np.array([3, 7, 399, 3], dtype=np.uint64)
                tmp = torch.empty(int(sizes[[3, 7, 399, 3]].sum()) + 1, dtype=torch.int64, device="cuda")
                oo = np.zeros(5, np.uint64)
                _lib.check(_lib.lib().vidc_roc_decode_lists(ctxs[i].h, r.h, 4, _lib.ptr(sel), _lib.ptr(tmp), _lib.ptr(oo)))
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    torch.cuda.synchronize()
    a, b = outs[0].cpu().numpy().view(np.uint64), outs[1].cpu().numpy().view(np.uint64)
    assert np.array_equal(a, b)
    for l in (0, 5, 399):
        assert np.array_equal(np.sort(a[int(off[l]):int(off[l + 1])]), lists[l])
    for c in ctxs:
        c.close()


# ---- row-per-list kernels (roc_grp.h: 16 lanes per list, four lists per wavefront): the automatic policy only picks them for
# calls with thousands of lists above 4096 ids; VIDC_FORCE_GRP=1 selects them for every list of 65 .. 131 072 ids.
@pytest.fixture
def force_grp(monkeypatch):
    monkeypatch.setenv("VIDC_FORCE_GRP", "1")


def test_row_kernels_boundaries_vs_oracle(roc, oracle, force_grp):
    """Sizes around every geometry switch of the row-per-list kernels (512-position blocks, the two- / three-level select at
    8192 ids, bucket bits at 2048 / 8192 / 16 384, row capacity at 32 768 / 65 536), several universes, ragged sizes inside
    one wavefront."""
    rng = np.random.default_rng(177)
    sizes = [65, 66, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 8704, 16383, 16384,
             16385, 20000, 32767, 32768, 32769, 40000, 100, 7000]
    for nbits in (16, 20, 24, 31):
        sz = [s for s in sizes if s <= (1 << nbits)]
        off, ids, lists = _random_lists(rng, sz, nbits=nbits)
        r = roc.encode(off, ids, want_perm=True)
        dec = r.decode_all().cpu().numpy().view(np.uint64)
        _check_against_oracle(oracle, r, off, lists, r.perm(), dec)
        assert r.last_decode_nonclean == 0


def test_row_kernels_longest_lists(roc, oracle, force_grp):
    """65 536 / 65 537 / 131 072-id lists (the top level's last group, the 96-member rows) and a neighbour in the same wavefront."""
    rng = np.random.default_rng(178)
    off, ids, lists = _random_lists(rng, [65536, 131072, 65537, 300], nbits=27)
    r = roc.encode(off, ids, want_perm=True)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    perm = r.perm()
    info = r.info()
    for l in (0, 3):  # (the oracle takes a second per 65 536-id list; lists beyond 65 536 ids are lossy in the reference, Q2)
        li = lists[l]
        e = oracle.roc_encode(li, oracle.list_precision(li))
        a, b = int(off[l]), int(off[l + 1])
        assert int(info["heads"][l]) == e["head"] and np.array_equal(r.words(l), e["words"])
        assert np.array_equal(perm[a:b], e["perm"]) and np.array_equal(dec[a:b], e["order"])


def test_row_kernels_dense_small_precision_and_fixed_precisions(roc, oracle, force_grp):
    """Dense lists (n close to the universe: tiny precision, many renormalisation pops, stream windows that drain fast) and
    explicit precisions above and below what the ids need (carry quirk of the uniform push)."""
    rng = np.random.default_rng(179)
    lists = [np.sort(rng.choice(m, size=n, replace=False)).astype(np.uint64)
             for n, m in ((65, 66), (300, 301), (1024, 1025), (1000, 1 << 10), (4096, 4097), (4096, 5000), (9000, 9001),
                          (9000, 1 << 14), (20000, 20001), (3000, 1 << 12))]
    off = np.concatenate([[0], np.cumsum([li.size for li in lists])]).astype(np.uint64)
    ids = np.concatenate(lists)
    r = roc.encode(off, ids, want_perm=True)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    _check_against_oracle(oracle, r, off, lists, r.perm(), dec)
    for P in (12, 17, 32):
        r = roc.encode(off, ids, precision_mode=P, want_perm=True)
        info = r.info()
        dec = r.decode_all().cpu().numpy().view(np.uint64)
        for l, li in enumerate(lists):
            e = oracle.roc_encode(li, P)
            assert int(info["heads"][l]) == e["head"] and np.array_equal(r.words(l), e["words"]), (P, l)
            want = oracle.roc_decode(e["head"], e["words"], li.size, P, e["mt_draws"])[0]
            assert np.array_equal(dec[int(off[l]):int(off[l + 1])], want), (P, l)


def test_row_kernels_hand_back_what_they_cannot_do(roc, oracle, force_grp):
    """Unsorted and duplicate-holding lists (the encoder samples POSITIONS of an ascending list: it must notice and hand the
    list to the sorting pass) and clustered ids (a full bucket row in the decoder -> VIDC_ST_RETRY): same bits as the oracle."""
    rng = np.random.default_rng(180)
    lists = []
    for n in (700, 5000, 9000):
        li = rng.choice(1 << 22, size=n, replace=False).astype(np.uint64)  # unsorted
        lists.append(li)
    dup = np.sort(rng.choice(1 << 22, size=6000, replace=False)).astype(np.uint64)
    dup[3000] = dup[2999]  # one duplicate: not strictly ascending
    lists.append(dup)
    lists.append(np.sort(rng.choice(1 << 22, size=6000, replace=False)).astype(np.uint64))  # a clean neighbour
    for n in (900, 5000, 12000):  # clustered: all but the maximum inside a sliver of the universe
        body = np.sort(rng.choice(20000, size=n - 1, replace=False)).astype(np.uint64) + 5
        lists.append(np.concatenate([body, [(1 << 22) - 1]]).astype(np.uint64))
    off = np.concatenate([[0], np.cumsum([li.size for li in lists])]).astype(np.uint64)
    ids = np.concatenate(lists)
    r = roc.encode(off, ids, want_perm=True)
    dec = r.decode_all().cpu().numpy().view(np.uint64)
    perm = r.perm()
    info = r.info()
    for l, li in enumerate(lists):
        a, b = int(off[l]), int(off[l + 1])
        srt = np.sort(li)
        e = oracle.roc_encode(srt, oracle.list_precision(srt))
        assert int(info["heads"][l]) == e["head"] and np.array_equal(r.words(l), e["words"]), l
        assert np.array_equal(li[perm[a:b]], e["order"]), l   # positions in the caller's (unsorted) order
        want = oracle.roc_decode(e["head"], e["words"], li.size, oracle.list_precision(srt), e["mt_draws"])[0]
        assert np.array_equal(dec[a:b], want), l
    req = [5, 0, 4, 7, 5]
    sub, sub_off = r.decode_lists(np.array(req, dtype=np.uint64))
    sub = sub.cpu().numpy().view(np.uint64)
    for i, l in enumerate(req):
        assert np.array_equal(sub[int(sub_off[i]):int(sub_off[i + 1])], dec[int(off[l]):int(off[l + 1])])


def test_row_kernels_match_wave_kernels_and_the_automatic_policy(roc, monkeypatch):
    """Same streams, permutations and decoded arrays from the row-per-list family and from the wave-per-list / lane families on
    ragged lists; then a call large enough (10 000 lists above 4096 ids) to take the row kernels by itself, against VIDC_NO_GRP=1."""
    rng = np.random.default_rng(181)
    sizes = np.concatenate([rng.integers(0, 9000, 400), rng.integers(8000, 40000, 40)])
    off, ids, _ = _random_lists(rng, sizes, nbits=26)
    got = {}
    for mode in ("grp", "nogrp"):
        monkeypatch.setenv("VIDC_FORCE_GRP", "1" if mode == "grp" else "0")
        monkeypatch.setenv("VIDC_NO_GRP", "0" if mode == "grp" else "1")
        r = roc.encode(off, ids, want_perm=True)
        info = r.info()
        got[mode] = (info["heads"], info["nwords"], info["mt_draws"], r.all_words(), r.perm(), r.decode_all().cpu().numpy().copy())
    for a, b in zip(got["grp"], got["nogrp"]):
        assert np.array_equal(a, b)
    monkeypatch.delenv("VIDC_FORCE_GRP")
    sizes = rng.integers(4097, 6000, 10000)
    off, ids, _ = _random_lists(rng, sizes, nbits=27)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VIDC_NO_GRP", mode)
        r = roc.encode(off, ids, want_perm=True)
        info = r.info()
        got[mode] = (info["heads"], info["nwords"], r.all_words(), r.perm(), r.decode_all().cpu().numpy().copy())
        assert r.last_decode_nonclean == 0
    for a, b in zip(got["0"], got["1"]):
        assert np.array_equal(a, b)


def test_wide_stream_context_takes_the_row_kernels_by_itself(roc, monkeypatch):
    """A context created with GPU_MAX_HW_QUEUES >= 8 (here: the VIDC_WIDE_STREAMS test hook) runs every kernel class of a call
    on its own stream and picks the row-per-list kernels for calls with thousands of lists above 4096 ids; same bits as a
    plain context with the row kernels switched off."""
    from vector_db_id_compression_amd import _lib

    rng = np.random.default_rng(182)
    sizes = np.concatenate([rng.integers(4097, 7000, 9000), rng.integers(0, 3000, 2000), [40000, 33000, 65536]])
    off, ids, _ = _random_lists(rng, sizes, nbits=28)
    monkeypatch.setenv("VIDC_WIDE_STREAMS", "1")
    wide = _lib.Context(0)
    monkeypatch.delenv("VIDC_WIDE_STREAMS")
    got = {}
    for mode in ("wide", "plain"):
        monkeypatch.setenv("VIDC_NO_GRP", "0" if mode == "wide" else "1")
        r = roc.encode(off, ids, want_perm=True, ctx=wide if mode == "wide" else None)
        info = r.info()
        got[mode] = (info["heads"], info["nwords"], r.all_words(), r.perm(), r.decode_all().cpu().numpy().copy())
        assert r.last_decode_nonclean == 0
    for a, b in zip(got["wide"], got["plain"]):
        assert np.array_equal(a, b)
    wide.close()


def test_lane_pair_decoder_matches_bucket_decoder(roc, oracle, force_lane, monkeypatch):
    """Lists of 257..512 ids decode on a PAIR of lanes with the ids in registers (k_roc_decode_lane_reg<192, true>: slot i >> 1 of
    the lane with the parity of step i, rank = sum of the two lanes); VIDC_NO_LANE_PAIR=1 sends them to the bucket-row decoder.
    Ragged sizes inside one wavefront, every size boundary (257, 383 / 384 / 385: the register / LDS slot switch at slot 192,
    511 / 512), dense lists that drain the stream window fastest: same decoded order, and the oracle's."""
    rng = np.random.default_rng(91)
    for nbits in (10, 13, 24, 31):
        sizes = rng.integers(257, 513, 500)
        sizes[:12] = [512, 512, 511, 385, 384, 383, 257, 258, 300, 448, 449, 512]
        sizes = np.minimum(sizes, (1 << nbits) - 1)
        off, ids, lists = _random_lists(rng, sizes, nbits=nbits)
        r = roc.encode(off, ids)
        monkeypatch.delenv("VIDC_NO_LANE_PAIR", raising=False)
        dec = r.decode_all().cpu().numpy().view(np.uint64)
        assert r.last_decode_nonclean == 0
        monkeypatch.setenv("VIDC_NO_LANE_PAIR", "1")
        dec2 = r.decode_all().cpu().numpy().view(np.uint64)
        monkeypatch.delenv("VIDC_NO_LANE_PAIR", raising=False)
        assert np.array_equal(dec, dec2)
        sub = list(range(14)) + [int(v) for v in rng.integers(0, len(lists), 16)]
        for l in sub:
            li = lists[l]
            e = oracle.roc_encode(li, oracle.list_precision(li))
            want = oracle.roc_decode(e["head"], e["words"], li.size, oracle.list_precision(li), e["mt_draws"])[0]
            assert np.array_equal(dec[int(off[l]):int(off[l + 1])], want), (nbits, l)
        got, goff = r.decode_lists(np.array(sub, dtype=np.uint64))
        got = got.cpu().numpy().view(np.uint64)
        for k, l in enumerate(sub):
            assert np.array_equal(got[int(goff[k]):int(goff[k + 1])], dec[int(off[l]):int(off[l + 1])])


def test_lane_quad_decoder_matches_bucket_decoder(roc, oracle, force_lane, monkeypatch):
    """Lists of 513..1024 ids decode on a QUAD of lanes with the ids in registers (k_roc_decode_lane_reg<192, 4>: slot i >> 2 of lane
    i & 3, rank = sum of the four lanes; opt-in through VIDC_LANE_QUAD=1), by default on the bucket-row decoder.  Ragged sizes inside one
    wavefront, the size boundaries (513, 767 / 768 / 769: the register / LDS slot switch at slot 192, 1023 / 1024), dense lists."""
    rng = np.random.default_rng(92)
    for nbits in (11, 14, 24, 31):
        sizes = rng.integers(513, 1025, 300)
        sizes[:12] = [1024, 1024, 1023, 769, 768, 767, 513, 514, 600, 896, 897, 1024]
        sizes = np.minimum(sizes, (1 << nbits) - 1)
        off, ids, lists = _random_lists(rng, sizes, nbits=nbits)
        r = roc.encode(off, ids)
        monkeypatch.setenv("VIDC_LANE_QUAD", "1")  # (opt-in: measured slower than the bucket rows, see roc.hip DecEnv)
        dec = r.decode_all().cpu().numpy().view(np.uint64)
        assert r.last_decode_nonclean == 0
        monkeypatch.delenv("VIDC_LANE_QUAD")
        dec2 = r.decode_all().cpu().numpy().view(np.uint64)
        assert np.array_equal(dec, dec2)
        monkeypatch.setenv("VIDC_LANE_QUAD", "1")
        sub = list(range(14)) + [int(v) for v in rng.integers(0, len(lists), 10)]
        for l in sub:
            li = lists[l]
            e = oracle.roc_encode(li, oracle.list_precision(li))
            want = oracle.roc_decode(e["head"], e["words"], li.size, oracle.list_precision(li), e["mt_draws"])[0]
            assert np.array_equal(dec[int(off[l]):int(off[l + 1])], want), (nbits, l)
        got, goff = r.decode_lists(np.array(sub, dtype=np.uint64))
        got = got.cpu().numpy().view(np.uint64)
        for k, l in enumerate(sub):
            assert np.array_equal(got[int(goff[k]):int(goff[k + 1])], dec[int(off[l]):int(off[l + 1])])
