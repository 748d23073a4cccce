"""Deterministic input recipes shared by tests/golden/make_golden.py and the parity tests.

Inputs only -- every expected output lives in tests/golden/roc_golden.json and was produced
by the reference codec (see make_golden.py).
"""
import numpy as np

MASK64 = (1 << 64) - 1


def splitmix64(seed, count):
    """Vectorised splitmix64 stream (uint64)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, count + 1, dtype=np.uint64)
        x = np.uint64(seed & MASK64) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def mt19937_raw(seed, count):
    """std::mt19937(seed) output words (init_genrand seeding)."""
    bg = np.random.MT19937()
    bg._legacy_seeding(int(seed))
    return bg.random_raw(count).astype(np.uint64)


def first_unique(stream, n):
    """First n distinct values of a stream, in order of first appearance."""
    vals, idx = np.unique(stream, return_index=True)
    order = np.sort(idx)[:n]
    assert order.size == n, "stream too short for n distinct values"
    return stream[order]


def distinct_uniform(seed, n, nbits):
    """n distinct values in [0, 2^nbits), pseudo-random order."""
    if n == 0:
        return np.zeros(0, dtype=np.uint64)
    assert n <= (1 << nbits)
    if n * 2 > (1 << nbits):  # dense: permute the whole universe
        keys = splitmix64(seed, 1 << nbits)
        return np.argsort(keys, kind="stable")[:n].astype(np.uint64)
    need = n
    while True:
        s = splitmix64(seed, need * 2 + 64) & np.uint64((1 << nbits) - 1)
        try:
            return first_unique(s, n)
        except AssertionError:
            need *= 2


def fnv1a64(data: bytes) -> str:
    h = 0xCBF29CE484222325
    for b in data:
        h ^= b
        h = (h * 0x100000001B3) & MASK64
    return f"{h:016x}"


def _fnv_fast(arr_u8: np.ndarray) -> str:
    # plain loop in python is slow for big buffers: process with python ints but bytes object iteration is C-fast enough
    return fnv1a64(arr_u8.tobytes())


def fnv_stream(head, words) -> str:
    """FNV-1a-64 over head (8 bytes LE) followed by each stack word (4 bytes LE) -- SURVEY KAT3 convention."""
    w = np.ascontiguousarray(words, dtype="<u4")
    return fnv1a64(int(head).to_bytes(8, "little") + w.tobytes())


def fnv_u64(values) -> str:
    return fnv1a64(np.ascontiguousarray(values, dtype="<u8").tobytes())


def make_ids(case):
    g = case["gen"]
    kind = g["kind"]
    if kind == "literal":
        return np.array(g["ids"], dtype=np.uint64)
    if kind == "test_codec":
        # custom_invlist_cpp/test_codec.cpp:60-82: mt19937(seed), x = mt() & mask, reject repeats
        n, nbits, seed = g["n"], g["nbits"], g["seed"]
        s = mt19937_raw(seed, n * 2 + 1024) & np.uint64((1 << nbits) - 1)
        return first_unique(s, n)
    if kind == "uniform":
        ids = distinct_uniform(g["seed"], g["n"], g["nbits"])
        if g.get("sorted"):
            ids = np.sort(ids)
        return ids
    if kind == "uniform_max":
        # distinct uniform ids below `max_id`, with max_id itself forced in (pow-2 quirk Q3 etc.)
        n, max_id = g["n"], g["max_id"]
        nbits = max(1, int(max_id - 1).bit_length()) if max_id > 1 else 1
        pool = distinct_uniform(g["seed"], min(1 << nbits, n * 2 + 8), nbits)
        pool = pool[pool < np.uint64(max_id)][: n - 1]
        assert pool.size == n - 1
        ids = np.concatenate([pool, np.array([max_id], dtype=np.uint64)])
        k = int(splitmix64(g["seed"] + 99, 1)[0] % np.uint64(n))
        ids[[k, n - 1]] = ids[[n - 1, k]]
        return ids
    if kind == "clustered":
        # ids packed in a narrow window near the top of a wide universe (skewed buckets)
        n, nbits, seed = g["n"], g["nbits"], g["seed"]
        width_bits = g["width_bits"]
        base = np.uint64((1 << nbits) - (1 << width_bits) - 7)
        return base + distinct_uniform(seed, n, width_bits)
    if kind == "two_clusters":
        n, nbits, seed = g["n"], g["nbits"], g["seed"]
        a = distinct_uniform(seed, n // 2, g["width_bits"])
        b = np.uint64((1 << nbits) - (1 << g["width_bits"]) - 3) + distinct_uniform(seed + 1, n - n // 2, g["width_bits"])
        ids = np.concatenate([a, b])
        keys = splitmix64(seed + 2, ids.size)
        return ids[np.argsort(keys, kind="stable")]
    if kind == "dups":
        # duplicates are outside the reference's intended domain but its tuple ordering is deterministic
        n, nbits, seed = g["n"], g["nbits"], g["seed"]
        return splitmix64(seed, n) & np.uint64((1 << nbits) - 1)
    if kind == "range":
        return np.arange(g["start"], g["start"] + g["n"], dtype=np.uint64)
    raise ValueError(kind)


def _c(name, gen, **kw):
    d = dict(name=name, gen=gen)
    d.update(kw)
    return d


CASES = [
    # SURVEY 8a known-answer vectors (test_codec.cpp:26-29 inputs, and a tiny one)
    _c("kat1_test_codec_main_xx", dict(kind="literal", ids=[12351235, 49024902, 17781778, 36663666]), precision=26,
       also_compress=True),
    _c("kat2_tiny_p4", dict(kind="literal", ids=[5, 0, 3, 9, 7, 1]), precision=4, also_compress=True),
    # test_codec.cpp:54-105 generator, explicit precision 20 like the test
    _c("kat3_test_codec_seed0", dict(kind="test_codec", n=65000, nbits=20, seed=0), precision=20),
    _c("kat3_test_codec_seed1", dict(kind="test_codec", n=65000, nbits=20, seed=1), precision=20),
    _c("kat3_test_codec_seed2", dict(kind="test_codec", n=65000, nbits=20, seed=2), precision=20),
    # degenerate / quirk cases
    _c("single_zero", dict(kind="literal", ids=[0])),                      # max_id = 0 -> log2(0) path, P = 0
    _c("single_one", dict(kind="literal", ids=[1])),                       # max_id = 1 -> P = 0, carry
    _c("pair_0_1_dense_mt_draw", dict(kind="literal", ids=[0, 1])),        # encoder-side mt19937 draw (Q5)
    _c("pair_1_0", dict(kind="literal", ids=[1, 0])),
    _c("single_big", dict(kind="literal", ids=[2147483647])),
    _c("q3_pow2_max_1024", dict(kind="literal", ids=[3, 1024, 7, 100])),   # Q3: lossy, decodes {4,0,7,100}
    _c("q3_pow2_max_65536", dict(kind="uniform_max", n=50, max_id=65536, seed=5)),
    _c("q3_pow2_max_2p20", dict(kind="uniform_max", n=90, max_id=1 << 20, seed=6)),
    _c("q3_pow2_max_2p30", dict(kind="uniform_max", n=33, max_id=1 << 30, seed=7)),
    _c("dense_all_of_2p6", dict(kind="uniform", n=64, nbits=6, seed=8)),
    _c("dense_all_of_2p3", dict(kind="range", start=0, n=8)),
    _c("dense_most_of_2p10", dict(kind="uniform", n=1000, nbits=10, seed=9)),
    _c("dups_small", dict(kind="dups", n=40, nbits=4, seed=10)),
    _c("dups_mid", dict(kind="dups", n=3000, nbits=10, seed=11)),
    # wave-boundary sizes
    _c("n63_p20", dict(kind="uniform", n=63, nbits=20, seed=12)),
    _c("n64_p20", dict(kind="uniform", n=64, nbits=20, seed=13)),
    _c("n65_p20", dict(kind="uniform", n=65, nbits=20, seed=14)),
    _c("n2_p31", dict(kind="uniform", n=2, nbits=31, seed=15)),
    _c("n3_p1", dict(kind="literal", ids=[1, 0])),
    _c("n17_p5", dict(kind="uniform", n=17, nbits=5, seed=16)),
    _c("n100_p7", dict(kind="uniform", n=100, nbits=7, seed=17)),
    _c("n1000_p20_sorted", dict(kind="uniform", n=1000, nbits=20, seed=18, sorted=True), also_compress=False),
    _c("n1000_p20", dict(kind="uniform", n=1000, nbits=20, seed=18)),
    _c("n1000_p30", dict(kind="uniform", n=1000, nbits=30, seed=19)),
    _c("n1000_p31", dict(kind="uniform", n=1000, nbits=31, seed=20)),
    _c("n4095_p13", dict(kind="uniform", n=4095, nbits=13, seed=21)),
    _c("n4096_p24", dict(kind="uniform", n=4096, nbits=24, seed=22)),
    _c("n4097_p17", dict(kind="uniform", n=4097, nbits=17, seed=23)),
    _c("n10000_p16", dict(kind="uniform", n=10000, nbits=16, seed=24)),
    _c("n20000_p28", dict(kind="uniform", n=20000, nbits=28, seed=25)),
    _c("n40000_p20", dict(kind="uniform", n=40000, nbits=20, seed=26)),
    _c("n65536_p20", dict(kind="uniform", n=65536, nbits=20, seed=27)),    # largest lossless n (Q2)
    _c("n65536_p16_dense", dict(kind="uniform", n=65536, nbits=16, seed=28)),
    # skewed value distributions (stress the decoder's bucketed rank structure)
    _c("clustered_n3000_p30_w12", dict(kind="clustered", n=3000, nbits=30, width_bits=12, seed=29)),
    _c("clustered_n20000_p31_w15", dict(kind="clustered", n=20000, nbits=31, width_bits=15, seed=30)),
    _c("two_clusters_n5000_p28", dict(kind="two_clusters", n=5000, nbits=28, width_bits=13, seed=31)),
    _c("consecutive_n5000", dict(kind="range", start=123456, n=5000)),
    # Q2: n > 65536 is lossy in the reference; the encoder stream and the reference's own
    # (wrong) decode are still deterministic and pinned here
    _c("q2_n65537_p20", dict(kind="uniform", n=65537, nbits=20, seed=32)),
    _c("q2_n70000_p20", dict(kind="uniform", n=70000, nbits=20, seed=33)),
    _c("q2_n140000_p24", dict(kind="uniform", n=140000, nbits=24, seed=34)),
]
