"""Synthetic inverted-list workloads: Zipf-sized lists over the id range 0..N-1.

S1  N = 1 000 000 ids, L = 1024 lists, Zipf s = 0.75   (BASELINE.json configs[1]; parity + timing)
S2  N = 1e9 ids,       L = 2^20 lists, Zipf s = 0.75, list sizes capped at 65 536 (roofline run)
S3  N = 1e6 nodes x K = 64 int32 neighbours, degree ~ U[32, 64], -1 padded (graph shape)

Every id 0..N-1 is assigned to exactly one list (like IVF add); each list holds its ids in
ascending order (Faiss add order).  Generation is deterministic for a given (N, L, s, seed).
"""
import numpy as np


def zipf_sizes(N, L, s, cap=None):
    """sz[i] = floor(N*w_i/sum w), w_i = (i+1)^-s; remainder handed out round-robin from list 0.
    With `cap`, sizes are clipped and the excess is redistributed over the uncapped lists."""
    w = (np.arange(1, L + 1, dtype=np.float64)) ** (-float(s))
    sz = np.floor(N * w / w.sum()).astype(np.int64)
    if cap is not None:
        for _ in range(64):
            over = sz > cap
            excess = int((sz[over] - cap).sum())
            sz[over] = cap
            if excess == 0:
                break
            free = ~over & (sz < cap)
            if not free.any():
                break
            wf = w * free
            add = np.floor(excess * wf / wf.sum()).astype(np.int64)
            sz += add
    rem = int(N - sz.sum())
    i = 0
    while rem > 0:
        if cap is None or sz[i % L] < cap:
            sz[i % L] += 1
            rem -= 1
        i += 1
    assert sz.sum() == N and (cap is None or sz.max() <= cap)
    return sz


def make_lists_numpy(N, L, s, seed=42, cap=None):
    """-> (offsets uint64[L+1], ids uint64[N]) on the host."""
    sz = zipf_sizes(N, L, s, cap)
    rng = np.random.Generator(np.random.PCG64(seed))
    assign = np.repeat(np.arange(L, dtype=np.int64), sz)
    rng.shuffle(assign)
    order = np.argsort(assign, kind="stable")  # ids of list 0 ascending, then list 1, ...
    offsets = np.concatenate([[0], np.cumsum(sz)]).astype(np.uint64)
    return offsets, order.astype(np.uint64)


def make_lists_torch(N, L, s, seed=42, cap=None, device="cuda"):
    """Same construction on the GPU (for S2-sized inputs). -> (offsets uint64 numpy, ids int64 CUDA tensor)."""
    import torch

    sz = zipf_sizes(N, L, s, cap)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    perm = torch.randperm(N, generator=g, device=device)  # perm[j] = id placed at slot j
    offsets = np.concatenate([[0], np.cumsum(sz)]).astype(np.uint64)
    # slot j belongs to list searchsorted(offsets, j): sort ids inside each list via one keyed sort
    bounds = torch.from_numpy(offsets[1:].astype(np.int64)).to(device)
    chunk = 1 << 26
    out = torch.empty(N, dtype=torch.int64, device=device)
    ends = np.cumsum(sz)
    start = 0
    while start < N:
        target = start + chunk
        if target >= N:
            end = N
        else:  # cut on a list boundary: the last list end <= target (lists are far shorter than a chunk)
            j = int(np.searchsorted(ends, target, side="right"))
            end = int(ends[j - 1]) if j > 0 and ends[j - 1] > start else int(ends[j])
        seg = perm[start:end]
        slot = torch.arange(start, end, device=device)
        lists = torch.searchsorted(bounds, slot, right=True)
        key = lists * (1 << 40) + seg  # N < 2^40
        out[start:end] = torch.sort(key).values & ((1 << 40) - 1)
        start = end
    return offsets, out


def workload(name, seed=42, device="cuda"):
    """Named workloads -> dict(name, offsets (numpy uint64), ids (numpy uint64 or CUDA int64 tensor), describe)."""
    if name == "s1":
        off, ids = make_lists_numpy(1_000_000, 1024, 0.75, seed)
        desc = "S1: 1M uint64 ids in 1024 Zipf(s=0.75) inverted lists"
    elif name == "s1_uniform":
        off, ids = make_lists_numpy(1_000_000, 1024, 0.0, seed)
        desc = "1M uint64 ids in 1024 equal-sized inverted lists"
    elif name == "s1_zipf1":
        off, ids = make_lists_numpy(1_000_000, 1024, 1.0, seed)
        desc = "1M uint64 ids in 1024 Zipf(s=1.0) inverted lists (two lists > 65536: lossy in the reference)"
    elif name == "c5":
        off, ids = make_lists_numpy(10_000_000, 65536, 0.75, seed, cap=65536)
        desc = "10M ids in 65536 Zipf(s=0.75) inverted lists (bigann10M IVF65k shape)"
    elif name == "uniform_16m":
        off, ids = make_lists_torch(1 << 24, 1 << 16, 0.0, seed, device=device)
        desc = "16M ids in 65536 equal-sized lists (256 ids each): throughput regime, no long chain"
    elif name == "uniform_64m_1k":
        off, ids = make_lists_torch(1 << 26, 1 << 16, 0.0, seed, device=device)
        desc = "64M ids in 65536 equal-sized lists (1024 ids each): throughput regime"
    elif name == "s2_64m":
        off, ids = make_lists_torch(1 << 26, 1 << 16, 0.75, seed, cap=65536, device=device)
        desc = "64M ids in 65536 Zipf(s=0.75) lists capped at 65536"
    elif name == "s2":
        off, ids = make_lists_torch(1_000_000_000, 1 << 20, 0.75, seed + 1, cap=65536, device=device)
        desc = "S2: 1B ids in 2^20 Zipf(s=0.75) lists capped at 65536"
    elif name.startswith("uniform:"):  # "uniform:<nlist>:<ids per list>" (dev sweeps)
        _, nl, per = name.split(":")
        off, ids = make_lists_torch(int(nl) * int(per), int(nl), 0.0, seed, device=device)
        desc = f"{int(nl) * int(per)} ids in {nl} equal-sized lists ({per} ids each)"
    else:
        raise ValueError(name)
    sizes = (off[1:] - off[:-1]).astype(np.int64)
    return dict(name=name, offsets=off, ids=ids, describe=desc, nlist=int(off.size - 1), ntotal=int(off[-1]),
                max_list=int(sizes.max()), median_list=int(np.median(sizes)))


def make_graph_rows(N=1_000_000, K=64, seed=44, dmin=32):
    """S3: int32 [N, K] neighbour rows, degree ~ U[dmin, K], distinct uniform ids, -1 padded (host numpy)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    deg = rng.integers(dmin, K + 1, size=N)
    rows = rng.integers(0, N, size=(N, K), dtype=np.int64)
    # make rows distinct cheaply: sort, bump duplicates, unsort by a random permutation key
    rows.sort(axis=1)
    dup = np.zeros_like(rows, dtype=bool)
    dup[:, 1:] = rows[:, 1:] == rows[:, :-1]
    while dup.any():
        rows[dup] = rng.integers(0, N, size=int(dup.sum()))
        rows.sort(axis=1)
        dup[:, 1:] = rows[:, 1:] == rows[:, :-1]
        dup[:, 0] = False
    keys = rng.random((N, K))
    rows = np.take_along_axis(rows, np.argsort(keys, axis=1), axis=1)
    mask = np.arange(K)[None, :] >= deg[:, None]
    rows[mask] = -1
    return rows.astype(np.int32)
