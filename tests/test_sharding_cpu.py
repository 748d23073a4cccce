"""CPU (gloo, world_size 2): list sharding + gather of decoded ids, with the CPU oracle standing in for the GPU codec."""
import os
import socket

import numpy as np
import pytest


def test_lpt_partition_balances_zipf_sizes():
    from vector_db_id_compression_amd import synth
    from vector_db_id_compression_amd.sharding import lpt_partition

    sz = synth.zipf_sizes(1_000_000, 1024, 0.75)
    for world in (1, 2, 4, 8):
        owner = lpt_partition(sz, world)
        load = np.array([sz[owner == r].sum() for r in range(world)])
        assert load.sum() == 1_000_000
        # the largest list (52114) bounds the imbalance
        assert load.max() - load.min() <= sz.max()
        assert load.max() <= 1_000_000 / world + sz.max()


class _OracleCodec:
    """decode_lists contract of RocLists, on the CPU oracle (test stand-in only)."""

    def __init__(self, offsets, ids):
        from oracle.pyoracle import Oracle

        self.o = Oracle()
        self.offsets = offsets
        self.streams = []
        for l in range(offsets.size - 1):
            li = ids[int(offsets[l]):int(offsets[l + 1])]
            P = self.o.list_precision(li) if li.size else 0
            e = self.o.roc_encode(li, P) if li.size else None
            self.streams.append((P, e, li.size))

    def decode_lists(self, local_nos):
        import torch

        parts, off = [], [0]
        for l in np.asarray(local_nos, dtype=np.int64):
            P, e, n = self.streams[int(l)]
            dec = self.o.roc_decode(e["head"], e["words"], n, P, e["mt_draws"])[0] if n else np.zeros(0, np.uint64)
            parts.append(dec.astype(np.int64))
            off.append(off[-1] + n)
        cat = np.concatenate(parts) if parts else np.zeros(0, np.int64)
        return torch.from_numpy(cat), np.array(off, dtype=np.uint64)


def _worker(rank, world, port, offsets, ids, req, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vector_db_id_compression_amd.sharding import ShardedInvLists

    sh = ShardedInvLists(offsets, ids, rank, world, _OracleCodec, device="cpu")
    out, off = sh.gather_ids(req, dst=0)
    if rank == 0:
        q.put((out.numpy(), off, sh.load))
    dist.barrier()
    dist.destroy_process_group()


def _worker_first_op(rank, world, port, offsets, ids, req, q):
    """The gather is the FIRST thing this process group is asked to do: the class itself must put a collective in front of the
    batched send / recv (undefined as a group's first call), on every rank, senders included."""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    real_ar, real_b = dist.all_reduce, dist.batch_isend_irecv

    def ar(*a, **k):
        calls.append("all_reduce")
        return real_ar(*a, **k)

    def b(ops):
        calls.append("batch:" + ",".join(o.op.__name__ for o in ops))
        return real_b(ops)

    dist.all_reduce, dist.batch_isend_irecv = ar, b
    from vector_db_id_compression_amd.sharding import ShardedInvLists

    sh = ShardedInvLists(offsets, ids, rank, world, _OracleCodec, device="cpu")
    out, off = sh.gather_ids(req, dst=0)
    out2, _ = sh.gather_ids(req, dst=0)  # (the second gather needs no collective)
    q.put((rank, calls, None if out is None else bool((out == out2).all())))
    dist.all_reduce, dist.batch_isend_irecv = real_ar, real_b
    dist.barrier()
    dist.destroy_process_group()


def test_gather_as_a_groups_first_operation_is_preceded_by_a_collective_on_every_rank():
    import torch.multiprocessing as mp

    from vector_db_id_compression_amd import synth

    offsets, ids = synth.make_lists_numpy(3000, 16, 0.75, seed=6)
    req = np.array([1, 0, 15, 7, 8, 2], dtype=np.int64)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_first_op, args=(r, 2, port, offsets, ids, req, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, calls, same in got:
        assert calls[0] == "all_reduce" and calls.count("all_reduce") == 1, (rank, calls)
        assert all(c.startswith("batch:") for c in calls[1:]) and len(calls) == 3, (rank, calls)
        # both sides use the batched form: receives on the destination, one send on the other rank
        assert calls[1] == ("batch:irecv" if rank == 0 else "batch:isend"), (rank, calls)
        if rank == 0:
            assert same is True


def test_two_rank_gather_matches_single_process_decode():
    import torch.multiprocessing as mp

    from oracle.pyoracle import Oracle
    from vector_db_id_compression_amd import synth

    offsets, ids = synth.make_lists_numpy(6000, 24, 0.75, seed=5)
    req = np.array([3, 0, 23, 7, 3, 11, 12], dtype=np.int64)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, offsets, ids, req, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, off, load = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o = Oracle()
    for i, l in enumerate(req):
        li = ids[int(offsets[l]):int(offsets[l + 1])]
        e = o.roc_encode(li, o.list_precision(li))
        assert np.array_equal(out[int(off[i]):int(off[i + 1])].astype(np.uint64), e["order"])
    assert load.sum() == 6000 and abs(int(load[0]) - int(load[1])) <= int((offsets[1:] - offsets[:-1]).max())


def test_shard_cut_converts_narrow_id_arrays_instead_of_reinterpreting_them():
    """A host id array that is not 8 bytes wide (int32 ids, as graph code often holds them) must reach the codec as the same
    VALUES in uint64 -- a view would halve the length and encode garbage."""
    from vector_db_id_compression_amd.sharding import ShardedInvLists

    rng = np.random.default_rng(4)
    sizes = rng.integers(0, 40, 50)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ids64 = rng.integers(0, 1 << 30, int(offsets[-1])).astype(np.uint64)
    seen = {}

    def capture(off, ids):
        seen["off"], seen["ids"] = off, ids
        return None

    for dtype in (np.int32, np.uint32, np.int64, np.uint64):
        sh = ShardedInvLists(offsets, ids64.astype(dtype), 1, 2, capture, device="cpu")
        assert seen["ids"].dtype == np.uint64 and seen["ids"].size == int(seen["off"][-1])
        want = np.concatenate([ids64[int(offsets[l]):int(offsets[l + 1])] for l in sh.my_lists]) if sh.my_lists.size else ids64[:0]
        assert np.array_equal(seen["ids"], want), dtype
    with pytest.raises(TypeError):
        ShardedInvLists(offsets, ids64.astype(np.float64), 0, 2, capture, device="cpu")


def _run_world(world, offsets, ids, req, dst=0):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_any, args=(r, world, port, offsets, ids, req, dst, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def _worker_any(rank, world, port, offsets, ids, req, dst, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vector_db_id_compression_amd.sharding import ShardedInvLists

    sh = ShardedInvLists(offsets, ids, rank, world, _OracleCodec, device="cpu")
    out, off = sh.gather_ids(req, dst=dst)
    # a second request on the same shards (the search loop calls gather_ids once per batch): only lists of ONE owner
    one_owner = np.nonzero(sh.owner == sh.owner[int(req[0])])[0][:3].astype(np.int64)
    out2, off2 = sh.gather_ids(one_owner, dst=dst)
    if rank == dst:
        q.put((out.numpy(), off, out2.numpy(), off2, one_owner, sh.load, sh.owner, np.bincount(sh.owner, minlength=world)))
    else:
        assert out is None and off is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_gather_at_world_size_4_and_8_with_zipf_sizes(world):
    """BASELINE configs[4] in small: Zipf-sized lists over 4 / 8 ranks (the widths the scaling run uses), a search-shaped request
    with repeated lists, owners that hold none of the requested lists, empty lists, and a root other than rank 0."""
    from oracle.pyoracle import Oracle
    from vector_db_id_compression_amd import synth

    offsets, ids = synth.make_lists_numpy(9000, 40, 0.75, seed=9)
    sizes = (offsets[1:] - offsets[:-1]).astype(np.int64)
    from vector_db_id_compression_amd.sharding import lpt_partition

    owner = lpt_partition(sizes, world)
    # request: every list of two owners (the others get nothing to send), the longest list three times, in shuffled order
    rng = np.random.default_rng(3)
    picked = np.nonzero((owner == 1) | (owner == world - 1))[0]
    req = np.concatenate([picked, [int(np.argmax(sizes))] * 3]).astype(np.int64)
    rng.shuffle(req)
    owners_hit = set(owner[req].tolist())
    assert len(owners_hit) < world  # some ranks own none of the requested lists
    dst = world - 1
    out, off, out2, off2, one_owner, load, owner_w, counts = _run_world(world, offsets, ids, req, dst=dst)
    assert np.array_equal(owner_w, owner)
    o = Oracle()

    def check(out, off, req):
        assert int(off[-1]) == int(sizes[req].sum()) == out.size
        for i, l in enumerate(req):
            li = ids[int(offsets[l]):int(offsets[l + 1])]
            if li.size == 0:
                assert off[i] == off[i + 1]
                continue
            e = o.roc_encode(li, o.list_precision(li))
            assert np.array_equal(out[int(off[i]):int(off[i + 1])].astype(np.uint64), e["order"]), (i, l)

    check(out, off, req)
    check(out2, off2, one_owner)
    assert load.sum() == 9000 and load.max() - load.min() <= sizes.max()
    assert counts.sum() == 40


def test_gather_with_an_empty_shard_and_empty_lists():
    """More ranks than non-empty lists: LPT leaves a rank without ids (an empty shard must encode, decode and take part in the
    gather), and empty lists travel as zero-length slices."""
    from vector_db_id_compression_amd.sharding import lpt_partition

    rng = np.random.default_rng(12)
    sizes = np.array([700, 0, 300, 0, 0], dtype=np.int64)  # 2 non-empty lists, 4 ranks
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ids = np.concatenate([np.sort(rng.choice(1 << 20, int(n), replace=False)) for n in sizes]).astype(np.uint64)
    owner = lpt_partition(sizes, 4)
    loads = np.bincount(owner, weights=sizes, minlength=4)
    assert (loads == 0).sum() >= 2  # at least two ranks hold no id at all
    req = np.array([1, 0, 4, 2, 0, 3], dtype=np.int64)
    out, off, out2, off2, one_owner, load, owner_w, counts = _run_world(4, offsets, ids, req, dst=0)
    from oracle.pyoracle import Oracle

    o = Oracle()
    for i, l in enumerate(req):
        li = ids[int(offsets[l]):int(offsets[l + 1])]
        got = out[int(off[i]):int(off[i + 1])].astype(np.uint64)
        if li.size == 0:
            assert got.size == 0
        else:
            assert np.array_equal(got, o.roc_encode(li, o.list_precision(li))["order"])
