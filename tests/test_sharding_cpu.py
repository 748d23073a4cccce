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
