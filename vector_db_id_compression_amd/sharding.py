"""Inverted lists sharded over the GPUs of one node.

Every list is an independent codec unit: encode and decode need no exchange, so each rank owns a subset of
the lists (balanced by total length, not by count: list sizes are Zipf-like) and compresses / decompresses it
with its own GPU.  The only communication is the variable-size gather of decoded ids to the rank that runs a
search -- one point-to-point message per owning rank over its direct xGMI link (RCCL send/recv; `gloo` in the
CPU tests).  The list -> (rank, local number) map is replicated (nlist * 8 bytes).
"""
import numpy as np


def lpt_partition(sizes, world):
    """Longest-processing-time greedy: lists sorted by length, each to the currently lightest rank.
    -> owner int32[nlist]"""
    sizes = np.asarray(sizes, dtype=np.int64)
    owner = np.zeros(sizes.size, dtype=np.int32)
    load = np.zeros(world, dtype=np.int64)
    for l in np.argsort(-sizes, kind="stable"):
        r = int(np.argmin(load))
        owner[l] = r
        load[r] += sizes[l]
    return owner


class ShardedInvLists:
    """One shard of a CSR set of lists per rank + gather of decoded ids.

    encode_fn(local_offsets uint64, local_ids uint64 numpy) -> codec object with
    decode_lists(local_list_nos) -> (ids tensor, out_offsets)   [RocLists / EfLists have exactly this]
    """

    def __init__(self, offsets, ids, rank, world, encode_fn, group=None, device="cuda"):
        offsets = np.asarray(offsets, dtype=np.uint64)
        self.rank, self.world, self.group, self.device = rank, world, group, device
        self.sizes = (offsets[1:] - offsets[:-1]).astype(np.int64)
        self.nlist = self.sizes.size
        self.owner = lpt_partition(self.sizes, world)
        # local numbering: lists of a rank in increasing global number
        self.local_no = np.zeros(self.nlist, dtype=np.int64)
        for r in range(world):
            mine = np.nonzero(self.owner == r)[0]
            self.local_no[mine] = np.arange(mine.size)
        mine = np.nonzero(self.owner == rank)[0]
        self.my_lists = mine
        loc_sizes = self.sizes[mine]
        self.local_offsets = np.concatenate([[0], np.cumsum(loc_sizes)]).astype(np.uint64)
        ids = np.asarray(ids)
        parts = [ids[int(offsets[l]):int(offsets[l + 1])] for l in mine]
        local_ids = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)
        self.codec = encode_fn(self.local_offsets, local_ids.astype(np.uint64))
        self.load = np.array([self.sizes[self.owner == r].sum() for r in range(world)], dtype=np.int64)

    def decode_local(self, list_nos):
        """Decode the requested GLOBAL list numbers this rank owns -> (tensor, per-list sizes)."""
        ln = np.asarray(list_nos, dtype=np.int64)
        ln = ln[self.owner[ln] == self.rank]
        ids, off = self.codec.decode_lists(self.local_no[ln].astype(np.uint64))
        return ln, ids, off

    def gather_ids(self, list_nos, dst=0):
        """Decoded ids of `list_nos` (global numbers, any owners) assembled on rank `dst`.
        -> (int64 tensor in request order, offsets) on dst, (None, None) elsewhere."""
        import torch
        import torch.distributed as dist

        ln = np.asarray(list_nos, dtype=np.int64)
        sizes = self.sizes[ln]
        req_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        mine, ids, off = self.decode_local(ln)
        ids = ids.to(self.device)
        if self.world == 1:
            out = torch.empty(int(req_off[-1]), dtype=torch.int64, device=self.device)
            self._scatter_into(out, ln, req_off, mine, ids, off)
            return out, req_off
        if self.rank != dst:
            if ids.numel():
                dist.send(ids.contiguous(), dst=dst, group=self.group)
            return None, None
        out = torch.empty(int(req_off[-1]), dtype=torch.int64, device=self.device)
        self._scatter_into(out, ln, req_off, mine, ids, off)
        for r in range(self.world):  # one message per owning rank, sizes known from the replicated map
            if r == dst:
                continue
            theirs = ln[self.owner[ln] == r]
            n = int(self.sizes[theirs].sum())
            if n == 0:
                continue
            buf = torch.empty(n, dtype=torch.int64, device=self.device)
            dist.recv(buf, src=r, group=self.group)
            toff = np.concatenate([[0], np.cumsum(self.sizes[theirs])]).astype(np.int64)
            self._scatter_into(out, ln, req_off, theirs, buf, toff)
        return out, req_off

    @staticmethod
    def _scatter_into(out, ln, req_off, owned, ids, off):
        """Copy the decoded lists `owned` (in that order inside `ids`) to their slots of the request."""
        pos_of = {}
        for i, l in enumerate(ln):
            pos_of.setdefault(int(l), []).append(i)
        for j, l in enumerate(owned):
            seg = ids[int(off[j]):int(off[j + 1])]
            for i in pos_of[int(l)]:
                out[int(req_off[i]):int(req_off[i + 1])] = seg
