#!/usr/bin/env python3
"""Differential fuzz of the hand-scheduled chain kernels (roc_u2.h; dev tool, run through gpurun): batches of long lists
(4 097 .. 70 000 ids, now and then up to 150 000: the reference's lossy regime) over universes 2^13 .. 2^20, dense lists,
duplicates, unsorted input, explicit precisions below / above what the ids need.  Encoded and decoded by the new kernels,
the round-1 bitmap kernels (VIDC_OLD_U=1) and the general kernels (VIDC_FORCE_GENERAL=1): streams, permutations and decoded
arrays must be identical; every list is also checked against the CPU oracle (stream + the reference's decode of it)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyoracle import Oracle  # noqa: E402  (dev tool: the checker)
from vector_db_id_compression_amd.codecs import RocLists  # noqa: E402

MODES = {"u2": {"VIDC_OLD_U": "0", "VIDC_FORCE_GENERAL": "0"},
         "u1": {"VIDC_OLD_U": "1", "VIDC_FORCE_GENERAL": "0"},
         "general": {"VIDC_OLD_U": "0", "VIDC_FORCE_GENERAL": "1"}}
# third argument "wide": universes 2^21 .. 2^31 -- the position-bitmap chain kernel (k_roc_encode_r2) against the general
# kernels (VIDC_NO_R2=1: through the normal classes, VIDC_FORCE_GENERAL=1: everything) and the oracle
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"
if WIDE:
    MODES = {"u2": {"VIDC_NO_R2": "", "VIDC_FORCE_GENERAL": "0"},
             "u1": {"VIDC_NO_R2": "1", "VIDC_FORCE_GENERAL": "0"},
             "general": {"VIDC_NO_R2": "", "VIDC_FORCE_GENERAL": "1"}}


def make_batch(rng):
    nbits = int(rng.integers(21, 32)) if WIDE else int(rng.integers(13, 21))
    nlist = int(rng.integers(1, 6))
    lists = []
    for _ in range(nlist):
        hi = 70000 if rng.random() < 0.9 else 150000
        s = int(min(rng.integers(4097, hi), (1 << nbits) - 3))
        r = rng.random()
        if r < 0.15:  # dense: nearly the whole universe
            li = rng.choice(s + int(rng.integers(1, 4)), size=s, replace=False)
        elif r < 0.22:  # duplicates
            li = rng.integers(0, 1 << nbits, size=s)
        elif r < 0.3:  # clustered
            c = rng.integers(0, 1 << nbits, size=8)
            li = np.unique((c[rng.integers(0, 8, size=s)] + rng.integers(0, 1 << max(nbits - 6, 1), size=s)) % (1 << nbits))
        else:
            li = rng.choice(1 << nbits, size=s, replace=False)
        li = np.sort(li).astype(np.uint64)
        if rng.random() < 0.1:
            rng.shuffle(li)
        lists.append(li)
    off = np.concatenate([[0], np.cumsum([li.size for li in lists])]).astype(np.uint64)
    ids = np.concatenate(lists)
    mode = -1
    r = rng.random()
    if r < 0.15:
        mode = -2
    elif r < 0.35:
        mode = int(rng.integers(max(nbits - 2, 0), min(nbits + 3, 32) + 1))
    return off, ids, lists, mode


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    rng = np.random.default_rng(seed)
    orc = Oracle()
    t0 = time.time()
    nb = nl = nid = nerr = 0
    while time.time() - t0 < budget:
        off, ids, lists, mode = make_batch(rng)
        want_perm = bool(rng.random() < 0.5) and all(np.all(li[1:] > li[:-1]) for li in lists)
        got = {}
        for name, env in MODES.items():
            os.environ.update(env)
            try:
                r = RocLists.encode(off, ids, precision_mode=mode, want_perm=want_perm)
                info = r.info()
                dec = r.decode_all().cpu().numpy().copy()
            except Exception as ex:  # domain errors (e.g. more than 1024 underflow words) must be the same in every family
                got[name] = str(ex)
                continue
            got[name] = (info["heads"], info["nwords"], info["precision"], info["mt_draws"], r.all_words(), dec,
                         r.perm() if want_perm else np.zeros(0))
        if any(isinstance(g, str) for g in got.values()):
            if len(set(map(str, got.values()))) != 1:
                print("ERROR in some families only, seed", seed, "batch", nb, "mode", mode, ":", got, flush=True)
                np.savez("gpurun_out/fuzz_chain_fail.npz", off=off, ids=ids, mode=mode)
                sys.exit(1)
            nerr += 1
            nb += 1
            continue
        for name in ("u1", "general"):
            for k, (a, b) in enumerate(zip(got["u2"], got[name])):
                if not np.array_equal(a, b):
                    print("MISMATCH u2 vs", name, "field", k, "seed", seed, "batch", nb, "mode", mode, flush=True)
                    np.savez("gpurun_out/fuzz_chain_fail.npz", off=off, ids=ids, mode=mode)
                    sys.exit(1)
        heads, nwords, prec, draws, words, dec, _ = got["u2"]
        woff = np.concatenate([[0], np.cumsum(nwords.astype(np.int64))])
        for l, li in enumerate(lists):
            P = int(prec[l])
            e = orc.roc_encode(li, P)
            assert int(heads[l]) == e["head"], (seed, nb, l)
            assert np.array_equal(words[woff[l]:woff[l + 1]], e["words"]), (seed, nb, l)
            ref = orc.roc_decode(e["head"], e["words"], li.size, P, e["mt_draws"])[0]
            assert np.array_equal(dec[int(off[l]):int(off[l + 1])].view(np.uint64), ref), (seed, nb, l)
        nb += 1
        nl += len(lists)
        nid += int(ids.size)
    print(f"fuzz_chain ok: seed {seed}, {nb} batches, {nl} lists, {nid} ids ({nerr} batches rejected identically by all families): chain / round-1 / general kernels identical, oracle identical", flush=True)


if __name__ == "__main__":
    main()
