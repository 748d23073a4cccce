#!/usr/bin/env python3
"""The id-compression part of the reference's large-scale driver (custom_invlist_cpp/search_ivf_qinco.py) on the in-repo IVF
harness: same switches (`--id_compression none|packed-bits|elias-fano|roc|wavelet-tree|wavelet-tree-1`,
`--defer_id_decoding`, `--id_decoding_1by1`, `--redo_search`, `--nprobe ...`, :382-395), same flow (:502-523: build the
container from index.invlists, print its size and build time, replace_invlists, run the searches) and a JSON result line.
QINCo itself (the neural re-ranking codec) is out of scope: codes are PQ.

    python tools/search_ivf.py --nb 10000000 --nlist 65536 --id_compression roc --defer_id_decoding --nprobe 4 16 64
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vector_db_id_compression_amd import custom_invlists  # noqa: E402
from vector_db_id_compression_amd.ivf import IVFIndex  # noqa: E402

if __name__ == "__main__":
    import torch

    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=32)
    ap.add_argument("--nlist", type=int, default=1024)
    ap.add_argument("--pq", type=int, default=8)
    ap.add_argument("--nq", type=int, default=100)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--id_compression", default="none", choices=custom_invlists.ID_COMPRESSION_CHOICES, help="How to compress the ids")
    ap.add_argument("--defer_id_decoding", default=False, action="store_true")
    ap.add_argument("--id_decoding_1by1", default=False, action="store_true")
    ap.add_argument("--redo_search", default=1, type=int, help="number of times to redo the search (to stabilize timings)")
    ap.add_argument("--nprobe", default=[1, 4, 16, 64], nargs="+", type=int)
    args = ap.parse_args()

    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    cent = torch.randn(256, args.d, generator=g, device="cuda") * 3
    xb = (cent[torch.randint(0, 256, (args.nb,), generator=g, device="cuda")] + torch.randn(args.nb, args.d, generator=g, device="cuda")).cpu().numpy()
    xq = (cent[torch.randint(0, 256, (args.nq,), generator=g, device="cuda")] + torch.randn(args.nq, args.d, generator=g, device="cuda")).cpu().numpy()
    res = {"args": vars(args)}
    t0 = time.time()
    index = IVFIndex(args.d, args.nlist, ("PQ", args.pq))
    index.train(xb)
    index.add(xb)
    index.parallel_mode = 3
    res["build_time"] = time.time() - t0
    if args.id_compression != "none":
        print("compressing ids with", args.id_compression)
        il, st = custom_invlists.apply_id_compression(index, args.id_compression)
        print(f"compressed ids size: {st['compressed_ids_size_in_bytes']} bytes, compressed in {st['id_compression_time']:.3f} s")
        res.update(st)
    res["searches"] = []
    for nprobe in args.nprobe:
        index.nprobe = nprobe
        times = []
        for _ in range(args.redo_search):
            torch.cuda.synchronize()
            t0 = time.time()
            if args.defer_id_decoding:
                D, I = index.search_defer_id_decoding(xq, args.k, decode_1by1=args.id_decoding_1by1)
            else:
                D, I = index.search(xq, args.k)
            torch.cuda.synchronize()
            times.append(time.time() - t0)
        res["searches"].append({"nprobe": nprobe, "t_search": float(np.min(times)), "checksum": int(I.sum())})
        print(res["searches"][-1], flush=True)
    print("JSON results:", json.dumps(res))
