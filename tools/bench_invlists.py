#!/usr/bin/env python3
"""Counterpart of the reference's custom_invlist_cpp/bench_invlists.py on the in-repo IVF harness (no Faiss here).

Same flow and CSV columns (bench_invlists.py:120-137): build the index, construct the four compressed inverted
lists, time `search_defer_id_decoding` for nprobe in {1, 4, 16}, k = 20.  Datasets are synthetic (sift1M / deep1M
are not available offline): index 0 = small smoke set, 1 = 1M x 128 ("sift1M shape"), 2 = 1M x 96 ("deep1M shape").

    python tools/bench_invlists.py 1 IVF1024,PQ16 [num_runs] [nq]
"""
import datetime
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vector_db_id_compression_amd.custom_invlists import AVAILABLE_COMPRESSED_IVFS  # noqa: E402
from vector_db_id_compression_amd.ivf import IVFIndex  # noqa: E402


class SyntheticDataset:
    def __init__(self, d, nt, nb, nq, seed=1338):
        rng = np.random.default_rng(seed)
        self.d, self.nt, self.nb, self.nq = d, nt, nb, nq
        centers = rng.normal(size=(256, d)).astype(np.float32) * 2

        def draw(n):
            return (centers[rng.integers(0, 256, n)] + rng.normal(size=(n, d))).astype(np.float32)

        self._t, self._b, self._q = draw(nt), draw(nb), draw(nq)

    def get_train(self):
        return self._t

    def get_database(self):
        return self._b

    def get_queries(self):
        return self._q


def get_ids_size(dataset, invlist, comp_method):
    return 8 * dataset.nb if comp_method is None else invlist.compressed_ids_size_in_bytes


def get_overhead_size(dataset, invlist, comp_method):
    if comp_method is None:
        return 0
    if comp_method in ["roc", "elias-fano"]:
        return invlist.overhead_in_bytes


def parse_index(d, index_str):
    ivf, code = index_str.split(",")
    nlist = int(ivf[3:])
    if code == "Flat":
        return IVFIndex(d, nlist, "Flat")
    assert code.startswith("PQ")
    return IVFIndex(d, nlist, ("PQ", int(code[2:].replace("np", ""))))


if __name__ == "__main__":
    import pandas as pd

    dataset_idx = int(sys.argv[1])
    index_str = str(sys.argv[2])
    num_runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    nq = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    now = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S-%f")
    AVAILABLE_DATASETS = [
        (SyntheticDataset, dict(d=32, nt=10_000, nq=nq, nb=1_000)),
        (SyntheticDataset, dict(d=128, nt=200_000, nq=nq, nb=1_000_000)),
        (SyntheticDataset, dict(d=96, nt=200_000, nq=nq, nb=1_000_000)),
    ]
    compression_methods = list(AVAILABLE_COMPRESSED_IVFS.keys())[:-1]
    search_time_params = dict(k=[20], nq=[None], nprobe=[1, 4, 16])
    results = []
    dataset_cls, dataset_kwargs = AVAILABLE_DATASETS[dataset_idx]
    dataset = dataset_cls(**dataset_kwargs)
    csv_path = Path(f"gpurun_out/results-online-ivf/ivf-results-{now}-{index_str}-{dataset_idx}.csv".replace(",", "_"))
    csv_path.parent.mkdir(parents=True, exist_ok=True)

    index = parse_index(dataset.d, index_str)
    index.train(dataset.get_train())
    index.add(dataset.get_database())
    index.parallel_mode = 3  # for deferred decoding
    t0 = time.time()
    invlists_comp = {m: AVAILABLE_COMPRESSED_IVFS[m](index.invlists) for m in compression_methods}
    print(f"built {len(invlists_comp)} compressed invlists in {time.time() - t0:.2f} s", flush=True)
    ref_invlists = index.invlists
    I_ref = {}
    for comp_method in [None, *compression_methods]:
        if comp_method is None:
            invlist = ref_invlists
            index.replace_invlists(ref_invlists, False)
        else:
            invlist = invlists_comp[comp_method]
            index.replace_invlists(invlist, False)
        decode_1by1 = comp_method in ("wavelet-tree", "packed-bits", None)
        for k in search_time_params["k"]:
            for nprobe in search_time_params["nprobe"]:
                index.nprobe = nprobe
                queries = dataset.get_queries()
                for run_id in range(num_runs):
                    t0 = time.time()
                    D, I = index.search_defer_id_decoding(queries, k=k, decode_1by1=decode_1by1)
                    dt_search = time.time() - t0
                    if comp_method is None:
                        I_ref[nprobe] = (D, I)
                    else:
                        # same results as uncompressed; ids may only differ inside groups of EQUAL distances (PQ codes
                        # collide on 1M vectors, and ROC / EF store the codes in a different order than add order)
                        Dr, Ir = I_ref[nprobe]
                        assert np.array_equal(D, Dr), (comp_method, nprobe)
                        diff = I != Ir
                        if diff.any():
                            qs, ks = np.nonzero(diff)
                            for q, kk in zip(qs, ks):
                                grp = D[q] == D[q, kk]
                                assert grp.sum() > 1 or kk == k - 1, (comp_method, nprobe, q, kk)
                                if kk < k - 1 or grp.sum() > 1:
                                    pass
                            print(f"   note: {len(set(qs))} queries have tied distances resolved in a different order", flush=True)
                    results.append({
                        "dt_search": dt_search, "nprobe": nprobe, "run_id": run_id, "index_str": index_str, "k": k,
                        "nq": queries.shape[0], "comp_method": comp_method or "ref", "dataset": f"Synthetic{dataset_idx}",
                        "ids_size": get_ids_size(dataset, invlist, comp_method),
                        "overhead_size": get_overhead_size(dataset, invlist, comp_method),
                        "nb": dataset.nb, "nt": dataset.nt,
                    })
                print(results[-1], flush=True)
    df = pd.DataFrame(results)
    df.to_csv(csv_path, index=False)
    print(f"Saved to {csv_path} with {len(results)} entries", flush=True)
    print(df.groupby(["comp_method", "nprobe"]).agg(dt=("dt_search", "median"), ids_size=("ids_size", "first")), flush=True)
