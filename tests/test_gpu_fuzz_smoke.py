"""The differential fuzzers of tools/ as short GPU tests, so that a plain `pytest -m gpu` run covers the kernel-family modes that
the library only takes under a switch (the driver's GPU run sees the default mode + the tests that force
a family; the other modes were only green in the builder's own evidence runs).

Each fuzzer runs in its own process for a few seconds with a fixed seed:
* fuzz_families.py: lane-per-list / wave-per-list / row-per-list / general ROC kernels (six family modes, switched through the
  library's environment switches between calls) must write identical streams, permutations and decoded arrays, and match the CPU
  oracle (codec.cpp:21-152 restated) on sampled lists;
* fuzz_chain.py (narrow: 13..20-bit universes, wide: 21..31 bits): the hand-scheduled chain kernels against the round-1 bitmap
  kernels, the general kernels and the oracle on lists of 4 097..150 000 ids (duplicates, unsorted input, the lossy regime);
* fuzz_ef_packed.py: Elias-Fano / packed-bits streams against the oracle's words;
* fuzz_graph_roc.py: ROC graph objects (random widths, node counts on both sides of the edge-count ordering, unaligned row arrays): the
  64-row tile / lane kernels against the wave-per-row kernels and the oracle;
* fuzz_wt.py: wavelet trees built through the partitioned scatter against the direct scatter and the input.
The long runs (minutes, other seeds) stay in tools/final_run.sh; their logs are profiles/r05*_long_fuzz.txt.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    ("fuzz_families.py", ["5", "8"], "fuzz ok"),
    ("fuzz_chain.py", ["5", "8"], "fuzz_chain ok"),
    ("fuzz_chain.py", ["6", "8", "wide"], "fuzz_chain ok"),
    ("fuzz_ef_packed.py", ["5", "6"], "fuzz ok"),
    ("fuzz_graph_roc.py", ["5", "8"], "fuzz ok"),
    ("fuzz_wt.py", ["5", "6"], "fuzz ok"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("script,argv,ok", CASES, ids=[" ".join([c[0]] + c[1]) for c in CASES])
def test_differential_fuzzer_is_green_for_a_few_seconds(script, argv, ok):
    env = dict(os.environ)
    # (the fuzzers set and clear the family switches themselves: start them from the defaults)
    for k in list(env):
        if k.startswith("VIDC_") and k not in ("VIDC_WIDE_STREAMS",):
            env.pop(k)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + argv, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert ok in r.stdout, tail
    assert "MISMATCH" not in r.stdout and "ERROR" not in r.stdout, tail
