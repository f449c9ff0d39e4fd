#!/usr/bin/env python3
"""Per-rank kernel time of ONE proof sharded over G ranks against the lone prover's, on one GPU (in-process transport under the
turnstile of csrc/comm.hip; see tests/test_gpu_sharded.py::sharded_profile).  Prints one JSON object per (m, G).
usage: python tools/sharded_profile.py [m=21,25] [G=2,4,8]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402  (torch's HIP runtime first, as tests/conftest.py does)

torch.cuda.is_available()
import oracle_lib  # noqa: E402
import provekit_amd  # noqa: E402
from test_gpu_sharded import sharded_profile  # noqa: E402

ms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "21,25").split(",")]
gs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,4,8").split(",")]
ctx = provekit_amd.Context(0)
for m in ms:
    for G in gs:
        want, got, _, report = sharded_profile(ctx, oracle_lib, m, G)
        report["transcripts_identical"] = all(p == want for p in got)
        print(json.dumps(report), flush=True)
