"""GPU: csrc/comm.hip's RCCL branch at G = 2, 4, 8 on a ONE-GPU box.

Real RCCL refuses two ranks on one device, so until a multi-GPU node runs tests/test_gpu_sharded.py's `rccl_set` tests the
RCCL side of the transport switch -- dlsym'd entry points, counts in elements of the stated datatype, the in-place all-reduce,
ncclCommInitAll (pk_ctx_create_set) and ncclGetUniqueId + ncclCommInitRank (pk_comm_init_rank), ncclCommAbort on a failing
rank -- had only met a communicator of ONE rank.  Here the library is pointed (PK_RCCL_LIB) at an in-process stand-in,
tests/stub_rccl/ (test infrastructure, never shipped), and the sharded commit / openings / proof run through THAT branch:
roots, openings and whole transcripts must equal the lone prover's.  A fresh process per run: the library resolves its RCCL once."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
STUB_DIR = os.path.join(HERE, "stub_rccl")
STUB = os.path.join(STUB_DIR, "libpk_stub_rccl.so")


def test_sharded_prover_through_the_rccl_branch_with_an_in_process_stand_in():
    if not os.path.exists(STUB):
        subprocess.check_call(["make", "-C", STUB_DIR])
    env = dict(os.environ, PK_RCCL_LIB=STUB, PK_STUB_RCCL_TIMEOUT_S="4", PK_COMM_TIMEOUT_S="2")
    out = subprocess.run([sys.executable, os.path.join(HERE, "rccl_stub_driver.py"), "2,4,8"], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RCCL_STUB_REPORT ")][-1]
    rep = json.loads(line[len("RCCL_STUB_REPORT "):])
    print(line)
    assert [c["G"] for c in rep["cases"]] == [2, 4, 8]
    for c in rep["cases"]:
        assert c["all_gathers"] > c["G"] and c["all_reduces"] >= c["G"]  # the proof's collectives went through the stand-in
    assert rep["stub_calls"]["init_all"] == 3 and rep["stub_calls"]["init_rank"] == 4 and rep["stub_calls"]["abort"] >= 2
    assert 1.5 < rep["hang_deadline_s"] < 30.0  # a collective that hangs on the stream costs its rank the deadline, not for ever
