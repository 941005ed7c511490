"""The boundary is a torch-free C ABI: a plain C program (examples/c_api_demo.c) links
libtristage.so, runs add/search with host pointers on the SYSTEM HIP runtime, and its
output is checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from helpers import check_topk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,d,k,nq", [(5000, 96, 50, 7), (60_000, 128, 100, 64)])
def test_plain_c_host_program(tmp_path, n, d, k, nq):
    exe = str(tmp_path / "c_api_demo")
    lib_dir = os.path.join(ROOT, "tristage-rag_amd")
    subprocess.run(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_api_demo.c"),
                    "-o", exe, "-L", lib_dir, "-ltristage", f"-Wl,-rpath,{lib_dir}", "-lm"], check=True)
    out = str(tmp_path / "out.bin")
    r = subprocess.run([exe, str(n), str(d), str(k), str(nq), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    raw = np.fromfile(out, dtype=np.uint8)
    o = 0
    D = raw[o:o + 4 * nq * k].view(np.float32).reshape(nq, k); o += 4 * nq * k
    I = raw[o:o + 8 * nq * k].view(np.int64).reshape(nq, k); o += 8 * nq * k
    corpus = raw[o:o + 4 * n * d].view(np.float32).reshape(n, d); o += 4 * n * d
    queries = raw[o:o + 4 * nq * d].view(np.float32).reshape(nq, d)
    check_topk(D, I, corpus, queries, k)
    assert ("path 1" in r.stdout) == (n >= 32768)      # fused filter path above 32768 rows
