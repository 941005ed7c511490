"""N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the row-shard
planning, id offsets, the single packed all-gather and the merge of
tristage_rag_amd.sharded.ShardedFlatIPIndex.  The per-rank scan and the merge
kernel are GPU code, so here they are replaced by oracle-backed doubles; what is
under test is the distributed plumbing, which must reproduce the unsharded top-k
exactly, for every world size, including ties across shard boundaries."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tristage_rag_amd.sharded import ShardedFlatIPIndex, shard_bounds


def test_shard_bounds_cover_rows_once():
    for n in (0, 1, 7, 8, 9, 1000, 10_000_000):
        for r in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, r, i) for i in range(r)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
    assert shard_bounds(10_000_000, 8, 3) == (3_750_000, 5_000_000)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, k, B, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from doubles import OracleIndex, oracle_merge
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(42)                     # same data on every rank
        base = rng.standard_normal((n // 4 + 1, d)).astype(np.float32)
        corpus = base[rng.integers(0, base.shape[0], size=n)]   # duplicates -> exact ties across shards
        queries = rng.standard_normal((B, d)).astype(np.float32)
        idx = ShardedFlatIPIndex(d, n, local_index=OracleIndex(d), merge_fn=oracle_merge)
        assert (idx.lo, idx.hi) == shard_bounds(n, world, rank)
        idx.add_global(corpus)
        assert idx.local_index.ntotal == idx.hi - idx.lo and idx.ntotal == n
        D, I = idx.search(torch.from_numpy(queries), k)
        np.save(os.path.join(out_dir, f"D{rank}.npy"), D.numpy())
        np.save(os.path.join(out_dir, f"I{rank}.npy"), I.numpy())
        if rank == 0:
            np.save(os.path.join(out_dir, "corpus.npy"), corpus)
            np.save(os.path.join(out_dir, "queries.npy"), queries)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,k", [(2, 501, 40), (3, 100, 60), (2, 5, 8)])
def test_sharded_search_equals_unsharded(tmp_path, world, n, k):
    from oracle import oracle
    d, B = 24, 6
    mp.spawn(_worker, args=(world, _free_port(), n, d, k, B, str(tmp_path)), nprocs=world, join=True)
    corpus, queries = np.load(tmp_path / "corpus.npy"), np.load(tmp_path / "queries.npy")
    D0, I0 = oracle.ip_topk(corpus, queries, k)
    for r in range(world):                                    # identical on every rank, equal to unsharded
        assert np.array_equal(np.load(tmp_path / f"I{r}.npy"), I0)
        np.testing.assert_array_equal(np.load(tmp_path / f"D{r}.npy"), D0)
