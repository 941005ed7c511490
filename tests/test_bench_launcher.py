"""bench.py's own launcher (`python bench.py --gpus N` with no torchrun around it): the plumbing on the
CPU — the ranks get the torchrun environment, rank 0's stdout is what the caller sees, one failing rank
fails the job and stops the others — and, under -m gpu, the gloo rehearsal of the 2-rank bench started
through exactly that code path."""
import io
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)     # top level imports the standard library only: no torch, no HIP
    return m


def test_launcher_module_imports_without_torch():
    code = ("import sys, importlib.util as u; s = u.spec_from_file_location('b', sys.argv[1]); m = u.module_from_spec(s); "
            "s.loader.exec_module(m); assert 'torch' not in sys.modules, 'the launcher must not import torch'")
    subprocess.run([sys.executable, "-c", code, BENCH], check=True)


def test_ranks_see_the_torchrun_environment():
    b = _bench_module()
    out, log = io.StringIO(), io.StringIO()
    child = ("import os, json; print('[lib] chatter on stdout'); print(json.dumps({k: os.environ.get(k) for k in "
             "('RANK','LOCAL_RANK','WORLD_SIZE','LOCAL_WORLD_SIZE','MASTER_ADDR','MASTER_PORT','HSA_ENABLE_IPC_MODE_LEGACY')}))")
    rc = b.spawn_ranks(3, [sys.executable, "-c", child], relay=out, log=log)
    assert rc == 0
    assert out.getvalue().count("\n") == 1 and "[rank 0] [lib] chatter" in log.getvalue()   # exactly one line reaches the caller
    mine = json.loads(out.getvalue())                       # rank 0's json line only
    assert mine["RANK"] == "0" and mine["LOCAL_RANK"] == "0" and mine["WORLD_SIZE"] == "3"
    assert mine["MASTER_ADDR"] == "127.0.0.1" and int(mine["MASTER_PORT"]) > 0 and mine["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    others = [json.loads(l.split("] ", 1)[1]) for l in log.getvalue().splitlines()
              if l.startswith("[rank ") and not l.startswith("[rank 0]") and "{" in l]
    assert sorted(o["RANK"] for o in others) == ["1", "2"]
    assert all(o["MASTER_PORT"] == mine["MASTER_PORT"] and o["WORLD_SIZE"] == "3" for o in others)


def test_one_failing_rank_fails_the_job_and_stops_the_others():
    b = _bench_module()
    out, log = io.StringIO(), io.StringIO()
    child = ("import os, sys, time\n"
             "if os.environ['RANK'] == '1': sys.exit(7)\n"
             "time.sleep(120)\n")              # a rank waiting in a collective for the dead one
    t0 = time.time()
    rc = b.spawn_ranks(2, [sys.executable, "-c", child], relay=out, log=log)
    assert rc == 7 and time.time() - t0 < 60
    assert "rank 1 exited with 7" in log.getvalue()


def test_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE=2" in r.stderr


def test_refuses_more_gpus_than_visible():
    """No launcher around it and fewer devices than --gpus: a non-zero exit and NO json line (round 2 printed a
    one-GPU line labelled by a stderr warning)."""
    import torch
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TS_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(have + 2), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and r.stdout.strip() == "" and "visible" in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` as the driver types it, TS_BENCH_BACKEND=gloo because the test box has one GPU
    (ranks share it, the exchange is staged through the host): two processes, a 2-rank process group, the line
    says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TS_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "6", "--warmup", "2", "--rows", "400000",
                        "--dim", "256", "--k", "100", "--no-encode-leg", "--no-pipeline-leg", "--no-read-probe"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["backend"] == "gloo"
    assert out["distinct_devices"] == 1 and "REHEARSAL" in out["config"]["workload"]
    assert out["value"] > 0 and out["scaling"] == "strong"


@pytest.mark.gpu
def test_bench_pipeline_gpus_2_runs_the_row_sharded_pipeline():
    """`python bench_pipeline.py --gpus 2 --store --ids --many 8` starts its own two ranks (gloo rehearsal: both on the
    one GPU of the test box) and times ShardedRetrievalPipeline.search_many: one JSON line from rank 0, n_gpus 2,
    rank 0 holding half of the rows / token matrices."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TS_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_pipeline.py"), "--gpus", "2", "--docs", "600", "--queries", "16",
                        "--store", "--ids", "--many", "8", "--stage1", "random:tiny", "--stage2", "random:tiny",
                        "--stage3", "random:tiny"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["array_path"]
    assert out["rank0_shard"]["rows"] == [0, 300] and out["rank0_shard"]["documents_total"] == 600
    assert out["rank0_shard"]["stage2_token_rows"] > 0 and out["rank0_shard"]["stage3_id_cache_documents"] == 300
