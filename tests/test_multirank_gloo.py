"""N>1 path on CPU: two gloo ranks run the sharding / timing plumbing bench.py uses under torch.distributed.run."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_partition_and_aggregate(tmp_path):
    env = dict(os.environ, OG_TEST_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "multirank_worker.py")]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, timeout=300)
    r = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert [x["world"] for x in r] == [2, 2]
    # stream ids: disjoint, contiguous, cover [0, 2n)
    assert (r[0]["lo"], r[0]["hi"]) == (0, 6) and (r[1]["lo"], r[1]["hi"]) == (6, 12)
    # different global streams -> different payloads and different PCM
    assert r[0]["pay_crc"] != r[1]["pay_crc"] and r[0]["crc"] != r[1]["crc"]
    # every frame of every rank decoded; the job's figure is all frames / the slowest rank's time, on both ranks
    assert r[0]["ok"] == r[1]["ok"] == 6 * 3
    for x in r:
        assert x["total"] == 36
        assert abs(x["dt_max"] - max(r[0]["dt"], r[1]["dt"])) < 1e-6
        assert abs(x["value"] - 36 / x["dt_max"]) < 1e-6
    assert r[1]["dt"] > r[0]["dt"]


def test_work_queue_scatter(tmp_path):
    """The one exchange step of the path: rank 0 ingests Ogg pages for the streams of both ranks, routes them by owner,
    demuxes them into decode steps and scatters the packed work; each rank must receive exactly its own share."""
    env = dict(os.environ, OG_TEST_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "multirank_pages_worker.py")]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, timeout=300)
    r = [json.load(open(tmp_path / f"pages_rank{k}.json")) for k in range(2)]
    for k, x in enumerate(r):
        assert x["rank"] == k and x["same"], x
        assert x["per_rank_same"], x  # raw pages scattered, demuxed by the receiving rank: the same work
        assert x["n_steps"] == 4 and x["counts"] == [30] * 4  # 30 streams per rank, 4 packets per page
        assert x["grouped"] and x["size"] == x["nbytes"]
    assert r[0]["crc"] != r[1]["crc"]  # different streams, different work


def test_bench_launcher_starts_the_ranks():
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks as a child job (before touching any GPU)
    and relays rank 0's line.  Run here with the gloo backend and no GPU work (--rendezvous-only)."""
    env = dict(os.environ, OPUSGPU_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_counted"] == 2


def test_bench_refuses_a_rank_count_that_is_not_gpus():
    """A launcher that started one rank for --gpus 2 must not yield an `n_gpus: 1` line: the run fails loudly."""
    env = dict(os.environ, OPUSGPU_DIST_BACKEND="gloo", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "refusing" in p.stderr
    assert not [x for x in p.stdout.splitlines() if x.startswith("{")]


def test_work_queue_scatter_eight_ranks_with_a_remainder(tmp_path):
    """BASELINE config 5's shape at world_size 8 (gloo, CPU): the per-rank share is NOT a multiple of the three modes (262,144 =
    3 x 87,381 + 1; here 100 = 3 x 33 + 1 streams per rank), rank 0 routes 800 pages by owner, scatters raw pages and packed work,
    and every rank must hold exactly its own share -- byte for byte what it would have built itself."""
    env = dict(os.environ, OG_TEST_OUT=str(tmp_path), OMP_NUM_THREADS="1", OG_TEST_STREAMS="100")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "multirank_pages_worker.py")]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, timeout=600)
    r = [json.load(open(tmp_path / f"pages_rank{k}.json")) for k in range(8)]
    for k, x in enumerate(r):
        assert x["rank"] == k and x["same"] and x["per_rank_same"], x
        assert x["n_steps"] == 4 and x["counts"] == [100] * 4 and x["grouped"] and x["size"] == x["nbytes"]
    assert len({x["crc"] for x in r}) == 8


def test_config5_page_arithmetic():
    """2,097,152 pages over 8 ranks, as bench.py's mixed_pages_2m shards them: page p belongs to global stream p, its owner is
    p // 262,144 (shard.Ranks.owner_of), its mode the local id mod 3.  Every page has exactly one owner, every rank 262,144 of
    them -- 87,382 SILK-NB, 87,381 hybrid, 87,381 CELT -- and nothing is trimmed (round 2 dropped 8 pages to make the share a
    multiple of three)."""
    import importlib.util

    import numpy as np
    spec = importlib.util.spec_from_file_location("og_shard_arith", os.path.join(ROOT, "esp32-opus-player_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    total, world = 2097152, 8
    per_rank = total // world
    assert per_rank * world == total and per_rank == 262144
    pages = np.arange(total, dtype=np.int64)
    owner = np.array([shard.Ranks.owner_of(None, int(p), per_rank) for p in (0, per_rank - 1, per_rank, total - 1)])
    assert owner.tolist() == [0, 0, 1, 7]
    owner = pages // per_rank  # (owner_of, vectorised)
    local = pages - owner * per_rank
    assert np.array_equal(np.bincount(owner, minlength=world), np.full(world, per_rank))
    for r in range(world):
        modes = np.bincount(local[owner == r] % 3, minlength=3)
        assert modes.tolist() == [87382, 87381, 87381] and modes.sum() == per_rank
        # the streams bench.py gives mode m on a rank: range(m, n, 3)
        assert [len(range(m, per_rank, 3)) for m in range(3)] == modes.tolist()
    sys.path.insert(0, ROOT)
    import bench
    assert bench.WORKLOADS["mixed_pages_2m"][3] == per_rank
