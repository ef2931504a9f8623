"""N > 1 path on the CPU: frames are sharded over ranks with no exchange, the only collective is the final gather of the
variable-length bitstreams (charls_amd/batch.py).  world_size 2 -- and 8, the size of the node the driver's scaling run uses:
the first 8-GPU run must not be the first 8-rank run --, gloo backend, spawned processes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from charls_amd import batch


def test_shard_range_covers_all_frames_once():
    for total in (0, 1, 7, 8, 256, 1001):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                a, b = batch.shard_range(total, r, world)
                assert 0 <= a <= b <= total
                seen.extend(range(a, b))
            assert seen == list(range(total))
            sizes = [batch.shard_range(total, r, world)[1] - batch.shard_range(total, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total_frames, result_queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = batch.shard_range(total_frames, rank, world)
        rng = np.random.default_rng(1234)
        all_sizes = rng.integers(5, 200, size=total_frames)
        pitch = 256
        # every rank fabricates the "encoded" frames it owns: frame f is sizes[f] bytes of value f
        mine = torch.zeros((hi - lo, pitch), dtype=torch.uint8)
        for i, f in enumerate(range(lo, hi)):
            mine[i, :all_sizes[f]] = f % 251
        parts, sizes = batch.gather_streams(mine, all_sizes[lo:hi].astype(np.uint64), dst=0, chunk_frames=3)
        assert [len(s) for s in sizes] == [batch.shard_range(total_frames, r, world)[1] - batch.shard_range(total_frames, r, world)[0]
                                           for r in range(world)]
        if rank == 0:
            f = 0
            for pieces, sz in zip(parts, sizes):
                got = 0
                for first, part in pieces:
                    assert first == got
                    for i in range(part.shape[0]):
                        assert int(sz[first + i]) == all_sizes[f]
                        assert (part[i, :int(sz[first + i])] == f % 251).all()
                        f += 1
                    got += part.shape[0]
                assert got == len(sz)
            assert f == total_frames
            result_queue.put("ok")
        else:
            assert parts is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total_frames", [5, 8])
def test_gather_streams_world_size_2(total_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"


def test_gather_streams_world_size_8():
    """Eight ranks, as on the 8-GPU node: 256 frames -- BASELINE configs[3] -- and a count that does not divide (rank shards of
    different sizes, two of them empty with 6 frames)."""
    for total_frames in (256, 6):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 8, port, total_frames, q)) for r in range(8)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
            assert p.exitcode == 0
        assert q.get(timeout=5) == "ok"


def test_bench_cfg4_over_eight_ranks():
    """`bench.py --workload cfg4 --gpus 8 --selftest-exchange`: the launcher starts eight ranks, 256 frames are split 32 a rank,
    rank 0 sees all of them, each from its owner."""
    import json
    r = _run_bench(["--gpus", "8", "--workload", "cfg4", "--selftest-exchange"], {"CHARLS_AMD_BENCH_BACKEND": "gloo"}, timeout=480)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert lines == [{"selftest": "exchange", "n_gpus": 8, "backend": "gloo", "ranks_seen_by_backend": 8, "ok": True,
                      "workload": "cfg4", "frames": 256}], r.stdout


def _run_bench(args, extra_env, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(_free_port()), **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` without a launcher: bench.py re-executes itself under torch.distributed.run with two ranks
    (gloo here, RCCL on a GPU box), the ranks exchange fabricated bitstreams of different sizes through
    batch.gather_streams and rank 0 reports the number of ranks the backend sees."""
    import json
    r = _run_bench(["--gpus", "2", "--selftest-exchange"], {"CHARLS_AMD_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert lines[0] == {"selftest": "exchange", "n_gpus": 2, "backend": "gloo", "ranks_seen_by_backend": 2, "ok": True}
    assert "nranks=2" in r.stderr


def test_bench_cfg4_shards_256_frames_over_the_ranks():
    """`bench.py --workload cfg4 --gpus 2`: BASELINE configs[3] as stated -- 256 frames split by batch.shard_range, every rank's
    bitstreams handed to rank 0 from a side thread (as the timed step does it); rank 0 sees all 256, each from its owner."""
    import json
    r = _run_bench(["--gpus", "2", "--workload", "cfg4", "--selftest-exchange"], {"CHARLS_AMD_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert lines == [{"selftest": "exchange", "n_gpus": 2, "backend": "gloo", "ranks_seen_by_backend": 2, "ok": True,
                      "workload": "cfg4", "frames": 256}], r.stdout


def test_bench_refuses_to_run_fewer_ranks_than_asked_for():
    """A box with fewer GPUs than --gpus is an error, not a silent single-rank run (there is no GPU here at all)."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--frames", "1"], {})
    assert r.returncode == 2
    assert "refusing" in r.stderr and not [x for x in r.stdout.splitlines() if x.startswith("{")]
