"""CPU test of the multi-GPU path of bench.py: world_size 2 over gloo.  The hot path shards by
independent blocks (no data-path collective); what is distributed is the shard assignment and
the reduction of {bytes, elapsed} that forms the aggregate GiB/s."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, q):
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard(1000, rank, world)
    # every rank generates ITS OWN blocks from the global block index (seeded by index)
    blocks = bench.gen_blocks(torch, torch.device("cpu"), 4, lo, chunk=2)
    u = float(blocks.numel())
    c = u / (2.0 + rank)                       # pretend compressed sizes
    wall = 1.0 + 0.5 * rank                    # rank 1 is the slow one
    tot_u, tot_c, wall_max = bench.reduce_totals(torch, dist, torch.device("cpu"), u, c, wall, True)
    q.put((rank, lo, hi, tot_u, tot_c, wall_max, int(blocks[0, :64].sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, u0, c0, w0, s0), (r1, lo1, hi1, u1, c1, w1, s1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 1000, 1000, 2000)            # contiguous, disjoint shards
    assert u0 == u1 == 2 * 4 * 65536                                 # SUM over ranks
    assert abs(c0 - (4 * 65536 / 2.0 + 4 * 65536 / 3.0)) < 1e-6 and c0 == c1
    assert w0 == w1 == 1.5                                           # MAX over ranks
    assert s0 != s1                                                  # different shards -> different data


def _mix_py(x):
    M = (1 << 64) - 1
    x &= M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
    return x ^ (x >> 31)


def _makedata_block(i):
    """block i of bench.gen_blocks, byte by byte the way /root/reference samples/makedata.c:51-68 fills its
    buffer: per-block seed 0x9E3779B97F4A7C15 ^ i, 33-symbol text in the first half,
    then copies of len in [16, len_max + 15] from dist in [1, min(dist_max, idx)] back"""
    import bench
    G, M = bench.GOLD, (1 << 64) - 1
    seed = _mix_py(i ^ G)

    def r31(counter):
        return _mix_py((seed + counter * G) & M) >> 33
    half = 32768
    buf = bytearray(bench.ALPHABET33[r31(j) % 33] for j in range(half))
    len_max = 10 + r31(1 << 20) % 240
    dist_max = 1 + r31((1 << 20) + 1) % 65536
    k = 0
    while len(buf) < 65536:
        ln = 16 + r31((2 << 20) + k) % len_max
        dist = 1 + r31((3 << 20) + k) % min(dist_max, len(buf))
        k += 1
        for _ in range(ln):
            if len(buf) < 65536:
                buf.append(buf[-dist])
    return bytes(buf), len_max, dist_max


def test_generator_is_the_makedata_recipe_per_block_index():
    import bench
    a = bench.gen_blocks(torch, torch.device("cpu"), 6, 8, chunk=4)
    b = bench.gen_blocks(torch, torch.device("cpu"), 3, 10, chunk=2)
    assert torch.equal(a[2:5], b)                                    # block i depends on i alone, not on the batch it is made in
    params = set()
    for j in range(6):
        exp, len_max, dist_max = _makedata_block(8 + j)
        assert a[j].numpy().tobytes() == exp, j
        assert 10 <= len_max <= 249 and 1 <= dist_max <= 65536
        params.add((len_max, dist_max))
    assert len(params) == 6                                          # every block draws its own len_max / dist_max
    # second half copies from what is in front of it -> compressible
    import zlib
    blk = a[1].numpy().tobytes()
    assert len(zlib.compress(blk, 1)) < 0.7 * len(blk)


def _worker_c5(rank, world, port, q):
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 11                                                       # odd on purpose: shards of 6 and 5
    lo, hi = bench.shard_strong(total, rank, world)
    blocks = bench.gen_mixed(torch, torch.device("cpu"), hi - lo, lo)
    kinds = [int(b.sum() == 0) for b in blocks]                      # zeros at global index % 4 == 0
    tot_u, _, wall_max = bench.reduce_totals(torch, dist, torch.device("cpu"), float(blocks.numel()), 1.0, 1.0 + rank, True)
    q.put((rank, lo, hi, kinds, tot_u, wall_max))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_strong_scaled_shards_of_the_mixed_batch():
    """config c5 (BASELINE configs[4]): one fixed batch, contiguous shards, every block index once"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_c5, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, k0, u0, w0), (r1, lo1, hi1, k1, u1, w1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 6, 6, 11)
    assert k0 == [1, 0, 0, 0, 1, 0] and k1 == [0, 0, 1, 0, 0]        # global index mod 4 decides the kind
    assert u0 == u1 == 11 * 65536 and w0 == w1 == 2.0


def test_mixed_batch_kinds():
    import bench
    import zlib
    b = bench.gen_mixed(torch, torch.device("cpu"), 8, 0).numpy()
    sizes = [len(zlib.compress(row.tobytes(), 1)) for row in b]
    assert sizes[0] < 400 and sizes[4] < 400                         # zeros
    assert sizes[3] > 65536 and sizes[7] > 65536                     # random bytes: stored path
    assert 30000 < sizes[1] < 50000 and 20000 < sizes[2] < 50000     # 33-symbol text; text + LZ copies


# ---- `python bench.py --gpus N` without a launcher starts its own ranks ----
def test_launcher_builds_the_drivers_command_and_relays_rank0(capsys):
    import bench

    class Done:
        returncode = 0
        stdout = '{"metric": "x", "n_gpus": 4}\n'
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return Done()
    rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"], visible=8, run=fake_run)
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]      # the caller's arguments, unchanged
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                            # dmabuf IPC for RCCL between processes
    assert capsys.readouterr().out == Done.stdout                                      # rank 0's line, relayed


def test_launcher_refuses_more_gpus_than_are_visible(capsys):
    import bench
    called = []
    rc = bench.launch_ranks(8, ["--gpus", "8"], visible=1, run=lambda *a, **k: called.append(a))
    assert rc == 2 and not called
    assert "8 GPUs requested, 1 visible" in capsys.readouterr().err


def test_bench_gpus_2_without_gpus_fails_with_a_clear_message():
    """the whole script: no launcher in the environment, fewer devices than asked for"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2
    assert "2 GPUs requested, 0 visible" in p.stderr
