"""world_size-2 gloo test (CPU): the N>1 path of bench.py -- round-robin block ownership with no
data-path collective, barrier + MAX-reduce of the elapsed time, whole-job aggregate on rank 0."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time, json, hashlib
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_lib
    from pg_cryogen_amd import shard
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ora = oracle_lib.Oracle()
    N, B = 12, 4096
    mine = list(shard.my_blocks(N, rank, world))
    dist.barrier(); t0 = time.perf_counter()
    # the data path: every rank round-trips only its own blocks (oracle as the stand-in codec on CPU)
    sums = torch.zeros(N, dtype=torch.int64)
    for i in mine:
        raw = ora.synth(0, i, B, i %% 5)
        c = ora.lz4_compress(raw, 1)
        r, out = ora.lz4_decompress(c, B)
        assert r == B and np.array_equal(out, raw)
        sums[i] = int.from_bytes(hashlib.sha256(out.tobytes()).digest()[:7], "little")
    dist.barrier(); el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(per_rank, el)                             # bench.py: every rank's own time next to the MAX
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    owners = torch.zeros(N, dtype=torch.int64); owners[mine] = 1
    dist.all_reduce(owners); dist.all_reduce(sums)            # verification only, not the data path
    if rank == 0:
        exp = [int.from_bytes(hashlib.sha256(ora.synth(0, i, B, i %% 5).tobytes()).digest()[:7], "little") for i in range(N)]
        print(json.dumps({"covered_once": bool((owners == 1).all()), "sums_ok": sums.tolist() == exp,
                          "elapsed_max": float(el[0]), "per_rank": [float(v[0]) for v in per_rank], "blocks": N}))
    dist.destroy_process_group()
""") % (ROOT, ROOT)


def test_round_robin_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                   "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                                  env=env, stderr=subprocess.STDOUT, timeout=300).decode()
    import json
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["covered_once"] and r["sums_ok"] and r["elapsed_max"] > 0
    assert len(r["per_rank"]) == 2 and max(r["per_rank"]) == r["elapsed_max"]


def test_shard_partition_properties():
    from pg_cryogen_amd import shard
    for n in (0, 1, 7, 64, 1000003 % 977):
        for w in (1, 2, 4, 8):
            seen = []
            for r in range(w):
                seen += list(shard.my_blocks(n, r, w))
                assert shard.my_count(n, r, w) == len(shard.my_blocks(n, r, w))
            assert sorted(seen) == list(range(n))
            assert all(shard.owner(i, w) == i % w for i in range(n))
