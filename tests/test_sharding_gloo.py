"""N>1 path on CPU: two gloo ranks take their blocks of a batch of independent streams from the
package's partition function (lamehip.shard_streams, the one bench.py calls per rank), each
encodes its block with the CPU checker standing in for the device (no data-path collective), and
the gathered per-stream digests equal a single-process run."""
import hashlib
import os
import subprocess
import sys

import helpers

WORKER = r'''
import hashlib, os, sys
sys.path.insert(0, os.environ["LH_TESTS"]); sys.path.insert(0, os.environ["LH_PKG"])
import torch, torch.distributed as dist
import helpers, lamehip
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
B = 7                                   # global batch (odd on purpose: blocks of 4 and 3)
lo, hi = lamehip.shard_streams(B, world, rank)
enc = lamehip.Encoder(44100, 128, require_device=False)
orc = helpers.Oracle()
mine = {}
for s in range(lo, hi):
    pcm = helpers.synth_stream(1000 + s, 4000)
    fr = orc.encode_frames(enc.config(), enc.tables(), pcm)
    mine[s] = hashlib.sha256(b"".join(bytes(f) for f in fr)).hexdigest()
gathered = [None] * world
dist.all_gather_object(gathered, mine)   # result collection only; the encode path has no collective
if rank == 0:
    allr = {}
    for g in gathered:
        assert not (set(allr) & set(g)), "shards overlap"
        allr.update(g)
    assert sorted(allr) == list(range(B)), "shards do not cover the batch"
    print("DIGEST " + hashlib.sha256("".join(allr[s] for s in range(B)).encode()).hexdigest())
dist.destroy_process_group()
'''


def test_two_rank_static_sharding_equals_single_process(tmp_path, oracle):
    import lamehip
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, LH_TESTS=os.path.join(helpers.ROOT, "tests"), LH_PKG=helpers.PKG,
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", str(w)],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:]
    digest = [l.split()[1] for l in out.stdout.splitlines() if l.startswith("DIGEST ")]
    assert len(digest) == 1
    enc = lamehip.Encoder(44100, 128, require_device=False)
    single = ""
    for s in range(7):
        fr = oracle.encode_frames(enc.config(), enc.tables(), helpers.synth_stream(1000 + s, 4000))
        single += hashlib.sha256(b"".join(bytes(f) for f in fr)).hexdigest()
    assert hashlib.sha256(single.encode()).hexdigest() == digest[0]


def test_shard_streams_partitions_exactly():
    import lamehip
    for total in (0, 1, 5, 1024, 8192, 8191):
        for world in (1, 2, 3, 8):
            blocks = [lamehip.shard_streams(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            for (a, b), (c, d) in zip(blocks, blocks[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert lamehip.shard_streams(8192, 8, 3) == (3072, 4096)       # BASELINE config[3]: 1024 per GPU
