"""The N>1 path on CPU: world_size-2 gloo run of the contiguous-chunk partition bench.py uses
(cute_nucleotides_amd/sharding.py).  Each rank encodes ITS chunk only (here with the oracle
standing in for the GPU kernel -- this test is about the partition, not the kernel), there is
no data-path collective, and the concatenation of the per-rank outputs must equal the encode
of the whole buffer."""
import os
import socket

import numpy as np
import pytest

from cute_nucleotides_amd import sharding


def test_partition_properties():
    for n_len in (0, 1, 31, 32, 16384, 16385, 100003, (1 << 20) + 13, 1 << 34, (1 << 36) + 7):
        for world in (1, 2, 3, 4, 8):
            parts = sharding.partition(n_len, world)
            assert len(parts) == world
            assert parts[0][0] == 0 and parts[-1][1] == n_len
            for (lo, hi), (lo2, _) in zip(parts, parts[1:]):
                assert hi == lo2
            for lo, hi in parts:
                assert lo <= hi
                if hi > lo:  # empty shards sit at (n_len, n_len)
                    assert lo % 32 == 0 and lo % sharding.SHARD_GRAN_NT == 0
            # only the last non-empty shard may have a ragged tail
            nonempty = [(lo, hi) for lo, hi in parts if hi > lo]
            for lo, hi in nonempty[:-1]:
                assert (hi - lo) % sharding.SHARD_GRAN_NT == 0
            # word ranges are disjoint and cover ceil(n/32)
            wr = [sharding.word_range(lo, hi) for lo, hi in nonempty]
            for (a, b), (c, d) in zip(wr, wr[1:]):
                assert b == c
            if nonempty:
                assert wr[0][0] == 0 and wr[-1][1] == (n_len + 31) // 32
    # weak scaling in bench.py: world x 2^34 -> every rank gets exactly 2^34
    assert sharding.partition(8 << 34, 8) == [(k << 34, (k + 1) << 34) for k in range(8)]
    with pytest.raises(ValueError):
        sharding.word_range(5, 64)


def _worker(rank, world, port, n_len, seed, q):
    import torch
    import torch.distributed as dist

    from oracle import cnt_oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.partition(n_len, world)[rank]
        assert (lo, hi) == sharding.shard_range_c(n_len, world, rank)  # the C library cuts at the same places
        mine = orc.fill_random_acgt(hi - lo, seed, first_nt=lo)  # each rank generates only its chunk
        bits = orc.n_to_bits_lut(mine)
        w_lo, w_hi = sharding.word_range(lo, hi)
        assert bits.size == w_hi - w_lo
        back = orc.bits_to_n_lut(bits, hi - lo)
        ok = torch.tensor([1 if np.array_equal(back, mine) else 0])
        dist.barrier()  # the only collectives: barrier + a MIN/MAX of scalars, as in bench.py
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, w_lo, bits.tobytes(), int(ok.item()), float(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_len", [(1 << 18) + 77, 16384 * 3])
def test_two_rank_gloo_shards_concatenate_to_the_whole(oracle, n_len):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, seed = 2, 1234
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_len, seed, q)) for r in range(world)]
    [p.start() for p in procs]
    results = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    whole = oracle.n_to_bits_lut(oracle.fill_random_acgt(n_len, seed))
    out = np.zeros_like(whole)
    for rank, w_lo, raw, ok, tmax in results:
        part = np.frombuffer(raw, dtype=np.uint64)
        out[w_lo : w_lo + part.size] = part
        assert ok == 1 and tmax == float(world)
    assert np.array_equal(out, whole)
