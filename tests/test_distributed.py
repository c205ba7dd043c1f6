"""world_size-2 `gloo` tests (CPU) of the N>1 path: shard map, packed all-gather collation and counter
all-reduce. No GPU kernels are involved: utterances are independent, so the distributed logic is exactly
this partition + collation."""
import importlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_align(i):
    rng = np.random.default_rng(i)
    n = int(rng.integers(0, 7))
    st = np.sort(rng.integers(0, 500, size=n)) / 50.0
    return st, st + 0.02 * int(rng.integers(1, 9))


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = importlib.import_module("whisper-char-alignment_amd.shard")
    lengths = [(i * 37) % 101 for i in range(n_items)]
    mine = shard.shard_indices(n_items, rank, world, lengths)
    local = {i: _fake_align(i) for i in mine}
    merged = shard.allgather_results(local, device=torch.device("cpu"))
    counters = shard.allreduce_counters(len(mine), 2 * len(mine), 3 * len(mine), device=torch.device("cpu"))
    # the CLI's --save_prediction dicts (infer_ali.py:118-119): every rank's shard must reach rank 0
    preds = shard.gather_predictions({i: dict(starts=[0.0], ends=[0.1 * i], texts=["w%d" % i], starts_hat=local[i][0], ends_hat=local[i][1],
                                              predwords=["w%d" % i, "<|endoftext|>"], fids="utt%d" % i) for i in mine})
    q.put((rank, mine, {k: (v[0].tolist(), v[1].tolist()) for k, v in merged.items()}, counters,
           None if preds is None else {k: (v["fids"], v["ends_hat"].tolist()) for k, v in preds.items()}))
    dist.destroy_process_group()


def _run_world(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(i for _, mine, _, _, _ in outs for i in mine)
    assert owned == list(range(n_items))  # a partition: every utterance exactly once
    sizes = [len(mine) for _, mine, _, _, _ in outs]
    assert max(sizes) - min(sizes) <= 1
    want = {i: tuple(a.tolist() for a in _fake_align(i)) for i in range(n_items)}
    for rank, _, merged, counters, preds in outs:
        assert merged == want  # every rank holds the full, identical collation
        assert counters == (n_items, 2 * n_items, 3 * n_items)
        if rank == 0:  # len(pkl) == len(dataset): nothing is lost with one rank per GPU
            assert sorted(preds) == list(range(n_items))
            assert all(preds[i] == ("utt%d" % i, want[i][1]) for i in range(n_items))
        else:
            assert preds is None
    return outs


def test_shard_and_collate_world2():
    _run_world(2, 23)


def test_shard_and_collate_world4_ragged_and_empty_shard():
    """n_items not divisible by the world size, and fewer items than ranks (a rank with an EMPTY shard must still take
    part in both collectives and contribute nothing)."""
    _run_world(4, 23)
    outs = _run_world(4, 3)
    assert sorted(len(mine) for _, mine, _, _, _ in outs) == [0, 1, 1, 1]


def test_pack_roundtrip_and_single_process_paths():
    shard = importlib.import_module("whisper-char-alignment_amd.shard")
    res = {5: (np.array([0.0, 0.7]), np.array([0.7, 1.38])), 2: (np.zeros(0), np.zeros(0)), 9: (np.array([1.5]), np.array([1.52]))}
    back = shard.unpack_results(shard.pack_results(res))
    assert sorted(back) == [2, 5, 9]
    for k in res:
        assert np.array_equal(back[k][0], res[k][0]) and np.array_equal(back[k][1], res[k][1])
    assert shard.unpack_results(shard.pack_results({})) == {}
    assert shard.allgather_results(res).keys() == res.keys()          # not initialised -> passthrough
    assert shard.allreduce_counters(1, 2, 3) == (1, 2, 3)
    # length-sorted round-robin: longest items are dealt first, one per rank
    idx0 = shard.shard_indices(6, 0, 2, lengths=[1, 9, 3, 7, 5, 2])
    idx1 = shard.shard_indices(6, 1, 2, lengths=[1, 9, 3, 7, 5, 2])
    assert idx0 == [1, 4, 5] and idx1 == [3, 2, 0]
