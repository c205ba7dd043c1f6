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


def _bench(args, env_extra):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_n_self_launches_n_ranks():
    """`python bench.py --gpus 2` run plainly (the driver's command shape, WORLD_SIZE unset) must start 2 ranks itself, run the
    product's collation over the process group and print exactly ONE line with n_gpus == 2 (gloo + --dry-run: no GPU here)."""
    r, recs = _bench(["--gpus", "2", "--steps", "3", "--batch", "5", "--warmup", "0", "--dry-run"], {"WCA_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(recs) == 1, r.stdout
    d = recs[0]
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["scaling"] == "weak"
    cfg = d["config"]
    assert cfg["dist_ranks"] == 2 and cfg["dist_backend"] == "gloo" and cfg["collated_utterances"] == 2 * 3 * 5
    assert cfg["collective_calls"]["all_gather"] == 2 and cfg["collective_calls"]["all_reduce"] == 1


def test_bench_gpus_n_refuses_when_devices_are_missing():
    """RCCL needs one device per rank: with fewer than N GPUs visible the launcher must fail loudly, not benchmark one GPU and
    label it (this container has none)."""
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible")
    r, recs = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0 and not recs
    assert "GPU(s) visible" in r.stderr


def test_force_dist_keeps_the_collectives_for_one_rank(monkeypatch):
    """WCA_FORCE_DIST=1: a single rank still runs all-gather / all-reduce (the GPU test drives exactly this over RCCL)."""
    shard = importlib.import_module("whisper-char-alignment_amd.shard")
    monkeypatch.setenv("WCA_FORCE_DIST", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        before = dict(shard.COLLECTIVE_CALLS)
        res = {3: (np.array([0.1, 0.5]), np.array([0.5, 0.9])), 0: (np.zeros(0), np.zeros(0))}
        back = shard.allgather_results(res, device=torch.device("cpu"))
        assert sorted(back) == [0, 3] and np.array_equal(back[3][1], res[3][1])
        assert shard.allreduce_counters(1, 2, 3, device=torch.device("cpu")) == (1, 2, 3)
        assert shard.COLLECTIVE_CALLS["all_gather"] == before["all_gather"] + 2
        assert shard.COLLECTIVE_CALLS["all_reduce"] == before["all_reduce"] + 1
        monkeypatch.delenv("WCA_FORCE_DIST")
        shard.allgather_results(res, device=torch.device("cpu"))
        assert shard.COLLECTIVE_CALLS["all_gather"] == before["all_gather"] + 2   # passthrough again
    finally:
        dist.destroy_process_group()


def test_abi_collation_retry_is_collective():
    """wca_allgather_results decides "does every shard fit" on the {size, capacity} pairs of ALL ranks
    (wca_collate_plan), so ranks with unequal shards and unequal first capacity guesses take the same branch and issue the same
    sequence of collectives. Two ranks are emulated by two threads that run the PRODUCT's WhisperAMD.allgather_packed against a stand-in
    for the two RCCL calls (a barrier-synchronised exchange) whose fit decision is the library's own wca_collate_plan: shards of
    160 000 and 100 000 bytes (capacities 160 000 and 100 000 on the first attempt: the old per-caller test let rank 0 go on to the
    payload all-gather while rank 1 retried the size gather)."""
    import ctypes as C
    import threading
    import types
    wca = importlib.import_module("whisper-char-alignment_amd")
    lib = wca._lib.load()
    i64 = C.c_int64
    # the pure decision, straight from libwca.so
    pad = i64(0)
    assert lib.wca_collate_plan((i64 * 2)(160000, 100000), (i64 * 2)(160000, 100000), 2, C.byref(pad)) == wca._lib.ERR_TOO_LONG and pad.value == 160000
    assert lib.wca_collate_plan((i64 * 2)(160000, 100000), (i64 * 2)(160000, 160000), 2, C.byref(pad)) == 0 and pad.value == 160000
    assert lib.wca_collate_plan((i64 * 2)(0, 0), (i64 * 2)(65536, 65536), 2, C.byref(pad)) == 0 and pad.value == 0
    assert lib.wca_collate_plan((i64 * 3)(5, 70000, 1), (i64 * 3)(65536, 70000, 65536), 3, C.byref(pad)) == wca._lib.ERR_TOO_LONG

    world = 2
    bar = threading.Barrier(world)
    board = {}
    log = {r: [] for r in range(world)}

    class FakeLib:
        """wca_allgather_results with the two ncclAllGather calls replaced by a barrier exchange between the threads."""
        def __init__(self, rank):
            self.rank = rank

        def wca_allgather_results(self, _h, packed_ptr, n_bytes, out_ptr, cap, sizes):
            r = self.rank
            log[r].append("gather {size, capacity}")
            board[("pair", r)] = (int(n_bytes), int(cap))
            bar.wait(timeout=30)
            pairs = [board[("pair", i)] for i in range(world)]
            bar.wait(timeout=30)
            for i in range(world):
                sizes[i] = pairs[i][0]
            pad_ = i64(0)
            rc = lib.wca_collate_plan((i64 * world)(*[p_[0] for p_ in pairs]), (i64 * world)(*[p_[1] for p_ in pairs]), world, C.byref(pad_))
            if rc != 0:
                return rc
            if pad_.value == 0:
                return 0
            log[r].append("gather payload %d" % pad_.value)
            board[("data", r)] = bytes((C.c_uint8 * int(n_bytes)).from_address(packed_ptr.value)) if n_bytes else b""
            bar.wait(timeout=30)
            for i in range(world):
                d = board[("data", i)]
                C.memmove(out_ptr.value + i * int(cap), d, len(d))
            bar.wait(timeout=30)
            return 0

    shards = [np.arange(160000, dtype=np.uint32).astype(np.uint8), (np.arange(100000, dtype=np.uint32) * 7).astype(np.uint8)]
    results, errors = {}, []

    def run(rank):
        try:
            me = types.SimpleNamespace(_comm=(rank, world), _lib=FakeLib(rank), _h=None, _bind_stream=lambda: None)
            results[rank] = wca.WhisperAMD.allgather_packed(me, shards[rank])
        except Exception as exc:  # noqa: BLE001
            errors.append((rank, repr(exc)))
            bar.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert not errors, errors
    assert log[0] == log[1] == ["gather {size, capacity}", "gather {size, capacity}", "gather payload 160000"], log
    for r in range(world):
        assert all(np.array_equal(results[r][i], shards[i]) for i in range(world))
