"""Utterance-level data parallelism: the `.scp` list (one utterance per line, dataset.py:25-31) shards
embarrassingly across ranks -- one process per GPU, no collective on the data path. Only the final
collation talks: one all-gather of packed per-utterance results and one all-reduce of the three scalar
evaluation counters the reference accumulates serially (infer_ali.py:53-55,123-132).
On MI355X the backend is "nccl" (= RCCL over xGMI); the same code runs over "gloo" in CPU tests.
"""
import os

import numpy as np
import torch


# collectives actually issued by this process (bench.py reports them: proof that the RCCL path ran, not the passthrough)
COLLECTIVE_CALLS = {"all_gather": 0, "all_reduce": 0, "gather_object": 0}


def _single_process(dist):
    """No process group, or one rank: nothing to exchange. WCA_FORCE_DIST=1 keeps the collectives even for one rank, so that the
    RCCL all-gather / all-reduce path itself can be exercised on a 1-GPU box."""
    if not dist.is_available() or not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and os.environ.get("WCA_FORCE_DIST") != "1"


def shard_indices(n_items, rank, world, lengths=None):
    """Indices owned by `rank`: item i of the (optionally length-sorted, longest first) order goes to
    rank i % world, which balances decoder / DTW work across GPUs."""
    if lengths is not None:
        order = np.argsort(-np.asarray(lengths), kind="stable")
    else:
        order = np.arange(n_items)
    return [int(i) for i in order[rank::world]]


def pack_results(results):
    """results: {utt_index: (start_times f64[n], end_times f64[n])} -> uint8 buffer
    [utt_index:int32][n:int32][starts:f64*n][ends:f64*n] ... (words are re-derivable from the text)."""
    chunks = []
    for idx in sorted(results):
        st, en = results[idx]
        st = np.ascontiguousarray(st, dtype=np.float64)
        en = np.ascontiguousarray(en, dtype=np.float64)
        assert st.shape == en.shape and st.ndim == 1
        chunks.append(np.array([idx, len(st)], dtype=np.int32).tobytes())
        chunks.append(st.tobytes())
        chunks.append(en.tobytes())
    return np.frombuffer(b"".join(chunks), dtype=np.uint8).copy()


def unpack_results(buf):
    out, pos = {}, 0
    raw = np.asarray(buf, dtype=np.uint8).tobytes()
    while pos < len(raw):
        idx, n = np.frombuffer(raw, dtype=np.int32, count=2, offset=pos)
        pos += 8
        st = np.frombuffer(raw, dtype=np.float64, count=int(n), offset=pos).copy()
        pos += 8 * int(n)
        en = np.frombuffer(raw, dtype=np.float64, count=int(n), offset=pos).copy()
        pos += 8 * int(n)
        out[int(idx)] = (st, en)
    return out


def allgather_results(local_results, device=None, engine=None):
    """Collates every rank's {utt_index: (starts, ends)} on all ranks. Two collectives: an all-gather of
    the packed byte counts, then one all-gather of the buffers padded to the maximum count.
    engine: a WhisperAMD whose RCCL communicator is up (engine.comm_init): the C ABI's wca_allgather_results does both
    collectives (ncclAllGather straight from libwca.so); default: torch.distributed on the initialised process group."""
    if engine is not None and engine.comm is not None:
        out = {}
        for part in engine.allgather_packed(pack_results(local_results)):
            out.update(unpack_results(part))
        COLLECTIVE_CALLS["all_gather"] += 2
        return out
    import torch.distributed as dist
    if _single_process(dist):
        return dict(local_results)
    world = dist.get_world_size()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl"
                                             else torch.device("cpu"))
    packed = torch.from_numpy(pack_results(local_results)).to(dev)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([packed.numel()], dtype=torch.int64, device=dev))
    COLLECTIVE_CALLS["all_gather"] += 2
    mx = int(sizes.max().item())
    padded = torch.zeros(max(mx, 1), dtype=torch.uint8, device=dev)
    padded[:packed.numel()] = packed
    gathered = torch.zeros(world * max(mx, 1), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, padded)
    gathered = gathered.cpu().numpy().reshape(world, max(mx, 1))
    out = {}
    for r in range(world):
        out.update(unpack_results(gathered[r, :int(sizes[r].item())]))
    return out


def allreduce_counters(corrects, total_preds, total_gts, device=None, engine=None):
    """Sums the evaluation counters over ranks (they are plain python ints in the reference)."""
    if engine is not None and engine.comm is not None:
        COLLECTIVE_CALLS["all_reduce"] += 1
        return engine.allreduce_counters(corrects, total_preds, total_gts)
    import torch.distributed as dist
    if _single_process(dist):
        return corrects, total_preds, total_gts
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl"
                                             else torch.device("cpu"))
    t = torch.tensor([corrects, total_preds, total_gts], dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    COLLECTIVE_CALLS["all_reduce"] += 1
    return tuple(int(v) for v in t.tolist())


def gather_predictions(local_predictions, dst=0):
    """Every rank's `all_predictions` dict ({utt index: {starts, ends, texts, starts_hat, ends_hat, predwords, fids}},
    infer_ali.py:118-119) merged on rank `dst` (returns None on the other ranks). The reference is single-process; with
    one rank per GPU each rank only holds its shard, and eval_ali.py must see the whole corpus."""
    import torch.distributed as dist
    if _single_process(dist):
        return dict(local_predictions)
    rank, world = dist.get_rank(), dist.get_world_size()
    bucket = [None] * world if rank == dst else None
    dist.gather_object(dict(local_predictions), bucket, dst=dst)
    COLLECTIVE_CALLS["gather_object"] += 1
    if rank != dst:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    return dict(sorted(merged.items()))
