"""Multi-GPU: one process per GPU, utterances sharded across ranks, ONE
collective -- an all-gather of the embedding shards (RCCL over xGMI on the GPU
box, gloo in the CPU tests) -- before scoring.  SURVEY.md 8(e).

The reference has no distributed code at all (only in-process
`torch.nn.DataParallel` in its trainers, train.py:40-41); utterances are
independent until scoring, so the partition is contiguous ranges and there is
no other data-path exchange.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world_size, rank):
    """Contiguous range [lo, hi) of rank `rank`: ceil(n / world) per rank, the last
    ranks may get fewer (148 642 over 8 -> 18 581 x 7 + 18 575)."""
    per = -(-n_items // world_size)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def shard_rows(n_items, world_size):
    """Rows every rank contributes to the all-gather (shards are padded to this)."""
    return -(-n_items // world_size)


def all_gather_embeddings(local, n_total, group=None):
    """local: [n_local, D] embeddings of this rank's contiguous shard.
    Returns [n_total, D] on every rank.  Shards are zero-padded to equal row
    counts so a single `all_gather_into_tensor` moves everything."""
    if not (dist.is_available() and dist.is_initialized()):
        return local[:n_total]
    world = dist.get_world_size(group)
    rows = shard_rows(n_total, world)
    send = local
    if local.shape[0] != rows:
        send = torch.zeros((rows, local.shape[1]), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of N ranks on one GPU (RCCL refuses two ranks on a device): gloo moves host memory
        parts = [torch.empty((rows, local.shape[1]), dtype=local.dtype) for _ in range(world)]
        dist.all_gather(parts, send.cpu().contiguous(), group=group)
        return torch.cat(parts)[:n_total].to(local.device)
    recv = torch.empty((world * rows, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    # ranks hold consecutive ranges of `rows` items, so the valid rows are a prefix
    return recv[:n_total]


def sharded_embed(embed_fn, items, n_total=None, group=None):
    """Run `embed_fn` on this rank's contiguous shard of `items` (anything sliceable
    along dim 0) and all-gather the result."""
    n_total = len(items) if n_total is None else n_total
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bounds(n_total, world, rank)
    local = embed_fn(items[lo:hi])
    return all_gather_embeddings(local, n_total, group)
