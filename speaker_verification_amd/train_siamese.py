"""Contrastive fine-tuning step for the Siamese set-up of `/root/reference/train_siamese.py:37-175`
(SURVEY.md 8f-4), one process per GPU.

The reference wraps the model in single-process `torch.nn.DataParallel` (`train_siamese.py:40-41`) and
cannot run as shipped (`c.LAMBDA` / `c.M` do not exist, `Siamese.forward` hard-codes `.cuda()`, Q20).
Here every rank holds a replica, computes the loss of `siamese.Siamese` on its shard of the pair batch,
and the 1.16 M-parameter gradient (4.66 MB) is averaged with ONE flat all-reduce (RCCL over xGMI on the
GPU box, gloo in the CPU tests) before the optimiser step: at this size a single bucket is
latency-bound (~tens of microseconds), so there is nothing to overlap with the backward pass.
"""
import torch
import torch.distributed as dist

from .siamese import Siamese


def allreduce_gradients(model, group=None):
    """Average the gradients of `model` over all ranks with one flat all-reduce."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)
    offset = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[offset:offset + n].view_as(g))
        offset += n


def siamese_train_step(model, criterion, optimizer, cubes_a, cubes_b, same, group=None):
    """One optimisation step on a batch of pairs.
    cubes_a / cubes_b: [n, 1, 20, 80, 40] feature cubes; same: [n] 1.0 when both come from one speaker.
    Embeddings are the 128-d FC5 outputs (`development=False`), the loss is `Siamese.forward`
    (contrastive + LAMBDA * sum of parameter norms).  Returns the local loss value."""
    model.train()
    optimizer.zero_grad(set_to_none=True)
    out_a = model(cubes_a, development=False)
    out_b = model(cubes_b, development=False)
    loss = criterion(model, same, out_a, out_b)
    loss.backward()
    allreduce_gradients(model, group)
    optimizer.step()
    return float(loss.detach())


def make_criterion(LAMBDA=0.001, M=2.0):
    return Siamese(LAMBDA, M)
