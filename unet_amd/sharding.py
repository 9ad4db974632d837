"""Data-parallel partitioning of a batch of utterances.

Eval-mode forwards have no cross-sample operation (BatchNorm uses running
statistics), so the batch splits into contiguous row ranges, one per rank
(one process per GPU), with no collective in the data path; weights are
regenerated or loaded identically on every rank.
"""


def shard_batch(total, rank, world):
    """[lo, hi) rows of rank ``rank`` of ``world``: contiguous, sizes differ by <= 1."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank {} of world {}".format(rank, world))
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
