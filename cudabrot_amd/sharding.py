"""Sample-stream sharding across the GPUs of one node.

The reference is single-GPU (``-d`` picks one device, cudabrot.cu:155).  Samples are independent and the
only shared state is the additive integer histogram, so the single seed-1337 XORWOW stream is sharded by
generator subsequence: rank r of N owns subsequences [r*T, (r+1)*T) and a full-resolution private
histogram; an N-GPU run of P passes is bit-identical to one GPU running N*T threads for P passes.  No
collective is on the data path -- the histograms are summed once (integer sum: order-free, exact) when
a checkpoint or the final image is written.
"""


def shard_subsequences(rank, world_size, threads_per_rank):
    """First subsequence id and thread count of ``rank``."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return rank * threads_per_rank, threads_per_rank


def reduce_histogram(hist_tensor, dst=0):
    """Sum the per-rank int64 histograms onto ``dst`` (torch.distributed; RCCL on GPUs, gloo on CPU).

    Counters are u64 carried as int64 bit patterns: two's-complement addition is the same bits.
    """
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(hist_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return hist_tensor
