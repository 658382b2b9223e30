"""Index sharding of a world across ranks (one process per GPU).

With the reference's semantics bodies never interact (src/solver.rs:3 takes ONE
&mut Rigid; src/world.rs:41-42 steps the two bodies independently), so the N-body
world shards by contiguous index range with no data-path collective.
"""


def shard_range(n, rank, world_size):
    """[first, first+count) of rank's contiguous shard; the first n % world_size ranks get one extra body."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank %r / world_size %r" % (rank, world_size))
    base, extra = divmod(n, world_size)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)
