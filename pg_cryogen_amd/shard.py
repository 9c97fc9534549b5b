"""Round-robin sharding of independent cryo blocks over the GPUs of one node
(BASELINE.json north_star: "round-robin dispatch, no RCCL collective required").
Block i of a job belongs to rank i mod N."""


def owner(block_index, world_size):
    return block_index % world_size


def my_blocks(n_total, rank, world_size):
    """global block indices handled by `rank`"""
    return range(rank, n_total, world_size)


def my_count(n_total, rank, world_size):
    return len(my_blocks(n_total, rank, world_size))
