"""Multi-GPU story of the inference path: whole videos are independent (reference inference_utils.py:28-48 resets
every piece of state at a video boundary), so they are dealt to GPUs with no data-path collective.
Longest-processing-time-first on the frame count keeps the makespan tight (DAVIS clip lengths vary ~3x)."""


def lpt_assign(lengths, n_shards):
    """lengths: {video: n_frames}.  Returns a list of n_shards lists of video names (deterministic)."""
    shards = [[] for _ in range(n_shards)]
    load = [0] * n_shards
    for name, n in sorted(lengths.items(), key=lambda kv: (-kv[1], kv[0])):
        i = min(range(n_shards), key=lambda s: (load[s], s))
        shards[i].append(name)
        load[i] += n
    return shards, load


def shard_for_rank(lengths, rank, world):
    shards, _ = lpt_assign(lengths, world)
    return sorted(shards[rank])
