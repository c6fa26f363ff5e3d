"""Contiguous-chunk partition of a nucleotide buffer over GPUs.

Word w depends only on nt [32w, 32w+32) (n_to_bits.rs:39-42), so the path shards into
independent units with no exchange step: GPU k of G gets nt [k*C, min(N,(k+1)*C)) with
C = ceil(N/G) rounded up to `gran` (a whole number of kernel tiles, hence of words), every
shard starts on a word boundary and only the last one has a tail.  The same arithmetic lives
in csrc/cute_nt.hip (cnt_n_to_bits_sharded); bench.py uses this copy for its one-process-
per-GPU launch.  No collective anywhere on the data path.
"""

SHARD_GRAN_NT = 16384  # one DIRECT/U=4 tile; multiple of 32


def shard_size(n_len, world, gran=SHARD_GRAN_NT):
    if world <= 0:
        raise ValueError("world must be positive")
    per = -(-n_len // world)
    return -(-per // gran) * gran


def partition(n_len, world, gran=SHARD_GRAN_NT):
    """[(lo, hi)] per rank, in nucleotides; empty shards are (n_len, n_len)."""
    per = shard_size(n_len, world, gran)
    return [(min(n_len, per * k), min(n_len, per * (k + 1))) for k in range(world)]


def word_range(lo, hi):
    """Packed-word range [w_lo, w_hi) a shard [lo, hi) of nucleotides owns (lo % 32 == 0)."""
    if lo % 32:
        raise ValueError("shard must start on a word boundary")
    return lo // 32, (hi + 31) // 32
