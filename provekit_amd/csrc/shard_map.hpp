// shard_map.hpp -- where a leaf of a commit sharded over the GPUs of a node lives: ONE definition, compiled for the
// device (tree.hip's kernels) and for the host (pk_shard_of_leaf / pk_shard_interleave_digests, which the CPU test-suite and
// host-side callers use), so the two cannot disagree.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace pk {

// leaf i -> the rank that encodes, hashes and later opens it, and its row in that rank's local codeword
__host__ __device__ __forceinline__ unsigned shard_rank_of_leaf(size_t i, unsigned G) { return (unsigned)(i % G); }
__host__ __device__ __forceinline__ size_t shard_local_row(size_t i, unsigned G) { return i / G; }
// an all-gather delivers rank g's `rows / G` local digests as block g: leaf i is element ...
__host__ __device__ __forceinline__ size_t shard_gathered_slot(size_t i, size_t rows, unsigned G) {
    return (size_t)shard_rank_of_leaf(i, G) * (rows / G) + shard_local_row(i, G);
}


// ---- the inner tree of a sharded commit (first option) -------------------------------------------------------
// Every rank holds ALL leaf digests after the all-gather.  For trees with at least SUBTREE_MIN_LOCAL leaves per rank, rank g
// builds only the contiguous subtree over leaves [g rows/G, (g+1) rows/G) -- heap nodes [c + g c/G, c + (g+1) c/G) of every
// level of c >= G nodes -- the G subtree roots (heap slots G .. 2G-1) are all-gathered and the top log2 G levels are hashed
// on every rank.  A pure function of (G, rows): commit and every later opening take the same decision.  Below the
// threshold the whole inner tree is built on every rank (the exchange would cost more than the hashing it saves).
constexpr size_t SUBTREE_MIN_LOCAL = (size_t)1 << 13;
__host__ __device__ __forceinline__ bool subtree_sharded(unsigned G, size_t rows) { return G > 1 && rows / G >= SUBTREE_MIN_LOCAL; }
// heap node x (1 <= x < 2 rows) of a subtree-sharded tree -> the rank that holds it; nodes above the subtree roots and the
// roots themselves (x < 2G) are replicated: rank 0 answers for them
__host__ __device__ __forceinline__ unsigned subtree_owner_of_node(size_t x, unsigned G) {
    if (x < 2 * (size_t)G) return 0;
    size_t c = 1;
    while (2 * c <= x) c <<= 1;  // the level of x holds c nodes, heap slots [c, 2c)
    return (unsigned)((x - c) / (c / G));
}

}  // namespace pk
