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

}  // namespace pk
