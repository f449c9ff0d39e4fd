/* pk_selftest.h -- test entry points EXPORTED BY libprovekit_hip.so that are not part of the drop-in's API: the host build of the
 * library's __host__ __device__ arithmetic (the very source the kernels compile, so the CPU suite checks it against the oracle without
 * a GPU), the two host-only pieces of the transcript, and the proof RNG.  Declared here, next to the probes, so that
 * include/provekit_hip.h holds only what a binder needs (tests/test_abi.py: product header + this header == the library's exports). */
#ifndef PK_SELFTEST_H
#define PK_SELFTEST_H
#include "provekit_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* self-test (host only, no device)
 * Runs the library's __host__ __device__ arithmetic (the same source the kernels compile) on the CPU:
 * op 0: a*b*2^-256 mod p (ark-ff mul)   1: Skyscraper v2 compress   2: v1 compress   3: from Montgomery
 * 4/5: the lazy 29-bit product / square followed by exact reduction.  n elements of 4 x u64 each. */
/* host-only pieces of the transcript: domain-separator tag, one sponge permutation on canonical (l, r) */
int pk_selftest_keccak_tag(const uint8_t *data, size_t len, uint8_t tag[32]);
int pk_selftest_permute(uint64_t l[4], uint64_t r[4]);
int pk_selftest_arith(int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
/* the proof RNG: one ChaCha block (host; RFC 8439 state layout, words 12-13 = counter, 14-15 = nonce; `rounds` = 20 for the
 * RFC's vectors, 12 is what the library runs -- rand's ThreadRng cipher) and the device draw of n uniform field elements
 * for (seed32, stream): elements 2j, 2j+1 take the first / second 254-bit candidate of blocks (counter j, nonce {stream,
 * attempt}), attempt = 0, 1, ... until the candidate is < p */
int pk_selftest_chacha(const uint8_t key[32], uint64_t counter, uint32_t n0, uint32_t n1, int rounds, uint8_t out[64]);
int pk_selftest_random_fe(pk_ctx *ctx, const uint8_t seed32[32], uint32_t stream, uint64_t *d_out, size_t n);
/* the NTT's register butterfly network (csrc/ntt_regs.hpp dft_regs<le, d>: a lane's 2^le values, le = 2 | 3; radix-2^d DIF over the top d
 * bits of the register index, d <= le; Shoup products by the powers of w_8) on the host: n_groups x 2^le values (any value below 1.2p:
 * the network's input contract) -> the outputs of each group, reduced exactly.  Output order is the network's own (bit-reversed
 * frequency digit).  tw (may be NULL): one multiplier below p per value, applied to the UNREDUCED outputs as the pass kernel applies its
 * twiddles. */
int pk_selftest_dft(const uint64_t *in, const uint64_t *tw, uint64_t *out, int le, int d, size_t n_groups);

/* hooks of the GPU suite (process-wide, 0 = off; the library reads no test switch from the environment): which = 0 the spin bound of a
 * latency-mode gated kernel (to reach its give-up path in milliseconds), 1 microseconds the host sleeps before publishing each gate's
 * challenge (a stalled host thread), 2 non-zero = pk_ctx_create_set takes the RCCL branch for a repeated device (with PK_RCCL_LIB naming the
 * in-process stand-in tests/stub_rccl: real RCCL refuses two ranks on one GPU) */
int pk_selftest_set_hook(int which, long value);

#ifdef __cplusplus
}
#endif
#endif /* PK_SELFTEST_H */
