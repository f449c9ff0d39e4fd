/* pk_probes.h -- libpk_probes.so: measurement probes and rejected prototypes, built from the product's headers and linked against
 * libprovekit_hip.so, but NOT part of the drop-in (nothing here is for a binder of include/provekit_hip.h).  Same conventions:
 * int status (0 ok, PK_ERR_*), pk_ctx from pk_ctx_create. */
#ifndef PK_PROBES_H
#define PK_PROBES_H
#include "provekit_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* the same ops run by a kernel on device buffers (device-vs-host codegen diff in the GPU suite) */
int pk_probe_arith_device(pk_ctx *ctx, int op, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n);
/* measurement aid (SURVEY 8d "measured_peak_modmul_per_s"): rate of register-resident 9x29-bit Montgomery squarings,
 * ilp (1|2|4) independent chains per lane, waves_per_simd (1..8) resident waves, iters squarings per chain */
int pk_probe_modmul_rate(pk_ctx *ctx, unsigned waves_per_simd, unsigned ilp, unsigned iters, double *modmul_per_s);
/* VERDICT r05 item 3, the ceiling of a hand-allocated Skyscraper square round by ABLATION: rounds per second of the product's round
 * (variant 0) and of the same round with the instructions assembly could fold away left out (1: the 8 additions of the round constant;
 * 2: also the zero-extensions of r into the 64-bit columns; 3: also the 8 doublings) -- wrong results, the product's multiply-adds and
 * dependency structure (tools/probes/probes.hip sq_round_ablated, tools/sq_round_ablation.py, profiles/r06_sq_round_ablation.json) */
int pk_probe_sq_round_rate(pk_ctx *ctx, int variant, unsigned waves_per_simd, unsigned ilp, unsigned iters, double *rounds_per_s);
/* PROTOTYPE, not on the product path (tools/probes/fe52.hpp): the reference's f64-FMA Montgomery square on 5 x 52-bit limbs
 * (skyscraper/block-multiplier/src/portable_simd.rs:17-196, utils.rs:66-147, constants.rs:100-133; round-toward-zero as
 * fp-rounding/src/lib.rs:57-78).  a: n values < 2^256 (4 x u64 each) -> out5: n x 5 limbs of x^2 * 2^-260 mod p, lazily
 * reduced (< 2^257).  Host: under fesetround(FE_TOWARDZERO); device: MODE.FP_ROUND set by the kernel; _rate_fp52: the same
 * probe as pk_probe_modmul_rate for this multiplier, so the two can be compared on one box. */
int pk_probe_fp52_sqr(const uint64_t *a, uint64_t *out5, size_t n);
/* PROTOTYPE probe, not on the product path: the wavefront-cooperative Skyscraper square round (limbs in lanes 0..8, v_readlane
 * broadcasts, DPP window shift) next to the lone lane's round (skyscraper29s.hpp sky_sq_round_s), `iters` rounds each on one
 * wavefront of an idle GPU: out[36] = coop l, coop r, lane l, lane r (9 x 29-bit limbs each); cycles[0..1] = s_memtime ticks of the
 * two loops inside one launch, cycles[2..3] = nanoseconds of each loop in a launch of its own (hipEvents).  north_star's
 * "one-wavefront-per-node Skyscraper rounds", measured. */
int pk_probe_coop_round(pk_ctx *ctx, const uint32_t l[9], const uint32_t r[9], unsigned iters, uint32_t out[36],
                           uint64_t cycles[4]);
int pk_probe_fp52_sqr_device(pk_ctx *ctx, const uint64_t *d_a, uint64_t *d_out5, size_t n);
int pk_probe_modmul_rate_fp52(pk_ctx *ctx, unsigned waves_per_simd, unsigned ilp, unsigned iters, double *modmul_per_s);
/* modular reduction as a constant-matrix product on the matrix core (v_mfma_i32_16x16x64_i8 over the
 * 8/8/8/5-bit digits a 29-bit limb already holds; tools/probes/probes.hip "Reduction on the matrix core") -- measured and
 * NOT adopted.  pk_probe_mfma_reduce: the product itself, exact (d_t_limbs: n x 18 limbs of 29 bits, d_out: n x 36 column sums
 * with sum_i out[i] 2^(29 (i/4) + 8 (i%4)) == t 2^-256 mod p, non-negative, < 2^270).  The two rate probes return squarings per
 * second: the matrix pipe fed for free, and the vector work that remains with the matrix products and lane movement free. */
/* one Fiat-Shamir round trip as pk_prove makes it (launch + stream synchronisation) against the same round trip
 * through a persistent kernel's pinned mailbox, `host_work_permutes` sponge permutations of host work in between; microseconds per
 * round over `rounds` dependent round trips (tools/roundtrip.py, profiles/r04_roundtrip.json). */
/* products by a constant per second, register-resident chains: the Montgomery product (mont261_29, shoup = 0) against the Shoup form
 * with a precomputed quotient (shoup261_29, shoup = 1: 143 multiply-adds instead of 162 + 9); csrc/fe29.hpp */
int pk_probe_constmul_rate(pk_ctx *ctx, unsigned waves_per_simd, unsigned ilp, unsigned iters, int shoup, double *modmul_per_s);
int pk_probe_roundtrip(pk_ctx *ctx, unsigned rounds, unsigned host_work_permutes, double *us_per_round_launch,
                          double *us_per_round_mailbox);
/* `launches` dependent one-wavefront kernels back to back on the context's stream, one synchronisation at the end: microseconds per
 * kernel boundary with no host in the loop */
int pk_probe_launch_chain(pk_ctx *ctx, unsigned launches, unsigned threads_per_launch, double *us_per_launch);
int pk_probe_mfma_reduce(pk_ctx *ctx, const uint32_t *d_t_limbs, int32_t *d_out, size_t n);
int pk_probe_mfma_reduce_rate(pk_ctx *ctx, unsigned waves_per_simd, unsigned iters, double *squarings_per_s);
int pk_probe_mfma_valu_rate(pk_ctx *ctx, unsigned waves_per_simd, unsigned ilp, unsigned iters, double *squarings_per_s);

#ifdef __cplusplus
}
#endif
#endif
