// witness.hip -- the R1CS witness builders on the device ("witness-builder solve on GPU").
//
// What it replaces: R1CSSolver::solve_witness_vec (provekit/prover/src/r1cs.rs:29-40), i.e. the sequential loop over
// `&[WitnessBuilder]` calling WitnessBuilderSolver::solve (provekit/prover/src/witness/witness_builder.rs:27-193) with its two
// helpers, DigitalDecompositionWitnessesSolver (witness/digits.rs:12-59) and SpiceWitnessesSolver (witness/ram.rs:13-47).  ACVM
// execution (NoirProofSchemeProver::generate_witness) stays on the host, as `north_star` asks: its result -- the ACIR witness map --
// is this module's input, next to the Fiat-Shamir challenges the host transcript draws (WitnessBuilder::Challenge reads the
// transcript and nothing else, so all challenges of a proof are known before the first builder runs).
//
// The reference runs the builders strictly in order; each may read any witness solved earlier.  Here the list is levelled once
// per scheme -- level(b) = 1 + max level of the builders that produce b's inputs -- and a proof runs one launch per level (runs of
// narrow levels share a launch: one workgroup walks them with a barrier in between).  A builder that writes many witnesses
// (digital decompositions, multiplicities, the Spice memory model) is expanded into one work item per written witness.
//   * Inverse: x^(p-2), 380 Montgomery products per element, one lane each (the reference inverts one by one as well).
//   * Multiplicities: a histogram by atomics in the level of the builder, the counts converted to field elements one phase later.
//   * Spice (read/write memory checking): the reference replays the memory operations one by one; here the operations of a block
//     are sorted by (address, position) -- rocPRIM's radix sort on 64-bit keys -- after which "the previous operation on my
//     address" is the left neighbour: read timestamps, old values and the final values / timestamps are all independent.
// The builder list arrives as the postcard bytes of `Vec<WitnessBuilder>` (provekit/common/src/witness/witness_builder.rs:33-117:
// the enum and every struct it contains are defined in the reference tree, so the serde layout is pinned by source; it is how the
// list sits inside a `.nps`).  Everything is exact field arithmetic: outputs are bit-identical to the reference's Vec<Option<F>>
// (unset entries are reported through the `is_set` mask; fill_witness's random filling stays with the caller).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <string>
#include <vector>

#define PK_BASE_PRIO 2
#include "ctx.hpp"
#include "fe29.hpp"
#include "feinv.hpp"
#include "reduce.hpp"

using namespace pk;

namespace {

enum : u32 {
    OP_CONST = 0, OP_ACIR, OP_SUM, OP_PRODUCT, OP_CHALLENGE, OP_IDX_LOGUP, OP_INVERSE, OP_PROD_LINEAR, OP_LOGUP, OP_SPICE_FACTOR,
    OP_BINOP_DENOM, OP_DIGIT, OP_DIGIT_CHECK, OP_HIST_RANGE, OP_HIST_BINOP, OP_COUNT_OUT
};
constexpr u32 NONE = 0xffffffffu;
constexpr u32 COW_CONST = 0x80000000u;  // ConstantOrR1CSWitness packed in one word: constant-table index | COW_CONST, or a witness index

// one written witness (or one check / histogram contribution)
struct WbItem {
    u32 op, out;
    u32 w[5];  // witness indices (meaning per op, see wb_eval)
    u32 k[4];  // constant-table indices, bit offsets, sizes
    u32 builder;  // index in the reference's list: error reports name it
};

// device-side error record: the lowest (builder << 4 | code) seen
enum : u32 { ERR_INVERSE_ZERO = 1, ERR_DIGIT_OVERFLOW = 2, ERR_MULTIPLICITY_RANGE = 3, ERR_SPICE_ADDRESS = 4 };

__device__ __forceinline__ void report(unsigned long long* err, u32 builder, u32 code) { atomicMin(err, ((unsigned long long)builder << 4) | code); }

__device__ __forceinline__ fe small_fe(u64 v) {  // FieldElement::from(u64)
    fe c = fe_zero();
    c.v[0] = (u32)v;
    c.v[1] = (u32)(v >> 32);
    return fe_to_montx(c);
}
// bits [start, start + len) of a canonical value as a field element (witness/digits.rs:33-59, le_bits_to_field)
__device__ __forceinline__ fe take_bits(const fe& canon, u32 start, u32 len) {
    fe r = fe_zero();
    for (u32 o = 0; o < 8; o++) {
        const u32 bit = start + 32 * o;
        if (32 * o >= len || bit >= 256) break;
        const u32 wi = bit >> 5, sh = bit & 31;
        u32 word = canon.v[wi] >> sh;
        if (sh && wi + 1 < 8) word |= canon.v[wi + 1] << (32 - sh);
        const u32 left = len - 32 * o;
        if (left < 32) word &= (1u << left) - 1u;
        r.v[o] = word;
    }
    return fe_to_montx(r);
}
__device__ __forceinline__ bool bits_above_zero(const fe& canon, u32 from) {  // all bits >= from are zero?
    for (u32 wi = 0; wi < 8; wi++) {
        const u32 lo = 32 * wi;
        if (lo + 32 <= from) continue;
        const u32 word = lo >= from ? canon.v[wi] : (canon.v[wi] >> (from - lo));
        if (word) return false;
    }
    return true;
}
// operand.inverse() (witness_builder.rs:66-69): safegcd on one lane, ~7x shorter than x^(p-2) (feinv.hpp); a level of the list
// costs at least one inversion's latency, so this is what the deep-but-wide lists are bound by
__device__ __forceinline__ fe fe_inverse(const fe& x) { return fe_inverse_mont(x); }
// counts[idx] += 1 for every active lane, the lanes of a wavefront that hit the same counter adding once: lookups are skewed in
// practice (the high bytes of small numbers are zero), and 10^6 atomics on ONE address serialise at the L2 (12 ms measured for 2^20
// lookups of one value, 0.5 ms when they are spread over 256)
__device__ __forceinline__ void count_one(u32* __restrict__ counts, u32 idx) {
    // which active lanes hold my idx: one ballot per bit of the counter index (< 2^28, build_program's cap on the tables)
    unsigned long long same = __ballot(1);
#pragma unroll
    for (int b = 0; b < 28; b++) {
        const bool bit = (idx >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        same &= bit ? m : ~m;
    }
    const unsigned lane = threadIdx.x & 63u;
    if (lane == (unsigned)(__ffsll((long long)same) - 1)) atomicAdd(counts + idx, (u32)__popcll(same));
}
__device__ __forceinline__ fe cow(const fe* __restrict__ W, const fe* __restrict__ K, u32 packed) {
    return (packed & COW_CONST) ? fe_load(K + (packed & ~COW_CONST)) : fe_load(W + packed);
}

__device__ void wb_eval(const WbItem& it, fe* __restrict__ W, unsigned char* __restrict__ is_set, const fe* __restrict__ K, const fe* __restrict__ acir,
                        const fe* __restrict__ challenges, const u32* __restrict__ extra, u32* __restrict__ counts, unsigned long long* err) {
    fe r = fe_zero();
    switch (it.op) {
        case OP_CONST: r = fe_load(K + it.k[0]); break;                       // witness_builder.rs:35-37
        case OP_ACIR: r = fe_load(acir + it.w[0]); break;                     // :38-44 (noir_to_native is the identity on the limbs)
        case OP_SUM: {                                                        // :45-60
            const u32* t = extra + it.w[0];
            for (u32 i = 0; i < it.w[1]; i++) {
                fe x = fe_load(W + t[2 * i + 1]);
                if (t[2 * i] != NONE) x = fe_mulx(fe_load(K + t[2 * i]), x);
                r = fe_add(r, x);
            }
            break;
        }
        case OP_PRODUCT: r = fe_mulx(fe_load(W + it.w[0]), fe_load(W + it.w[1])); break;  // :61-65
        case OP_CHALLENGE: r = fe_load(challenges + it.w[0]); break;                        // :99-103
        case OP_IDX_LOGUP:  // :70-84  sz - (index_coeff * index + rs * value)
            r = fe_sub(fe_load(W + it.w[0]), fe_add(fe_mulx(fe_load(K + it.k[0]), fe_load(W + it.w[1])), fe_mulx(fe_load(W + it.w[2]), fe_load(W + it.w[3]))));
            break;
        case OP_INVERSE: {  // :66-69
            const fe x = fe_load(W + it.w[0]);
            if (fe_eq(x, fe_zero())) report(err, it.builder, ERR_INVERSE_ZERO);
            r = fe_inverse(x);
            break;
        }
        case OP_PROD_LINEAR:  // :113-120  (a x + b)(c y + d)
            r = fe_mulx(fe_add(fe_mulx(fe_load(K + it.k[0]), fe_load(W + it.w[0])), fe_load(K + it.k[1])),
                        fe_add(fe_mulx(fe_load(K + it.k[2]), fe_load(W + it.w[1])), fe_load(K + it.k[3])));
            break;
        case OP_LOGUP:  // :104-112  sz - value_coeff * value
            r = fe_sub(fe_load(W + it.w[0]), fe_mulx(fe_load(K + it.k[0]), fe_load(W + it.w[1])));
            break;
        case OP_SPICE_FACTOR: {  // :124-141  sz - (addr * addr_w + rs * value + rs * rs * timer * timer_w)
            const fe rs = fe_load(W + it.w[1]);
            fe t = fe_mulx(fe_load(K + it.k[0]), fe_load(W + it.w[2]));
            t = fe_add(t, fe_mulx(rs, fe_load(W + it.w[3])));
            t = fe_add(t, fe_mulx(fe_mulx(fe_mulx(rs, rs), fe_load(K + it.k[1])), fe_load(W + it.w[4])));
            r = fe_sub(fe_load(W + it.w[0]), t);
            break;
        }
        case OP_BINOP_DENOM: {  // :145-171  sz - (lhs + rs * rhs + rs_sqrd * output)
            fe t = cow(W, K, it.w[3]);
            t = fe_add(t, fe_mulx(fe_load(W + it.w[1]), cow(W, K, it.w[4])));
            t = fe_add(t, fe_mulx(fe_load(W + it.w[2]), cow(W, K, it.k[0])));
            r = fe_sub(fe_load(W + it.w[0]), t);
            break;
        }
        case OP_DIGIT: r = take_bits(fe_from_montx(fe_load(W + it.w[0])), it.k[0], it.k[1]); break;  // digits.rs:17-29
        case OP_DIGIT_CHECK:                                                                          // digits.rs:52-56
            if (!bits_above_zero(fe_from_montx(fe_load(W + it.w[0])), it.k[0])) report(err, it.builder, ERR_DIGIT_OVERFLOW);
            return;
        case OP_HIST_RANGE: {  // witness_builder.rs:85-98: value.into_bigint().0[0] as index
            const fe c = fe_from_montx(fe_load(W + it.w[0]));
            const u64 v = (u64)c.v[0] | ((u64)c.v[1] << 32);
            if (v >= it.k[1]) report(err, it.builder, ERR_MULTIPLICITY_RANGE);
            else count_one(counts, it.k[0] + (u32)v);
            return;
        }
        case OP_HIST_BINOP: {  // :172-191  index = (lhs << BINOP_ATOMIC_BITS) + rhs
            const fe a = fe_from_montx(cow(W, K, it.w[0])), b = fe_from_montx(cow(W, K, it.w[1]));
            const u64 idx = ((((u64)a.v[0] | ((u64)a.v[1] << 32))) << 8) + ((u64)b.v[0] | ((u64)b.v[1] << 32));
            if (idx >= 65536) report(err, it.builder, ERR_MULTIPLICITY_RANGE);
            else count_one(counts, it.k[0] + (u32)idx);
            return;
        }
        case OP_COUNT_OUT:  // FieldElement::from(*count).  The histogram was built by atomics (at the L2): read it there too, past an L1
                            // that may hold the line from an earlier phase of the same launch
            r = small_fe(__hip_atomic_load(counts + it.k[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            break;
        default: return;
    }
    fe_store(W + it.out, r);
    is_set[it.out] = 1;
}

// one phase (all items independent)
__global__ __launch_bounds__(256) void wb_phase_kernel(const WbItem* __restrict__ items, size_t n, fe* W, unsigned char* is_set, const fe* K, const fe* acir,
                                                       const fe* challenges, const u32* extra, u32* counts, unsigned long long* err) {
    PK_LATENCY_PRIO();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wb_eval(items[i], W, is_set, K, acir, challenges, extra, counts, err);
}
// a run of consecutive narrow phases in one launch: a single workgroup walks them, a barrier between phases
constexpr u32 NARROW = 1024;
__global__ __launch_bounds__(NARROW) void wb_narrow_run_kernel(const WbItem* __restrict__ items, const u32* __restrict__ phase_begin, u32 first_phase,
                                                               u32 n_phases, fe* W, unsigned char* is_set, const fe* K, const fe* acir,
                                                               const fe* challenges, const u32* extra, u32* counts, unsigned long long* err) {
    PK_LATENCY_PRIO();
    for (u32 p = first_phase; p < first_phase + n_phases; p++) {
        const u32 b = phase_begin[p], e = phase_begin[p + 1];
        if (b + threadIdx.x < e) wb_eval(items[b + threadIdx.x], W, is_set, K, acir, challenges, extra, counts, err);
        __threadfence_block();
        __syncthreads();
    }
}

// ---- Spice (witness/ram.rs:13-47) ------------------------------------------------------------------------------------------------
struct SpiceOp {
    u32 addr, value, out_old, out_ts;  // witness indices: address; the value this operation leaves in memory (Load: value read,
                                       // Store: new value); where the old value goes (Store) or NONE; where the read timestamp goes
};
__global__ __launch_bounds__(256) void spice_keys_kernel(const SpiceOp* __restrict__ ops, u32 n_ops, u32 memory_length, const fe* __restrict__ W,
                                                         unsigned long long* __restrict__ keys, u32 builder, unsigned long long* err) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_ops) return;
    const fe c = fe_from_montx(fe_load(W + ops[k].addr));  // addr.into_bigint().0[0] as usize
    u64 a = (u64)c.v[0] | ((u64)c.v[1] << 32);
    if (a >= memory_length) {
        report(err, builder, ERR_SPICE_ADDRESS);
        a = 0;
    }
    keys[k] = (a << 32) | k;
}
__global__ __launch_bounds__(256) void spice_init_finals_kernel(u32 memory_length, u32 initial_start, u32 rv_start, u32 rt_start, fe* W, unsigned char* is_set) {
    const u32 a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= memory_length) return;
    fe_store(W + rv_start + a, fe_load(W + initial_start + a));  // untouched addresses keep their initial value, timestamp 0
    is_set[rv_start + a] = is_set[initial_start + a];
    fe_store(W + rt_start + a, fe_zero());
    is_set[rt_start + a] = 1;
}
// sorted by (address, position): the left neighbour with the same address is the previous operation on that address
__global__ __launch_bounds__(256) void spice_resolve_kernel(const SpiceOp* __restrict__ ops, u32 n_ops, const unsigned long long* __restrict__ sorted,
                                                            u32 initial_start, u32 rv_start, u32 rt_start, fe* W, unsigned char* is_set) {
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_ops) return;
    const u64 key = sorted[s];
    const u32 a = (u32)(key >> 32), k = (u32)key;
    const bool has_prev = s > 0 && (u32)(sorted[s - 1] >> 32) == a;
    const u32 prev = has_prev ? (u32)sorted[s - 1] : 0;
    const SpiceOp op = ops[k];
    fe_store(W + op.out_ts, small_fe(has_prev ? (u64)prev + 1 : 0));  // rt_final[addr] before this operation
    is_set[op.out_ts] = 1;
    if (op.out_old != NONE) {  // Store: old value = what the previous operation left there, or the initial value
        const u32 src = has_prev ? ops[prev].value : initial_start + a;
        fe_store(W + op.out_old, fe_load(W + src));
        is_set[op.out_old] = is_set[src];
    }
    const bool last = s + 1 == n_ops || (u32)(sorted[s + 1] >> 32) != a;
    if (last) {
        fe_store(W + rv_start + a, fe_load(W + op.value));
        is_set[rv_start + a] = is_set[op.value];
        fe_store(W + rt_start + a, small_fe((u64)k + 1));
    }
}

// ---- long sums (witness_builder.rs:45-60 with thousands of terms: a LogUp grand sum has one per lookup) --------------------------
// A lane that walks 10^5 terms holds its level for 10^5 dependent products.  Sums longer than SUM_HEAVY terms leave the item list:
// a workgroup per SUM_CHUNK terms forms a partial sum, a workgroup per sum adds the partials.  Exact field additions: the order
// does not show in the result.
constexpr u32 SUM_HEAVY = 128, SUM_CHUNK = 1024;
struct SumChunk {
    u32 begin, end;  // term numbers (pairs of P.extra)
};
struct HeavySum {
    u32 builder, level, out, chunk0, n_chunks;
};
__global__ __launch_bounds__(RED_THREADS) void heavy_sum_part_kernel(const SumChunk* __restrict__ chunks, const u32* __restrict__ extra, const fe* __restrict__ W,
                                                                     const fe* __restrict__ K, fe* __restrict__ partials) {
    PK_LATENCY_PRIO();
    __shared__ uint4 smem[16];
    const SumChunk c = chunks[blockIdx.x];
    fe acc = fe_zero();
    for (u32 t = c.begin + threadIdx.x; t < c.end; t += RED_THREADS) {
        fe x = fe_load(W + extra[2 * t + 1]);
        if (extra[2 * t] != NONE) x = fe_mulx(fe_load(K + extra[2 * t]), x);
        acc = fe_add(acc, x);
    }
    wide w[1] = {wide_zero()};
    wide_add_fe(w[0], acc);
    const fe sum = block_reduce_wide<1>(w, smem);
    if (threadIdx.x == 0) fe_store(partials + blockIdx.x, sum);
}
__global__ __launch_bounds__(RED_THREADS) void heavy_sum_final_kernel(const HeavySum* __restrict__ sums, const fe* __restrict__ partials, u32 chunk_base,
                                                                      fe* __restrict__ W, unsigned char* __restrict__ is_set) {
    PK_LATENCY_PRIO();
    __shared__ uint4 smem[16];
    const HeavySum hs = sums[blockIdx.x];
    fe acc = fe_zero();
    for (u32 j = threadIdx.x; j < hs.n_chunks; j += RED_THREADS) acc = fe_add(acc, fe_load(partials + (hs.chunk0 - chunk_base) + j));
    wide w[1] = {wide_zero()};
    wide_add_fe(w[0], acc);
    const fe sum = block_reduce_wide<1>(w, smem);
    if (threadIdx.x == 0) {
        fe_store(W + hs.out, sum);
        is_set[hs.out] = 1;
    }
}

// ---- postcard(Vec<WitnessBuilder>) ---------------------------------------------------------------------------------------------
constexpr uint64_t PK_MAX_WITNESS_INDEX = 1ull << 27;
// work items one program may expand to (a builder that writes many witnesses becomes one item per witness): a scheme holds at
// most 2^26 witnesses, so a list from untrusted bytes that asks for more than this is refused before anything is allocated for it
constexpr uint64_t PK_MAX_PROGRAM_ITEMS = 1ull << 27;
// ... and never out of proportion to the input: every witness a real list writes is read by a later builder (a table entry by its
// LogUp denominator, a digit by its range check), i.e. costs bytes of its own, so the expansion of an honest list stays within a
// small multiple of its length; 2^20 covers the fixed-size tables (a 2^16-entry binop table per builder) of short lists
inline uint64_t program_item_budget(size_t input_bytes) {
    const uint64_t prop = (1ull << 20) + 64ull * (uint64_t)input_bytes;
    return prop < PK_MAX_PROGRAM_ITEMS ? prop : PK_MAX_PROGRAM_ITEMS;
}
struct Reader {
    const uint8_t* p;
    size_t n, off = 0;
    bool ok = true;
    uint64_t varint() {
        uint64_t v = 0;
        for (unsigned shift = 0; shift < 70; shift += 7) {
            if (off >= n) return ok = false, 0;
            const uint8_t b = p[off++];
            if (shift == 63 && b > 1) return ok = false, 0;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        return ok = false, 0;
    }
    u32 index() {  // a usize that must be a witness index, an ACIR index or a table size: below 2^27 (a scheme holds at most 2^26
                   // witnesses, pk_scheme_create), so a corrupted list cannot ask for tables of gigabytes
        const uint64_t v = varint();
        if (v >= PK_MAX_WITNESS_INDEX) ok = false;
        return (u32)v;
    }
    // serde_ark: bytes(32) = varint(32) | canonical little-endian (provekit/common/src/utils/serde_ark.rs:11-30)
    bool field(fe& out) {
        if (varint() != 32 || !ok || n - off < 32) return ok = false;
        memcpy(out.v, p + off, 32);
        off += 32;
        fe red = fe_reduce_any(out);
        if (memcmp(red.v, out.v, 32) != 0) return ok = false;  // Fp::deserialize_compressed rejects values >= p
        return true;
    }
};

struct SpiceBlock {
    u32 builder, level, memory_length, initial_start, rv_start, rt_start;
    std::vector<SpiceOp> ops;
};

struct Program {
    size_t n_builders = 0, n_witnesses = 0, n_challenges = 0, n_acir = 0;  // n_witnesses / n_acir: 1 + the largest index used
    std::vector<WbItem> items;      // sorted by phase
    std::vector<u32> phase_begin;   // n_phases + 1
    std::vector<fe> consts;         // Montgomery
    std::vector<u32> extra;
    std::vector<SpiceBlock> spice;  // sorted by level
    std::vector<HeavySum> heavy_sums;  // sorted by level; their chunks in the same order
    std::vector<SumChunk> sum_chunks;
    std::vector<std::pair<u32, u32>> heavy_terms;  // per heavy sum before sorting: (first term, number of terms)
    size_t n_counts = 0;
    uint64_t expanded = 0;  // work items + Spice cells asked for so far (PK_MAX_PROGRAM_ITEMS)
    std::string error;
};

struct Parsed {  // a builder before levelling
    std::vector<u32> reads, writes;
    std::vector<u32> copies;  // witnesses copied as Options, not unwrapped (Spice values): None is legal and stays None
    std::vector<WbItem> main, second;  // second = the phase after main (COUNT_OUT)
    int spice = -1, heavy_sum = -1;
};

u32 add_const(Program& P, const fe& canon) {
    P.consts.push_back(fe_to_montx(canon));
    return (u32)(P.consts.size() - 1);
}

bool read_cow(Reader& rd, Program& P, Parsed& b, u32& packed) {  // ConstantOrR1CSWitness (witness/mod.rs:29-33)
    const uint64_t tag = rd.varint();
    if (tag == 0) {
        fe c;
        if (!rd.field(c)) return false;
        packed = add_const(P, c) | COW_CONST;
    } else if (tag == 1) {
        packed = rd.index();
        b.reads.push_back(packed);
    } else {
        return rd.ok = false;
    }
    return rd.ok;
}

// one builder (witness_builder.rs:33-117, variants in declaration order)
bool parse_builder(Reader& rd, Program& P, u32 bi, Parsed& b) {
    // `count` more work items: refused once the program's total passes PK_MAX_PROGRAM_ITEMS (checked BEFORE the caller expands)
    auto budget = [&](uint64_t count) {
        const uint64_t cap = program_item_budget(rd.n);
        if (count > cap || P.expanded + count > cap) return rd.ok = false;
        P.expanded += count;
        return true;
    };
    auto item = [&](u32 op, u32 out) {
        WbItem it{};
        it.op = op;
        it.out = out;
        it.builder = bi;
        for (auto& x : it.w) x = 0;
        for (auto& x : it.k) x = 0;
        if (out != NONE) b.writes.push_back(out);
        return it;
    };
    auto rdw = [&](u32& dst) {
        dst = rd.index();
        b.reads.push_back(dst);
    };
    const uint64_t tag = rd.varint();
    if (!rd.ok) return false;
    fe c;
    switch (tag) {
        case 0: {  // Constant(ConstantTerm(idx, c))
            WbItem it = item(OP_CONST, rd.index());
            if (!rd.field(c)) return false;
            it.k[0] = add_const(P, c);
            b.main.push_back(it);
            break;
        }
        case 1: {  // Acir(idx, acir_idx)
            WbItem it = item(OP_ACIR, rd.index());
            it.w[0] = rd.index();
            P.n_acir = std::max<size_t>(P.n_acir, (size_t)it.w[0] + 1);
            b.main.push_back(it);
            break;
        }
        case 2: {  // Sum(idx, Vec<SumTerm(Option<F>, usize)>)
            WbItem it = item(OP_SUM, rd.index());
            const uint64_t n = rd.varint();
            if (!rd.ok || n > rd.n - rd.off) return rd.ok = false;
            it.w[0] = (u32)P.extra.size();
            it.w[1] = (u32)n;
            for (uint64_t i = 0; i < n; i++) {
                const uint64_t some = rd.varint();
                u32 coef = NONE;
                if (some == 1) {
                    if (!rd.field(c)) return false;
                    coef = add_const(P, c);
                } else if (some != 0) {
                    return rd.ok = false;
                }
                u32 w;
                rdw(w);
                P.extra.push_back(coef);
                P.extra.push_back(w);
            }
            if (n > SUM_HEAVY) {  // summed by workgroups between the two phases of its level, not by one lane
                b.heavy_sum = (int)P.heavy_sums.size();
                P.heavy_sums.push_back(HeavySum{bi, 0, it.out, 0, 0});
                P.heavy_terms.push_back({it.w[0] / 2, (u32)n});
            } else {
                b.main.push_back(it);
            }
            break;
        }
        case 3: {  // Product(idx, a, b)
            WbItem it = item(OP_PRODUCT, rd.index());
            rdw(it.w[0]);
            rdw(it.w[1]);
            b.main.push_back(it);
            break;
        }
        case 4: {  // MultiplicitiesForRange(start, range_size, Vec<usize>)
            const u32 start = rd.index(), range = rd.index();
            const uint64_t n = rd.varint();
            if (!rd.ok || n > rd.n - rd.off) return rd.ok = false;
            // the table is expanded to one work item per entry: an absurd size from untrusted bytes must not drive host allocations
            if (range > (1u << 26) || P.n_counts + range > (1ull << 28)) return rd.ok = false;
            if ((uint64_t)start + range > PK_MAX_WITNESS_INDEX || !budget(n + range)) return rd.ok = false;
            const u32 base = (u32)P.n_counts;
            P.n_counts += range;
            for (uint64_t i = 0; i < n; i++) {
                WbItem it = item(OP_HIST_RANGE, NONE);
                rdw(it.w[0]);
                it.k[0] = base;
                it.k[1] = range;
                b.main.push_back(it);
            }
            for (u32 i = 0; i < range; i++) {
                WbItem it = item(OP_COUNT_OUT, start + i);
                it.k[0] = base + i;
                b.second.push_back(it);
            }
            break;
        }
        case 5: {  // Challenge(idx)
            WbItem it = item(OP_CHALLENGE, rd.index());
            it.w[0] = (u32)P.n_challenges++;
            b.main.push_back(it);
            break;
        }
        case 6: {  // IndexedLogUpDenominator(idx, sz, WitnessCoefficient(coeff, index), rs, value)
            WbItem it = item(OP_IDX_LOGUP, rd.index());
            rdw(it.w[0]);
            if (!rd.field(c)) return false;
            it.k[0] = add_const(P, c);
            rdw(it.w[1]);
            rdw(it.w[2]);
            rdw(it.w[3]);
            b.main.push_back(it);
            break;
        }
        case 7: {  // Inverse(idx, operand)
            WbItem it = item(OP_INVERSE, rd.index());
            rdw(it.w[0]);
            b.main.push_back(it);
            break;
        }
        case 8: {  // ProductLinearOperation(idx, (x, a, b), (y, c, d))
            WbItem it = item(OP_PROD_LINEAR, rd.index());
            for (int t = 0; t < 2; t++) {
                rdw(it.w[t]);
                for (int q = 0; q < 2; q++) {
                    if (!rd.field(c)) return false;
                    it.k[2 * t + q] = add_const(P, c);
                }
            }
            b.main.push_back(it);
            break;
        }
        case 9: {  // LogUpDenominator(idx, sz, WitnessCoefficient(coeff, value))
            WbItem it = item(OP_LOGUP, rd.index());
            rdw(it.w[0]);
            if (!rd.field(c)) return false;
            it.k[0] = add_const(P, c);
            rdw(it.w[1]);
            b.main.push_back(it);
            break;
        }
        case 10: {  // DigitalDecomposition(DigitalDecompositionWitnesses) (witness/digits.rs:10-21 of common)
            std::vector<u32> log_bases, values;
            uint64_t n = rd.varint();
            if (!rd.ok || n > rd.n - rd.off) return rd.ok = false;
            for (uint64_t i = 0; i < n; i++) log_bases.push_back(rd.index());
            const u32 declared = rd.index();
            n = rd.varint();
            if (!rd.ok || n > rd.n - rd.off) return rd.ok = false;
            for (uint64_t i = 0; i < n; i++) values.push_back(rd.index());
            const u32 first = rd.index();
            (void)rd.index();  // num_witnesses
            if (!rd.ok || declared != values.size()) return rd.ok = false;
            // one item per (digit, value): zero-width digits are legal, so only the COUNT of bases bounds the product -- cap it
            // (256 one-bit digits is the finest decomposition that reads any bit), keep every written index a witness index
            // (64-bit arithmetic: first + bases * values must not wrap) and charge the expansion to the program's budget
            if (log_bases.size() > 256) return rd.ok = false;
            const uint64_t dd_items = (uint64_t)log_bases.size() * values.size();
            if ((uint64_t)first + dd_items > PK_MAX_WITNESS_INDEX || !budget(dd_items + values.size())) return rd.ok = false;
            u32 total = 0;
            for (u32 lb : log_bases) {
                if (lb > 256 || total + lb > 256) return rd.ok = false;  // field_to_le_bits yields 256 bits: a longer slice panics
                total += lb;
            }
            for (size_t i = 0; i < values.size(); i++) {
                b.reads.push_back(values[i]);
                u32 start = 0;
                for (size_t d = 0; d < log_bases.size(); d++) {
                    WbItem it = item(OP_DIGIT, first + (u32)(d * values.size() + i));
                    it.w[0] = values[i];
                    it.k[0] = start;
                    it.k[1] = log_bases[d];
                    start += log_bases[d];
                    b.main.push_back(it);
                }
                WbItem ck = item(OP_DIGIT_CHECK, NONE);
                ck.w[0] = values[i];
                ck.k[0] = total;
                b.main.push_back(ck);
            }
            break;
        }
        case 11: {  // SpiceMultisetFactor(idx, sz, rs, (addr, addr_w), value, (timer, timer_w))
            WbItem it = item(OP_SPICE_FACTOR, rd.index());
            rdw(it.w[0]);
            rdw(it.w[1]);
            if (!rd.field(c)) return false;
            it.k[0] = add_const(P, c);
            rdw(it.w[2]);
            rdw(it.w[3]);
            if (!rd.field(c)) return false;
            it.k[1] = add_const(P, c);
            rdw(it.w[4]);
            b.main.push_back(it);
            break;
        }
        case 12: {  // SpiceWitnesses (witness/ram.rs:19-38 of common)
            SpiceBlock sb{};
            sb.builder = bi;
            sb.memory_length = rd.index();
            sb.initial_start = rd.index();
            // three reads / writes per cell are recorded below: charged to the budget before anything is pushed
            if ((uint64_t)sb.initial_start + sb.memory_length > PK_MAX_WITNESS_INDEX || !budget(3ull * sb.memory_length)) return rd.ok = false;
            const uint64_t n = rd.varint();
            if (!rd.ok || n > rd.n - rd.off || !budget(n)) return rd.ok = false;
            for (uint64_t i = 0; i < n; i++) {
                const uint64_t kind = rd.varint();
                SpiceOp op{};
                if (kind == 0) {  // Load(addr, value, read_timestamp)
                    op.addr = rd.index();
                    op.value = rd.index();
                    op.out_old = NONE;
                    op.out_ts = rd.index();
                } else if (kind == 1) {  // Store(addr, old_value, new_value, read_timestamp)
                    op.addr = rd.index();
                    op.out_old = rd.index();
                    op.value = rd.index();
                    op.out_ts = rd.index();
                    b.writes.push_back(op.out_old);
                } else {
                    return rd.ok = false;
                }
                b.reads.push_back(op.addr);     // witness[*addr].unwrap()
                b.copies.push_back(op.value);   // rv_final[addr] = witness[*value]: an Option, copied as it is
                b.writes.push_back(op.out_ts);
                sb.ops.push_back(op);
            }
            sb.rv_start = rd.index();
            sb.rt_start = rd.index();
            (void)rd.index();  // first_witness_idx
            (void)rd.index();  // num_witnesses
            if (!rd.ok) return false;
            if ((uint64_t)sb.rv_start + sb.memory_length > PK_MAX_WITNESS_INDEX || (uint64_t)sb.rt_start + sb.memory_length > PK_MAX_WITNESS_INDEX)
                return rd.ok = false;
            for (u32 a = 0; a < sb.memory_length; a++) {
                b.copies.push_back(sb.initial_start + a);
                b.writes.push_back(sb.rv_start + a);
                b.writes.push_back(sb.rt_start + a);
            }
            b.spice = (int)P.spice.size();
            P.spice.push_back(std::move(sb));
            break;
        }
        case 13: {  // BinOpLookupDenominator(idx, sz, rs, rs_sqrd, lhs, rhs, output)
            WbItem it = item(OP_BINOP_DENOM, rd.index());
            rdw(it.w[0]);
            rdw(it.w[1]);
            rdw(it.w[2]);
            if (!read_cow(rd, P, b, it.w[3]) || !read_cow(rd, P, b, it.w[4]) || !read_cow(rd, P, b, it.k[0])) return false;
            b.main.push_back(it);
            break;
        }
        case 14: {  // MultiplicitiesForBinOp(idx, Vec<(CoW, CoW)>)
            const u32 start = rd.index();
            const uint64_t n = rd.varint();
            if (!rd.ok || n > rd.n - rd.off) return rd.ok = false;
            if (P.n_counts + 65536 > (1ull << 28)) return rd.ok = false;
            if ((uint64_t)start + 65536 > PK_MAX_WITNESS_INDEX || !budget(n + 65536)) return rd.ok = false;
            const u32 base = (u32)P.n_counts;
            P.n_counts += 65536;  // 2^(2 * BINOP_ATOMIC_BITS)
            for (uint64_t i = 0; i < n; i++) {
                WbItem it = item(OP_HIST_BINOP, NONE);
                if (!read_cow(rd, P, b, it.w[0]) || !read_cow(rd, P, b, it.w[1])) return false;
                it.k[0] = base;
                b.main.push_back(it);
            }
            for (u32 i = 0; i < 65536; i++) {
                WbItem it = item(OP_COUNT_OUT, start + i);
                it.k[0] = base + i;
                b.second.push_back(it);
            }
            break;
        }
        default: return rd.ok = false;
    }
    return rd.ok;
}

// decode + level.  Returns false with P.error set on malformed input or a list the reference itself would panic on.
bool build_program(const uint8_t* bytes, size_t len, Program& P, size_t* consumed) {
    Reader rd{bytes, len};
    const uint64_t n = rd.varint();
    if (!rd.ok || n > len) return P.error = "postcard Vec<WitnessBuilder>: truncated", false;
    P.n_builders = (size_t)n;
    std::vector<Parsed> B((size_t)n);
    for (size_t i = 0; i < (size_t)n; i++)
        if (!parse_builder(rd, P, (u32)i, B[i])) return P.error = "postcard Vec<WitnessBuilder>: builder " + std::to_string(i) + " is malformed", false;
    if (consumed) *consumed = rd.off;
    size_t nw = 0;
    for (auto& b : B) {
        for (u32 w : b.reads) nw = std::max<size_t>(nw, (size_t)w + 1);
        for (u32 w : b.copies) nw = std::max<size_t>(nw, (size_t)w + 1);
        for (u32 w : b.writes) nw = std::max<size_t>(nw, (size_t)w + 1);
    }
    P.n_witnesses = nw;
    // level(b) = 1 + max level of the producers of its inputs; an input nobody produced EARLIER in the list is a `None`
    // the reference would unwrap (panic): refuse the list instead.  The reference's solver runs the list in order, so a witness may
    // be written more than once (the last writer wins) and a Spice block may copy a `None` that a later builder solves: both keep
    // their sequential meaning here because a writer is also levelled after the previous writer of its witness (write after write) and
    // after every earlier reader or copier of the version it replaces (write after read) -- there is one witness array, no renaming.
    std::vector<int> producer(nw, -1);
    std::vector<int> reader_level(nw, -1);  // highest level among the builders that read / copied the CURRENT version (or the None)
    std::vector<u32> level(B.size(), 0);
    u32 max_level = 0;
    for (size_t i = 0; i < B.size(); i++) {
        u32 lv = 0;
        for (u32 w : B[i].reads) {
            if (producer[w] < 0) return P.error = "builder " + std::to_string(i) + " reads witness " + std::to_string(w) + " before it is solved", false;
            lv = std::max(lv, level[(size_t)producer[w]] + 1);
        }
        for (u32 w : B[i].copies)
            if (producer[w] >= 0) lv = std::max(lv, level[(size_t)producer[w]] + 1);
        {  // one builder writing the same witness twice would race inside its level: not a list the compiler emits
            std::vector<u32> ws(B[i].writes);
            std::sort(ws.begin(), ws.end());
            if (std::adjacent_find(ws.begin(), ws.end()) != ws.end())
                return P.error = "builder " + std::to_string(i) + " writes witness " + std::to_string(*std::adjacent_find(ws.begin(), ws.end())) + " twice", false;
        }
        for (u32 w : B[i].writes) {
            if (producer[w] >= 0 && (size_t)producer[w] != i) lv = std::max(lv, level[(size_t)producer[w]] + 1);
            if (reader_level[w] >= 0) lv = std::max(lv, (u32)reader_level[w] + 1);
        }
        for (u32 w : B[i].reads) reader_level[w] = std::max(reader_level[w], (int)lv);
        for (u32 w : B[i].copies) reader_level[w] = std::max(reader_level[w], (int)lv);
        for (u32 w : B[i].writes) {
            producer[w] = (int)i;
            reader_level[w] = -1;  // a new version: nobody has read it yet
        }
        level[i] = lv;
        max_level = std::max(max_level, lv);
        if (B[i].spice >= 0) P.spice[(size_t)B[i].spice].level = lv;
        if (B[i].heavy_sum >= 0) P.heavy_sums[(size_t)B[i].heavy_sum].level = lv;
    }
    // phase = 2 * level (+1 for the second-phase items of multiplicity builders); counting sort of the items by phase
    const size_t n_phases = 2 * ((size_t)max_level + 1);
    std::vector<size_t> cnt(n_phases + 1, 0);
    for (size_t i = 0; i < B.size(); i++) {
        cnt[2 * level[i] + 1] += B[i].main.size();
        if (2 * level[i] + 2 <= n_phases) cnt[2 * level[i] + 2] += B[i].second.size();
    }
    for (size_t p = 1; p <= n_phases; p++) cnt[p] += cnt[p - 1];
    if (cnt[n_phases] >= 0x7fffffffull) return P.error = "too many work items", false;
    P.items.resize(cnt[n_phases]);
    P.phase_begin.assign(cnt.begin(), cnt.end());
    std::vector<size_t> cur(cnt.begin(), cnt.end() - 1);
    for (size_t i = 0; i < B.size(); i++) {
        for (auto& it : B[i].main) P.items[cur[2 * level[i]]++] = it;
        for (auto& it : B[i].second) P.items[cur[2 * level[i] + 1]++] = it;
    }
    // items of one phase are independent: grouped by variant, a wavefront runs one case of wb_eval's switch instead of all of them
    // one after the other (and an Inverse's 40 us are not paid by 63 lanes that only needed a product)
    for (size_t ph = 0; ph + 1 < P.phase_begin.size(); ph++)
        std::stable_sort(P.items.begin() + P.phase_begin[ph], P.items.begin() + P.phase_begin[ph + 1],
                         [](const WbItem& a, const WbItem& b) { return a.op < b.op; });
    std::stable_sort(P.spice.begin(), P.spice.end(), [](const SpiceBlock& a, const SpiceBlock& b) { return a.level < b.level; });
    {  // heavy sums by level, each with its chunks (consecutive, in the sums' order)
        std::vector<size_t> order(P.heavy_sums.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return P.heavy_sums[a].level < P.heavy_sums[b].level; });
        std::vector<HeavySum> sorted;
        for (size_t i : order) {
            HeavySum hs = P.heavy_sums[i];
            const u32 t0 = P.heavy_terms[i].first, n = P.heavy_terms[i].second;
            hs.chunk0 = (u32)P.sum_chunks.size();
            for (u32 t = 0; t < n; t += SUM_CHUNK) P.sum_chunks.push_back(SumChunk{t0 + t, t0 + std::min(n, t + SUM_CHUNK)});
            hs.n_chunks = (u32)P.sum_chunks.size() - hs.chunk0;
            sorted.push_back(hs);
        }
        P.heavy_sums.swap(sorted);
    }
    return true;
}

}  // namespace

struct pk_witness_program {
    Program P;
    WbItem* d_items = nullptr;
    u32* d_phase_begin = nullptr;
    fe* d_consts = nullptr;
    u32* d_extra = nullptr;
    u32* d_counts = nullptr;
    unsigned long long* d_err = nullptr;
    std::vector<SpiceOp*> d_spice_ops;
    HeavySum* d_heavy_sums = nullptr;
    SumChunk* d_sum_chunks = nullptr;
    fe* d_sum_partials = nullptr;
    unsigned long long *d_keys = nullptr, *d_sorted = nullptr;  // sized for the longest Spice block
    void* d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
};

namespace pk {
void witness_program_shape(const pk_witness_program* p, size_t* n_witnesses, size_t* n_challenges, size_t* n_acir) {
    *n_witnesses = p->P.n_witnesses;
    *n_challenges = p->P.n_challenges;
    *n_acir = p->P.n_acir;
}
}  // namespace pk

extern "C" {

// the distinct ACIR witness indices the list's Acir builders read, ascending (host only; the program keeps its item list)
int pk_witness_program_acir_reads(const pk_witness_program* p, uint32_t* idx, size_t cap, size_t* n) {
    if (!p || !n) return PK_ERR_BAD_ARG;
    std::vector<u32> v;
    for (const WbItem& it : p->P.items)
        if (it.op == OP_ACIR) v.push_back(it.w[0]);
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    *n = v.size();
    if (idx && cap >= v.size() && !v.empty()) memcpy(idx, v.data(), 4 * v.size());
    return PK_OK;
}

// Where should this list be solved?  Measured on MI355X (profiles/r03_witness_bench.jsonl): a level costs ~2.6 us of
// launch / barrier / dependent-load latency whatever its width, an item ~1 ns of throughput (2^20 builders in 1.08 ms); one host
// core runs the reference's sequential loop at ~60 ns per builder.  A chain-shaped list (depth ~ length: 16 k builders = 43 ms here,
// ~1 ms there) belongs on the host; a real constraint system's list (depth << length) on the device.
int pk_witness_program_placement(const pk_witness_program* p, size_t* n_levels, size_t* n_items, double* est_device_us, double* est_host_us,
                                 int* prefer_host) {
    if (!p) return PK_ERR_BAD_ARG;
    const size_t levels = p->P.phase_begin.empty() ? 0 : (p->P.phase_begin.size() - 1) / 2;
    const double dev = 2.6 * (double)levels + 1e-3 * (double)p->P.items.size() + 10.0;
    const double host = 0.06 * (double)p->P.n_builders;
    if (n_levels) *n_levels = levels;
    if (n_items) *n_items = p->P.items.size();
    if (est_device_us) *est_device_us = dev;
    if (est_host_us) *est_host_us = host;
    if (prefer_host) *prefer_host = dev > host ? 1 : 0;
    return PK_OK;
}

int pk_witness_program_destroy(pk_ctx* ctx, pk_witness_program* p) {
    PK_ENTER(ctx);
    if (!p) return PK_OK;
    (void)wait_ctx(ctx);
    (void)hipFree(p->d_items);
    (void)hipFree(p->d_phase_begin);
    (void)hipFree(p->d_consts);
    (void)hipFree(p->d_extra);
    (void)hipFree(p->d_counts);
    (void)hipFree(p->d_err);
    for (auto q : p->d_spice_ops) (void)hipFree(q);
    (void)hipFree(p->d_heavy_sums);
    (void)hipFree(p->d_sum_chunks);
    (void)hipFree(p->d_sum_partials);
    (void)hipFree(p->d_keys);
    (void)hipFree(p->d_sorted);
    (void)hipFree(p->d_sort_tmp);
    delete p;
    return PK_OK;
}

// host only: decode and level the list, report its shape (no device needed: the CPU test-suite checks the codec with it)
int pk_witness_builders_inspect(const uint8_t* bytes, size_t len, size_t* n_builders, size_t* n_witnesses, size_t* n_challenges, size_t* n_acir,
                                size_t* n_levels, size_t* n_items, size_t* consumed, char* err, size_t err_cap) {
    if (!bytes) return PK_ERR_BAD_ARG;
    try {
        Program P;
        size_t used = 0;
        if (!build_program(bytes, len, P, &used)) {
            if (err && err_cap) snprintf(err, err_cap, "%s", P.error.c_str());
            return PK_ERR_BAD_ARG;
        }
        if (n_builders) *n_builders = P.n_builders;
        if (n_witnesses) *n_witnesses = P.n_witnesses;
        if (n_challenges) *n_challenges = P.n_challenges;
        if (n_acir) *n_acir = P.n_acir;
        if (n_levels) *n_levels = (P.phase_begin.size() - 1) / 2;
        if (n_items) *n_items = P.items.size();
        if (consumed) *consumed = used;
        return PK_OK;
    } catch (const std::bad_alloc&) {
        return PK_ERR_OOM;
    }
}

int pk_witness_builders_from_postcard(pk_ctx* ctx, const uint8_t* bytes, size_t len, pk_witness_program** out, size_t* n_witnesses,
                                      size_t* n_challenges, size_t* n_acir) {
    if (!ctx || !out) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    *out = nullptr;
    PK_REQUIRE(ctx, bytes, "null pointer");
    pk_witness_program* p = nullptr;
    try {
        p = new pk_witness_program();
        if (!build_program(bytes, len, p->P, nullptr)) {
            int rc = set_err(ctx, PK_ERR_BAD_ARG, "%s", p->P.error.c_str());
            delete p;
            return rc;
        }
    } catch (const std::bad_alloc&) {
        delete p;
        return set_err(ctx, PK_ERR_OOM, "host memory exhausted while decoding the witness builders");
    }
    Program& P = p->P;
    auto up = [&](void** d, const void* h, size_t bytes_) -> bool {
        if (hipMalloc(d, bytes_ ? bytes_ : 4) != hipSuccess) return false;
        return !bytes_ || hipMemcpy(*d, h, bytes_, hipMemcpyHostToDevice) == hipSuccess;
    };
    bool ok = up((void**)&p->d_items, P.items.data(), P.items.size() * sizeof(WbItem)) &&
              up((void**)&p->d_phase_begin, P.phase_begin.data(), P.phase_begin.size() * 4) && up((void**)&p->d_consts, P.consts.data(), P.consts.size() * 32) &&
              up((void**)&p->d_extra, P.extra.data(), P.extra.size() * 4) && hipMalloc((void**)&p->d_counts, (P.n_counts ? P.n_counts : 1) * 4) == hipSuccess &&
              hipMalloc((void**)&p->d_err, 8) == hipSuccess;
    if (!P.heavy_sums.empty())
        ok = ok && up((void**)&p->d_heavy_sums, P.heavy_sums.data(), P.heavy_sums.size() * sizeof(HeavySum)) &&
             up((void**)&p->d_sum_chunks, P.sum_chunks.data(), P.sum_chunks.size() * sizeof(SumChunk)) &&
             hipMalloc((void**)&p->d_sum_partials, P.sum_chunks.size() * 32) == hipSuccess;
    size_t longest = 0;
    for (auto& sb : P.spice) {
        SpiceOp* d = nullptr;
        ok = ok && up((void**)&d, sb.ops.data(), sb.ops.size() * sizeof(SpiceOp));
        p->d_spice_ops.push_back(d);
        longest = std::max(longest, sb.ops.size());
    }
    if (ok && longest) {
        ok = hipMalloc((void**)&p->d_keys, 8 * longest) == hipSuccess && hipMalloc((void**)&p->d_sorted, 8 * longest) == hipSuccess;
        if (ok) ok = hipcub::DeviceRadixSort::SortKeys(nullptr, p->sort_tmp_bytes, p->d_keys, p->d_sorted, (int)longest, 0, 64, ctx->stream) == hipSuccess;
        if (ok) ok = hipMalloc(&p->d_sort_tmp, p->sort_tmp_bytes ? p->sort_tmp_bytes : 4) == hipSuccess;
    }
    if (!ok) {
        pk_witness_program_destroy(ctx, p);
        return set_err(ctx, PK_ERR_OOM, "device allocation of the witness program failed");
    }
    if (n_witnesses) *n_witnesses = P.n_witnesses;
    if (n_challenges) *n_challenges = P.n_challenges;
    if (n_acir) *n_acir = P.n_acir;
    *out = p;
    return PK_OK;
}

int pk_witness_solve(pk_ctx* ctx, pk_witness_program* p, const uint64_t* d_acir, size_t n_acir, const uint64_t* challenges, size_t n_challenges,
                     uint64_t* d_witness, size_t n_witness, uint8_t* d_is_set) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, p && d_witness && d_is_set && (d_acir || p->P.n_acir == 0) && (challenges || p->P.n_challenges == 0), "null pointer");
    Program& P = p->P;
    PK_REQUIRE(ctx, n_acir >= P.n_acir, "the ACIR witness vector is shorter than the largest ACIR index the builders read");
    PK_REQUIRE(ctx, n_challenges == P.n_challenges, "one challenge per WitnessBuilder::Challenge, in list order");
    PK_REQUIRE(ctx, n_witness >= P.n_witnesses, "the witness vector is shorter than the largest index the builders touch");
    // challenges travel through the pinned mailbox (read directly by the kernels)
    fe* m_chal = nullptr;
    int rc = mail_alloc(ctx, 32 * (n_challenges ? n_challenges : 1), (void**)&m_chal);
    if (rc) return rc;
    if (n_challenges) memcpy(m_chal, challenges, 32 * n_challenges);
    PK_HIP(ctx, hipMemsetAsync(d_witness, 0, 32 * n_witness, ctx->stream));
    PK_HIP(ctx, hipMemsetAsync(d_is_set, 0, n_witness, ctx->stream));
    PK_HIP(ctx, hipMemsetAsync(p->d_counts, 0, (P.n_counts ? P.n_counts : 1) * 4, ctx->stream));
    PK_HIP(ctx, hipMemsetAsync(p->d_err, 0xff, 8, ctx->stream));
    fe* W = (fe*)d_witness;
    const size_t n_phases = P.phase_begin.size() - 1;
    size_t sp = 0, hs = 0;  // next Spice block, next heavy sum
    auto block_at = [&](size_t phase) {  // does a Spice block or a heavy sum run right before this (odd) phase?
        return (phase & 1) == 1 && ((sp < P.spice.size() && P.spice[sp].level == phase / 2) || (hs < P.heavy_sums.size() && P.heavy_sums[hs].level == phase / 2));
    };
    {
        ProfScope prof(ctx, "witness_builders");
        for (size_t ph = 0; ph < n_phases;) {
            const size_t n = P.phase_begin[ph + 1] - P.phase_begin[ph];
            // Spice blocks of level L run between phase 2L (main items of that level) and phase 2L + 1
            const bool spice_here = (ph & 1) == 1 && sp < P.spice.size() && P.spice[sp].level == ph / 2;
            if (spice_here) {
                for (; sp < P.spice.size() && P.spice[sp].level == ph / 2; sp++) {
                    SpiceBlock& sb = P.spice[sp];
                    const u32 K = (u32)sb.ops.size(), M = sb.memory_length;
                    if (M) spice_init_finals_kernel<<<(M + 255) / 256, 256, 0, ctx->stream>>>(M, sb.initial_start, sb.rv_start, sb.rt_start, W, d_is_set);
                    if (K) {
                        spice_keys_kernel<<<(K + 255) / 256, 256, 0, ctx->stream>>>(p->d_spice_ops[sp], K, M, W, p->d_keys, sb.builder, p->d_err);
                        size_t tmp = p->sort_tmp_bytes;
                        PK_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(p->d_sort_tmp, tmp, p->d_keys, p->d_sorted, (int)K, 0, 64, ctx->stream));
                        spice_resolve_kernel<<<(K + 255) / 256, 256, 0, ctx->stream>>>(p->d_spice_ops[sp], K, p->d_sorted, sb.initial_start, sb.rv_start, sb.rt_start, W,
                                                                                     d_is_set);
                    }
                }
            }
            if ((ph & 1) == 1 && hs < P.heavy_sums.size() && P.heavy_sums[hs].level == ph / 2) {  // the long sums of this level
                size_t he = hs;
                while (he < P.heavy_sums.size() && P.heavy_sums[he].level == ph / 2) he++;
                const u32 c0 = P.heavy_sums[hs].chunk0, c1 = P.heavy_sums[he - 1].chunk0 + P.heavy_sums[he - 1].n_chunks;
                heavy_sum_part_kernel<<<c1 - c0, RED_THREADS, 0, ctx->stream>>>(p->d_sum_chunks + c0, p->d_extra, W, p->d_consts, p->d_sum_partials + c0);
                heavy_sum_final_kernel<<<(unsigned)(he - hs), RED_THREADS, 0, ctx->stream>>>(p->d_heavy_sums + hs, p->d_sum_partials + c0, c0, W, d_is_set);
                hs = he;
            }
            if (n == 0) {
                ph++;
                continue;
            }
            if (n <= NARROW) {  // a run of narrow phases in one launch (no Spice block may fall inside the run)
                size_t run = 1;
                while (ph + run < n_phases && P.phase_begin[ph + run + 1] - P.phase_begin[ph + run] <= NARROW && !block_at(ph + run)) run++;
                wb_narrow_run_kernel<<<1, NARROW, 0, ctx->stream>>>(p->d_items, p->d_phase_begin, (u32)ph, (u32)run, W, d_is_set, p->d_consts, (const fe*)d_acir, m_chal,
                                                                  p->d_extra, p->d_counts, p->d_err);
                ph += run;
            } else {
                wb_phase_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(p->d_items + P.phase_begin[ph], n, W, d_is_set, p->d_consts, (const fe*)d_acir,
                                                                                     m_chal, p->d_extra, p->d_counts, p->d_err);
                ph++;
            }
        }
    }
    PK_LAUNCH_CHECK(ctx);
    unsigned long long err = ~0ull;
    PK_HIP(ctx, hipMemcpyAsync(&err, p->d_err, 8, hipMemcpyDeviceToHost, ctx->stream));
    rc = sync_stream(ctx);
    if (rc) return rc;
    if (err != ~0ull) {  // the reference panics at these points; the lowest builder index is the one it would reach first
        static const char* what[] = {"", "inverse of zero (Inverse: operand.inverse().unwrap())", "Higher order bits are not zero (DigitalDecomposition)",
                                     "value outside the multiplicity table", "memory address outside the Spice block"};
        const unsigned code = (unsigned)(err & 15);
        return set_err(ctx, PK_ERR_UNSATISFIED, "witness builder %llu: %s", err >> 4, code < 5 ? what[code] : "error");
    }
    return PK_OK;
}

}  // extern "C"
