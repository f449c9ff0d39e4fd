// provekit_hip.hpp -- C++17 host side ABOVE the C ABI (provekit_hip.h), header only.
//
// The reference's prover is compiled code (Rust); its toolchain is absent here, so this is the host mirror a C++ caller
// links instead of the Rust shim of INTEGRATION.md: the same type names, argument meaning and error behaviour as the
// reference interfaces for this path, each member citing the one it mirrors.  Everything here is plumbing over pk_* calls;
// no arithmetic lives in this file.
//
//   provekit::FieldElement        ark-ff Fp256<MontBackend> in memory: 4 x u64 little-endian, Montgomery (common/src/lib.rs:19)
//   provekit::Error               anyhow::Error at the crate boundary: what() carries pk_last_error()
//   provekit::SparseMatrix, R1CS  provekit_common::{SparseMatrix, R1CS} (common/src/sparse_matrix.rs:12-27, r1cs.rs)
//   provekit::WhirConfig          the WhirConfig fields the prover consumes (tooling/provekit-gnark/src/gnark_config.rs:32-57)
//   provekit::WhirR1CSScheme      {m, m_0, whir_witness, whir_for_hiding_spartan} + WhirR1CSProver::prove
//                                 (common/src/whir_r1cs.rs:17-39, prover/src/whir_r1cs.rs:36-100)
//   provekit::WhirR1CSProof       {transcript} (common/src/whir_r1cs.rs:43-46)
//   provekit::SkyscraperCRH, SkyscraperTwoToOne, MerkleTree
//                                 the ark CRHScheme / TwoToOneCRHScheme / MerkleTree<SkyscraperMerkleConfig> plug-ins
//                                 (common/src/skyscraper/whir.rs:30-86); digests are canonical 32-byte values
//   provekit::SkyscraperPoW       spongefish_pow::PowStrategy {new, check, solve} (common/src/skyscraper/pow.rs:14-30)
//   provekit::compress_many       skyscraper::CompressManyFn (skyscraper/core/src/lib.rs:26)
//   provekit::WitnessBuilders     R1CSSolver::solve_witness_vec over a postcard-encoded &[WitnessBuilder]
//                                 (prover/src/r1cs.rs:29-40, prover/src/witness/witness_builder.rs:27-193)
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "provekit_hip.h"

namespace provekit {

using FieldElement = std::array<uint64_t, 4>;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// one device, one stream; not thread-safe, one per prover thread (SURVEY 8b "Threading")
class Context {
   public:
    explicit Context(int device = 0) {
        int rc = pk_ctx_create(device, &ctx_);
        if (rc) throw Error(rc, "pk_ctx_create failed: no HIP device / bad ordinal (libprovekit_hip has no CPU fallback)");
    }
    ~Context() {
        if (ctx_) pk_ctx_destroy(ctx_);
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    pk_ctx* get() const { return ctx_; }
    void check(int rc) const {
        if (rc) throw Error(rc, pk_last_error(ctx_));
    }
    void sync() const { check(pk_ctx_sync(ctx_)); }
    void set_hash_version(int v) const { check(pk_ctx_set_hash_version(ctx_, v)); }

   private:
    pk_ctx* ctx_ = nullptr;
};

// Vec<FieldElement> resident in HBM
class DeviceVec {
   public:
    DeviceVec(const Context& c, size_t n) : c_(&c), n_(n) { c.check(pk_malloc(c.get(), 32 * (n ? n : 1), &p_)); }
    DeviceVec(const Context& c, const std::vector<FieldElement>& v) : DeviceVec(c, v.size()) {
        if (!v.empty()) c.check(pk_memcpy_h2d(c.get(), p_, v.data(), 32 * v.size()));
    }
    ~DeviceVec() {
        if (p_) pk_free(c_->get(), p_);
    }
    DeviceVec(DeviceVec&& o) noexcept : c_(o.c_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
    DeviceVec(const DeviceVec&) = delete;
    DeviceVec& operator=(const DeviceVec&) = delete;
    uint64_t* data() const { return static_cast<uint64_t*>(p_); }
    size_t size() const { return n_; }
    std::vector<FieldElement> to_host() const {
        std::vector<FieldElement> v(n_);
        if (n_) c_->check(pk_memcpy_d2h(c_->get(), v.data(), p_, 32 * n_));
        return v;
    }

   private:
    const Context* c_;
    void* p_ = nullptr;
    size_t n_;
};

// same fields as the reference struct; `values` index the interner (sparse_matrix.rs:12-27)
struct SparseMatrix {
    size_t num_rows = 0, num_cols = 0;
    std::vector<uint32_t> new_row_indices, col_indices, values;
};

class R1CS {
   public:
    R1CS(const Context& c, const SparseMatrix& a, const SparseMatrix& b, const SparseMatrix& cc, const std::vector<FieldElement>& interner)
        : c_(&c), num_constraints_(a.num_rows), num_witnesses_(a.num_cols) {
        const SparseMatrix* ms[3] = {&a, &b, &cc};
        pk_sparse_matrix mats[3];
        for (int k = 0; k < 3; k++) {
            if (ms[k]->num_rows != num_constraints_ || ms[k]->num_cols != num_witnesses_) throw Error(PK_ERR_BAD_ARG, "matrix shape mismatch");
            mats[k] = {ms[k]->new_row_indices.data(), ms[k]->col_indices.data(), ms[k]->values.data(), ms[k]->col_indices.size()};
        }
        c.check(pk_r1cs_create(c.get(), num_constraints_, num_witnesses_, mats, interner.empty() ? nullptr : interner[0].data(), interner.size(), &h_));
    }
    ~R1CS() {
        if (h_) pk_r1cs_destroy(c_->get(), h_);
    }
    R1CS(const R1CS&) = delete;
    R1CS& operator=(const R1CS&) = delete;
    size_t num_constraints() const { return num_constraints_; }
    size_t num_witnesses() const { return num_witnesses_; }
    pk_r1cs* get() const { return h_; }
    // R1CSSolver::test_witness_satisfaction (prover/src/r1cs.rs:41-60): throws "Constraint {row} failed" / "Witness size does not match"
    void test_witness_satisfaction(const DeviceVec& witness) const {
        int64_t row = -1;
        c_->check(pk_r1cs_test_witness_satisfaction(c_->get(), h_, witness.data(), witness.size(), &row));
    }

   private:
    const Context* c_;
    pk_r1cs* h_ = nullptr;
    size_t num_constraints_, num_witnesses_;
};

struct WhirConfig {
    unsigned n_vars = 0, batch_size = 2, folding_factor = 4, starting_log_inv_rate = 1;
    std::vector<unsigned> num_queries, ood_samples;
    std::vector<double> pow_bits;
    unsigned final_queries = 0;
    double final_pow_bits = 0.0;
    unsigned commitment_ood_samples = 1;

    double final_folding_pow_bits = 0.0;

    // new_whir_config_for_size(num_variables, batch_size) (provekit/r1cs-compiler/src/whir_r1cs.rs:38-53): WhirConfig::new
    // with security 128, ConjectureList, fold 4, rate 1/2, pow_bits = default_max_pow(n, 1), through the library's
    // restatement of whir's derivation (pk_whir_config_derive).  test_pow >= 0 (TESTS ONLY) flattens every grinding
    // difficulty to a cheaper value.
    static WhirConfig for_size(unsigned n_vars, double test_pow = -1.0, unsigned batch = 2) {
        pk_whir_config s{};
        if (int rc = pk_whir_config_derive(n_vars, batch, 4, 1, 128, -1, &s)) throw Error(rc, "pk_whir_config_derive failed");
        WhirConfig c;
        c.n_vars = s.n_vars;
        c.batch_size = s.batch_size;
        c.folding_factor = s.folding_factor;
        c.starting_log_inv_rate = s.starting_log_inv_rate;
        for (unsigned r = 0; r < s.n_rounds; r++) {
            c.num_queries.push_back(s.num_queries[r]);
            c.ood_samples.push_back(s.ood_samples[r]);
            c.pow_bits.push_back(test_pow >= 0.0 ? test_pow : s.pow_bits[r]);
        }
        c.final_queries = s.final_queries;
        c.final_pow_bits = test_pow >= 0.0 ? test_pow : s.final_pow_bits;
        c.commitment_ood_samples = s.commitment_ood_samples;
        c.final_folding_pow_bits = s.final_folding_pow_bits;
        return c;
    }
    // new_whir_config_for_size(next_power_of_two(4 m_0) + 1, 2) (r1cs-compiler/src/whir_r1cs.rs:31-34)
    static WhirConfig for_hiding_spartan(unsigned m_0, double test_pow = -1.0) {
        unsigned nb = 0;
        while ((1u << nb) < 4 * m_0) nb++;
        return for_size(nb + 1, test_pow);
    }

    pk_whir_config to_c() const {
        if (num_queries.size() > PK_MAX_WHIR_ROUNDS || ood_samples.size() != num_queries.size() || pow_bits.size() != num_queries.size())
            throw Error(PK_ERR_BAD_ARG, "WhirConfig: per-round vectors disagree");
        pk_whir_config s{};
        s.n_vars = n_vars;
        s.batch_size = batch_size;
        s.folding_factor = folding_factor;
        s.starting_log_inv_rate = starting_log_inv_rate;
        s.n_rounds = (unsigned)num_queries.size();
        for (size_t i = 0; i < num_queries.size(); i++) {
            s.num_queries[i] = num_queries[i];
            s.ood_samples[i] = ood_samples[i];
            s.pow_bits[i] = pow_bits[i];
        }
        s.final_queries = final_queries;
        s.final_pow_bits = final_pow_bits;
        s.commitment_ood_samples = commitment_ood_samples;
        s.final_folding_pow_bits = final_folding_pow_bits;
        return s;
    }
};

struct WhirR1CSProof {
    std::vector<uint8_t> transcript;
};

class WhirR1CSScheme {
   public:
    unsigned m, m_0;
    WhirConfig whir_witness, whir_for_hiding_spartan;

    // binds the scheme to an uploaded R1CS; the ensure!() checks of prove (prover/src/whir_r1cs.rs:43-54) that depend only
    // on the scheme and the R1CS fire here ("R1CS witness length exceeds scheme capacity", "... constraints exceed ...")
    WhirR1CSScheme(const Context& c, const R1CS& r1cs, unsigned m_, unsigned m_0_, WhirConfig w, WhirConfig b)
        : m(m_), m_0(m_0_), whir_witness(std::move(w)), whir_for_hiding_spartan(std::move(b)), c_(&c), n_witness_(r1cs.num_witnesses()) {
        pk_whir_config cw = whir_witness.to_c(), cb = whir_for_hiding_spartan.to_c();
        c.check(pk_scheme_create(c.get(), r1cs.get(), r1cs.num_constraints(), r1cs.num_witnesses(), m, m_0, &cw, &cb, &h_));
    }
    ~WhirR1CSScheme() {
        if (h_) pk_scheme_destroy(c_->get(), h_);
    }
    WhirR1CSScheme(const WhirR1CSScheme&) = delete;
    WhirR1CSScheme& operator=(const WhirR1CSScheme&) = delete;

    // WhirR1CSProver::prove(&self, &R1CS, Vec<FieldElement>) -> Result<WhirR1CSProof>: the witness moves to the device and
    // the proof string comes back; "Unexpected witness length for R1CS instance" is thrown for a wrong length.
    // Randomness: like the reference (thread_rng) every proof draws fresh masks from the OS CSPRNG.  The overloads taking a
    // TestSeed inject the 256-bit key instead: a test hook for reproducible transcripts only.
    using TestSeed = std::array<uint8_t, 32>;
    static TestSeed test_seed(uint64_t v) {
        TestSeed s{};
        for (int i = 0; i < 8; i++) s[i] = (uint8_t)(v >> (8 * i));
        return s;
    }
    WhirR1CSProof prove(const std::vector<FieldElement>& witness, const TestSeed* seed = nullptr) const {
        DeviceVec d(*c_, witness);
        return prove(d, seed);
    }
    WhirR1CSProof prove(const DeviceVec& d_witness, const TestSeed* seed = nullptr) const {
        WhirR1CSProof p;
        p.transcript.resize((size_t)4 << 20);  // proofs of this scheme are a few hundred KiB (268,756 B at the poseidon size)
        p.transcript.resize(prove_into(d_witness, seed, p.transcript));
        return p;
    }
    // proves into a caller buffer sized once (no size query): the steady-state form
    size_t prove_into(const DeviceVec& d_witness, const TestSeed* seed, std::vector<uint8_t>& buf) const {
        size_t len = 0;
        c_->check(pk_prove(c_->get(), h_, d_witness.data(), d_witness.size(), seed ? seed->data() : nullptr, buf.data(), buf.size(), &len));
        return len;
    }
    pk_scheme* get() const { return h_; }
    const Context& context() const { return *c_; }
    // WhirR1CSScheme::create_io_pattern (provekit/common/src/whir_r1cs.rs:28-39): the bytes of the IO pattern in force; a caller
    // that holds the reference's own (`create_io_pattern().as_bytes()`) installs them -- refused unless they declare the
    // operations the prover performs ("... IO pattern ..." is thrown, pk_scheme_set_io_pattern)
    std::string create_io_pattern() const { return domain_separator(); }
    void set_io_pattern(const std::string& bytes) { c_->check(pk_scheme_set_io_pattern(c_->get(), h_, (const uint8_t*)bytes.data(), bytes.size())); }
    std::string domain_separator() const {
        size_t n = 0;
        pk_scheme_domain_separator(h_, nullptr, 0, &n);
        std::string s(n, '\0');
        pk_scheme_domain_separator(h_, s.data(), n, &n);
        return s;
    }

   private:
    const Context* c_;
    pk_scheme* h_ = nullptr;
    size_t n_witness_;
};

using Digest = std::array<uint64_t, 4>;  // canonical little-endian limbs, as the transcript carries them (whir.rs:96-102)

// SkyscraperCRH::evaluate (whir.rs:30-48): leaf.iter().copied().reduce(compress) over Montgomery field elements
struct SkyscraperCRH {
    static Digest evaluate(const Context& c, const std::vector<FieldElement>& leaf) {
        if (leaf.empty()) throw Error(PK_ERR_BAD_ARG, "empty leaf");  // the reference unwraps the reduce() of an empty iterator
        DeviceVec d(c, leaf), out(c, 1);
        c.check(pk_leaf_hash(c.get(), d.data(), 1, leaf.size(), PK_LEAF_MAJOR, out.data()));
        return out.to_host()[0];
    }
};
// SkyscraperTwoToOne::{evaluate, compress} (whir.rs:53-74)
struct SkyscraperTwoToOne {
    static Digest compress(const Context& c, const Digest& left, const Digest& right) {
        std::vector<uint8_t> msg(64), h(32);
        std::memcpy(msg.data(), left.data(), 32);
        std::memcpy(msg.data() + 32, right.data(), 32);
        c.check(pk_compress_many_host(c.get(), msg.data(), 64, h.data(), 32));
        Digest d;
        std::memcpy(d.data(), h.data(), 32);
        return d;
    }
};
// ark MerkleTree::new over leaves already on the device + generate_multi_proof (wire form of types.go:17-22)
class MerkleTree {
   public:
    // leaves: n_leaves x width field elements, leaf-major; n_leaves must be a power of two (ark asserts the same)
    MerkleTree(const Context& c, const DeviceVec& leaves, size_t n_leaves, size_t width) : c_(&c), n_(n_leaves), width_(width) {
        if (leaves.size() != n_leaves * width) throw Error(PK_ERR_BAD_ARG, "leaves.size() != n_leaves * width");
        c.check(pk_tree_from_leaves(c.get(), leaves.data(), n_leaves, width, PK_LEAF_MAJOR, reinterpret_cast<uint8_t*>(root_.data()), &t_));
    }
    ~MerkleTree() {
        if (t_) pk_tree_destroy(c_->get(), t_);
    }
    MerkleTree(const MerkleTree&) = delete;
    MerkleTree& operator=(const MerkleTree&) = delete;
    Digest root() const { return root_; }
    // sorted, de-duplicated leaf indices -> ark-serialized MultiPath bytes (what the "merkle_proof" hint carries)
    std::vector<uint8_t> generate_multi_proof(const std::vector<uint64_t>& indices, std::vector<FieldElement>* opened_leaves = nullptr) const {
        const size_t k = indices.size();
        size_t logn = 0;
        while (((size_t)1 << logn) < n_) logn++;
        const size_t plen = logn ? logn - 1 : 0;
        std::vector<FieldElement> lv(k * width_), sib(k ? k : 1), paths((k && plen) ? k * plen : 1);
        c_->check(pk_tree_open(c_->get(), t_, indices.data(), k, 1, lv.empty() ? nullptr : lv[0].data(), sib[0].data(), paths[0].data()));
        size_t len = 0;
        pk_multipath_serialize(indices.data(), k, plen, sib[0].data(), paths[0].data(), nullptr, 0, &len);
        std::vector<uint8_t> out(len);
        int rc = pk_multipath_serialize(indices.data(), k, plen, sib[0].data(), paths[0].data(), out.data(), out.size(), &len);
        if (rc) throw Error(rc, "pk_multipath_serialize failed");
        if (opened_leaves) *opened_leaves = std::move(lv);
        return out;
    }

   private:
    const Context* c_;
    pk_tree* t_ = nullptr;
    size_t n_, width_;
    Digest root_{};
};

// spongefish_pow::PowStrategy for Skyscraper (common/src/skyscraper/pow.rs:14-30)
class SkyscraperPoW {
   public:
    // new(challenge, bits): "bits must be smaller than 60" (pow.rs:16)
    SkyscraperPoW(const Context& c, const std::array<uint8_t, 32>& challenge, double bits) : c_(&c), challenge_(challenge), bits_(bits) {
        if (!(bits >= 0.0 && bits < 60.0)) throw Error(PK_ERR_BAD_ARG, "bits must be smaller than 60");
    }
    bool check(uint64_t nonce) const {
        int ok = 0;
        c_->check(pk_pow_check(c_->get(), challenge_.data(), bits_, nonce, &ok));
        return ok != 0;
    }
    std::optional<uint64_t> solve() const {
        uint64_t nonce = 0;
        c_->check(pk_pow_solve(c_->get(), challenge_.data(), bits_, &nonce));
        return nonce;
    }

   private:
    const Context* c_;
    std::array<uint8_t, 32> challenge_;
    double bits_;
};

// R1CSSolver::solve_witness_vec (prover/src/r1cs.rs:29-40): the builder list is handed over once (postcard of
// Vec<WitnessBuilder>, the form it has inside a .nps), levelled by data dependence and kept on the device; a proof supplies the
// ACIR witness map as a dense vector and the challenges the transcript draws for the Challenge builders, in list order.
// A list the reference would panic on at creation ("reads witness .. before it is solved") or while solving ("inverse of zero",
// "Higher order bits are not zero", an index out of range) throws Error with that text.
class WitnessBuilders {
   public:
    WitnessBuilders(const Context& c, const std::vector<uint8_t>& postcard_builders) : c_(&c) {
        c.check(pk_witness_builders_from_postcard(c.get(), postcard_builders.data(), postcard_builders.size(), &p_, &n_witnesses_, &n_challenges_, &n_acir_));
    }
    ~WitnessBuilders() {
        if (p_) pk_witness_program_destroy(c_->get(), p_);
    }
    WitnessBuilders(const WitnessBuilders&) = delete;
    WitnessBuilders& operator=(const WitnessBuilders&) = delete;
    size_t num_challenges() const { return n_challenges_; }
    size_t num_acir_witnesses() const { return n_acir_; }
    size_t num_witnesses_touched() const { return n_witnesses_; }
    pk_witness_program* get() const { return p_; }
    // -> Vec<Option<FieldElement>> of length num_witnesses (fill_witness's random filling of the None entries stays with the caller)
    std::vector<std::optional<FieldElement>> solve_witness_vec(const std::vector<FieldElement>& acir_witness_values,
                                                               const std::vector<FieldElement>& challenges, size_t num_witnesses) const {
        if (num_witnesses < n_witnesses_) num_witnesses = n_witnesses_;
        DeviceVec acir(*c_, acir_witness_values), wit(*c_, num_witnesses);
        void* d_set = nullptr;
        c_->check(pk_malloc(c_->get(), num_witnesses ? num_witnesses : 1, &d_set));
        int rc = pk_witness_solve(c_->get(), p_, acir.data(), acir_witness_values.size(), challenges.empty() ? nullptr : challenges[0].data(),
                                  challenges.size(), wit.data(), num_witnesses, static_cast<uint8_t*>(d_set));
        std::vector<uint8_t> set(num_witnesses);
        if (!rc && num_witnesses) rc = pk_memcpy_d2h(c_->get(), set.data(), d_set, num_witnesses);
        pk_free(c_->get(), d_set);
        c_->check(rc);
        const std::vector<FieldElement> w = wit.to_host();
        std::vector<std::optional<FieldElement>> out(num_witnesses);
        for (size_t i = 0; i < num_witnesses; i++)
            if (set[i]) out[i] = w[i];
        return out;
    }

   private:
    const Context* c_;
    pk_witness_program* p_ = nullptr;
    size_t n_witnesses_ = 0, n_challenges_ = 0, n_acir_ = 0;
};

// the witness transcript (create_witness_io_pattern + seed_witness_merlin, prover/src/noir_proof_scheme.rs:94-133) and the
// challenge each WitnessBuilder::Challenge draws from it, in list order.  Host only.
inline std::vector<FieldElement> witness_challenges(size_t num_constraints, size_t num_witnesses, const std::vector<FieldElement>& public_inputs,
                                                    size_t n_challenges) {
    std::vector<FieldElement> out(n_challenges);
    int rc = pk_witness_challenges(num_constraints, num_witnesses, public_inputs.empty() ? nullptr : public_inputs[0].data(), public_inputs.size(),
                                   n_challenges ? out[0].data() : nullptr, n_challenges);
    if (rc) throw Error(rc, "pk_witness_challenges");
    return out;
}

// NoirProofSchemeProver::prove after ACVM execution (noir_proof_scheme.rs:69-91) in one call: witness transcript, witness builders,
// fill_witness, WhirR1CSProver::prove -- the R1CS witness is born and stays on the device.  acir_witness_values: the ACIR witness
// map as a dense vector indexed by ACIR witness index; public_acir_idx: Circuit::public_inputs().indices().
inline WhirR1CSProof noir_prove(const WhirR1CSScheme& scheme, const WitnessBuilders& builders, const DeviceVec& acir_witness_values,
                                const std::vector<uint32_t>& public_acir_idx, const WhirR1CSScheme::TestSeed* seed = nullptr) {
    WhirR1CSProof p;
    p.transcript.resize((size_t)4 << 20);
    size_t len = 0;
    scheme.context().check(pk_noir_prove(scheme.context().get(), scheme.get(), builders.get(), acir_witness_values.data(), acir_witness_values.size(),
                                         public_acir_idx.data(), public_acir_idx.size(), seed ? seed->data() : nullptr, p.transcript.data(),
                                         p.transcript.size(), &len));
    p.transcript.resize(len);
    return p;
}

// skyscraper::CompressManyFn = fn(&[u8] /*64 n*/, &mut [u8] /*32 n*/); the reference panics on a length mismatch
// (generic.rs:18-25), this throws
inline void compress_many(const Context& c, const std::vector<uint8_t>& messages, std::vector<uint8_t>& hashes) {
    c.check(pk_compress_many_host(c.get(), messages.data(), messages.size(), hashes.data(), hashes.size()));
}

}  // namespace provekit
