// prove_demo.cpp -- the C++ host side (include/provekit_hip.hpp) end to end, no Python in the loop:
// build a satisfiable R1CS, upload it, check the witness, prove, exercise the error paths, write the proof.
//
//   prove_demo <m> <m_0> <num_constraints> <num_inputs> <seed> <out_prefix>
//
// Row i of the instance: (sum a z)(sum b z) = z[1 + num_inputs + i]; A and B read only the constant and the inputs, so the
// outputs are one Hadamard product -- formed on the device with the library's own field kernels.
// Writes <out_prefix>.transcript (the WhirR1CSProof string) and <out_prefix>.ds (the domain separator);
// tests/test_gpu_cpp_host.py hands both to the independent verifier.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "provekit_hip.hpp"

using namespace provekit;

static uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    if (argc != 7) {
        std::fprintf(stderr, "usage: %s m m_0 num_constraints num_inputs seed out_prefix\n", argv[0]);
        return 2;
    }
    const unsigned m = std::atoi(argv[1]), m_0 = std::atoi(argv[2]);
    const size_t nc = std::strtoull(argv[3], nullptr, 10), n_in = std::strtoull(argv[4], nullptr, 10);
    uint64_t rng = std::strtoull(argv[5], nullptr, 10);
    const std::string prefix = argv[6];
    const size_t nw = 1 + n_in + nc;
    try {
        Context ctx(0);
        // interner: small canonical constants, converted to Montgomery by the library (FieldElement::new)
        const uint64_t small[8] = {1, 2, 3, 5, 7, 11, 13, 17};
        std::vector<FieldElement> canon(8, FieldElement{0, 0, 0, 0});
        for (int i = 0; i < 8; i++) canon[i][0] = small[i];
        DeviceVec d_canon(ctx, canon), d_int(ctx, 8);
        ctx.check(pk_fe_to_mont(ctx.get(), d_canon.data(), d_int.data(), 8));
        std::vector<FieldElement> interner = d_int.to_host();

        SparseMatrix A, B, Cm;
        for (SparseMatrix* M : {&A, &B}) {
            M->num_rows = nc;
            M->num_cols = nw;
            for (size_t i = 0; i < nc; i++) {
                M->new_row_indices.push_back((uint32_t)M->col_indices.size());
                uint32_t c0 = (uint32_t)(splitmix(rng) % (1 + n_in - 2));
                for (uint32_t k = 0; k < 3; k++) {  // three distinct ascending columns among [0, 1 + n_in)
                    M->col_indices.push_back(c0 + k > n_in ? (uint32_t)n_in : c0 + k);
                    M->values.push_back((uint32_t)(splitmix(rng) % 8));
                }
            }
        }
        Cm.num_rows = nc;
        Cm.num_cols = nw;
        for (size_t i = 0; i < nc; i++) {
            Cm.new_row_indices.push_back((uint32_t)i);
            Cm.col_indices.push_back((uint32_t)(1 + n_in + i));
            Cm.values.push_back(0);  // coefficient 1
        }
        R1CS r1cs(ctx, A, B, Cm, interner);

        // witness: [1 | inputs < 2^250 (valid Montgomery images) | outputs]
        std::vector<FieldElement> z(nw, FieldElement{0, 0, 0, 0});
        z[0] = interner[0];
        for (size_t i = 1; i <= n_in; i++) z[i] = {splitmix(rng), splitmix(rng), splitmix(rng), splitmix(rng) >> 6};
        DeviceVec d_z(ctx, z), d_az(ctx, nc), d_bz(ctx, nc);
        ctx.check(pk_r1cs_matvec(ctx.get(), r1cs.get(), 0, 0, d_z.data(), d_az.data()));
        ctx.check(pk_r1cs_matvec(ctx.get(), r1cs.get(), 1, 0, d_z.data(), d_bz.data()));
        ctx.check(pk_fe_mul(ctx.get(), d_az.data(), d_bz.data(), d_z.data() + 4 * (1 + n_in), nc));  // outputs in place
        r1cs.test_witness_satisfaction(d_z);

        WhirR1CSScheme scheme(ctx, r1cs, m, m_0, WhirConfig::for_size(m, 8.0), WhirConfig::for_hiding_spartan(m_0, 8.0));
        const auto seed = WhirR1CSScheme::test_seed(42);  // test hook: reproducible transcript
        WhirR1CSProof proof = scheme.prove(d_z, &seed);
        WhirR1CSProof again = scheme.prove(d_z, &seed);
        if (proof.transcript != again.transcript) throw Error(-100, "same witness and seed gave different transcripts");
        WhirR1CSProof fresh = scheme.prove(d_z);  // production form: masks from the OS CSPRNG, a different transcript every time
        if (fresh.transcript == proof.transcript) throw Error(-101, "fresh randomness reproduced the seeded transcript");

        // error behaviour of the reference's ensure!() / test_witness_satisfaction
        int seen = 0;
        try {
            DeviceVec shorter(ctx, nw - 1);
            scheme.prove(shorter);
        } catch (const Error& e) {
            seen += std::string(e.what()).find("Unexpected witness length") != std::string::npos;
        }
        try {
            std::vector<FieldElement> bad = d_z.to_host();
            bad[nw - 1][0] ^= 1;
            DeviceVec d_bad(ctx, bad);
            r1cs.test_witness_satisfaction(d_bad);
        } catch (const Error& e) {
            seen += e.code == PK_ERR_UNSATISFIED && std::string(e.what()) == "Constraint " + std::to_string(nc - 1) + " failed";
        }
        try {
            WhirR1CSScheme too_small(ctx, r1cs, 4, m_0, WhirConfig::for_size(4), WhirConfig::for_hiding_spartan(m_0));
        } catch (const Error& e) {
            seen += std::string(e.what()).find("exceeds scheme capacity") != std::string::npos;
        }
        // the Merkle plug-ins agree with each other: a 4-leaf tree's root == the two-level chain of CRH / TwoToOne calls
        {
            std::vector<FieldElement> lf(z.begin() + 1, z.begin() + 1 + 4 * 5);  // 4 leaves of width 5
            DeviceVec d_lf(ctx, lf);
            MerkleTree tree(ctx, d_lf, 4, 5);
            Digest h[4];
            for (int i = 0; i < 4; i++) h[i] = SkyscraperCRH::evaluate(ctx, std::vector<FieldElement>(lf.begin() + 5 * i, lf.begin() + 5 * i + 5));
            Digest want = SkyscraperTwoToOne::compress(ctx, SkyscraperTwoToOne::compress(ctx, h[0], h[1]), SkyscraperTwoToOne::compress(ctx, h[2], h[3]));
            if (want != tree.root()) throw Error(-102, "MerkleTree root != chained CRH / TwoToOne compressions");
            std::vector<FieldElement> opened;
            std::vector<uint8_t> mp = tree.generate_multi_proof({1, 2}, &opened);
            if (opened.size() != 10 || mp.size() < 8 * 4 + 32 * 2) throw Error(-103, "generate_multi_proof shape");
        }
        // the witness builders (R1CSSolver::solve_witness_vec): a four-builder list written as postcard by hand --
        // Constant(0, 1), Acir(1, 0), Product(2, 1, 1), Inverse(3, 2) -- must give w2 = x^2 and w3 * w2 = 1, leave w4 unset
        {
            std::vector<uint8_t> pc = {4};  // Vec length
            auto varint = [&](uint64_t v) {
                do {
                    uint8_t b = v & 0x7f;
                    v >>= 7;
                    pc.push_back(v ? b | 0x80 : b);
                } while (v);
            };
            varint(0); varint(0); varint(32);  // Constant(ConstantTerm(0, 1)): serde_ark = bytes(32), canonical little-endian
            pc.push_back(1);
            for (int i = 1; i < 32; i++) pc.push_back(0);
            varint(1); varint(1); varint(0);              // Acir(1, 0)
            varint(3); varint(2); varint(1); varint(1);   // Product(2, 1, 1)
            varint(7); varint(3); varint(2);              // Inverse(3, 2)
            WitnessBuilders wb(ctx, pc);
            const FieldElement x = z[1];
            auto w = wb.solve_witness_vec({x}, {}, 5);
            DeviceVec d_a(ctx, std::vector<FieldElement>{x, *w[3]}), d_b(ctx, std::vector<FieldElement>{x, *w[2]}), d_o(ctx, 2);
            ctx.check(pk_fe_mul(ctx.get(), d_a.data(), d_b.data(), d_o.data(), 2));
            const auto prod = d_o.to_host();
            if (!w[0] || *w[0] != interner[0] || !w[1] || *w[1] != x || !w[2] || *w[2] != prod[0] || prod[1] != interner[0] || w[4])
                throw Error(-104, "witness builders: Constant / Acir / Product / Inverse / None pattern");
            // NoirProofSchemeProver::prove after ACVM: with every witness an ACIR value (Constant(0, 1), Acir(j, j - 1)) nothing is
            // filled, so the one-call form must reproduce scheme.prove's seeded transcript byte for byte
            pc.clear();
            varint(nw);
            varint(0); varint(0); varint(32);
            pc.push_back(1);
            for (int i = 1; i < 32; i++) pc.push_back(0);
            for (size_t j = 1; j < nw; j++) { varint(1); varint(j); varint(j - 1); }
            WitnessBuilders all_acir(ctx, pc);
            const std::vector<FieldElement> zh = d_z.to_host();
            DeviceVec d_acir(ctx, std::vector<FieldElement>(zh.begin() + 1, zh.end()));
            if (noir_prove(scheme, all_acir, d_acir, {0, 1}, &seed).transcript != proof.transcript)
                throw Error(-105, "noir_prove differs from prove on the same witness and seed");
            if (witness_challenges(nc, nw, {z[1], z[2]}, 2) == witness_challenges(nc, nw, {z[1], z[1]}, 2))
                throw Error(-106, "the witness transcript ignores a public input");
            try {  // a list the reference would panic on: Product reads witness 9, which nobody solves
                std::vector<uint8_t> bad = {1, 3, 2, 9, 9};
                WitnessBuilders refuse(ctx, bad);
            } catch (const Error& e) {
                seen += std::string(e.what()).find("before it is solved") != std::string::npos;
            }
        }
        SkyscraperPoW pow(ctx, std::array<uint8_t, 32>{1, 2, 3}, 10.0);
        const uint64_t nonce = *pow.solve();
        seen += pow.check(nonce) ? 1 : 0;
        if (seen != 5) throw Error(-101, "error-path checks: " + std::to_string(seen) + " of 5");

        std::ofstream(prefix + ".transcript", std::ios::binary).write((const char*)proof.transcript.data(), (std::streamsize)proof.transcript.size());
        const std::string ds = scheme.domain_separator();
        std::ofstream(prefix + ".ds", std::ios::binary).write(ds.data(), (std::streamsize)ds.size());
        std::printf("ok transcript_bytes=%zu constraints=%zu witnesses=%zu pow_nonce=%llu\n", proof.transcript.size(), nc, nw, (unsigned long long)nonce);
        return 0;
    } catch (const Error& e) {
        std::fprintf(stderr, "provekit::Error %d: %s\n", e.code, e.what());
        return 1;
    }
}
