#!/usr/bin/env python3
"""Time-boxed search for the domain separator (spongefish IO pattern) behind the reference's own proof
tooling/provekit-bench/benches/poseidon-1000.np  (VERDICT r03 item 1b).

The sponge IV is Keccak(io_pattern bytes); the order of sponge operations is pinned by the in-tree Go verifier, so the
only unknowns are the label strings (they live in whir @3e7f8c2 / spongefish, both absent) and -- because the fixture
predates the Skyscraper v2 switch -- the permutation the sponge ran at the time.  It is all-or-nothing: a candidate is
right iff replaying the proof's first absorbs gives the blinding commitment's OOD point that gen_fixture_whir.py
recovered from the proof by algebra (tests/golden/fixture_whir.json: blinding.ood_point).

BUILD CONTAINER ONLY (reads /root/reference).  Writes the list of patterns tried to profiles/r04_iopattern_search.json.
"""
import hashlib
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import pyref as pr  # noqa: E402
import verifier as V  # noqa: E402

P = pr.P
FIXTURE = "/root/reference/tooling/provekit-bench/benches/poseidon-1000.np"


# ------------------------------------------------------------------ permutation candidates
def permute_v2(l, r):
    return pr.permute(l, r)


def _v1_rounds(l, r):
    sq, bar, RC = pr.sq, pr.bar, pr.RC
    l, r = (r + sq(l)) % P, l
    l, r = (r + sq(l) + RC[1]) % P, l
    l, r = (r + bar(l) + RC[2]) % P, l
    l, r = (r + bar(l) + RC[3]) % P, l
    l, r = (r + sq(l) + RC[4]) % P, l
    l, r = (r + sq(l) + RC[5]) % P, l
    l, r = (r + bar(l) + RC[6]) % P, l
    l, r = (r + bar(l) + RC[7]) % P, l
    l, r = (r + sq(l) + RC[8]) % P, l
    l, r = (r + sq(l)) % P, l
    return l, r


def permute_v1(l, r):  # the ten v1 rounds (v1.rs:19-32) without the feed-forward
    return _v1_rounds(l % P, r % P)


def permute_v1_ff(l, r):  # ... with it (state[0] = compress(l, r))
    a, b = _v1_rounds(l % P, r % P)
    return (a + l) % P, b


def permute_v1_c(l, r):  # state = [compress(l, r), l]
    a, _ = _v1_rounds(l % P, r % P)
    return (a + l) % P, l % P


def permute_v1_c2(l, r):  # state = [compress(l, r), r]
    a, _ = _v1_rounds(l % P, r % P)
    return (a + l) % P, r % P


PERMS = {"v2": permute_v2, "v1": permute_v1, "v1+ff": permute_v1_ff, "v1:[c,l]": permute_v1_c, "v1:[c,r]": permute_v1_c2}


# ------------------------------------------------------------------ IO pattern builder
class Pattern:
    def __init__(self, proto, sep=b"\0"):
        self.b = bytearray(proto)
        self.sep = sep

    def op(self, kind, count, label):
        self.b += self.sep + kind.encode() + (str(count).encode() if count is not None else b"") + label.encode()
        return self

    def A(self, n, label):
        return self.op("A", n, label)

    def S(self, n, label):
        return self.op("S", n, label)

    def H(self, label):
        return self.op("H", None, label)


def build(m_0, cfg_w, cfg_b, L, opt):
    """L: label dict; opt: structural options"""
    p = Pattern(opt["proto"])

    def challenge_bytes(n, label):
        p.S(-(-n // 15), label)

    def pow_(bits):
        if bits > 0:
            challenge_bytes(32, L["pow_queries"])
            p.A(8, "pow-nonce")

    def add_ood(n, batch=1):
        if n > 0:
            p.S(n, L["ood_query"])
            if opt["ood_ans_split"] and batch > 1:
                for _ in range(batch):
                    p.A(n, L["ood_ans"])
            else:
                p.A(n * batch, L["ood_ans"])

    def sumcheck(k, bits):
        for _ in range(k):
            p.A(3, L["sumcheck_poly"])
            p.S(1, L["folding_randomness"])
            pow_(bits)

    def commit_statement(c):
        p.A(1, L["merkle_digest"])
        add_ood(c.commitment_ood_samples, c.batch_size)
        if opt["batching_at"] == "commit" and c.batch_size > 1:
            p.S(1, L["batching_randomness"])

    def whir_proof(c):
        if opt["batching_at"] == "prove" and c.batch_size > 1:
            p.S(1, L["batching_randomness"])
        p.S(1, L["initial_combination_randomness"])
        sumcheck(c.folding_factor, 0)
        domain = 1 << (c.n_vars + c.starting_log_inv_rate)
        for r in range(len(c.num_queries)):
            folded = domain >> c.folding_factor
            nbytes = ((folded * 2 - 1).bit_length() - 1 + 7) // 8
            p.A(1, L["merkle_digest"])
            add_ood(c.ood_samples[r])
            if opt["pow_first"]:
                pow_(c.pow_bits[r])
            challenge_bytes(c.num_queries[r] * nbytes, L["stir_queries"])
            if not opt["pow_first"]:
                pow_(c.pow_bits[r])
            if opt["hints"]:
                p.H("stir_answers")
                p.H("merkle_proof")
            p.S(1, L["combination_randomness"])
            sumcheck(c.folding_factor, 0)
            domain >>= 1
        folded = domain >> c.folding_factor
        nbytes = ((folded * 2 - 1).bit_length() - 1 + 7) // 8
        final_vars = c.n_vars - c.folding_factor * (len(c.num_queries) + 1)
        p.A(1 << final_vars, L["final_coeffs"])
        if opt["pow_first"]:
            pow_(c.final_pow_bits)
        challenge_bytes(nbytes * c.final_queries, L["final_queries"])
        if not opt["pow_first"]:
            pow_(c.final_pow_bits)
        if opt["hints"]:
            p.H("stir_answers")
            p.H("merkle_proof")
        sumcheck(final_vars, c.final_folding_pow_bits)
        if opt["hints"]:
            p.H("deferred_weight_evaluations")

    commit_statement(cfg_w)
    p.S(m_0, "rand")
    commit_statement(cfg_b)
    p.A(1, "Sum of G over boolean hypercube")
    p.S(1, "Rho")
    for _ in range(m_0):
        p.A(4, "Sumcheck Polynomials")
        p.S(1, "Sumcheck Random")
    p.A(2, "Polynomial sums")
    whir_proof(cfg_b)
    if opt["hints"]:
        p.H("claimed_evaluations")
    whir_proof(cfg_w)
    return bytes(p.b)


# ------------------------------------------------------------------ the check
def iv_variants(pattern: bytes):
    tag = V.keccak_tag(pattern)
    yield "keccak-duplex/le", int.from_bytes(tag, "little") % P
    yield "keccak-duplex/be", int.from_bytes(tag, "big") % P
    for name, h in (("sha3-256", hashlib.sha3_256), ("sha256", hashlib.sha256)):
        d = h(pattern).digest()
        yield name + "/le", int.from_bytes(d, "little") % P
        yield name + "/be", int.from_bytes(d, "big") % P


def replay_to_blinding_ood(t, iv, perm, m_0, batching_at_commit, iv_pos=1):
    st = [0, 0]
    st[iv_pos] = iv
    absorb_pos, squeeze_pos = 0, 1

    def absorb(x):
        nonlocal st, absorb_pos, squeeze_pos
        if absorb_pos == 1:
            st = list(perm(*st))
            absorb_pos = 0
        st[0] = x
        absorb_pos, squeeze_pos = 1, 1

    def squeeze():
        nonlocal st, absorb_pos, squeeze_pos
        if squeeze_pos == 1:
            squeeze_pos = absorb_pos = 0
            st = list(perm(*st))
        squeeze_pos = 1
        return st[0]

    sc = lambda off: int.from_bytes(t[off : off + 32], "little")
    absorb(sc(0))
    squeeze()  # witness OOD point
    absorb(sc(32))
    absorb(sc(64))
    if batching_at_commit:
        squeeze()
    for _ in range(m_0):
        squeeze()
    absorb(sc(96))
    return squeeze()


def main():
    from provekit_amd.file import read_np
    from provekit_amd.scheme import WhirConfig, blinding_config_for

    t = read_np(FIXTURE)
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixture_whir.json")))
    want = int(fx["blinding"]["ood_point"], 16)
    m, m_0 = 21, 20
    cfg_w, cfg_b = WhirConfig.for_size(m), blinding_config_for(m_0)

    label_opts = {
        "merkle_digest": ["merkle_digest"],
        "ood_query": ["ood_query"],
        "ood_ans": ["ood_ans"],
        "batching_randomness": ["batching_randomness", "batching_rand", "batching randomness", "batch_randomness"],
        "initial_combination_randomness": ["initial_combination_randomness"],
        "sumcheck_poly": ["sumcheck_poly"],
        "folding_randomness": ["folding_randomness"],
        "pow_queries": ["pow_queries", "pow-queries"],
        "stir_queries": ["stir_queries"],
        "combination_randomness": ["combination_randomness"],
        "final_coeffs": ["final_coeffs"],
        "final_queries": ["final_queries"],
    }
    struct_opts = {
        "proto": ["🌪️".encode(), "🌪".encode()],
        "ood_ans_split": [False, True],
        "batching_at": ["commit", "prove", "none"],
        "pow_first": [True, False],
        "hints": [True, False],
    }
    tried, hit, t0 = [], None, time.time()
    lkeys, skeys = list(label_opts), list(struct_opts)
    wide = "--wide" in sys.argv
    if wide:  # phase B: a wider label grid, the HEAD-style IV (Keccak duplex tag, either byte order) only
        import ctypes as C

        from provekit_amd._lib import lib

        def fast_tag(pat):
            out = (C.c_uint8 * 32)()
            lib.pk_selftest_keccak_tag(pat, len(pat), out)
            return bytes(out)

        label_opts.update({
            "merkle_digest": ["merkle_digest", "merkle_root", "root"],
            "ood_query": ["ood_query", "ood_queries"],
            "ood_ans": ["ood_ans", "ood_answers"],
            "initial_combination_randomness": ["initial_combination_randomness", "combination_randomness"],
            "pow_queries": ["pow_queries", "pow-queries", "pow_challenge"],
            "stir_queries": ["stir_queries", "stir_challenges"],
            "final_coeffs": ["final_coeffs", "final_coefficients"],
        })
        struct_opts["proto"] = ["🌪️".encode()]
        struct_opts["hints"] = [True]
        for lv in itertools.product(*label_opts.values()):
            L = dict(zip(lkeys, lv))
            for sv in itertools.product(*struct_opts.values()):
                opt = dict(zip(skeys, sv))
                pat = build(m_0, cfg_w, cfg_b, L, opt)
                tag = fast_tag(pat)
                for ivname, iv in (("keccak-duplex/le", int.from_bytes(tag, "little") % P), ("keccak-duplex/be", int.from_bytes(tag, "big") % P)):
                    for pname, perm in PERMS.items():
                        for b_at_commit in (True, False):
                            if replay_to_blinding_ood(t, iv, perm, m_0, b_at_commit, 1) == want:
                                hit = dict(labels=L, options={k: (v.decode() if isinstance(v, bytes) else v) for k, v in opt.items()}, iv=ivname,
                                           permutation=pname, batching_squeeze_at_commit=b_at_commit, pattern_hex=pat.hex())
                                print("HIT", hit)
                tried.append(hashlib.sha256(pat).hexdigest()[:12])
        out = dict(phase="B (wide label grid, Keccak-duplex IV in either byte order, 5 permutations, batching squeeze at commit or not)",
                   label_grid=label_opts, structural_grid={k: [(x.decode() if isinstance(x, bytes) else x) for x in v] for k, v in struct_opts.items()},
                   patterns_tried=len(tried), seconds=round(time.time() - t0, 1), hit=hit)
        json.dump(out, open(os.path.join(ROOT, "profiles", "r04_iopattern_search_wide.json"), "w"), indent=1, ensure_ascii=False)
        print(f"wide: {len(tried)} patterns, hit = {hit is not None} ({time.time() - t0:.0f} s)")
        return
    for lv in itertools.product(*label_opts.values()):
        L = dict(zip(lkeys, lv))
        for sv in itertools.product(*struct_opts.values()):
            opt = dict(zip(skeys, sv))
            pat = build(m_0, cfg_w, cfg_b, L, opt)
            for ivname, iv in iv_variants(pat):
                for pname, perm in PERMS.items():
                    for iv_pos in (1, 0):
                        for b_at_commit in (True, False):
                            got = replay_to_blinding_ood(t, iv, perm, m_0, b_at_commit, iv_pos)
                            if got == want:
                                hit = dict(labels=L, options={k: (v.decode() if isinstance(v, bytes) else v) for k, v in opt.items()}, iv=ivname,
                                           permutation=pname, iv_pos=iv_pos, batching_squeeze_at_commit=b_at_commit, pattern_hex=pat.hex())
                                print("HIT", hit)
            tried.append(dict(sha256=hashlib.sha256(pat).hexdigest()[:16], len=len(pat),
                              labels={k: v for k, v in L.items() if len(label_opts[k]) > 1},
                              options={k: (v.decode() if isinstance(v, bytes) else v) for k, v in opt.items()}))
            if hit:
                break
        if hit:
            break
    out = dict(fixture="tooling/provekit-bench/benches/poseidon-1000.np", target="blinding.ood_point (tests/golden/fixture_whir.json)",
               patterns_tried=len(tried), iv_derivations=[n for n, _ in iv_variants(b"")], permutations=list(PERMS), iv_positions=[1, 0],
               seconds=round(time.time() - t0, 1), hit=hit, example_pattern=build(m_0, cfg_w, cfg_b, {k: v[0] for k, v in label_opts.items()},
                                                                            {k: v[0] for k, v in struct_opts.items()}).decode("utf-8", "replace"),
               tried=tried)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r04_iopattern_search.json"), "w"), indent=1, ensure_ascii=False)
    print(f"{len(tried)} patterns x {len(list(iv_variants(b'')))} IVs x {len(PERMS)} permutations x 2 x 2: hit = {hit is not None} ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
