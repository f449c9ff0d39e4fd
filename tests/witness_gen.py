"""Random but valid WitnessBuilder lists for the X4 tests: every builder reads only witnesses solved earlier in the list, every
variant of the enum occurs, some witness indices are never written (the reference leaves them None)."""
import random

from provekit_amd.witness import WitnessBuilder as WB

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def random_program(seed, n_builders, chain=False, with_big=True):
    """-> (builders, acir values (ints), challenges (ints), num_witnesses).  chain=True makes every scalar builder depend on the
    previous one (a deep, narrow dependence graph); otherwise inputs are drawn from everything solved so far (wide levels)."""
    rnd = random.Random(seed)
    builders, acir, challenges = [], [], []
    nxt = [0]
    solved = []      # witness indices with arbitrary field values
    small = []       # witness indices known to hold values < 2^8 (ACIR bytes, digits)
    nonzero = []     # witness indices known to be non-zero

    def fresh(n=1):
        i = nxt[0]
        nxt[0] += n
        return i

    def fe():
        return rnd.randrange(P)

    def pick():
        return solved[-1] if chain and solved else rnd.choice(solved)

    def new_acir(value):
        i = fresh()
        builders.append(WB.Acir(i, len(acir)))
        acir.append(value)
        return i

    one = fresh()
    builders.append(WB.Constant(one, 1))
    solved.append(one)
    nonzero.append(one)
    for _ in range(8):
        solved.append(new_acir(fe()))
    for _ in range(24):
        small.append(new_acir(rnd.randrange(256)))
    for _ in range(3):
        i = fresh()
        builders.append(WB.Challenge(i))
        challenges.append(fe())
        solved.append(i)
    fresh(2)  # two indices nobody ever writes: None in the reference, is_set = 0 here
    kinds = ["sum", "product", "inverse", "idx_logup", "logup", "prod_linear", "spice_factor", "binop_denom", "const", "acir", "challenge"]
    if chain:  # only the variants whose first operand is the previous builder's output: depth ~ length
        kinds = ["sum", "product", "idx_logup", "logup", "prod_linear", "spice_factor", "binop_denom"]
    while len(builders) < n_builders:
        k = rnd.choice(kinds)
        i = fresh()
        if k == "sum":
            terms = [(None if rnd.random() < 0.3 else fe(), pick() if t == 0 else rnd.choice(solved)) for t in range(rnd.randrange(1, 6))]
            builders.append(WB.Sum(i, terms))
        elif k == "product":
            builders.append(WB.Product(i, pick(), rnd.choice(solved)))
        elif k == "inverse":
            src = rnd.choice(nonzero)
            builders.append(WB.Inverse(i, src))
            nonzero.append(i)
        elif k == "idx_logup":
            builders.append(WB.IndexedLogUpDenominator(i, rnd.choice(solved), fe(), pick(), rnd.choice(solved), rnd.choice(solved)))
        elif k == "logup":
            builders.append(WB.LogUpDenominator(i, pick(), fe(), rnd.choice(solved)))
        elif k == "prod_linear":
            builders.append(WB.ProductLinearOperation(i, pick(), fe(), fe(), rnd.choice(solved), fe(), fe()))
        elif k == "spice_factor":
            builders.append(WB.SpiceMultisetFactor(i, pick(), rnd.choice(solved), fe(), rnd.choice(solved), rnd.choice(solved), fe(), rnd.choice(solved)))
        elif k == "binop_denom":
            cw = lambda: ("c", fe()) if rnd.random() < 0.4 else ("w", rnd.choice(solved))
            builders.append(WB.BinOpLookupDenominator(i, pick(), rnd.choice(solved), rnd.choice(solved), cw(), cw(), cw()))
        elif k == "const":
            builders.append(WB.Constant(i, fe()))
        elif k == "acir":
            builders.append(WB.Acir(i, len(acir)))
            acir.append(fe())
        else:
            builders.append(WB.Challenge(i))
            challenges.append(fe())
        solved.append(i)
    if with_big:
        # digital decomposition of values < 2^20 into mixed bases [8, 8, 4]; the digits are small witnesses again
        vals = [new_acir(rnd.randrange(1 << 20)) for _ in range(37)]
        first = fresh(3 * len(vals))
        builders.append(WB.DigitalDecomposition([8, 8, 4], vals, first))
        small += list(range(first, first + 2 * len(vals)))
        solved += list(range(first, first + 3 * len(vals)))
        # a wide decomposition crossing word boundaries: 4 digits of 61 bits + the rest
        wide = [new_acir(fe()) for _ in range(5)]
        first = fresh(5 * len(wide))
        builders.append(WB.DigitalDecomposition([61, 61, 61, 61, 12], wide, first))
        # multiplicities of byte-sized values in a range of 256 and of pairs in the 2^16 bin-op table
        start = fresh(256)
        builders.append(WB.MultiplicitiesForRange(start, 256, [rnd.choice(small) for _ in range(300)]))
        solved += list(range(start, start + 256))
        start = fresh(65536)
        cw = lambda: ("c", rnd.randrange(256)) if rnd.random() < 0.2 else ("w", rnd.choice(small))
        builders.append(WB.MultiplicitiesForBinOp(start, [(cw(), cw()) for _ in range(500)]))
        solved += list(range(start, start + 64))
        # Spice: a memory block of 16 cells, 200 loads / stores at byte addresses mod 16
        M = 16
        addrs = [new_acir(rnd.randrange(M)) for _ in range(40)]
        init = fresh(M)
        for a in range(M):
            builders.append(WB.Acir(init + a, len(acir)))
            acir.append(fe())
        ops = []
        for _ in range(200):
            a = rnd.choice(addrs)
            if rnd.random() < 0.5:
                ops.append(("load", a, rnd.choice(solved), fresh()))
            else:
                old, ts = fresh(), fresh()
                ops.append(("store", a, old, rnd.choice(solved), ts))
        rv, rt = fresh(M), fresh(M)
        builders.append(WB.SpiceWitnesses(M, init, ops, rv, rt))
        for op in ops:  # the timestamps and old values feed later builders
            solved.append(op[-1])
        # something downstream of each big builder
        for _ in range(20):
            i = fresh()
            builders.append(WB.Product(i, rnd.choice(solved), rnd.choice(solved)))
            solved.append(i)
    return builders, acir, challenges, nxt[0] + 3
