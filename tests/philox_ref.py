"""Plain-Python Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) for the
device generator's known-answer test.  Checked below against the paper's published test vectors."""
M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter, key):
    c0, c1, c2, c3 = counter
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xffffffff, p1 & 0xffffffff, ((p0 >> 32) ^ c3 ^ k1) & 0xffffffff, p0 & 0xffffffff
        k0, k1 = (k0 + W0) & 0xffffffff, (k1 + W1) & 0xffffffff
    return [c0, c1, c2, c3]


# Random123 kat_vectors, philox4x32 10 rounds
assert philox4x32_10((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
assert philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
assert philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
    [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
