"""SURVEY 8f-3 (opt-in, UNPINNED): the reference's own seeded stream for the draws that precede sample_beta
(main.rs:289-319: StdRng::seed_from_u64, Uniform<f64>, statrs Exp), restated from the published algorithms of
rand 0.8.5 / rand_core 0.6 / rand_chacha 0.3 / statrs 0.16 -- crates that are neither vendored by the reference nor
available here.  What CAN be checked without them: the ChaCha block function against RFC 7539's vector, and the library
against an independent pure-Python restatement of the same published algorithms (host-only code: no GPU needed)."""
import math
import struct

import numpy as np

M32 = 0xFFFFFFFF


def _rotl(x, k):
    return ((x << k) | (x >> (32 - k))) & M32


def chacha_block(key, counter, stream, rounds):
    st = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(key) + [counter & M32, counter >> 32, stream & M32, stream >> 32]
    x = list(st)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 7)
    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M32 for a, b in zip(x, st)]


class StdRng:
    """rand 0.8.5 StdRng (ChaCha12) after SeedableRng::seed_from_u64"""

    def __init__(self, state):
        self.key = []
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & 0xFFFFFFFFFFFFFFFF
            xs = (((state >> 18) ^ state) >> 27) & M32
            rot = state >> 59
            self.key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & M32)
        self.counter, self.buf, self.index = 0, [0] * 64, 64

    def _refill(self, index_after):
        self.buf = sum((chacha_block(self.key, self.counter + b, 0, 12) for b in range(4)), [])
        self.counter += 4
        self.index = index_after

    def next_u64(self):
        if self.index < 63:
            v = (self.buf[self.index + 1] << 32) | self.buf[self.index]
            self.index += 2
            return v
        if self.index >= 64:
            self._refill(2)
            return (self.buf[1] << 32) | self.buf[0]
        lo = self.buf[63]
        self._refill(1)
        return (self.buf[0] << 32) | lo

    def gen_f64(self):
        return (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)

    def uniform01(self):
        bits = (self.next_u64() >> 12) | 0x3FF0000000000000
        return (struct.unpack("<d", struct.pack("<Q", bits))[0] - 1.0) * 1.0 + 0.0


def zig_tables():
    R, V = 7.69711747013104972, 0.0039496598225815571993
    x = [0.0] * 257
    x[0], x[1] = V / math.exp(-R), R
    for i in range(2, 256):
        x[i] = -math.log(V / x[i - 1] + math.exp(-x[i - 1]))
    return x, [math.exp(-v) for v in x]


def exp1(rng, xt, ft):
    while True:
        bits = rng.next_u64()
        i = bits & 0xff
        u = (bits >> 11) / 9007199254740992.0
        x = u * xt[i]
        if x < xt[i + 1]:
            return x
        if i == 0:
            return 7.69711747013104972 - math.log(rng.gen_f64())
        if ft[i + 1] + (ft[i] - ft[i + 1]) * rng.gen_f64() < math.exp(-x):
            return x


def selection(seed, G, prop_positive, pos_lambda, neg_lambda):
    out = [0.0] * G
    if not prop_positive >= 0.0:
        return out
    rng = StdRng(seed)
    xt, ft = zig_tables()
    for g in range(G):
        if rng.uniform01() <= prop_positive:
            s = exp1(rng, xt, ft) / pos_lambda
        else:
            s = exp1(rng, xt, ft) / neg_lambda
            while s > 1.0:
                s = exp1(rng, xt, ft) / neg_lambda
            s = -1.0 * s
        out[g] = s
    return out


def test_chacha_block_rfc7539_vector(pa):
    # RFC 7539 section 2.3.2: key 00..1f, block counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00, 20 rounds
    lib = pa.load()
    key = np.array([0x03020100, 0x07060504, 0x0b0a0908, 0x0f0e0d0c, 0x13121110, 0x17161514, 0x1b1a1918, 0x1f1e1d1c], np.uint32)
    out = np.zeros(16, np.uint32)
    lib.ps_chacha_block(key, (0x09000000 << 32) | 1, 0x4a000000, 20, out)
    want = [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
            0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    assert out.tolist() == want
    assert chacha_block(key.tolist(), (0x09000000 << 32) | 1, 0x4a000000, 20) == want
    # 12 rounds (StdRng): library = the independent restatement
    lib.ps_chacha_block(key, 5, 0, 12, out)
    assert out.tolist() == chacha_block(key.tolist(), 5, 0, 12)


def test_chacha12_and_chacha8_published_zero_key_vectors(pa):
    # draft-strombergson-chacha-test-vectors-01, TC1 (256-bit key of zeros, IV of zeros, block 0): the published keystreams of
    # the 12-round core -- the one behind rand 0.8's StdRng -- and of the 8- and 20-round ones.  An EXTERNAL vector for the
    # round count the library actually uses (RFC 7539 only pins 20 rounds); it pins the block function,
    # not rand's buffering / seed expansion, which stay unpinned.
    import struct
    lib = pa.load()
    key = np.zeros(8, np.uint32)
    out = np.zeros(16, np.uint32)
    want = {8: "3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e",
            12: "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f",
            20: "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"}
    for rounds, hexstr in want.items():
        lib.ps_chacha_block(key, 0, 0, rounds, out)
        assert struct.pack("<16I", *out.tolist()).hex()[:64] == hexstr
        assert struct.pack("<16I", *chacha_block([0] * 8, 0, 0, rounds)).hex()[:64] == hexstr


def test_reference_selection_stream_matches_the_python_restatement(pa):
    lib = pa.load()
    for seed, G, pp, pl, nl in [(0, 4000, 0.3, 10.0, 10.0), (12345678901234567, 700, 0.0, 2.0, 0.7), (7, 300, 1.0, 50.0, 10.0),
                                (3, 10, -0.1, 10.0, 10.0)]:
        out = np.zeros(G)
        assert lib.ps_reference_selection_coefficients(seed, G, pp, pl, nl, out) == 0
        assert out.tolist() == selection(seed, G, pp, pl, nl)
        if pp >= 0.0:
            assert (out <= 0).mean() > 0 or pp == 1.0
            assert (out >= -1.0).all()              # negatives are redrawn while > 1 (main.rs:309-311)
    # distribution: Exp(lambda) has mean 1 / lambda
    out = np.zeros(200000)
    lib.ps_reference_selection_coefficients(1, out.size, 1.0, 4.0, 10.0, out)
    assert abs(out.mean() - 0.25) < 0.004 and abs(out.var() - 1 / 16) < 0.004


def test_reference_stream_is_opt_in(pa):
    # the default stays the build's Philox stream; the flag only moves the selection coefficients
    a = pa.selection_coefficients(5, 100, 0.4, 10.0, 10.0)
    out = np.zeros(100)
    pa.load().ps_reference_selection_coefficients(5, 100, 0.4, 10.0, 10.0, out)
    assert not np.array_equal(a, out)
    assert pa.make_params().reference_seed_stream == 0
