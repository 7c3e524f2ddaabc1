"""Generates the constants of bevy_gaussian_splatting_amd/csrc/exact_log.h (mpmath, 200 bits).

For j = the top five mantissa bits of x's significand m2 in [1, 2): j >= 14 means the significand is halved
(m in [0.71875, 1), exponent + 1), so m lies in [0.71875, 1.4375) and x = 2^e * m. c_j is 1 / (interval centre)
rounded to 10 significant bits (m * c_j is exact in binary64), except that the two intervals touching 1 take
c = 1 (so that ln of an x next to 1 keeps its RELATIVE accuracy). The table holds -ln(c_j) as a double-double.
ln 2 is split into a 44-bit head (e * head is exact for |e| < 256) and a tail.
Run: python scripts/exact_log/make_table.py   (prints the C initialisers)"""
import mpmath as mp

mp.mp.prec = 200


def to_double(x):
    return float(x)  # mpmath rounds to nearest


def head_bits(x, bits):
    m, e = mp.frexp(x)
    return mp.ldexp(mp.nint(mp.ldexp(m, bits)), int(e) - bits)


rows = []
for j in range(32):
    lo, hi = 1 + mp.mpf(j) / 32, 1 + mp.mpf(j + 1) / 32
    if j >= 14:
        lo, hi = lo / 2, hi / 2
    if j in (0, 31):
        c = mp.mpf(1)
    else:
        c = head_bits(1 / ((lo + hi) / 2), 10)
    worst = max(abs(lo * c - 1), abs(hi * c - 1))
    assert worst <= mp.mpf(2) ** -5, (j, worst)
    nl = -mp.log(c)
    nl_hi = mp.mpf(to_double(nl))
    nl_lo = mp.mpf(to_double(nl - nl_hi))
    rows.append((float(c), float(nl_hi), float(nl_lo)))

ln2 = mp.log(2)
ln2_hi = head_bits(ln2, 44)
ln2_lo = mp.mpf(to_double(ln2 - ln2_hi))
print("LN2_HI", float(ln2_hi).hex(), "LN2_LO", float(ln2_lo).hex())
for c, h, l in rows:
    print("    {%s, %s, %s}," % (c.hex(), h.hex(), l.hex()))
print("coefficients 2/(2k+1):", [float(mp.mpf(2) / (2 * k + 1)).hex() for k in range(1, 8)])
