"""Study for a later round (CPU only, numpy): can the senone kernel's log-add chain run without
the step offsets in its second operand?

Today (ssw_k1b_senone.inc, quad_chains): x' <- min(x', y') + (T - tab[|x' - y'|]) on two 16-bit
halves per register, where x' = x + T k and y' = y + T (k - 1) carry the same offset so that
nothing is ever subtracted.  y' = spread(weight bytes) + (relative score + T (k - 1)): the add
happens AFTER the bytes are spread to 16-bit lanes, two packed adds per dword of four weights,
because weight + relative score + offset can pass 255 (158 + 96 + 14 on en-us / fr-fr).

Alternative: leave y without an offset -- y = weight + relative score <= 254 fits a byte, ONE
32-bit add per dword before the spread -- and fold minimum and table into one signed 16-bit table
    G[d] = T - tab[|d|] - max(d, 0),        x'_(k+1) = x'_k + G[x'_k - y_k - T k]   (mod 2^16)
indexed by the true difference d = x - y (the look-up's immediate offset takes the - T k), added
with a packed add that wraps per half.  This script checks on random and extreme inputs that
the chain so written returns exactly fast_logmath_add's values plus T k, and that the halves
never interfere.  Per pair and step it costs what the present form costs (two differences, two
look-ups, two packed adds instead of packed min + add3); what it saves is 9 of the 18 packed adds
of a quad and frame (120 vector instructions today).  Round 5, on the device side: NOT
worth building.  A 2-byte table is read at byte address 2 d; the SDWA subtracts deliver d, and
every way of doubling it (a shift per look-up, a packed subtract + two v_mad_u32_u16, the whole
chain in doubled units -- which needs the spread weights doubled, the packed instruction this
was to remove) costs per look-up what the byte-domain add saves per pair (DESIGN.md, round 5,
item 9).

    python tools/studies/senone_g_table.py
"""
import numpy as np

TAB = np.zeros(1024, np.int64)
TAB[:29] = [7, 6, 6, 5, 5, 5, 4, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1][:29]
T = int(TAB.max())


def ref_chain(w, rel):
    """fast_logmath_add chain of the reference (tied_mgau_common.h:100-117) over 4 codewords"""
    x = w[:, 0] + rel[:, 0]
    for k in range(1, 4):
        y = w[:, k] + rel[:, k]
        d = np.abs(x - y)
        x = np.minimum(x, y) - TAB[np.minimum(d, 1023)]
    return x


def g_chain(w, rel):
    """the same with the signed table, 16-bit wrap-around arithmetic, y without offsets"""
    dd = np.arange(-1024, 1024)
    G = (T - TAB[np.minimum(np.abs(dd), 1023)] - np.maximum(dd, 0)) & 0xffff      # as uint16
    x = (w[:, 0] + rel[:, 0]) & 0xffff                                            # x'_1, offset 0
    for k in range(1, 4):
        yb = w[:, k] + rel[:, k]
        assert yb.max() <= 255, "byte-domain add would wrap"
        d = ((x - yb - T * (k - 1)) & 0xffff).astype(np.int64)                    # x' - y - T (k-1)
        d = np.where(d >= 0x8000, d - 0x10000, d)                                 # signed view
        x = (x + G[d + 1024]) & 0xffff
    x = np.where(x >= 0x8000, x - 0x10000, x)
    return x - 3 * T                                                              # offsets cancel in score - best


def main():
    rng = np.random.default_rng(1)
    n = 4_000_000
    for wmax, name in ((158, "weights <= 158 (en-us, fr-fr)"), (159, "weights <= 159")):
        w = rng.integers(0, wmax + 1, size=(n, 4))
        rel = np.sort(rng.integers(0, 97, size=(n, 4)), axis=1)
        rel[:, 0] = 0
        # extremes: ties, maxima, tiny values that dip below zero
        ext = np.array([[0, 0, 0, 0], [wmax] * 4, [0, wmax, 0, wmax], [1, 0, 0, 0], [3, 3, 3, 3]])
        w[:len(ext)] = ext
        rel[:len(ext)] = 0
        a, b = ref_chain(w, rel), g_chain(w, rel)
        print(name, "chains:", n, "differences:", int((a != b).sum()), "min value", int(a.min()))
        assert (a == b).all()


if __name__ == "__main__":
    main()
