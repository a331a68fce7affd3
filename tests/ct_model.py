"""Integer model of the secret-scalar kernels' recoding and addition schedule (no field arithmetic).

The kernels of kernels_ct.hpp (fixed base) and kernels_coz.hpp with CT = true (variable base) use
incomplete mixed additions and resolve accumulator == +-entry by selects -- but only in the windows
where the magnitudes of the partial sums allow such a collision at all.  This module restates the
schedules on plain integers so that tests can (a) check the window bounds by exhaustive enumeration on
small parameters and (b) find, for the real curves, the scalars that DO hit the collision selects, so
that the GPU tests exercise them.
"""


def booth_digits(k: int, sb: int, w: int):
    """Signed w-bit window digits d_i in [-2^(w-1), 2^(w-1)] of the sb-byte scalar k, sum d_i 2^(w i) = k
    (booth_digit<W, SB> in kernels_unsat.hpp: window i covers bits w*i - 1 .. w*i + w - 1)."""
    assert 0 <= k < 1 << (8 * sb)
    nwin = (8 * sb + 1 + w - 1) // w
    out = []
    for i in range(nwin):
        wv = ((k << 1) >> (w * i)) & ((1 << (w + 1)) - 1)
        neg = wv >> w
        m = ((1 << (w + 1)) - wv - 1) if neg else wv
        d = (m >> 1) + (m & 1)
        out.append(-d if neg else d)
    assert sum(d << (w * i) for i, d in enumerate(out)) == k
    return out


def fixed_base_unsafe_windows(sb: int, w: int, nbits: int) -> int:
    """ct_unsafe_windows() of kernels_ct.hpp: windows i with w (i + 1) >= nbits, counted from the top."""
    nwin = (8 * sb + 1 + w - 1) // w
    return nwin - ((nbits + w - 1) // w) + 1


def fixed_base_events(k: int, sb: int, w: int, n: int):
    """Windows are added LOW to HIGH.  Returns [(window, 'twice' | 'cancel')] for every window whose entry
    e G meets an accumulator s G with s = +-e (mod n), s != 0 (mod n)."""
    ev = []
    acc = 0
    for i, d in enumerate(booth_digits(k, sb, w)):
        e = d << (w * i)
        if d != 0 and acc % n != 0:
            if (acc - e) % n == 0:
                ev.append((i, "twice"))
            elif (acc + e) % n == 0:
                ev.append((i, "cancel"))
        acc += e
    return ev


def var_base_full_windows(sb: int, wb: int, nbits: int) -> int:
    """coz_full_windows() of kernels_coz.hpp for a curve of prime order: the bottom windows that resolve
    accumulator == +-entry."""
    return (8 * sb - nbits + 1) // wb + 1


def var_base_events(k: int, sb: int, wb: int, n: int):
    """The ladder: accumulator = top digit; per window (HIGH to LOW) wb doublings, then + d P.  Returns the
    windows in which 2^wb S = +-d (mod n) with S != 0 (mod n)."""
    ev = []
    ds = booth_digits(k, sb, wb)
    s = ds[-1]
    for i in range(len(ds) - 2, -1, -1):
        s <<= wb
        d = ds[i]
        if d != 0 and s % n != 0:
            if (s - d) % n == 0:
                ev.append((i, "twice"))
            elif (s + d) % n == 0:
                ev.append((i, "cancel"))
        s += d
    assert s == k
    return ev


def collision_scalars_fixed_base(sb: int, w: int, n: int, limit: int = 64):
    """Scalars below 2^(8 sb) whose fixed-base schedule hits a collision select, by construction:
    cancel needs k = 0 (mod n) with a non-zero lower part, twice needs k = 2 e (mod n).  Up to `limit` of each."""
    nwin = (8 * sb + 1 + w - 1) // w
    top = 1 << (8 * sb)
    found = {}
    count = {"twice": 0, "cancel": 0}

    def consider(k, want):
        if 0 < k < top and k not in found and count[want] < limit:
            ev = fixed_base_events(k, sb, w, n)
            if ev:
                found[k] = ev
                for _, kind in ev:
                    count[kind] += 1

    for j in range(1, 2 + top // n):
        consider(j * n, "cancel")
    for i in range(nwin - 1, max(nwin - 4, -1), -1):
        for d in range(1, (1 << (w - 1)) + 1):
            e = d << (w * i)
            j0 = -((2 * e) // n) - 1
            for j in range(j0, j0 + 4 + top // n):
                consider(2 * e + j * n, "twice")
    return found


def collision_scalars_var_base(sb: int, wb: int, n: int, limit: int = 64):
    """Scalars whose ladder hits a collision select: k = n + 2 d, k = j n (cancel at the last window), ..."""
    found = {}
    top = 1 << (8 * sb)
    for j in range(1, 3 + top // n):
        for d in range(-(1 << (wb - 1)) * 40, (1 << (wb - 1)) * 40 + 1):
            k = j * n + d
            if 0 < k < top and k not in found:
                ev = var_base_events(k, sb, wb, n)
                if ev:
                    found[k] = ev
                    if len(found) >= limit:
                        return found
    return found


# ---- the endomorphism ladder with secret scalars (kernels_coz.hpp, GLV = true, CT = true; bls12_381_g1) -------------
def glv_digits(v: int, nwin: int, wb: int):
    """Booth digits of one half of the split scalar over nwin windows (the booth lambda of the kernel)."""
    out = []
    for i in range(nwin):
        wv = ((v << 1) >> (wb * i)) & ((1 << (wb + 1)) - 1)
        neg = wv >> wb
        m = ((1 << (wb + 1)) - wv - 1) if neg else wv
        d = (m >> 1) + (m & 1)
        out.append(-d if neg else d)
    assert sum(d << (wb * i) for i, d in enumerate(out)) == v, "the half must fit nwin windows"
    return out


def glv_events(k: int, x2: int, r: int, kbits: int, wb: int):
    """k = k1 + k2 x^2 (k2 = floor(k / x^2)); [x^2]P = -sigma(P) is one table lookup away.  The ladder: accumulator =
    top digit of k1, + top digit of k2 times x^2; per window (HIGH to LOW) wb doublings, + d1, + d2 x^2.  Returns
    [(window, half, 'twice' | 'cancel')] for every addition whose entry meets an accumulator equal to +- it (mod r)
    while neither is the neutral element."""
    k2, k1 = divmod(k, x2)
    nwin = (kbits + 1 + wb - 1) // wb
    d1s, d2s = glv_digits(k1, nwin, wb), glv_digits(k2, nwin, wb)
    ev = []
    s = 0

    def add(win, half, e):
        nonlocal s
        if e % r != 0 and s % r != 0:
            if (s - e) % r == 0:
                ev.append((win, half, "twice"))
            elif (s + e) % r == 0:
                ev.append((win, half, "cancel"))
        s += e

    for win in range(nwin - 1, -1, -1):
        if win != nwin - 1:
            s <<= wb
        add(win, 0, d1s[win])
        add(win, 1, d2s[win] * x2)
    assert s == k
    return ev


BLS_X = 0xD201000000010000  # |x| of BLS12-381; r = x^4 - x^2 + 1
BLS_X2 = BLS_X * BLS_X
BLS_R = BLS_X2 * BLS_X2 - BLS_X2 + 1


def collision_scalars_glv(x2: int, r: int, sbits: int, kbits: int, wb: int, limit: int = 48):
    """Scalars below 2^sbits whose endomorphism ladder hits a collision select: k = j r + a + b x^2 for small a, b."""
    found = {}
    span = (1 << (wb - 1)) * 2 + 1
    for j in range(1, 1 + (1 << sbits) // r):
        for b in range(-span, span + 1):
            for a in range(-span, span + 1):
                k = j * r + a + b * x2
                if 0 < k < (1 << sbits) and k not in found:
                    ev = glv_events(k, x2, r, kbits, wb)
                    if ev:
                        found[k] = ev
                        if len(found) >= limit:
                            return found
    return found
