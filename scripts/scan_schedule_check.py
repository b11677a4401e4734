"""The level schedule of run_scan (lqg_amd/csrc/lqg_scan_inst.hip) for windows of 25 .. 64, restated on symbolic elements: an element is
the RANGE of steps it stands for, a combine must join adjacent ranges, a level reads only what earlier levels (or the input) wrote into
the buffer it reads, and at the end element k of buffer `a` must be the prefix [0, k] — for two side-by-side sequences of any lengths,
any number of systems, in the default order (Brent-Kung around a ping-pong scan of the block totals) and plain Brent-Kung.  Prints the
number of levels and of rounds (256 concurrent combines).   python scripts/scan_schedule_check.py"""


def run(len0, len1, n_sys, order=0, conc=256):
    lens = [len0, len1]
    longest = max(lens)
    a = [[(i, i) for i in range(n)] for n in lens]
    b = [[None] * n for n in lens]
    rounds = levels = 0

    def comb(x, y):
        assert x is not None and y is not None, "a level read an element nobody wrote"
        assert x[1] + 1 == y[0], (x, y)
        return (x[0], y[1])

    def level(frm, to, d, k0, ks, cnt):
        nonlocal rounds, levels
        if sum(cnt) == 0:
            return
        levels += 1
        rounds += -(-sum(cnt) * n_sys // conc)
        for s in range(2):
            new = {}
            for i in range(cnt[s]):
                k = k0 + i * ks
                assert k < lens[s]
                if k < d:
                    if to[s] is not frm[s]:
                        new[k] = frm[s][k]
                else:
                    new[k] = comb(frm[s][k - d], frm[s][k])
            for k, v in new.items():
                to[s][k] = v

    B = 2
    if order == 2:
        B = 2 * longest
    else:
        while B < longest and n_sys * (lens[0] // B + lens[1] // B) > conc:
            B *= 2
    nb_max = max(lens[0] // B, lens[1] // B)
    mid, dj = 0, 1
    while dj < nb_max:
        mid, dj = mid + 1, dj * 2
    top = 1
    while 2 * top < B and 4 * top <= longest:
        top *= 2
    d = 1
    while d <= top:
        flip = 2 * d == B and mid & 1
        level(a, b if flip else a, d, 2 * d - 1, 2 * d, [lens[0] // (2 * d), lens[1] // (2 * d)])
        d *= 2
    if mid:
        cur, oth = (b, a) if mid & 1 else (a, b)
        dj = 1
        while dj < nb_max:
            level(cur, oth, dj * B, B - 1, B, [lens[0] // B, lens[1] // B])
            cur, oth = oth, cur
            dj *= 2
        assert cur is a
    d = top
    while d >= 1:
        level(a, a, d, 3 * d - 1, 2 * d, [(n - d) // (2 * d) if n > d else 0 for n in lens])
        d //= 2
    for s in range(2):
        assert all(a[s][k] == (0, k) for k in range(lens[s]))
    return B, levels, rounds


if __name__ == "__main__":
    for T in list(range(2, 140)) + [255, 256, 257, 500, 511, 512, 513, 1000, 1067]:
        for n_sys in (1, 2, 3, 13, 64, 400):
            for order in (0, 2):
                run(T + 1, T, n_sys, order)
                run(T, 0, n_sys, order)
    for args in ((501, 500, 1), (500, 0, 1), (501, 500, 13), (500, 0, 13)):
        print(args, "block, levels, rounds: default", run(*args), "plain Brent-Kung", run(*args, order=2))
    print("OK")
