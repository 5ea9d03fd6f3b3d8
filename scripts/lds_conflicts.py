#!/usr/bin/env python
"""Bank-conflict calculator for the LDS images of csrc/netvlad.hip's fused kernels, by the rules of
MI355X_MICROARCH.md (LDS): 64 banks of 4 bytes; ds_read_b128 is served in four fixed 16-lane
groups, ds_read_b64 / ds_read_b64_tr_b16 / ds_write_b64 in two 32-lane halves (ds_write_b64: four
contiguous 16-lane groups); lanes of a group that touch the same bank at different addresses
serialise.  Prints the extra LDS cycles per instruction for every access pattern of the kernels
(0 everywhere = conflict-free).  Pure host arithmetic; run anywhere."""
import itertools

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
HALVES = [list(range(0, 32)), list(range(32, 64))]
QUARTERS = [list(range(16 * k, 16 * k + 16)) for k in range(4)]


def extra_cycles(addr_of_lane, nbytes, groups):
    """Sum over lane groups of (max distinct addresses on one bank - 1)."""
    extra = 0
    for grp in groups:
        per_bank = {}
        for l in grp:
            a = addr_of_lane(l)
            for w in range(nbytes // 4):
                per_bank.setdefault(((a // 4) + w) % 64, set()).add(a + 4 * w)
        extra += max(len(v) for v in per_bank.values()) - 1
    return extra


# ---- x tile of the fused kernels: [32 locations][512 channels] bf16, 16-byte unit
#      u(loc, c) = 64 loc + (c ^ (loc & 15)),  c = 16-byte chunk of the row (0..63)
def x_unit(loc, c):
    return 64 * loc + (c ^ (loc & 15))


def pi(k):
    """location (within the 32 of a step) of contraction index k = 8 g + e of the aggregation."""
    g, e = k >> 3, k & 7
    return 16 * (g >> 1) + 2 * (4 * (g & 1) + (e & 3)) + (e >> 2)


def check_x_tile():
    worst = 0
    # logits: B fragment of (subtile t, k-step s): lane (i, g) reads unit (16 t + i, 4 s + g)
    for t, s in itertools.product(range(2), range(16)):
        worst = max(worst, extra_cycles(lambda l: 16 * x_unit(16 * t + (l & 15), 4 * s + (l >> 4)), 16,
                                        B128_GROUPS))
    print('x tile, ds_read_b128 row fragments (logits):        worst extra cycles', worst)
    worst = 0
    # aggregation: A fragment (channel tile ct, half h = e >> 2): lane 4 q + p of 16-lane group g
    # reads 8 bytes at row pi(8 g + 4 h + q), channels 16 ct + 4 p .. + 3
    for ct, h in itertools.product(range(32), range(2)):
        def addr(l):
            g, q, p = l >> 4, (l >> 2) & 3, l & 3
            return 16 * x_unit(pi(8 * g + 4 * h + q), 2 * ct + (p >> 1)) + 8 * (p & 1)
        worst = max(worst, extra_cycles(addr, 8, HALVES))
    print('x tile, ds_read_b64_tr_b16 (aggregation A operand): worst extra cycles', worst)
    assert sorted(pi(k) for k in range(32)) == list(range(32))


# ---- per-wave coefficient image: [plane 3][32 locations][16 clusters] bf16, rows of 32 bytes;
#      written 8 bytes per lane (location 16 t + i, clusters 4 g .. + 3), read transposed
def cf_addr(plane, loc, cl, ld):
    return plane * 32 * ld + loc * ld + 2 * cl


def check_cf(ld):
    worst_w = worst_r = 0
    for t in range(2):
        worst_w = max(worst_w, extra_cycles(lambda l: cf_addr(0, 16 * t + (l & 15), 4 * (l >> 4), ld), 8,
                                            QUARTERS))
    for h in range(2):
        def addr(l):
            g, q, p = l >> 4, (l >> 2) & 3, l & 3
            return cf_addr(0, pi(8 * g + 4 * h + q), 4 * p, ld)
        worst_r = max(worst_r, extra_cycles(addr, 8, HALVES))
    print('cf image, row stride %3d B: ds_write_b64 extra %d, ds_read_b64_tr_b16 extra %d'
          % (ld, worst_w, worst_r))


if __name__ == '__main__':
    check_x_tile()
    for ld in (32, 40, 48, 64, 72):
        check_cf(ld)
