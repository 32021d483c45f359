"""Round-3 experiment (time-boxed): is the reference's origin pointer on an exactly periodic block
T = u^k a function of the rotation order of u?  Needs the compiled reference (oracle/_ref, this container only).

Result (seed 5, 54 184 random primitive u over 2..4 letters, |u| 2..9, k 2..6): the pointer's offset j inside the k
equal rows is deterministic in (u, k) but NOT a function of (k, order of u's rotations): 226 of 16 164 such classes
have two different offsets.  j is 1 or 2 in 93 % of the cases with k >= 3 and is never decided by k alone.  The k
equal rotations inherit their order from the k equal type-B* suffixes that induce them (divbwt.c:1634-1699 keeps the
relative order of equal keys); those are ordered by sssort's multikey introsort, whose three-way partition
(divbwt.c:412-545) moves equal elements around depending on what else shares their two-character bucket -- i.e. on the
whole string.  No rule short of running that sort; the documented divergence stays."""
import collections
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as L  # noqa: E402


def main(cases=60000, seed=5):
    R = L.ref()

    def ref_idx(T):
        out = (C.c_uint8 * (len(T) + 8))()
        return R.ref_bwt(bytes(T), len(T), out)

    rng = random.Random(seed)
    by_rot, by_u, dist = collections.defaultdict(set), collections.defaultdict(set), collections.Counter()
    n_cases = 0
    for _ in range(cases):
        p, a = rng.randint(2, 9), rng.randint(2, 4)
        u = bytes(rng.randrange(a) + 97 for _ in range(p))
        if any(u == u[d:] + u[:d] for d in range(1, p)):
            continue
        k = rng.randint(2, 6)
        T = u * k
        rots = sorted(range(len(T)), key=lambda i: (T[i:] + T[:i], i))
        r0 = min(r for r, i in enumerate(rots) if i % p == 0)
        j = ref_idx(T) - r0
        assert 0 <= j < k
        by_rot[(k, tuple(sorted(range(p), key=lambda i: u[i:] + u[:i])))].add(j)
        by_u[(k, u)].add(j)
        dist[(k, j)] += 1
        n_cases += 1
    print("cases", n_cases)
    print("(k, rotation order of u) classes:", len(by_rot), "with more than one offset:", sum(len(v) > 1 for v in by_rot.values()))
    print("(k, u) classes with more than one offset (non-determinism):", sum(len(v) > 1 for v in by_u.values()))
    print("offset histogram (k, j):", sorted(dist.items()))


if __name__ == "__main__":
    main()
