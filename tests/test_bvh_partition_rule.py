"""The closed form the GPU BVH builder uses for the reference's in-place partition (src/core/bvh.rs:212-232), restated
in Python and compared with the loop itself: every 0/1 array of up to 11 elements, and random ones up to 200."""
import random


def partition_loop(big):
    idx = list(range(len(big)))
    left, right = 0, len(big)
    while left < right:
        if not big[idx[left]]:
            left += 1
        else:
            right -= 1
            idx[left], idx[right] = idx[right], idx[left]
    return idx, left


def partition_rule(big):
    """Where every element ends up, each from prefix counts alone (kb_scatter / kb_move in csrc/rt_bvh.hip)."""
    n = len(big)
    pb = [0] * (n + 1)
    for k in range(n):
        pb[k + 1] = pb[k] + (1 if big[k] else 0)
    small_total = n - pb[n]
    n_front = small_total + (1 if small_total < n and big[small_total] else 0)
    front_big_pos, back_small_pos = {}, {}
    for k in range(n):
        if k < n_front:
            if big[k]:
                front_big_pos[pb[k]] = k
        elif not big[k]:
            back_small_pos[(n - 1 - k) - (pb[n] - pb[k + 1]) + 1] = k
    out = [None] * n
    for k in range(n):
        if k < n_front:
            dest = k if not big[k] else (n - 1 if pb[k] == 0 else back_small_pos[pb[k]] - 1)
        elif big[k]:
            dest = k - 1
        else:
            dest = front_big_pos[(n - 1 - k) - (pb[n] - pb[k + 1])]
        assert out[dest] is None
        out[dest] = k
    return out, small_total


def test_rule_equals_loop():
    for n in range(12):
        for bits in range(1 << n):
            big = [(bits >> k) & 1 for k in range(n)]
            assert partition_rule(big) == partition_loop(big), big
    rng = random.Random(1)
    for _ in range(3000):
        n, p = rng.randint(1, 200), rng.random()
        big = [1 if rng.random() < p else 0 for _ in range(n)]
        assert partition_rule(big) == partition_loop(big), big
