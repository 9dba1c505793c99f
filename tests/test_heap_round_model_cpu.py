"""The device queue's pop, as arithmetic (csrc/grow_kernels.hpp: heap_descend_round, heap_remove_top_wave), against the textbook form.

The kernel decides all 63 nodes of a round's subtree with one vector comparison, marks the path with bit masks (a node is TAKEN if its
parent has two children and goes its way; ON THE PATH if it and every node above it are taken) and places the queue's last element from
the path in registers.  This file restates that arithmetic in numpy integers -- lane for lane, mask for mask -- and checks that a pop
leaves the array exactly as std::pop_heap's moves do (__adjust_heap to the leaf along `comp(right, left)`, then __push_heap), on random
queues with heavy ties, every size from 1 to a few thousand, and every split between "LDS levels" and the levels below.  (The device
code itself is tested on the GPU against the shared host queue: tests/test_gpu_round4.py, tests/test_gpu_parity.py.)"""
import numpy as np
import pytest

ROUND_LEVELS = 6


def pop_textbook(key):
    """std::pop_heap's element moves with comp(a, b) = a.sim > b.sim (smallest on top); returns the array without its old top."""
    key = list(key)
    length = len(key) - 1
    if length == 0:
        return []
    value = key[length]
    hole, child = 0, 0
    while child < (length - 1) // 2:                       # __adjust_heap
        child = 2 * (child + 1)
        if key[child][0] > key[child - 1][0]:              # comp(first[secondChild], first[secondChild - 1])
            child -= 1
        key[hole] = key[child]
        hole = child
    if length % 2 == 0 and child == (length - 2) // 2:
        child = 2 * (child + 1)
        key[hole] = key[child - 1]
        hole = child - 1
    parent = (hole - 1) // 2                               # __push_heap
    while hole > 0 and key[parent][0] > value[0]:
        key[hole] = key[parent]
        hole = parent
        parent = (hole - 1) // 2
    key[hole] = value
    return key[:length]


def lane_constants():
    lanes = np.arange(64)
    depth = np.floor(np.log2(lanes + 1)).astype(int)
    parent = np.where(lanes > 0, (lanes - 1) >> 1, 0)
    is_left = (lanes & 1) != 0
    line = np.zeros(64, dtype=object)
    for lane in range(64):
        if 1 <= lane <= 62:
            m, x = 0, lane
            while x >= 1:
                m |= 1 << x
                x = (x - 1) >> 1
            line[lane] = m
        else:
            line[lane] = (1 << 64) - 1                     # never matched
    return depth, parent, is_left, line


DEPTH, PARENT, IS_LEFT, LINE = lane_constants()


def descend_round(key, hole, half, last, nl, value, single):
    """heap_descend_round: returns (new hole, placed).  key: list of (sim, id) tuples, modified in place."""
    two = goes_right = 0
    r0 = [None] * 64
    for lane in range(63):
        d = DEPTH[lane]
        idx0 = (hole << d) + lane
        if lane >= 1 and d <= nl and idx0 <= last:
            r0[lane] = key[idx0]
        if d < nl and idx0 < half:                          # `dec`: both children exist, the round may step below the node
            two |= 1 << lane
            kid = 2 * idx0 + 1
            if not (key[kid + 1][0] > key[kid][0]):         # comp(right, left) picks the left child
                goes_right |= 1 << lane
    six = nl >= 6 and hole + 1 <= ((last + 1) >> 6)
    r6 = [key[(hole << 6) + 63 + lane] if six and (hole << 6) + 63 + lane <= last else None for lane in range(64)]
    takes_left, takes_right = (two & ~goes_right) & 0xFFFFFFFF, (two & goes_right) & 0xFFFFFFFF
    taken_mask = 0
    for lane in range(64):
        src = takes_left if IS_LEFT[lane] else takes_right
        if (src >> PARENT[lane]) & 1:
            taken_mask |= 1 << lane
    pm = 0
    for lane in range(64):
        if (taken_mask & LINE[lane]) == LINE[lane]:
            pm |= 1 << lane
    assert pm != 0
    deep = pm.bit_length() - 1
    r = bin(pm).count("1")
    c6 = -1
    if r == 5 and (two >> deep) & 1:
        c6 = 2 * deep + 1 + ((goes_right >> deep) & 1) - 63
        r = 6
    cur = (hole << 6) + 63 + c6 if c6 >= 0 else (hole << r) + deep
    placed, move6, stop6, stop = False, c6 >= 0, False, 64
    if cur >= half and cur != single and value is not None:
        if c6 >= 0 and not (r6[c6][0] > value[0]):
            placed = stop6 = True
        else:
            greater = sum(1 << lane for lane in range(64) if r0[lane] is not None and r0[lane][0] > value[0])
            stay = pm & ~greater
            if stay:
                placed, stop, move6 = True, stay.bit_length() - 1, False
    stores = []
    for lane in range(64):
        if (pm >> lane) & 1 and lane <= stop:
            idx0 = (hole << DEPTH[lane]) + lane
            stores.append(((idx0 - 1) >> 1, r0[lane]))
            if lane == stop:
                stores.append((idx0, value))
    if move6:
        idx6 = (hole << 6) + 63 + c6
        stores.append(((idx6 - 1) >> 1, r6[c6]))
        if stop6:
            stores.append((idx6, value))
    for i, v in stores:                                     # (all loads of the round came before its stores)
        key[i] = v
    return cur, placed


def pop_model(key, top_levels):
    key = list(key)
    length = len(key) - 1
    if length == 0:
        return []
    value = key[length]
    half, last = (length - 1) // 2, length - 1
    single = (length - 2) // 2 if length % 2 == 0 else -1
    hole, placed = 0, False
    have_value = length < (1 << top_levels) - 1             # the last element lies in "LDS"; otherwise the first round below asks for it
    while hole < half and not placed:
        in_lds = top_levels - (hole + 1).bit_length()       # levels below the hole that lie in "LDS"
        nl = min(in_lds, ROUND_LEVELS) if in_lds >= 1 else ROUND_LEVELS
        have_value = have_value or in_lds < 1
        hole, placed = descend_round(key, hole, half, last, nl, value if have_value else None, single)
    if not placed:
        if hole == single:
            child = 2 * (hole + 1)
            key[hole] = key[child - 1]
            hole = child - 1
        while hole > 0 and key[(hole - 1) // 2][0] > value[0]:      # heap_sift_up_wave
            key[hole] = key[(hole - 1) // 2]
            hole = (hole - 1) // 2
        key[hole] = value
    return key[:length]


def random_heap(rng, n, levels):
    sims = rng.integers(0, levels, n).astype(np.float32) * 0.25
    heap = []
    for k, s in enumerate(sims):                            # std::push_heap
        heap.append((float(s), k))
        hole = len(heap) - 1
        v = heap[hole]
        while hole > 0 and heap[(hole - 1) // 2][0] > v[0]:
            heap[hole] = heap[(hole - 1) // 2]
            hole = (hole - 1) // 2
        heap[hole] = v
    return heap


@pytest.mark.parametrize("top_levels", [0, 1, 3, 6, 12, 13])
def test_mask_arithmetic_of_a_pop_makes_the_moves_of_pop_heap(top_levels):
    rng = np.random.default_rng(70 + top_levels)
    for n, levels in [(1, 2), (2, 2), (3, 2), (4, 3), (7, 2), (64, 4), (65, 1000), (127, 3), (128, 2), (129, 50), (1000, 5), (4097, 100000),
                      (5000, 3)]:
        heap = random_heap(rng, n, levels)
        for _ in range(min(n, 300)):                        # drain from the top: sizes n, n-1, ...
            want = pop_textbook(heap)
            got = pop_model(heap, top_levels)
            assert got == want, (top_levels, n, len(heap))
            heap = want
