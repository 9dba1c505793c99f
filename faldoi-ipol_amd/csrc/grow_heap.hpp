// grow_heap.hpp -- the candidate queue of the local growing, usable from host and device code.
//
// The reference keeps its candidates in std::priority_queue<SparseOF, std::vector<SparseOF>,
// CompareSparseOF> (src/energy_structures.h:13-30; comparator: a.sim_node > b.sim_node, i.e. the
// smallest energy is on top).  Which of several equal-energy candidates comes out first is a
// property of the binary-heap algorithm behind that container, and the growing order (hence the
// flow) depends on it, so the queue here performs the same element moves as the standard
// push_heap / pop_heap the reference resolves to (GNU libstdc++):
//   push: append, then sift the hole up while parent.sim > value.sim;
//   pop : take the last element out as `value`, move the hole from the root to a leaf always
//         following the child that wins the comparison (the left one on ties), then sift `value`
//         up from that leaf.
// tests/ replays random push / pop sequences with many ties against the real std::priority_queue.
#pragma once

namespace faldoi {

struct alignas(16) Candidate {      // SparseOF without the unused `occluded`, 16 bytes
    float sim;                      // sim_node
    float u, v;
    unsigned ij;                    // column | row << 16
};

__host__ __device__ inline Candidate make_candidate(int i, int j, float u, float v, float sim)
{
    Candidate c;
    c.sim = sim; c.u = u; c.v = v; c.ij = (unsigned)i | ((unsigned)j << 16);
    return c;
}

__host__ __device__ inline void heap_sift_up(Candidate *h, int hole, const Candidate &value)
{
    while (hole > 0) {
        const int parent = (hole - 1) / 2;
        const Candidate pc = h[parent];
        if (!(pc.sim > value.sim)) break;
        h[hole] = pc;
        hole = parent;
    }
    h[hole] = value;
}

__host__ __device__ inline void heap_push(Candidate *h, int &size, const Candidate &value)
{
    heap_sift_up(h, size, value);
    ++size;
}

// precondition: size > 0
__host__ __device__ inline Candidate heap_pop(Candidate *h, int &size)
{
    const Candidate top = h[0];
    const int len = --size;                     // length of the heap that remains
    if (len == 0) return top;
    const Candidate value = h[len];
    int hole = 0, child = 0;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        const Candidate r = h[child], l = h[child - 1];
        if (r.sim > l.sim) { --child; h[hole] = l; } else { h[hole] = r; }
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h[hole] = h[child - 1];
        hole = child - 1;
    }
    heap_sift_up(h, hole, value);
    return top;
}

}  // namespace faldoi
