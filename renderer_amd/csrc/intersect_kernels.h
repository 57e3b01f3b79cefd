// intersect_kernels.h — the closest-hit traversal kernel (gfx950, wave64).
//
// Replaces Shader/Intersection.frag:25-84 (GLSL stack traversal) and the CPU recursion
// KDTree.cpp:295-339. One lane = one ray. The reference visits BOTH children of every node whose
// box is hit and keeps the smallest t (ties: first triangle inside a leaf, KDTree.cpp:305; the
// RIGHT subtree between siblings, KDTree.cpp:319-320). The result is therefore
//     argmin t over { triangles whose every ancestor box passes the reference box test },
// with ties won by the LATER leaf in pre-order and, inside one leaf, by the EARLIER triangle.
// This kernel computes the same argmin in a different order: nearer child first, far child pushed
// on a per-lane stack in LDS together with its slab-entry distance, and subtrees whose entry
// distance lies beyond the closest accepted hit are skipped. The tie rule is applied explicitly
// through the slot index (slots are leaf-ordered), so the visit order does not change the answer.
//
// Execution shape (what the first profile asked for: 15.6 of 64 lanes were busy):
//   * persistent waves: the grid is sized to the chip, not to the batch; each wave draws rays from a
//     global queue in chunks of RAY_CHUNK (one atomic per chunk) and a lane whose ray is finished
//     takes the next ray instead of idling until the slowest ray of its wave is done;
//   * "while-while" traversal: all lanes descend inner nodes until each has reached a leaf (or run
//     out of nodes), then all lanes test their leaf's triangles together. Leaves are no longer
//     processed in the same iteration as other lanes' inner nodes, where they kept ~10 % of the lanes
//     busy for most of the time.
//   * node boxes are decided by the slab classifier of mirt_dev.h (no divisions; ~4x fewer
//     instructions than the reference's six-plane test) whenever that decision is provably the
//     reference's; a box it cannot prove is entered anyway and marks the path "ambiguous"; a
//     triangle accepted under such a path is first confirmed by the exact reference box test on every
//     ancestor. Rays outside the classifier's error model (zero / extreme components) use the exact
//     test throughout. The instrumented instantiation (COUNT) also runs the exact test beside every
//     proven decision and counts disagreements (must be 0; tests/test_gpu_intersect.py).
#pragma once
#include "mirt_dev.h"

#pragma clang fp contract(off)

namespace mirt {

constexpr uint32_t RAY_CHUNK = 256;  // rays a wave reserves per queue atomic

// MODE: MIRT_MODE_*   COUNT: accumulate box/triangle test counts   SLOT_OUT: write the slot index
// (internal path-tracing use) instead of the global triangle id.
// `queue` is one zero-initialised word per launch: the index of the next unassigned ray.
template <int MODE, bool COUNT, bool SLOT_OUT>
__global__ void __launch_bounds__(256)
k_intersect(DevScene sc, const float* __restrict__ O, const float* __restrict__ D,
            const int* __restrict__ aux, uint32_t n, uint32_t plane_stride, float* __restrict__ t_out,
            int* __restrict__ id_out, unsigned long long* __restrict__ counters,
            uint32_t* __restrict__ queue) {
    extern __shared__ uint32_t lds[];
    const uint32_t bs = blockDim.x;
    uint32_t* stk_node = lds + threadIdx.x;  // [level * bs]
    float* stk_lb = (float*)(lds + (sc.depth + 1) * bs) + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    unsigned long long c_box = 0, c_tri = 0;
    // wave-uniform ray pool [pool_next, pool_end)
    uint32_t pool_next = 0, pool_end = 0;
    bool exhausted = false;

    // per-lane traversal state
    bool active = false;
    uint32_t idx = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
    int a = 0, contact = -1, inside = -1;
    float best = -1.f;
    int best_slot = -1;
    int sp = 0;
    int cur = -1, c0 = 0, c1 = 0;  // current node and its child / leaf words; cur < 0: no node left
    bool amb = false;              // the path to `cur` holds a box the classifier could not prove
    RayPre pre;
    pre.inv = mk3(0, 0, 0);
    pre.a_ray = 0.f;
    pre.fast = false;
    unsigned long long c_amb = 0, c_bad = 0, c_walk = 0;

    // one node box: hit / entry distance / "unproven" flag
    auto test_box = [&](const NodeRec& r, bool& hit, float& lb, bool& unproven) {
        if (pre.fast) {
            const int code = slab_classify(r.lo, r.hi, o, pre, lb);
            hit = code != BOX_MISS;
            unproven = code == BOX_AMBIGUOUS;
            if (COUNT) {
                if (unproven) {
                    c_amb++;
                } else if (box_test(r.lo, r.hi, o, d).hit != hit) {
                    c_bad++;
                }
            }
        } else {
            const BoxHit b = box_test(r.lo, r.hi, o, d);
            hit = b.hit;
            lb = b.lb;
            unproven = false;
        }
    };

    // next node from the stack (entries whose box starts beyond the closest hit are dropped)
    auto pop = [&]() {
        cur = -1;
        while (sp > 0) {
            sp--;
            const float lb = stk_lb[sp * bs];
            if (prune(lb, best_slot >= 0, best)) continue;
            const uint32_t w = stk_node[sp * bs];
            cur = (int)(w & 0x7FFFFFFFu);
            amb = (w >> 31) != 0;
            const NodeRec r = sc.nodes[cur];
            c0 = __float_as_int(r.lo.w);
            c1 = __float_as_int(r.hi.w);
            break;
        }
    };

    for (;;) {
        // ---- refill: lanes without a ray take the next rays of the wave's pool ---------------------
        const unsigned long long need = __ballot(!active);
        if (need) {
            if (pool_next == pool_end && !exhausted) {
                uint32_t base = 0;
                if (lane == (uint32_t)__ffsll((long long)need) - 1u) base = atomicAdd(queue, RAY_CHUNK);
                base = __shfl(base, __ffsll((long long)need) - 1);
                if (base >= n) {
                    exhausted = true;
                } else {
                    pool_next = base;
                    pool_end = min(base + RAY_CHUNK, n);
                }
            }
            const uint32_t avail = pool_end - pool_next;
            const uint32_t want = __popcll(need);
            const uint32_t take = min(avail, want);
            const uint32_t rank = __popcll(need & lt_mask);
            if (!active && rank < take) {
                idx = pool_next + rank;
                o = mk3(O[idx], O[plane_stride + idx], O[2 * plane_stride + idx]);
                d = mk3(D[idx], D[plane_stride + idx], D[2 * plane_stride + idx]);
                if (MODE != MIRT_MODE_UNCONDITIONAL) {
                    a = aux[idx];
                    contact = unpack_contact(a);
                    inside = unpack_inside(a);
                }
                best = -1.f;
                best_slot = -1;
                sp = 0;
                pre = ray_prepare(sc, o, d);
                const NodeRec r = sc.nodes[0];
                if (COUNT) c_box++;
                bool rh, ru;
                float rlb;
                test_box(r, rh, rlb, ru);
                if (rh) {
                    cur = 0;
                    amb = ru;
                    c0 = __float_as_int(r.lo.w);
                    c1 = __float_as_int(r.hi.w);
                } else {
                    cur = -1;
                }
                active = true;
            }
            pool_next += take;
        }
        if (!__any(active)) {
            if (exhausted || n == 0) break;
            continue;
        }

        // ---- inner phase: descend until this lane stands on a leaf or has no node left ------------
        while (active && cur >= 0 && c0 >= 0) {
            const int left = cur + 1, right = c0;
            const NodeRec rl = sc.nodes[left];
            const NodeRec rr = sc.nodes[right];
            if (COUNT) c_box += 2;
            bool hl, hr, ul, ur;
            float lbl, lbr;
            test_box(rl, hl, lbl, ul);
            test_box(rr, hr, lbr, ur);
            const bool have = best_slot >= 0;
            hl = hl && !prune(lbl, have, best);
            hr = hr && !prune(lbr, have, best);
            ul |= amb;
            ur |= amb;
            if (hl && hr) {
                const bool left_first = lbl <= lbr;
                stk_node[sp * bs] = (left_first ? (uint32_t)right : (uint32_t)left) |
                                    ((left_first ? ur : ul) ? 0x80000000u : 0u);
                stk_lb[sp * bs] = left_first ? lbr : lbl;
                sp++;
                cur = left_first ? left : right;
                amb = left_first ? ul : ur;
                c0 = __float_as_int(left_first ? rl.lo.w : rr.lo.w);
                c1 = __float_as_int(left_first ? rl.hi.w : rr.hi.w);
            } else if (hl) {
                cur = left;
                amb = ul;
                c0 = __float_as_int(rl.lo.w);
                c1 = __float_as_int(rl.hi.w);
            } else if (hr) {
                cur = right;
                amb = ur;
                c0 = __float_as_int(rr.lo.w);
                c1 = __float_as_int(rr.hi.w);
            } else {
                pop();
            }
        }

        // ---- leaf phase: every lane that stands on a leaf tests its triangles -----------------------
        if (active && cur >= 0) {
            const int lbeg = ~c0, lend = c1;
            for (int s = lbeg; s < lend; s++) {
                const TriRec T = sc.tris[s];
                if (COUNT) c_tri++;
                const float t = tri_test(T, o, d);
                if (!(t >= 0.f)) continue;
                const bool wins = best_slot < 0 || t < best || (t == best && best_slot < lbeg);
                if (!wins) continue;
                if (MODE == MIRT_MODE_LAST_TID) {
                    if (__float_as_int(sc.vns[s].n1.w) == a) continue;
                } else if (MODE == MIRT_MODE_NO_SELF) {
                    const VnRec V = sc.vns[s];
                    const int hobj = __float_as_int(V.n0.w);
                    const f3 pos = o + d * t;
                    const f3 nn = world_normal(T, V, sc.objects[hobj].has_vn != 0, pos);
                    if (!legal_noself(dot3(d, nn), hobj, contact, inside)) continue;
                }
                if (amb) {
                    // an unproven box lies on the way here: the reference reaches this triangle only if
                    // its own box test passes on every ancestor (KDTree.cpp:297)
                    if (COUNT) c_walk++;
                    bool ok = true;
                    for (int nd = cur; nd >= 0 && ok; nd = sc.parents[nd]) {
                        const NodeRec r = sc.nodes[nd];
                        ok = box_test(r.lo, r.hi, o, d).hit;
                    }
                    if (!ok) continue;
                }
                best = t;
                best_slot = s;
            }
            pop();
        }

        // ---- retire finished rays -----------------------------------------------------------------
        if (active && cur < 0) {
            t_out[idx] = best_slot >= 0 ? best : -1.f;
            if (SLOT_OUT)
                id_out[idx] = best_slot;
            else
                id_out[idx] = best_slot >= 0 ? __float_as_int(sc.vns[best_slot].n1.w) : -1;
            active = false;
        }
    }

    if (COUNT) {
        // wave-level reduction, then one atomic per wave
        for (int off = 32; off > 0; off >>= 1) {
            c_box += __shfl_down(c_box, off);
            c_tri += __shfl_down(c_tri, off);
            c_amb += __shfl_down(c_amb, off);
            c_bad += __shfl_down(c_bad, off);
            c_walk += __shfl_down(c_walk, off);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], c_box);
            atomicAdd(&counters[1], c_tri);
            atomicAdd(&counters[2], c_amb);
            atomicAdd(&counters[3], c_bad);
            atomicAdd(&counters[4], c_walk);
        }
    }
}

}  // namespace mirt
