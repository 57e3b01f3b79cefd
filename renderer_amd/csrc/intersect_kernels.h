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
// Execution shape:
//   * persistent waves: the grid is sized to the chip, not to the batch; each wave draws rays from a
//     global queue in chunks of RAY_CHUNK (one atomic per chunk) and a lane whose ray is finished
//     takes the next ray instead of idling until the slowest ray of its wave is done;
//   * "while-while" traversal: all lanes descend inner nodes until each has reached a leaf (or run
//     out of nodes), then all lanes test triangles of their leaf together;
//   * one 64-byte PairRec load per inner step brings both children's boxes and links (breadth-first
//     pair order: the top of the tree is contiguous and stays in L1);
//   * node boxes are decided by the slab classifier of mirt_dev.h (no divisions) whenever that
//     decision is provably the reference's; a box it cannot prove is entered anyway and marks the
//     path "ambiguous"; if a triangle under such a path would become the closest hit, the ray is
//     handed to k_intersect_exact (a second, small launch that traces the deferred rays with the
//     exact reference box test throughout; ~3e-5 of the rays on the Cornell batch). Rays outside the
//     classifier's error model (zero / extreme components) are deferred the same way.
//     The instrumented instantiation (COUNT) also runs the exact test beside every proven decision
//     and counts disagreements (must be 0; tests/test_gpu_intersect.py).
#pragma once
#include "mirt_dev.h"

#pragma clang fp contract(off)

namespace mirt {

#ifndef MIRT_RAY_CHUNK
#define MIRT_RAY_CHUNK 256
#endif
#ifndef MIRT_LEAF_BATCH
#define MIRT_LEAF_BATCH 8
#endif
constexpr uint32_t RAY_CHUNK = MIRT_RAY_CHUNK;  // rays a wave reserves per queue atomic
constexpr int LEAF_BATCH = MIRT_LEAF_BATCH;     // triangles a lane tests per leaf phase before the wave regroups

struct Audit {
    unsigned long long box = 0, tri = 0, amb = 0, bad = 0, walk = 0;
    unsigned long long w_inner = 0, w_leaf = 0, w_outer = 0;  // wave-level iteration counts (lane 0)
};

// NoSelfIntersection / last_tid acceptance of a candidate hit (the part after the triangle test)
template <int MODE>
__device__ __forceinline__ bool candidate_legal(const DevScene& sc, const TriRec& T, int s, f3 o, f3 d,
                                                float t, int a, int contact, int inside) {
    if (MODE == MIRT_MODE_LAST_TID) return __float_as_int(sc.vns[s].n1.w) != a;
    if (MODE == MIRT_MODE_NO_SELF) {
        const VnRec V = sc.vns[s];
        const int hobj = __float_as_int(V.n0.w);
        const f3 pos = o + d * t;
        const f3 nn = world_normal(T, V, sc.objects[hobj].has_vn != 0, pos);
        return legal_noself(dot3(d, nn), hobj, contact, inside);
    }
    return true;
}

// Traces the deferred rays (indices in `list`, count in *list_n) with the exact reference box test
// throughout: pre-order node records, same ordering / pruning / tie rules as the main kernel.
template <int MODE, bool COUNT, bool SLOT_OUT>
__global__ void __launch_bounds__(256)
k_intersect_exact(DevScene sc, const float* __restrict__ O, const float* __restrict__ D,
                  const int* __restrict__ aux, uint32_t plane_stride, float* __restrict__ t_out,
                  int* __restrict__ id_out, unsigned long long* __restrict__ counters,
                  const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_n) {
    extern __shared__ uint32_t lds[];
    const uint32_t bs = blockDim.x;
    uint32_t* stk_node = lds + threadIdx.x;
    float* stk_lb = (float*)(lds + (sc.depth + 1) * bs) + threadIdx.x;
    const uint32_t count = *list_n;
    Audit au;
    for (uint32_t k = blockIdx.x * bs + threadIdx.x; k < count; k += gridDim.x * bs) {
        const uint32_t idx = list[k];
        const f3 o = mk3(O[idx], O[plane_stride + idx], O[2 * plane_stride + idx]);
        const f3 d = mk3(D[idx], D[plane_stride + idx], D[2 * plane_stride + idx]);
        int a = 0;
        if (MODE != MIRT_MODE_UNCONDITIONAL) a = aux[idx];
        const int contact = unpack_contact(a), inside = unpack_inside(a);
        float best = -1.f;
        int best_slot = -1, sp = 0, cur = -1;
        {
            const NodeRec r = sc.nodes[0];
            if (COUNT) au.box++;
            if (box_test(r.lo, r.hi, o, d).hit) cur = 0;
        }
        while (cur >= 0) {
            const NodeRec r = sc.nodes[cur];
            const int c0 = __float_as_int(r.lo.w);
            bool pop = false;
            if (c0 < 0) {
                const int lbeg = ~c0, lend = __float_as_int(r.hi.w);
                for (int s = lbeg; s < lend; s++) {
                    const TriRec T = sc.tris[s];
                    if (COUNT) au.tri++;
                    const float t = tri_test(T, o, d);
                    if (!(t >= 0.f)) continue;
                    if (!(best_slot < 0 || t < best || (t == best && best_slot < lbeg))) continue;
                    if (!candidate_legal<MODE>(sc, T, s, o, d, t, a, contact, inside)) continue;
                    best = t;
                    best_slot = s;
                }
                pop = true;
            } else {
                const NodeRec rl = sc.nodes[cur + 1];
                const NodeRec rr = sc.nodes[c0];
                if (COUNT) au.box += 2;
                const BoxHit bl = box_test(rl.lo, rl.hi, o, d);
                const BoxHit br = box_test(rr.lo, rr.hi, o, d);
                const bool have = best_slot >= 0;
                const bool hl = bl.hit && !prune(bl.lb, have, best);
                const bool hr = br.hit && !prune(br.lb, have, best);
                if (hl && hr) {
                    const bool lf = bl.lb <= br.lb;
                    stk_node[sp * bs] = lf ? (uint32_t)c0 : (uint32_t)(cur + 1);
                    stk_lb[sp * bs] = lf ? br.lb : bl.lb;
                    sp++;
                    cur = lf ? cur + 1 : c0;
                } else if (hl) {
                    cur = cur + 1;
                } else if (hr) {
                    cur = c0;
                } else {
                    pop = true;
                }
            }
            if (pop) {
                cur = -1;
                while (sp > 0) {
                    sp--;
                    if (prune(stk_lb[sp * bs], best_slot >= 0, best)) continue;
                    cur = (int)stk_node[sp * bs];
                    break;
                }
            }
        }
        t_out[idx] = best_slot >= 0 ? best : -1.f;
        if (SLOT_OUT)
            id_out[idx] = best_slot;
        else
            id_out[idx] = best_slot >= 0 ? __float_as_int(sc.vns[best_slot].n1.w) : -1;
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            au.box += __shfl_down(au.box, off);
            au.tri += __shfl_down(au.tri, off);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters[0], au.box);
            atomicAdd(&counters[1], au.tri);
        }
    }
}

// MODE: MIRT_MODE_*   COUNT: audit counters (never timed)   SLOT_OUT: write the slot index (internal
// path-tracing use) instead of the global triangle id.
// `queue`: two zero-initialised words per launch: [0] index of the next unassigned ray, [1] number
// of rays deferred to k_intersect_exact (their indices go to `defer`, capacity n).
template <int MODE, bool COUNT, bool SLOT_OUT>
__global__ void __launch_bounds__(256)
k_intersect(DevScene sc, const float* __restrict__ O, const float* __restrict__ D,
            const int* __restrict__ aux, uint32_t n, uint32_t plane_stride, float* __restrict__ t_out,
            int* __restrict__ id_out, unsigned long long* __restrict__ counters,
            uint32_t* __restrict__ queue, uint32_t* __restrict__ defer) {
    extern __shared__ uint32_t lds[];
    const uint32_t bs = blockDim.x;
    uint32_t* stk_node = lds + threadIdx.x;  // [level * bs]: child link
    float* stk_lb = (float*)(lds + (sc.depth + 1) * bs) + threadIdx.x;  // entry distance | amb << 31
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    Audit au;
    uint32_t pool_next = 0, pool_end = 0;  // wave-uniform ray pool
    bool exhausted = false;

    // per-lane traversal state. `link` is the current child link: >= 0 pair index (inner node),
    // < 0 leaf (~link = begin | count << 24); has = false: nothing left to visit.
    bool active = false, has = false, amb = false;
    uint32_t idx = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
    int a = 0, contact = -1, inside = -1;
    float best = -1.f;
    int best_slot = -1, sp = 0, link = 0;
    int leaf_pos = 0, leaf_end = 0, leaf_beg = 0;  // triangles still to test in the current leaf
    RayPre pre;
    pre.inv = mk3(0, 0, 0);
    pre.a_ray = 0.f;
    pre.fast = false;

    auto classify = [&](const float4 lo, const float4 hi, float& lb, bool& unproven) -> bool {
        const int code = slab_classify(lo, hi, o, pre, lb);
        unproven = code == BOX_AMBIGUOUS;
        if (COUNT) {
            au.box++;
            if (unproven)
                au.amb++;
            else if (box_test(lo, hi, o, d).hit != (code == BOX_HIT))
                au.bad++;
        }
        return code != BOX_MISS;
    };
    auto enter = [&](int lk) {  // start visiting child link lk
        link = lk;
        has = true;
        if (lk < 0) {
            const uint32_t v = (uint32_t)~lk;
            leaf_beg = leaf_pos = (int)(v & 0xFFFFFFu);
            leaf_end = leaf_pos + (int)(v >> 24);
        }
    };
    auto pop = [&]() {  // next node from the stack; entries beyond the closest hit are dropped
        has = false;
        while (sp > 0) {
            sp--;
            const uint32_t w = __float_as_uint(stk_lb[sp * bs]);
            if (prune(__uint_as_float(w & 0x7FFFFFFFu), best_slot >= 0, best)) continue;
            amb = (w >> 31) != 0;
            enter((int)stk_node[sp * bs]);
            break;
        }
    };

    for (;;) {
        // ---- refill: lanes without a ray take the next rays of the wave's pool ---------------------
        const unsigned long long need = __ballot(!active);
        if (need) {
            if (pool_next == pool_end && !exhausted) {
                uint32_t base = 0;
                const int leader = __ffsll((long long)need) - 1;
                if ((int)lane == leader) base = atomicAdd(queue, RAY_CHUNK);
                base = __shfl(base, leader);
                if (base >= n) {
                    exhausted = true;
                } else {
                    pool_next = base;
                    pool_end = min(base + RAY_CHUNK, n);
                }
            }
            const uint32_t take = min(pool_end - pool_next, (uint32_t)__popcll(need));
            const uint32_t rank = __popcll(need & lt_mask);
            if (!active && rank < take) {
                idx = pool_next + rank;
                o = mk3(O[idx], O[plane_stride + idx], O[2 * plane_stride + idx]);
                d = mk3(D[idx], D[plane_stride + idx], D[2 * plane_stride + idx]);
                a = 0;
                if (MODE != MIRT_MODE_UNCONDITIONAL) a = aux[idx];
                contact = unpack_contact(a);
                inside = unpack_inside(a);
                best = -1.f;
                best_slot = -1;
                sp = 0;
                has = false;
                amb = false;
                active = true;
                pre = ray_prepare(sc, o, d);
                if (pre.fast) {
                    float lb;
                    bool un;
                    if (classify(sc.root_lo, sc.root_hi, lb, un)) {
                        amb = un;
                        enter(__float_as_int(sc.root_lo.w));
                    }
                } else {
                    defer[atomicAdd(queue + 1, 1u)] = idx;  // outside the classifier's error model
                    active = false;
                }
            }
            pool_next += take;
        }
        if (!__any(active)) {
            if (exhausted) break;
            continue;
        }

        // ---- inner phase: descend until this lane stands on a leaf or has no node left ------------
        for (;;) {
            const bool go = active && has && link >= 0;
            if (COUNT) {  // same loop, written so that whole-wave iterations can be counted
                if (!__any(go)) break;
                if (lane == 0) au.w_inner++;
                if (!go) continue;
            } else if (!go) {
                break;
            }
            const PairRec P = sc.pairs[link];
            float lbl, lbr;
            bool ul, ur;
            bool hl = classify(P.llo, P.lhi, lbl, ul);
            bool hr = classify(P.rlo, P.rhi, lbr, ur);
            const bool have = best_slot >= 0;
            hl = hl && !prune(lbl, have, best);
            hr = hr && !prune(lbr, have, best);
            ul |= amb;
            ur |= amb;
            const int kl = __float_as_int(P.llo.w), kr = __float_as_int(P.rlo.w);
            if (hl && hr) {
                const bool lf = lbl <= lbr;
                stk_node[sp * bs] = (uint32_t)(lf ? kr : kl);
                stk_lb[sp * bs] = __uint_as_float(__float_as_uint(lf ? lbr : lbl) | ((lf ? ur : ul) ? 0x80000000u : 0u));
                sp++;
                amb = lf ? ul : ur;
                enter(lf ? kl : kr);
            } else if (hl) {
                amb = ul;
                enter(kl);
            } else if (hr) {
                amb = ur;
                enter(kr);
            } else {
                pop();
            }
        }

        // ---- leaf phase: up to LEAF_BATCH triangles of this lane's leaf ------------------------------
        if (COUNT) {
            int trips = (active && has) ? min(LEAF_BATCH, leaf_end - leaf_pos) : 0;
            for (int off = 32; off > 0; off >>= 1) trips = max(trips, __shfl_xor(trips, off));
            if (lane == 0) {
                au.w_leaf += trips;
                au.w_outer++;
            }
        }
        if (active && has) {
            const int stop = min(leaf_pos + LEAF_BATCH, leaf_end);
            for (int s = leaf_pos; s < stop; s++) {
                const TriRec T = sc.tris[s];
                if (COUNT) au.tri++;
                const float t = tri_test(T, o, d);
                if (!(t >= 0.f)) continue;
                if (!(best_slot < 0 || t < best || (t == best && best_slot < leaf_beg))) continue;
                if (!candidate_legal<MODE>(sc, T, s, o, d, t, a, contact, inside)) continue;
                if (amb) {
                    // the closest hit would rest on a box the classifier could not prove: retrace
                    // this ray with the exact test (k_intersect_exact)
                    if (COUNT) au.walk++;
                    defer[atomicAdd(queue + 1, 1u)] = idx;
                    active = false;
                    break;
                }
                best = t;
                best_slot = s;
            }
            leaf_pos = stop;
            if (active && leaf_pos >= leaf_end) pop();
        }

        // ---- retire finished rays -----------------------------------------------------------------
        if (active && !has) {
            t_out[idx] = best_slot >= 0 ? best : -1.f;
            if (SLOT_OUT)
                id_out[idx] = best_slot;
            else
                id_out[idx] = best_slot >= 0 ? __float_as_int(sc.vns[best_slot].n1.w) : -1;
            active = false;
        }
    }

    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            au.box += __shfl_down(au.box, off);
            au.tri += __shfl_down(au.tri, off);
            au.amb += __shfl_down(au.amb, off);
            au.bad += __shfl_down(au.bad, off);
            au.walk += __shfl_down(au.walk, off);
        }
        if (lane == 0) {
            atomicAdd(&counters[5], au.w_inner);
            atomicAdd(&counters[6], au.w_leaf);
            atomicAdd(&counters[7], au.w_outer);
            atomicAdd(&counters[0], au.box);
            atomicAdd(&counters[1], au.tri);
            atomicAdd(&counters[2], au.amb);
            atomicAdd(&counters[3], au.bad);
            atomicAdd(&counters[4], au.walk);
        }
    }
}

}  // namespace mirt
