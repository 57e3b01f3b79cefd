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
#pragma once
#include "mirt_dev.h"

#pragma clang fp contract(off)

namespace mirt {

// MODE: MIRT_MODE_*   COUNT: accumulate box/triangle test counts   SLOT_OUT: write the slot index
// (internal path-tracing use) instead of the global triangle id.
template <int MODE, bool COUNT, bool SLOT_OUT>
__global__ void __launch_bounds__(256)
k_intersect(DevScene sc, const float* __restrict__ O, const float* __restrict__ D,
            const int* __restrict__ aux, const uint32_t* __restrict__ n_ptr, uint32_t n_max,
            uint32_t plane_stride, float* __restrict__ t_out, int* __restrict__ id_out,
            unsigned long long* __restrict__ counters) {
    extern __shared__ uint32_t lds[];
    const uint32_t bs = blockDim.x;
    uint32_t* stk_node = lds + threadIdx.x;            // [level * bs]
    float* stk_lb = (float*)(lds + sc.depth * bs + bs) + threadIdx.x;

    const uint32_t n = n_ptr ? min(*n_ptr, n_max) : n_max;
    const uint32_t i = blockIdx.x * bs + threadIdx.x;
    unsigned long long c_box = 0, c_tri = 0;
    if (i < n) {
        const f3 o = mk3(O[i], O[plane_stride + i], O[2 * plane_stride + i]);
        const f3 d = mk3(D[i], D[plane_stride + i], D[2 * plane_stride + i]);
        int a = 0;
        if (MODE != MIRT_MODE_UNCONDITIONAL) a = aux[i];
        const int contact = unpack_contact(a), inside = unpack_inside(a);

        float best = -1.f;
        int best_slot = -1;
        int sp = 0;
        int cur = -1;
        {
            const NodeRec r = sc.nodes[0];
            if (COUNT) c_box++;
            if (box_test(r.lo, r.hi, o, d).hit) cur = 0;
        }
        while (cur >= 0) {
            const NodeRec r = sc.nodes[cur];
            const int c0 = __float_as_int(r.lo.w);
            bool pop = false;
            if (c0 < 0) {  // leaf
                const int lbeg = ~c0, lend = __float_as_int(r.hi.w);
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
                    best = t;
                    best_slot = s;
                }
                pop = true;
            } else {  // inner: left = cur + 1, right = c0
                const NodeRec rl = sc.nodes[cur + 1];
                const NodeRec rr = sc.nodes[c0];
                if (COUNT) c_box += 2;
                const BoxHit bl = box_test(rl.lo, rl.hi, o, d);
                const BoxHit br = box_test(rr.lo, rr.hi, o, d);
                const bool have = best_slot >= 0;
                const bool hl = bl.hit && !prune(bl.lb, have, best);
                const bool hr = br.hit && !prune(br.lb, have, best);
                if (hl && hr) {
                    const bool left_first = bl.lb <= br.lb;
                    stk_node[sp * bs] = left_first ? (uint32_t)c0 : (uint32_t)(cur + 1);
                    stk_lb[sp * bs] = left_first ? br.lb : bl.lb;
                    sp++;
                    cur = left_first ? cur + 1 : c0;
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
                    const float lb = stk_lb[sp * bs];
                    if (prune(lb, best_slot >= 0, best)) continue;
                    cur = (int)stk_node[sp * bs];
                    break;
                }
            }
        }
        t_out[i] = best_slot >= 0 ? best : -1.f;
        if (SLOT_OUT)
            id_out[i] = best_slot;
        else
            id_out[i] = best_slot >= 0 ? __float_as_int(sc.vns[best_slot].n1.w) : -1;
    }
    if (COUNT) {
        // wave-level reduction, then one atomic per wave
        for (int off = 32; off > 0; off >>= 1) {
            c_box += __shfl_down(c_box, off);
            c_tri += __shfl_down(c_tri, off);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters[0], c_box);
            atomicAdd(&counters[1], c_tri);
        }
    }
}

}  // namespace mirt
