// Reconstruction of the reverted round-1 experiment: k_verify walking the winner's chain TWO levels per
// step, records holding a node's own box and its parent's box in pair layout. Built twice: -DCAPPED=0 / 1.
#include "../../renderer_amd/csrc/intersect_kernels.h"
using namespace mirt;
struct NodeRec2 { float4 a0, b0, a1, b1; };  // own box (b0.z = bits(parent)), parent's box (b1.z = bits(grandparent))
template <bool SLOT_OUT>
__global__ void __launch_bounds__(256)
k_verify2(DevScene sc, const NodeRec2* __restrict__ rec2, const float* __restrict__ O, const float* __restrict__ D, uint32_t n,
          uint32_t plane_stride, int* __restrict__ id_out, uint32_t* __restrict__ list, uint32_t* __restrict__ list_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool redo = false;
    if (i < n) {
        const f3 o = mk3(O[i], O[plane_stride + i], O[2 * plane_stride + i]);
        const f3 d = mk3(D[i], D[plane_stride + i], D[2 * plane_stride + i]);
        const RayPre pre = ray_prepare(sc, o, d);
        const RayPre2 pre2 = ray_pairs(o, pre);
        const int slot = id_out[i];
        if (!pre.fast || slot == -2) {
            redo = true;
        } else if (slot >= 0) {
            int node = sc.slot_node[slot];
#if CAPPED
            for (int step = 0; node >= 0; step++) {
                if (step > 64) { redo = true; break; }
#else
            while (node >= 0) {
#endif
                const NodeRec2 R = rec2[node];
                float lb;
                const int c0 = slab_classify_pair(R.a0, R.b0, pre2, pre, lb);
                const int c1 = slab_classify_pair(R.a1, R.b1, pre2, pre, lb);
                const bool ok0 = c0 == BOX_HIT || (c0 == BOX_AMBIGUOUS && box_test(pair_lo(R.a0, R.b0), pair_hi(R.a0, R.b0), o, d).hit);
                const bool ok1 = c1 == BOX_HIT || (c1 == BOX_AMBIGUOUS && box_test(pair_lo(R.a1, R.b1), pair_hi(R.a1, R.b1), o, d).hit);
                if (!(ok0 && ok1)) { redo = true; break; }
                node = __float_as_int(R.b1.z);
            }
            if (!redo && !SLOT_OUT) id_out[i] = __float_as_int(sc.vns[slot].n1.w);
        } else if (!SLOT_OUT) {
            id_out[i] = -1;
        }
    }
    const unsigned long long m = __ballot(redo);
    if (m) {
        const uint32_t lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(list_count, (uint32_t)__popcll(m));
        base = __shfl(base, leader);
        if (redo) list[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
    }
}
template __global__ void k_verify2<false>(DevScene, const NodeRec2*, const float*, const float*, uint32_t, uint32_t, int*, uint32_t*, uint32_t*);
