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
//     global queue in chunks (64 / 128 / 256 rays by launch size and mode, one atomic per chunk) and a
//     lane whose ray is finished takes the next ray of the chunk instead of idling until the slowest ray
//     of its wave is done;
//   * register budget by instantiation: the fast-tree kernels are built for 7 waves per SIMD (latency of
//     the node loads is what they wait for), the exact ones for 5; the per-lane stack is addressed through
//     pointers with explicit address spaces (LDS levels first, a global spill area behind them);
//   * "while-while" traversal: all lanes descend inner nodes until each has reached a leaf (or run
//     out of nodes), then all lanes test triangles of their leaf together;
//   * one 64-byte PairRec load per inner step brings both children's boxes and links (breadth-first
//     pair order: the top of the tree is contiguous and stays in L1);
//   * node boxes are decided by the slab classifier of mirt_dev.h (no divisions) whenever that
//     decision is provably the reference's; a box it cannot prove (~8e-5 of the tests on the Cornell
//     batch) and every box of a ray outside its error model (zero / extreme components) is decided by
//     the reference arithmetic itself, in a branch whole waves normally skip.
//     The instrumented instantiation (COUNT) also runs the exact test beside every proven decision
//     and counts disagreements (must be 0; tests/test_gpu_intersect.py);
//   * in NO_SELF mode the predicate (shading normal, ~300 instructions for smooth-shaded objects) is
//     evaluated once per leaf pass for all lanes that found a candidate, not inside the triangle loop.
#pragma once
#include "mirt_dev.h"

#pragma clang fp contract(off)

namespace mirt {

#ifndef MIRT_RAY_CHUNK
#define MIRT_RAY_CHUNK 64
#endif
#ifndef MIRT_RAY_CHUNK_COHERENT
#define MIRT_RAY_CHUNK_COHERENT 256
#endif
#ifndef MIRT_RAY_CHUNK_INCOHERENT
#define MIRT_RAY_CHUNK_INCOHERENT 256
#endif
#ifndef MIRT_RAY_CHUNK_MIN_RAYS
#define MIRT_RAY_CHUNK_MIN_RAYS (1u << 20)
#endif
#ifndef MIRT_LEAF_BATCH
#define MIRT_LEAF_BATCH 8
#endif
// Rays a wave reserves per queue atomic. Same-address atomics of the whole chip are served one per ~11 ns (per 64-byte line), so the
// chunk size caps a launch at chunk / 11 ns (64 rays: 6 Grays/s, below what camera rays reach), and a wave that
// waits for its atomic idles. In the fast-tree kernels the chunk is a pool in LDS that the four waves of a block
// consume together in pieces of <= 64 rays (s_pool below), so the end of a launch is balanced at that granularity
// whatever the chunk; 4 Mi Cornell camera rays: 0.41 ms with private 256-ray chunks, 0.36 with a static round-robin
// (no atomics, no balancing: bounce rays 1.90 ms instead of 1.75), 0.33 with shared pools (bounce rays 1.70).
// The exact kernels keep private chunks. Small launches draw 64 so that every wave of the grid gets work.
constexpr uint32_t RAY_CHUNK_SMALL = MIRT_RAY_CHUNK;
// The fast-tree kernels' loop does one kind of work per iteration, chosen by majority vote of the wave's lanes (the
// exact kernels keep the while-while form: every lane descends until all stand on a leaf, then all test triangles).
#ifndef MIRT_GUARD
#define MIRT_GUARD 0  // debug build: wild indices are reported (printf) and made harmless
#endif
#ifndef MIRT_REFILL_MIN
#define MIRT_REFILL_MIN 24  // idle lanes that trigger a refill (a refill also happens whenever no lane has work left)
#endif
// a traversal step runs when (lanes on inner nodes) * NODE_WEIGHT >= (lanes on leaves) * LEAF_WEIGHT, else a leaf pass.
// Swept on the 12.6 M-ray bounce batches (Cornell / 1 M triangles, ms): 1:1 3.45 / 5.62, 2:1 3.26 / 5.63, 3:1 3.31 / 5.90,
// 4:1 3.36 / 6.14, 1:2 3.74 / 5.92 (leaf batch 3-4, refill at 16 idle lanes; the while-while form: 3.52 / 7.19).
// Again with the triangle test split (edge signs per trip, plane part once per pass): 2:1 2.47 / 4.42, 3:2 2.45 / 4.30,
// 1:1 2.49 / 4.29; with single-triangle leaves: 2:1 1.83 / 3.49, 3:2 1.79 / 3.42, 1:1 1.79 / 3.38.
#ifndef MIRT_VOTE_NODE_WEIGHT
#define MIRT_VOTE_NODE_WEIGHT 3
#endif
// triangles per leaf pass (a lane goes on when its next node is a leaf again): bounce rays 1, coherent rays 2.
// Swept, Cornell / 1 M triangles / fog, mode-2 call in ms: 1 1.79 / 3.42 / 1.64, 2 1.82 / 3.55 / 1.70, 4 1.77 / 3.44 / 1.63;
// 4 Mi camera rays: 1 0.42 / 0.755 / 0.48, 2 0.405 / 0.745 / 0.457.
#ifndef MIRT_LEAF_REPS
#define MIRT_LEAF_REPS 1
#endif
#ifndef MIRT_LEAF_REPS_COHERENT
#define MIRT_LEAF_REPS_COHERENT 2
#endif
#ifndef MIRT_NODE_REPS
#define MIRT_NODE_REPS 3  // traversal steps per node pass (lanes that reach a leaf sit the rest out)
#endif
#ifndef MIRT_VOTE_LEAF_WEIGHT
#define MIRT_VOTE_LEAF_WEIGHT 2
#endif
#ifndef MIRT_FAST_BUDGET
#define MIRT_FAST_BUDGET 16
#endif
constexpr int FAST_AMBIGUOUS_BUDGET = MIRT_FAST_BUDGET;  // k_intersect<FAST>: unproven boxes a ray may enter before it is handed to the exact kernel
// triangles a lane tests per leaf phase before the wave regroups (the exact kernels; a fast-tree leaf is one triangle)
constexpr int LEAF_BATCH_EXACT = MIRT_LEAF_BATCH;

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
        const f3 nn = shading_normal<false>(sc, T, V, s, pos);
        return legal_noself(dot3(d, nn), hobj, contact, inside);
    }
    return true;
}

// MODE: MIRT_MODE_*   COUNT: audit counters (never timed)   SLOT_OUT: write the slot index (internal
// path-tracing use) instead of the global triangle id.
// `queue`: one zero-initialised word per launch, the index of the next unassigned ray.
// ANYHIT: visibility query (MCRenderer::testVisibility, MCRenderer.cpp:106-118): instead of the
// closest hit, vis_out[i] = 1 unless some legal hit has t - limit[i] < -EPSILON (macros.h:5), which
// is exactly "closest legal hit is nearer than the limit" because the rounded difference is
// monotone in t. The traversal stops at the first such hit and never enters boxes beyond the limit.
// FAST: walk the fast tree (DevScene.fpairs) instead of the reference tree. Every box decision only has
// to be conservative there (a box is skipped only when the classifier PROVES the reference test
// misses it; proven hits and unproven boxes are both entered), the result is written as (t, SLOT) and
// k_verify decides whether it is the reference's answer. Rays outside the classifier's error model are
// not traversed at all (k_verify sends them to the exact kernel).
// BUMP: the scene has bump textures (SceneObject.cpp:250-283): the predicate's normal is the perturbed one.
// Only the NO_SELF instantiations differ; scenes without bump textures never run them.
// ANYHIT + FAST: an occluder found in the fast tree's superset occludes if its chain of reference boxes
// passes (k_verify); no occluder in the superset means none in the reference's set. The kernel leaves the
// occluder's slot in id_out for k_verify; rejected rays are answered by the exact any-hit kernel.
// ray_list != nullptr: the queue hands out positions of ray_list, which holds the ray indices (the
// exact re-run of the rays k_verify rejected); the launch does nothing unless the list holds more than
// list_min rays (shorter lists are k_rerun's).
#ifndef MIRT_BLOCK
#define MIRT_BLOCK 256  // lanes per block of the traversal kernels (swept: 128 / 256 / 512)
#endif
#ifndef MIRT_WAVES_PER_EU
#define MIRT_WAVES_PER_EU 5
#endif
#ifndef MIRT_WAVES_PER_EU_FAST
#define MIRT_WAVES_PER_EU_FAST 7
#endif
template <int MODE, bool COUNT, bool SLOT_OUT, bool ANYHIT = false, bool FAST = false, bool BUMP = false>
__global__ void __launch_bounds__(MIRT_BLOCK)
__attribute__((amdgpu_waves_per_eu(FAST && !COUNT ? MIRT_WAVES_PER_EU_FAST : MIRT_WAVES_PER_EU,
                                   FAST && !COUNT ? MIRT_WAVES_PER_EU_FAST : MIRT_WAVES_PER_EU)))
k_intersect(DevScene sc, const float* __restrict__ O, const float* __restrict__ D,
            const int* __restrict__ aux, uint32_t n_direct, const uint32_t* __restrict__ n_ptr,
            uint32_t plane_stride, float* __restrict__ t_out,
            int* __restrict__ id_out, unsigned long long* __restrict__ counters,
            uint32_t* __restrict__ queue, uint32_t lds_levels, uint2* __restrict__ spill,
            const float* __restrict__ limit = nullptr, unsigned char* __restrict__ vis_out = nullptr,
            const uint32_t* __restrict__ ray_list = nullptr, uint32_t list_min = 0,
            const uint32_t* __restrict__ off_ptr = nullptr) {
    extern __shared__ uint32_t lds[];
    constexpr int LEAF_BATCH = LEAF_BATCH_EXACT;
    const PairRec* __restrict__ pairs = FAST ? sc.fpairs : sc.pairs;
    const float4 root_lo = FAST ? sc.froot_lo : sc.root_lo, root_hi = FAST ? sc.froot_hi : sc.root_hi;
    // ray count: given, or (wavefront path tracer) left in device memory by the previous shade pass
    const uint32_t n = n_ptr ? min(*n_ptr, n_direct) : n_direct;
    // the batch starts at entry *off_ptr of every array (the path tracer's two-ended compaction, pt_kernels.h k_shade)
    if (off_ptr) {
        const uint32_t off = *off_ptr;
        O += off;
        D += off;
        if (aux) aux += off;
        if (t_out) t_out += off;
        if (id_out) id_out += off;
    }
    // re-run of a LONG list only (launch_intersect): lists of at most list_min rays belong to k_rerun
    if (ray_list != nullptr && n <= list_min) return;
    const uint32_t bs = blockDim.x;
    const uint32_t RAY_CHUNK = n < MIRT_RAY_CHUNK_MIN_RAYS ? RAY_CHUNK_SMALL
                               : (MODE == MIRT_MODE_NO_SELF ? MIRT_RAY_CHUNK_INCOHERENT : MIRT_RAY_CHUNK_COHERENT);
    // per-lane stack, [level][lane] of (child link, entry distance): the first
    // lds_levels levels in LDS, deeper ones (rarely reached) in the global spill area
    // (address spaces spelled out: a generic pointer chosen between the two at run time costs an aperture test
    // per access, and trips this compiler's verifier at 7 waves per SIMD)
    // (entries move as one 64-bit scalar, link | entry distance << 32: a vector type would be copied through a
    // generic reference and bring the generic pointer back)
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    typedef __attribute__((address_space(1))) unsigned long long glb_u64;
    lds_u64* stk = (lds_u64*)lds + threadIdx.x;
    // (the deep levels' address is put together where it is used, rarely: two more VGPRs saved)
    const size_t deep_stride = (size_t)gridDim.x * bs;
    // (blockDim.x is a power of two: mirt.hip launches 64 .. 256 lanes per block)
    const uint32_t level_shift = 3u + (31u - (uint32_t)__builtin_clz(bs)), level_bytes = 1u << level_shift;
    const uint32_t lds_bytes = lds_levels << level_shift;
    auto stk_deep = [&](uint32_t level) -> glb_u64* {
        return (glb_u64*)spill + ((size_t)blockIdx.x * bs + threadIdx.x) + (size_t)level * deep_stride;
    };
    // (the lane index and a lane's rank inside a ballot come from v_mbcnt where they are needed: kept in registers
    // they cost three VGPRs of a budget of 72 at 7 waves per SIMD)
    auto lane_id = [&]() -> uint32_t { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); };
    auto rank_in = [&](unsigned long long m) -> uint32_t {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };
    const float INF = __builtin_inff();

    Audit au;
    uint32_t pool_next = 0, pool_end = 0;  // wave-uniform ray pool
    bool exhausted = false, first_chunk = true;
    const uint32_t wave_id = blockIdx.x * (bs >> 6) + (threadIdx.x >> 6), total_waves = gridDim.x * (bs >> 6);
    // VOTE loop: the wave's pool lives in LDS (end << 32 | next) so that the other waves of the block can take rays from
    // it: a wave draws RAY_CHUNK rays from the global queue with ONE atomic (same-address atomics of a whole chip are
    // served one per ~11 ns per 64-byte line, whoever asks: 64 rays per draw would cap a launch at 6 Grays/s), and the block consumes
    // them in pieces of at most 64, so that the end of a launch is balanced at that granularity, not at RAY_CHUNK.
    static_assert(MIRT_BLOCK >= 64 && MIRT_BLOCK <= 256 && (MIRT_BLOCK & (MIRT_BLOCK - 1)) == 0,
                  "the shared ray pools (s_pool, the steal loop's mask) are written for 1, 2 or 4 waves per block");
    __shared__ unsigned long long s_pool[4];
    __shared__ int s_dry;  // a wave of the block found the global queue empty
    const uint32_t wib = threadIdx.x >> 6, nwb = bs >> 6;
    bool dry = false;      // nothing left for this wave to take, anywhere
    if (FAST && total_waves * RAY_CHUNK >= n) exhausted = true;  // the static first chunks cover the batch: no draw can succeed
    if (FAST) {
        if ((threadIdx.x & 63u) == 0) {
            const uint32_t b0 = wave_id * RAY_CHUNK;
            s_pool[wib] = b0 < n ? ((unsigned long long)min(b0 + RAY_CHUNK, n) << 32) | b0 : 0ull;
        }
        if (threadIdx.x < 4 && threadIdx.x >= nwb) s_pool[threadIdx.x] = 0ull;
        if (threadIdx.x == 0) s_dry = 0;
        __syncthreads();
    }

    // per-lane traversal state. `link` is the current child link: >= 0 pair index (inner node),
    // < 0 leaf (~link = begin | count << 24); has = false: nothing left to visit.
    bool active = false, has = false;
    uint32_t idx = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
    int a = 0;
    float best = -1.f, thr = INF;  // thr: entry distances beyond it cannot beat the closest hit
    // FAST: entry distance of the box being visited. A fast-tree leaf is one triangle's own box, and the reference's
    // t = orthoDist / (n.d) of a sliver (an fp32 plane normal off by 1e-3) or of a grazing hit can lie IN FRONT of that
    // box by more than the prune slack; pruning against such a t would skip the triangle's twin in a tie, or a triangle
    // whose own t is as far off. accept() therefore prunes against max(t, entry distance of the accepted triangle's box).
    // The entry distances themselves are slab_miss_pair's one-FMA parameters, each off by up to 3/16 a_ray (far from
    // the origin that is much more than the prune slack: |o| = 300, 1/d = 13 gives 1e-3), so thr carries a_ray / 2.
    float cur_lb = 0.f;
    float xslack = 0.f;  // exact kernels: exact_prune_slack() of this ray
    int best_slot = -1, link = 0;
    uint32_t sp = 0;  // stack depth as the BYTE offset of the next free LDS level (level * blockDim * 8)
    int leaf_pos = 0, leaf_end = 0;  // triangles still to test in the current leaf
    int unproven = 0;                // FAST: boxes entered without proof so far
    float lim = 0.f;                 // ANYHIT: distance limit of this ray
    bool occluded = false;
    RayPre pre;
    pre.inv = mk3(0, 0, 0);
    pre.a_ray = 0.f;
    pre.fast = false;
    pre.near = false;
    RayPre2 pre2 = ray_pairs(o, d, pre);

    // does the box pass the reference test? Decided by the slab classifier when it can prove the
    // answer, by the reference arithmetic itself (box_test) otherwise: rarely taken, so whole waves
    // skip it. Rays outside the classifier's error model (pre.fast false) always take the exact test.
    auto resolve = [&](const float4 lo, const float4 hi, int code, float& lb) -> bool {
        if (COUNT) au.box++;
        if (!pre.fast || code == BOX_AMBIGUOUS) {
            if (COUNT) {
                au.walk++;
                if (pre.fast) au.amb++;
            }
            const BoxHit b = box_test(lo, hi, o, d);
            lb = b.lb;
            return b.hit;
        }
        if (COUNT && box_test(lo, hi, o, d).hit != (code == BOX_HIT)) au.bad++;
        return code == BOX_HIT;
    };
    auto enter = [&](int lk, float lb = 0.f) {  // start visiting child link lk, whose box the ray enters at lb
        link = lk;
        has = true;
        if (FAST) cur_lb = lb;
        const uint32_t v = ~(uint32_t)lk;  // leaf fields; meaningless (and unused) for an inner node
        leaf_pos = (int)(v & 0xFFFFFFu);
        leaf_end = leaf_pos + (int)(v >> 24);
    };
    auto pop = [&]() {  // next node from the stack; entries beyond the closest hit are dropped
        has = false;
        while (sp > 0) {
            sp -= level_bytes;
            // (two loads in two address spaces, on purpose: a select between the pointers would go through a
            // generic pointer, and its aperture test trips this compiler at higher occupancy targets)
            unsigned long long packed;
            if (sp < lds_bytes)
                packed = *(lds_u64*)((__attribute__((address_space(3))) char*)stk + sp);
            else
                packed = *stk_deep((sp - lds_bytes) >> level_shift);
            const uint2 e = make_uint2((uint32_t)packed, (uint32_t)(packed >> 32));
            if (__uint_as_float(e.y) > thr) continue;
            enter((int)e.x, __uint_as_float(e.y));
            break;
        }
    };
    auto accept = [&](float t, int s) {  // a legal candidate that beats the closest hit so far
        if (ANYHIT) {  // (only candidates nearer than the limit get here) one occluder is enough
            occluded = true;
            best_slot = s;  // (FAST: k_verify checks this occluder's chain of reference boxes)
            has = false;
            return;
        }
        best = t;
        best_slot = s;
        thr = FAST ? __builtin_fmaf(fmaxf(t, cur_lb), 1.0001f, 1e-5f + 0.5f * pre.a_ray) : t * 1.0001f + 1e-5f + xslack;
    };

    // The fast-tree kernels (incoherent rays, deep walks) take the vote loop; the exact kernels (camera rays on small
    // scenes, re-runs) keep the while-while form below, which measured faster for coherent rays (Cornell camera
    // rays 0.63 vs 0.67 ms).
    constexpr bool VOTE = FAST;
    if constexpr (VOTE) {
    // ---- one loop, one kind of work per iteration, chosen by majority ---------------------------------------
    // Each iteration the wave does EITHER one traversal step for the lanes standing on an inner node OR one leaf
    // pass (up to LEAF_BATCH triangles + the predicate) for the lanes standing on a leaf, whichever more lanes
    // want; the others wait one iteration. The while-while form this replaces let every lane descend until ALL
    // stood on a leaf: on the 1 M-triangle scene the slowest lane of a wave needs ~29 steps per descent against
    // a mean of 8, i.e. 27 % of the lanes did work in a traversal step (43 % on the Cornell scene).
    for (;;) {
        // ---- refill: lanes without a ray take the next rays of the wave's pool, once enough of them are idle ----
        const unsigned long long need = __ballot(!active);
        if (need && !dry && ((int)__popcll(need) >= MIRT_REFILL_MIN || !__any(active && has))) {
            const uint32_t want = (uint32_t)__popcll(need);
            const int leader = __ffsll((long long)need) - 1;
            uint32_t nb = 0, take = 0;  // the piece this wave gets: `take` rays from ray nb on
            // the wave's own pool first, then the other waves' of the block
            for (uint32_t k = 0; k < nwb && take == 0; k++) {
                const uint32_t j = (wib + k) & (nwb - 1);  // (1, 2 or 4 waves per block)
                const unsigned long long v = __hip_atomic_load(&s_pool[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((uint32_t)v < (uint32_t)(v >> 32)) {
                    unsigned long long got = 0;
                    if ((int)lane_id() == leader) got = atomicAdd(&s_pool[j], (unsigned long long)want);
                    const uint32_t b = __shfl((uint32_t)got, leader), e = __shfl((uint32_t)(got >> 32), leader);
                    if (b < e) {  // (else: emptied by another wave since the look at v)
                        nb = b;
                        take = min(e - b, want);
                    }
                }
            }
            if (take == 0) {  // all empty: draw from the global queue into the own pool, unless that is known to be empty
                if (!exhausted && __hip_atomic_load(&s_dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) exhausted = true;
                if (!exhausted) {
                    uint32_t base = 0;
                    if ((int)lane_id() == leader) base = atomicAdd(queue, RAY_CHUNK);
                    base = __shfl(base, leader) + total_waves * RAY_CHUNK;
                    if (base >= n) {
                        exhausted = true;
                        if ((int)lane_id() == leader) __hip_atomic_store(&s_dry, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        const uint32_t e = min(base + RAY_CHUNK, n);
                        nb = base;
                        take = min(e - base, want);
                        if ((int)lane_id() == leader)  // (only the owner writes a pool; the others look and add, atomically)
                            __hip_atomic_store(&s_pool[wib], ((unsigned long long)e << 32) | (base + take), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (take == 0) dry = true;  // (a wave of the block that publishes a pool after this keeps it to itself)
            }
            const uint32_t pool_next = nb;
            const uint32_t rank = rank_in(need);
            if (!active && rank < take) {
                idx = pool_next + rank;
#if MIRT_GUARD
                if (idx >= n) { printf("GUARD idx %u n %u nb %u take %u want %u\n", idx, n, nb, take, want); idx = 0; }
#endif
                if (ray_list) idx = ray_list[idx];
                o = mk3(O[idx], O[plane_stride + idx], O[2 * plane_stride + idx]);
                d = mk3(D[idx], D[plane_stride + idx], D[2 * plane_stride + idx]);
                a = 0;
                if (MODE != MIRT_MODE_UNCONDITIONAL) a = aux[idx];
                best = -1.f;
                thr = INF;
                if (ANYHIT) {
                    lim = limit[idx];
                    occluded = false;
                }
                best_slot = -1;
                unproven = 0;
                sp = 0;
                has = false;
                active = true;
                pre = ray_prepare(sc, o, d);
                pre2 = ray_pairs(o, d, pre);
                // nothing beyond the limit can occlude (+ the rounding of this kernel's entry distances, see cur_lb, and
                // the reach of the reference's edge signs in t, the exact kernels' exact_prune_slack: 2 a_ray for a ray
                // that takes this route)
                if (ANYHIT) thr = __builtin_fmaf(lim, 1.0001f, 1e-5f + 2.f * pre.a_ray);
                float lb;
                const int code = slab_classify(root_lo, root_hi, box_is_flat(root_lo, root_hi), o, pre, lb);
                if (FAST) {
                    if (COUNT) au.box++;
                    if (pre.fast && pre.near && code != BOX_MISS) enter(__float_as_int(root_lo.w));
                } else if (resolve(root_lo, root_hi, code, lb))
                    enter(__float_as_int(root_lo.w));
            }
        }
        if (!__any(active)) {
            if (dry) break;
            continue;
        }

        const bool want_node = active && has && link >= 0;
        const bool want_leaf = active && has && link < 0;
        const int n_node = (int)__popcll(__ballot(want_node)), n_leaf = (int)__popcll(__ballot(want_leaf));
        if (n_node > 0 && n_node * MIRT_VOTE_NODE_WEIGHT >= n_leaf * MIRT_VOTE_LEAF_WEIGHT) {
            // ---- one traversal step for the lanes on an inner node ------------------------------------------
            if (COUNT && lane_id() == 0) au.w_inner++;
#pragma unroll
            for (int rep = 0; rep < MIRT_NODE_REPS; rep++)
            if (active && has && link >= 0) {
#if MIRT_GUARD
                if (link >= max(sc.n_tris, 1)) { printf("GUARD link %d idx %u sp %u\n", link, idx, sp); link = 0; }
                if ((sp >> level_shift) > (uint32_t)sc.depth + 1) { printf("GUARD sp %u idx %u\n", sp, idx); sp = 0; }
#endif
                const PairRec P = pairs[link];
                float lbl, lbr;
                // conservative: only a proven miss is skipped (slab_miss_pair, mirt_dev.h)
                // (boxes entered on the margin alone are counted in the first step of a node pass only: a ray the
                // budget is for enters such boxes at every step, and the bookkeeping is ~9 instructions of ~60)
                bool ul, ur, vl, vr;
                if (COUNT || rep == 0) {
                    vl = !slab_miss_pair<true>(P.la, P.lb, pre2, pre, lbl, ul);
                    vr = !slab_miss_pair<true>(P.ra, P.rb, pre2, pre, lbr, ur);
                } else {
                    vl = !slab_miss_pair<false>(P.la, P.lb, pre2, pre, lbl, ul);
                    vr = !slab_miss_pair<false>(P.ra, P.rb, pre2, pre, lbr, ur);
                }
                if (COUNT) {  // audit: a reported miss that the reference test does not share must never happen
                    au.box += 2;
                    au.amb += (int)ul + (int)ur;
                    if (!vl && box_test(pair_lo(P.la, P.lb), pair_hi(P.la, P.lb), o, d).hit) au.bad++;
                    if (!vr && box_test(pair_lo(P.ra, P.rb), pair_hi(P.ra, P.rb), o, d).hit) au.bad++;
                }
                // A ray whose margin is wide (a nearly axis-parallel direction: |1/d| ~ 1e9) can prove almost no
                // miss and would walk the whole tree: after FAST_AMBIGUOUS_BUDGET boxes entered on the margin alone
                // it is given up and left to the exact kernel (slot -2 tells k_verify).
                if (COUNT || rep == 0) {
                    unproven += (int)ul + (int)ur;
                    if (unproven > FAST_AMBIGUOUS_BUDGET) {
                        vl = vr = false;
                        sp = 0;
                        best_slot = -2;
                    }
                }
                vl &= !(lbl > thr);
                vr &= !(lbr > thr);
                const int kl = __float_as_int(P.lb.z), kr = __float_as_int(P.rb.z);
                const bool pick_l = vl & (!vr | (lbl <= lbr));  // nearer child first
                if (vl & vr) {
                    const unsigned long long e = (unsigned long long)(uint32_t)(pick_l ? kr : kl) |
                                                 ((unsigned long long)__float_as_uint(pick_l ? lbr : lbl) << 32);
                    if (sp < lds_bytes)
                        *(lds_u64*)((__attribute__((address_space(3))) char*)stk + sp) = e;
                    else
                        *stk_deep((sp - lds_bytes) >> level_shift) = e;
                    sp += level_bytes;
                }
                if (vl | vr)
                    enter(pick_l ? kl : kr, pick_l ? lbl : lbr);
                else
                    pop();
            }
        } else if (n_leaf > 0) {
            // ---- one leaf pass: the fast tree's leaves are single triangles (build_fast_tree). The edge signs first
            // (tri_inside); for the lanes whose line passes through their triangle the distance (one division), the
            // tie rule and, in NO_SELF mode, the predicate. A lane whose next node is a leaf again takes part in the
            // next repetition of the pass. ------------------------------------------------------------------------
#pragma unroll
            for (int rep = 0; rep < (MODE == MIRT_MODE_NO_SELF ? MIRT_LEAF_REPS : MIRT_LEAF_REPS_COHERENT); rep++) {
                const bool on_leaf = active && has && link < 0;
                if (rep > 0 && !__any(on_leaf)) break;
                if (COUNT && lane_id() == 0) au.w_leaf++;
                int s = (int)(~(uint32_t)link & 0xFFFFFFu);
#if MIRT_GUARD
                if (on_leaf && s >= sc.n_tris) { printf("GUARD leaf slot %d link %d idx %u\n", s, link, idx); s = 0; }
#endif
                bool through = false;
                if (on_leaf) {
                    if (COUNT) au.tri++;
                    const TriRec* __restrict__ T = sc.tris + s;
                    through = tri_inside(T->x, T->y, T->z, pre2);
                }
                if (through) {
                    const TriRec T = sc.tris[s];
                    const float t = tri_plane(T.n, tri_v2(T), o, d);
                    bool hit = t >= 0.f;
                    if (ANYHIT && t - lim >= -1e-5f) hit = false;  // at or beyond the limit: no occluder
                    // (a tie goes to the larger rank, DevScene::tie_rank: the position of a fast-tree leaf says
                    // nothing about the reference's order of leaves and triangles)
                    bool better = best_slot < 0 || t < best;
                    if (hit && !better && t == best) better = sc.tie_rank[s] > sc.tie_rank[best_slot];
                    hit = hit && better;
                    if (MODE == MIRT_MODE_NO_SELF) {
                        if (hit) {
                            const VnRec V = sc.vns[s];
                            const f3 pos = o + d * t;
                            const float dn = noself_dn<BUMP>(sc, T, V, s, pos, d);
                            if (legal_noself(dn, __float_as_int(V.n0.w), unpack_contact(a), unpack_inside(a)))
                                accept(t, s);
                        }
                    } else if (hit && !(MODE == MIRT_MODE_LAST_TID && __float_as_int(sc.vns[s].n1.w) == a)) {
                        accept(t, s);
                    }
                }
                if (on_leaf && has) pop();  // (an any-hit ray that found its occluder has nothing left to visit)
            }
            if (COUNT && lane_id() == 0) au.w_outer++;
        }

        // ---- retire finished rays -----------------------------------------------------------------
        if (ANYHIT) {
            if (active && !has) {
                vis_out[idx] = occluded ? 0 : 1;
                // FAST: the occluder's slot, -1 for "nothing in the superset" (final), -2 for a ray given up
                if (FAST) id_out[idx] = best_slot;
                active = false;
            }
        } else if (active && !has) {
            t_out[idx] = best_slot >= 0 ? best : -1.f;
            if (SLOT_OUT || FAST)
                id_out[idx] = best_slot;
            else
                id_out[idx] = best_slot >= 0 ? __float_as_int(sc.vns[best_slot].n1.w) : -1;
            active = false;
        }
    }
    } else {
    // ---- the exact kernels (the reference tree, every box decided as the reference decides it): while-while ---------
    for (;;) {
        // ---- refill: lanes without a ray take the next rays of the wave's pool ---------------------
        const unsigned long long need = __ballot(!active);
        if (need) {
            if (pool_next == pool_end && !exhausted) {
                // a wave's FIRST chunk is its own (wave index), without touching the queue: a grid sized for
                // an upper bound of the count (wavefront bounces, the exact re-run of a short list) would
                // otherwise send thousands of atomics to one address just to learn there is nothing to do
                uint32_t base = wave_id * RAY_CHUNK;
                if (!first_chunk) {
                    const int leader = __ffsll((long long)need) - 1;
                    if ((int)lane_id() == leader) base = atomicAdd(queue, RAY_CHUNK);
                    base = __shfl(base, leader) + total_waves * RAY_CHUNK;
                }
                first_chunk = false;
                if (base >= n) {
                    exhausted = true;
                } else {
                    pool_next = base;
                    pool_end = min(base + RAY_CHUNK, n);
                }
            }
            const uint32_t take = min(pool_end - pool_next, (uint32_t)__popcll(need));
            const uint32_t rank = rank_in(need);
            if (!active && rank < take) {
                idx = pool_next + rank;
                if (ray_list) idx = ray_list[idx];
                o = mk3(O[idx], O[plane_stride + idx], O[2 * plane_stride + idx]);
                d = mk3(D[idx], D[plane_stride + idx], D[2 * plane_stride + idx]);
                a = 0;
                if (MODE != MIRT_MODE_UNCONDITIONAL) a = aux[idx];
                best = -1.f;
                thr = INF;
                if (ANYHIT) {
                    lim = limit[idx];
                    occluded = false;
                }
                best_slot = -1;
                unproven = 0;
                sp = 0;
                has = false;
                active = true;
                pre = ray_prepare(sc, o, d);
                pre2 = ray_pairs(o, d, pre);
                xslack = exact_prune_slack(sc, o, pre);
                if (ANYHIT) thr = lim * 1.0001f + 1e-5f + xslack;  // nothing beyond the limit can occlude
                float lb;
                const int code = slab_classify(root_lo, root_hi, box_is_flat(root_lo, root_hi), o, pre, lb);
                if (resolve(root_lo, root_hi, code, lb)) enter(__float_as_int(root_lo.w));
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
                if (lane_id() == 0) au.w_inner++;
                if (!go) continue;
            } else if (!go) {
                break;
            }
            const PairRec P = pairs[link];
            float lbl, lbr;
            const int cl = slab_classify_pair(P.la, P.lb, pre2, pre, lbl);
            const int cr = slab_classify_pair(P.ra, P.rb, pre2, pre, lbr);
            bool vl = cl == BOX_HIT, vr = cr == BOX_HIT;
            if (COUNT || !pre.fast || cl == BOX_AMBIGUOUS || cr == BOX_AMBIGUOUS) {
                vl = resolve(pair_lo(P.la, P.lb), pair_hi(P.la, P.lb), cl, lbl);
                vr = resolve(pair_lo(P.ra, P.rb), pair_hi(P.ra, P.rb), cr, lbr);
            }
            vl &= !(lbl > thr);
            vr &= !(lbr > thr);
            const int kl = __float_as_int(P.lb.z), kr = __float_as_int(P.rb.z);
            const bool pick_l = vl & (!vr | (lbl <= lbr));  // nearer child first
            if (vl & vr) {
                const unsigned long long e = (unsigned long long)(uint32_t)(pick_l ? kr : kl) |
                                             ((unsigned long long)__float_as_uint(pick_l ? lbr : lbl) << 32);
                if (sp < lds_bytes)
                    *(lds_u64*)((__attribute__((address_space(3))) char*)stk + sp) = e;
                else
                    *stk_deep((sp - lds_bytes) >> level_shift) = e;
                sp += level_bytes;
            }
            if (vl | vr)
                enter(pick_l ? kl : kr);
            else
                pop();
        }

        // ---- leaf phase: up to LEAF_BATCH triangles of this lane's leaf. In NO_SELF mode a lane
        // stops at its first candidate; the predicate (shading normal) is evaluated once per pass for
        // all such lanes together instead of inside the triangle loop. --------------------------------
        if (COUNT) {
            int trips = (active && has) ? min(LEAF_BATCH, leaf_end - leaf_pos) : 0;
            for (int off = 32; off > 0; off >>= 1) trips = max(trips, __shfl_xor(trips, off));
            if (lane_id() == 0) {
                au.w_leaf += trips;
                au.w_outer++;
            }
        }
        {
            const int stop = (active && has) ? min(leaf_pos + LEAF_BATCH, leaf_end) : 0;
            const int leaf_beg = (int)(~(uint32_t)link & 0xFFFFFFu);
            bool more = active && has;
            for (;;) {
                int cand = -1;
                float cand_t = 0.f;
                if (more) {
                    int s = leaf_pos;
                    for (; s < stop; s++) {
                        const TriRec T = sc.tris[s];
                        if (COUNT) au.tri++;
                        const float t = tri_test(T, o, d);
                        if (!(t >= 0.f)) continue;
                        if (ANYHIT && t - lim >= -1e-5f) continue;  // at or beyond the limit: no occluder
                        if (!(best_slot < 0 || t < best || (t == best && best_slot < leaf_beg))) continue;
                        if (MODE == MIRT_MODE_NO_SELF) {
                            cand = s;
                            cand_t = t;
                            s++;
                            break;
                        }
                        if (MODE == MIRT_MODE_LAST_TID && __float_as_int(sc.vns[s].n1.w) == a) continue;
                        accept(t, s);
                        if (ANYHIT && !has) break;
                    }
                    leaf_pos = s;
                }
                if (MODE != MIRT_MODE_NO_SELF) break;
                if (cand >= 0) {
                    const VnRec V = sc.vns[cand];
                    const f3 pos = o + d * cand_t;
                    const float dn = noself_dn<BUMP>(sc, sc.tris[cand], V, cand, pos, d);
                    if (legal_noself(dn, __float_as_int(V.n0.w), unpack_contact(a), unpack_inside(a)))
                        accept(cand_t, cand);
                }
                more = active && has && leaf_pos < stop;
                if (!__any(more)) break;
            }
        }
        if (active && has && leaf_pos >= leaf_end) pop();

        // ---- retire finished rays -----------------------------------------------------------------
        if (ANYHIT) {
            if (active && !has) {
                vis_out[idx] = occluded ? 0 : 1;
                active = false;
            }
        } else if (active && !has) {
            t_out[idx] = best_slot >= 0 ? best : -1.f;
            if (SLOT_OUT)
                id_out[idx] = best_slot;
            else
                id_out[idx] = best_slot >= 0 ? __float_as_int(sc.vns[best_slot].n1.w) : -1;
            active = false;
        }
    }

    }  // !VOTE

    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            au.box += __shfl_down(au.box, off);
            au.tri += __shfl_down(au.tri, off);
            au.amb += __shfl_down(au.amb, off);
            au.bad += __shfl_down(au.bad, off);
            au.walk += __shfl_down(au.walk, off);
        }
        if (lane_id() == 0) {
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

// Decides whether the (t, slot) k_intersect<FAST> left for a ray is the reference's answer. The fast
// kernel returns argmin t over a SUPERSET S of the reference's candidate set A (triangles passing the
// triangle test and the predicate whose leaf box and every ancestor box pass the reference box test):
// its conservative box test never skips a box the reference would enter. So: a miss in S is a miss;
// a winner whose whole chain of reference boxes (leaf up to the root) passes the reference test is in
// A and is therefore A's winner too (same total order: t, then the slot tie rule). Anything else, and
// every ray outside the classifier's error model, goes on the list the exact kernel re-runs.
// SLOT_OUT false: a confirmed slot is replaced by its global triangle id. counters[0] += box tests.
// CONNECT (the visibility query of PathTracerTest): a ray with no occluder in the superset is visible, finally: its
// pending light connection is added to its path's radiance here (ConnectArgs, mirt_dev.h).
template <bool SLOT_OUT, bool COUNT, bool CONNECT = false>
__global__ void __launch_bounds__(256)
k_verify(DevScene sc, const float* __restrict__ O, const float* __restrict__ D, uint32_t n_direct,
         const uint32_t* __restrict__ n_ptr, uint32_t plane_stride, const float* __restrict__ t_out,
         int* __restrict__ id_out, uint32_t* __restrict__ list, uint32_t* __restrict__ list_count,
         unsigned long long* __restrict__ counters, const uint32_t* __restrict__ off_ptr = nullptr,
         ConnectArgs ca = ConnectArgs{}) {
    const uint32_t n = n_ptr ? min(*n_ptr, n_direct) : n_direct;
    if (off_ptr) {  // (as in k_intersect; the list holds indices from there on, and k_rerun counts from there too)
        const uint32_t off = *off_ptr;
        O += off;
        D += off;
        t_out += off;
        id_out += off;
    }
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool redo = false;
    unsigned long long boxes = 0;
    if (i < n) {
        const f3 o = mk3(O[i], O[plane_stride + i], O[2 * plane_stride + i]);
        const f3 d = mk3(D[i], D[plane_stride + i], D[2 * plane_stride + i]);
        const RayPre pre = ray_prepare(sc, o, d);
        const f3 c = mk3(-o.x * pre.inv.x, -o.y * pre.inv.y, -o.z * pre.inv.z);
        int slot = id_out[i];
#if MIRT_GUARD
        if (slot >= sc.n_tris || slot < -2) { printf("GUARD verify slot %d ray %u\n", slot, (uint32_t)i); slot = -2; }
#endif
        if (!pre.fast || !pre.near) {  // outside the classifier's model, or too far out for the fast tree's padded leaves
            redo = true;
        } else if (slot == -2) {  // the fast kernel gave this ray up
            redo = true;
        } else if (slot >= 0) {
            int node = sc.slot_node[slot];
            // (the step bound can never bind: the host rejects trees deeper than MIRT_MAX_TREE_DEPTH and builds the
            // parent links itself; it is there so that no state of the data can keep a wave in this loop)
            for (int step = 0; node >= 0; step++) {
                if (step > MIRT_MAX_TREE_DEPTH + 1) {
                    redo = true;
                    break;
                }
                // (the record as four 16-byte vectors and the jump entry chosen by selects: `N.jump[face]` on a struct
                // copy made the compiler keep the six entries in LDS, written and read back with bank conflicts, which
                // was 12 % of this kernel's wave cycles)
                const float4* __restrict__ rec = reinterpret_cast<const float4*>(sc.rnodes + node);
                struct { float4 lo, hi; } N;
                N.lo = rec[0];
                N.hi = rec[1];
                const float4 j0 = rec[2], j1 = rec[3];  // jump[0..3], jump[4..5] + padding
                if (COUNT) boxes++;
                // most boxes of a chain are left through a face well clear of the box's edges, or are flat and crossed
                // well inside: proven in ~30 instructions; the rest takes the full classifier, then the reference test
                int axis;
                bool via_flat;
                if (slab_face_proven(N.lo, N.hi, __float_as_int(N.hi.w) != 0, c, pre, axis, via_flat)) {
                    if (COUNT && !box_test(N.lo, N.hi, o, d).hit)
                        atomicAdd(&counters[3], 1ull);  // audit: a proven exit face the reference test does not share (must be 0)
                    // the reference accepts this box through that face, hence every ancestor that shares the face's
                    // plane bit for bit and is no narrower on the other two axes (NodeRec::jump): go on above them
                    const float da = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
                    // jump[2 * axis + (1 for the max plane)]: the ray leaves through the max plane when it runs upwards
                    const float jlo = axis == 0 ? j0.x : (axis == 1 ? j0.z : j1.x);
                    const float jhi = axis == 0 ? j0.y : (axis == 1 ? j0.w : j1.y);
                    // (a flat box's accepted plane is its min and its max plane: either run of inheriting ancestors is
                    // proven, so the one that ends higher up - the smaller node index, -1 = above the root - is taken)
                    const int next = via_flat ? min(__float_as_int(jlo), __float_as_int(jhi)) : __float_as_int(da > 0.f ? jhi : jlo);
                    if (COUNT) {  // audit: every ancestor skipped must pass the reference test as well (must be 0)
                        for (int a2 = __float_as_int(N.lo.w); a2 >= 0 && a2 != next; a2 = __float_as_int(sc.rnodes[a2].lo.w))
                            if (!box_test(sc.rnodes[a2].lo, sc.rnodes[a2].hi, o, d).hit) atomicAdd(&counters[3], 1ull);
                    }
                    node = next;
                    continue;
                }
                float lb;
                const int code = slab_classify(N.lo, N.hi, __float_as_int(N.hi.w) != 0, o, pre, lb);
                if (code != BOX_HIT && !(code == BOX_AMBIGUOUS && box_test(N.lo, N.hi, o, d).hit)) {
                    redo = true;
                    break;
                }
                node = __float_as_int(N.lo.w);
            }
            if (!redo && !SLOT_OUT) id_out[i] = __float_as_int(sc.vns[slot].n1.w);
        } else {
            if (!SLOT_OUT) id_out[i] = -1;
            if (CONNECT) connect_add(ca, i);
        }
    }
    // wave-aggregated append
    const unsigned long long m = __ballot(redo);
    if (m) {
        const uint32_t lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(list_count, (uint32_t)__popcll(m));
        base = __shfl(base, leader);
        if (redo) list[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) boxes += __shfl_down(boxes, off);
        if ((threadIdx.x & 63) == 0 && boxes) atomicAdd(&counters[0], boxes);
    }
}

// The exact re-run of the rays k_verify listed, one WAVE per ray. The list is short (~1e-5 of a batch)
// but its rays are the slow ones: every box goes through the reference arithmetic (six IEEE divisions),
// and with one lane per ray the launch lasts as long as the slowest ray's ~200 dependent steps (0.42 ms
// measured on the 12.6 M ray Cornell batch for 157 rays). Here the 64 lanes test 64 boxes of ONE ray per
// step: a stack of pair links in LDS, up to 32 pairs popped per step, lane l tests child (l & 1) of pair
// (l >> 1), surviving inner children are pushed back, a lane whose child is a leaf tests its triangles.
// No distance pruning: the answer is the plain definition, argmin over { legal triangle hits whose every
// ancestor box passes box_test } in the order (t, later leaf link, earlier slot) of the header comment,
// so the visit order is free.
// Stack bound: batch steps (pop k <= 32, push <= 2k) only while sp <= CAP - 192, single-pair depth-first
// steps above that, which add at most depth + 1 (<= MIRT_MAX_TREE_DEPTH + 1) entries.
constexpr int RERUN_STACK = 2048;
// Longest list k_rerun takes. Measured on the Cornell batch: 157 rays take 39 us wave-wide and 424 us with one lane per
// ray (the slowest ray's ~200 dependent exact box tests); the wave-wide time grows with count / 1024 waves, the other
// barely, and they cross near 10^4 rays.
constexpr uint32_t RERUN_WIDE_MAX = 8192;
// ANYHIT: the visibility query's re-run (launch_visibility): vis_out = 1 unless the closest legal hit lies nearer than
// limit by more than EPSILON, which is "some legal hit does" because the rounded difference is monotone in t.
template <bool SLOT_OUT, int MODE = MIRT_MODE_NO_SELF, bool ANYHIT = false>
__global__ void __launch_bounds__(64)
k_rerun(DevScene sc, const float* __restrict__ O, const float* __restrict__ D, const int* __restrict__ aux,
        uint32_t plane_stride, float* __restrict__ t_out, int* __restrict__ id_out,
        const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count, uint32_t list_max,
        const uint32_t* __restrict__ off_ptr = nullptr, const float* __restrict__ limit = nullptr,
        unsigned char* __restrict__ vis_out = nullptr) {
    __shared__ int stack[RERUN_STACK];
    if (off_ptr) {
        const uint32_t off = *off_ptr;
        O += off;
        D += off;
        if (aux) aux += off;
        if (t_out) t_out += off;
        if (id_out) id_out += off;
    }
    const uint32_t lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t count = *list_count;
    // one wave per ray pays for short lists only; a long one (a batch of axis-parallel rays, which the classifier's
    // model excludes wholesale) goes to the lane-per-ray exact kernel launched next (launch_intersect)
    if (count > list_max) return;
    for (uint32_t r = blockIdx.x; r < count; r += gridDim.x) {
        const uint32_t idx = list[r];
        const f3 o = mk3(O[idx], O[plane_stride + idx], O[2 * plane_stride + idx]);
        const f3 d = mk3(D[idx], D[plane_stride + idx], D[2 * plane_stride + idx]);
        const int a = MODE != MIRT_MODE_UNCONDITIONAL ? aux[idx] : 0;
        const int contact = unpack_contact(a), inside = unpack_inside(a);
        float best = 0.f;
        int best_slot = -1, best_leaf = -1;
        int sp = 0;
        int my_leaf = 0;  // leaf link this lane has to test (0: none; leaf links are negative)
        if (box_test(sc.root_lo, sc.root_hi, o, d).hit) {
            const int lk = __float_as_int(sc.root_lo.w);
            if (lk >= 0) {
                if (lane == 0) stack[0] = lk;
                sp = 1;
            } else if (lane == 0) {
                my_leaf = lk;
            }
        }
        __syncthreads();
        for (;;) {
            // ---- leaves found in the previous step (or the root leaf) --------------------------------
            if (my_leaf < 0) {
                const uint32_t v = ~(uint32_t)my_leaf;
                const int beg = (int)(v & 0xFFFFFFu), end = beg + (int)(v >> 24);
                for (int s = beg; s < end; s++) {
                    const TriRec T = sc.tris[s];
                    const float t = tri_test(T, o, d);
                    if (!(t >= 0.f)) continue;
                    if (!(best_slot < 0 || t < best || (t == best && best_leaf < beg))) continue;
                    if (MODE == MIRT_MODE_NO_SELF) {
                        const VnRec V = sc.vns[s];
                        const f3 nn = shading_normal<false>(sc, T, V, s, o + d * t);
                        if (!legal_noself(dot3(d, nn), __float_as_int(V.n0.w), contact, inside)) continue;
                    } else if (MODE == MIRT_MODE_LAST_TID) {
                        if (__float_as_int(sc.vns[s].n1.w) == a) continue;
                    }
                    best = t;
                    best_slot = s;
                    best_leaf = beg;
                }
                my_leaf = 0;
            }
            if (sp == 0) break;
            // ---- one step: k pairs off the top of the stack, one child box per lane ---------------------
            const int k = sp > RERUN_STACK - 192 ? 1 : min(32, sp);
            const bool mine = (int)(lane >> 1) < k;
            bool hit = false;
            int lk = 0;
            if (mine) {
                const PairRec P = sc.pairs[stack[sp - 1 - (int)(lane >> 1)]];
                const float4 ca = (lane & 1) ? P.ra : P.la, cb = (lane & 1) ? P.rb : P.lb;
                hit = box_test(pair_lo(ca, cb), pair_hi(ca, cb), o, d).hit;
                lk = __float_as_int(cb.z);
            }
            __syncthreads();  // every lane has read its stack entry
            sp -= k;
            const bool inner = hit && lk >= 0;
            const unsigned long long m = __ballot(inner);
            if (inner) {
                const int pos = sp + __popcll(m & lt_mask);
                if (pos < RERUN_STACK) stack[pos] = lk;  // (cannot fail: see the bound above)
            }
            sp = min(sp + (int)__popcll(m), RERUN_STACK);
            if (hit && lk < 0) my_leaf = lk;
            __syncthreads();
        }
        // ---- the wave's winner: smallest t, then the later leaf link, then the earlier slot ---------------
        for (int off = 32; off > 0; off >>= 1) {
            const float ot = __shfl_xor(best, off);
            const int os = __shfl_xor(best_slot, off), ol = __shfl_xor(best_leaf, off);
            const bool take = os >= 0 && (best_slot < 0 || ot < best ||
                                          (ot == best && (ol > best_leaf || (ol == best_leaf && os < best_slot))));
            if (take) {
                best = ot;
                best_slot = os;
                best_leaf = ol;
            }
        }
        if (ANYHIT) {
            if (lane == 0) vis_out[idx] = (best_slot >= 0 && best - limit[idx] < -1e-5f) ? 0 : 1;
        } else if (lane == 0) {
            t_out[idx] = best_slot >= 0 ? best : -1.f;
            if (SLOT_OUT)
                id_out[idx] = best_slot;
            else
                id_out[idx] = best_slot >= 0 ? __float_as_int(sc.vns[best_slot].n1.w) : -1;
        }
    }
}

// every visible shadow ray's light connection (the visibility routes without k_verify: no fast tree, bump textures)
__global__ void __launch_bounds__(256)
k_connect_all(ConnectArgs ca, const unsigned char* __restrict__ vis, const uint32_t* __restrict__ n_ptr, uint32_t n_max) {
    const uint32_t n = n_ptr ? min(*n_ptr, n_max) : n_max;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && vis[j]) connect_add(ca, j);
}

// the light connections of the shadow rays the visibility query re-ran (their answers are in vis now)
__global__ void __launch_bounds__(256)
k_connect_list(ConnectArgs ca, const unsigned char* __restrict__ vis, const uint32_t* __restrict__ list,
               const uint32_t* __restrict__ list_count) {
    const uint32_t count = *list_count;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < count; r += gridDim.x * blockDim.x) {
        const uint32_t j = list[r];
        if (vis[j]) connect_add(ca, j);
    }
}

}  // namespace mirt
