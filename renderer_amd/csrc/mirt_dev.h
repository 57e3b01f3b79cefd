// mirt_dev.h — device-side data layout and exact-arithmetic geometry helpers (gfx950 only).
//
// Everything that decides WHICH triangle a ray hits is evaluated with the same operations, in the
// same order and with the same roundings as the reference CPU path (KDTree.cpp:34-52, 267-293;
// NoSelfIntersectionCondition.cpp:12-27; SimpleShape.cpp:300-339): fp32, one rounding per
// multiply and add (no contraction to FMA), correctly rounded division and square root (hipcc's
// default for HIP). This file is compiled with -ffp-contract=off and repeats the request below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mirt.h"

#pragma clang fp contract(off)

namespace mirt {

// ---- HBM layout ------------------------------------------------------------------------------
// reference-tree node record, 64 B, pre-order: what k_verify walks from a leaf up to the root
//   lo = (min.x, min.y, min.z, bits(parent index, -1 for the root))   hi = (max.x, max.y, max.z, bits(flat))
//   jump[face], face = 2 * axis + (1 for the max plane): the nearest ancestor that does NOT inherit this node's
//   acceptance through that face, -1 when every ancestor up to the root does. An ancestor inherits it when its plane on
//   that face is this node's plane bit for bit and its bounds on the other two axes contain this node's: the reference
//   test (KDTree.cpp:40-49) then computes the very same t = (plane - o) / d and the very same point o + t d for that
//   face, and a point inside the node's other two slabs (inclusive compares) is inside the wider ones. So "the
//   reference accepts this node through face F" implies, exactly and with no margin, that it accepts every ancestor
//   below jump[F]: k_verify continues there instead of at the parent (built on the host from the caller's boxes with
//   both conditions checked, never assumed).
struct NodeRec {
    float4 lo, hi;
    int jump[6];
    int pad[2];
};
// traversal record, 64 B: the two children of one inner node, in BREADTH-FIRST order of the inner
// nodes (the top of the tree is contiguous). Per child two float4, with each axis' two planes side
// by side so that the slab parameters come out of packed fp32 instructions (v_pk_add_f32 /
// v_pk_mul_f32 on the register pair a dwordx4 load leaves behind, no moves):
//   a = (min.x, max.x, min.y, max.y)    b = (min.z, max.z, bits link, bits flat)
//   flat != 0: the box has zero extent along at least one axis (slab_classify)
//   link >= 0: index of the child's own PairRec (inner node)
//   link <  0: leaf, ~link = first slot | (triangle count << 24)
struct PairRec {
    float4 la, lb, ra, rb;
};
// the (min.xyz, link) / (max.xyz, flat) view of one child, as box_test and the root record use it
__host__ __device__ __forceinline__ float4 pair_lo(const float4 a, const float4 b) { return make_float4(a.x, a.z, b.x, b.z); }
__host__ __device__ __forceinline__ float4 pair_hi(const float4 a, const float4 b) { return make_float4(a.y, a.w, b.y, b.w); }
__host__ __device__ __forceinline__ void pair_pack(const float4 lo, const float4 hi, float4* a, float4* b) {
    *a = make_float4(lo.x, hi.x, lo.y, hi.y);
    *b = make_float4(lo.z, hi.z, lo.w, hi.w);
}
// triangle record in SLOT (leaf) order, 64 B (one per half cache line). Per axis one float4 that holds the
// corner coordinates as the two register pairs the packed edge arithmetic of tri_inside() reads (edge 0 and
// edge 1 side by side in v_pk_mul_f32 / v_pk_add_f32, no moves), then the unit plane normal the reference
// recomputes per test (KDTree.cpp:283-284), precomputed on the host with identical arithmetic:
//   x = (v0.x, v1.x, v1.x, v2.x)   y, z likewise   n = (n.x, n.y, n.z, 0)
struct TriRec {
    float4 x, y, z, n;
};
__host__ __device__ __forceinline__ TriRec tri_pack(const float v0[3], const float v1[3], const float v2[3], const float n[3]) {
    TriRec T;
    T.x = make_float4(v0[0], v1[0], v1[0], v2[0]);
    T.y = make_float4(v0[1], v1[1], v1[1], v2[1]);
    T.z = make_float4(v0[2], v1[2], v1[2], v2[2]);
    T.n = make_float4(n[0], n[1], n[2], 0.f);
    return T;
}
// per-slot shading data, 64 B, touched only for candidate hits that would win (predicate) and in
// the shade kernel: corner normals (normal-matrix transformed, not normalised); for a triangle of an
// object without vertex normals n0.xyz is its face normal normalize((v1-v0) x (v2-v0))
// (SimpleShape.cpp:300-339), rounded on the host exactly as the reference rounds it per query.
// w: the terms of the reference's barycentric weights that depend on the triangle alone, computed on the host
// with the reference's operations in the reference's order (world_normal): with b1 = v1 - v0, b2 = v2 - v0,
//   w = (l1 = |b1|, d12 = b1 . b2, d12 * d12 - l1 * l1 * l2 * l2, l1 * l1)
struct VnRec {
    float4 n0, n1, n2;  // .w: n0 bits(object id), n1 bits(global tid), n2 bits(object has vertex normals)
    float4 w;
};

// corner texture coordinates of one slot (SimpleShape::vertexTexCoordList through the face's indices)
struct VtRec {
    float4 t0, t1, t2;  // xyz used
};
// one texture (Texture.h:14-42): texels in HBM, the first three rows of each matrix
struct DevTexture {
    const float* texels;
    int width, height, channels, pad;
    float coord[12];  // coordTransform rows 0..2, row-major
    float color[12];  // colorTransform rows 0..2
};

struct DevObject {
    int type, has_vn, volumetric, pad;
    float color[3];
    float coeff;
    float ior;
    float decay[3];
    float sigma_s[3];  // homogeneous medium (SceneVPMObject.h:13-14): ds, dt = ds + da, HG g
    float sigma_t[3];
    float g;
    float color2[3];   // MIRT_OBJ_PHONG: glossy colour (color = diffuse)
    float emit_pdf_a;  // emissive objects: weight / totalArea of the emission sampler (SceneObject.cpp:11-21,
                       // Scene.cpp:19-33), the area density PathTracerTest.cpp:38 uses for MIS
    int tex_diffuse, tex_gcolor, tex_gcoeff, tex_pad;  // MirtObjectTex: indices into DevScene.textures, -1 none
};
// emission sampler tables (Scene::preprocessEmissionSampler, Scene.cpp:193-204): one record per
// emissive object in scene order; its triangles' running areas live in DevScene.light_area / the
// slots of those triangles in DevScene.light_slot at [first, first + count)
struct DevLight {
    float weight_sum;   // running sum of the un-normalised weights up to and including this object
    float total_area;   // SceneObject::totalArea
    float origin_prob;  // weight / totalArea after normalizeEmissionWeight (SceneObject.cpp:156)
    int first, count, object;
    int pad[2];
};
// One volumetric object's own tree (Scene::objKDTrees, Scene.cpp:287-320) for Scene::findInsideObject: its nodes in
// PRE-ORDER with an escape index, walked without a stack (pt_kernels.h trace_object): these trees are small (a fog
// cube has 12 triangles) and are queried from inside the shade kernel, where a per-lane stack meant 528 B of scratch
// memory per lane for every path of a scene that holds a medium.
//   lo = (min.xyz, bits(escape)): the node that follows this node's whole subtree (VolRoot::first + count at the end)
//   hi = (max.xyz, 0); leaf_begin / leaf_end: the leaf's triangle SLOTS (behind the scene's own), -1 for an inner node
//   parent: index of the parent node in DevScene.vnodes (-1 for the root); flat: the box has zero extent along an axis
// A SMALL tree (VolRoot::small: at most MIRT_VOL_SMALL triangles) is not walked at all: pt_kernels.h trace_small tests
// every triangle and then the chain of boxes above the few that are hit.
struct VolNode {
    float4 lo, hi;
    int leaf_begin, leaf_end, parent, flat;
};
struct VolRoot {
    int object, first, count, small;  // nodes [first, first + count) of DevScene.vnodes
};
#ifndef MIRT_VOL_SMALL
#define MIRT_VOL_SMALL 32
#endif

// PathTracerTest's light connections (pt_kernels.h k_shade<NEE>): the pending contribution c[j] of shadow ray j is added
// to its path's radiance L[pid] when the ray turns out unoccluded. The visibility query's own kernels do that where
// they learn the answer (k_verify for a ray with no occluder in the fast tree's superset, k_connect_list for the few
// rays that were re-run), so the shadow rays are not streamed through a kernel of their own once more.
struct ConnectArgs {
    const float* c;       // [3][c_stride] pending contributions
    uint32_t c_stride;
    const uint32_t* pid;  // path id of the shadow ray; the bits from gen_shift up select the buffer of the L ring
    float* L;             // [gens][3][L_stride]
    uint32_t L_stride, pid_mask, gen_shift;
};
__device__ __forceinline__ void connect_add(const ConnectArgs& a, uint32_t j) {
    const uint32_t p = a.pid[j];
    float* L = a.L + (size_t)(p >> a.gen_shift) * 3u * (size_t)a.L_stride;
    const uint32_t px = p & a.pid_mask;
    L[px] += a.c[j];
    L[a.L_stride + px] += a.c[a.c_stride + j];
    L[2 * a.L_stride + px] += a.c[2 * a.c_stride + j];
}

struct DevScene {
    const PairRec* pairs;  // breadth-first child pairs: the classifier-path traversal
    float4 root_lo, root_hi;  // root box; root_lo.w = bits(link of the root)
    const TriRec* tris;
    const VnRec* vns;
    const DevObject* objects;
    const VolRoot* vols;  // per-object trees of the volumetric objects, ascending object index
    const VolNode* vnodes;
    int n_vols;
    int n_nodes, n_tris, n_objects, depth;
    float bound[3];       // max |coordinate| of the root box per axis
    // fast tree (mirt.hip build_fast_tree): child pairs of a surface-area hierarchy over the reference
    // tree's leaves, walked by k_intersect<FAST>; rnodes / slot_node let k_verify prove the winner's
    // chain of reference boxes afterwards
    const PairRec* fpairs;
    float4 froot_lo, froot_hi;
    const NodeRec* rnodes;
    const int* slot_node;      // reference leaf node of every slot of the scene tree
    // distance ties in the fast-tree kernels, whose leaves are single triangles: of two hits at the same distance the
    // one with the larger rank is the reference's answer (between leaves the later one, KDTree.cpp:319-320; inside
    // a leaf the earlier triangle, KDTree.cpp:305): rank = leaf begin + (leaf end - 1 - position in the leaf)
    const int* tie_rank;
    const DevLight* lights;    // emission sampler (PathTracerTest's light connections)
    const float* light_area;   // running triangle areas per light object (SceneObject::areaValues)
    const int* light_slot;     // slot of each light triangle
    int n_lights;
    float light_total_weight;  // SurfaceSampler::totalWeight
    const VtRec* vts;          // per slot, or nullptr when the scene has no texture coordinates
    const DevTexture* textures;
    const struct DevBump* bumps;  // bump textures of the objects that have one (VnRec.n2.w bits 8..)
    int has_bump;
    int fast_ok;          // node bounds finite, < 2^60, extent per axis 0 or >= 2^-36: the slab classifier may be used
    float far_o;          // 4 x the largest |coordinate| of the root box: rays that start farther out than this are not
                          // walked through the fast tree (RayPre::near; tri_fast_box in mirt.hip sizes its pad for this)
};

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 neg(f3 a) { return f3{-a.x, -a.y, -a.z}; }
// nvVector.h:501-507 order: ((x*x) + y*y) + z*z, each product and sum rounded
__device__ __forceinline__ float dot3(f3 a, f3 b) {
    float r = a.x * b.x;
    r = r + a.y * b.y;
    r = r + a.z * b.z;
    return r;
}
__device__ __forceinline__ f3 cross3(f3 l, f3 r) {
    return f3{l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x};
}
__device__ __forceinline__ float len3(f3 a) { return sqrtf(dot3(a, a)); }
// nvVector.h:376-380, 522-526: zero vectors (and NaN lengths) are left alone
__device__ __forceinline__ f3 normalize3(f3 a) {
    float r = len3(a);
    if (r > 0.f) return f3{a.x / r, a.y / r, a.z / r};
    return a;
}
__device__ __forceinline__ f3 div3(f3 a, float d) {
    if (d == 0.f) return a;
    return f3{a.x / d, a.y / d, a.z / d};
}
__device__ __forceinline__ float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// ---- node box test ---------------------------------------------------------------------------
// `hit` is exactly KDTree::Node::BoundingBox::intersect(ray) >= 0 (KDTree.cpp:34-52); `lb` is a
// lower bound of the parametric distance of any point of the box (slab entry, from the same six
// quotients), used only to skip subtrees that cannot beat the current closest hit.
struct BoxHit {
    bool hit;
    float lb;
};
__device__ __forceinline__ BoxHit box_test(const float4 lo, const float4 hi, f3 o, f3 d) {
    bool hit = false;
    float lb = 0.f;
    // axis x: planes lo.x / hi.x, other axes y, z
    if (d.x != 0.f) {
        float t0 = (lo.x - o.x) / d.x;
        float t1 = (hi.x - o.x) / d.x;
        float py = o.y + t0 * d.y, pz = o.z + t0 * d.z;
        hit |= (t0 >= 0.f) & (lo.y <= py) & (hi.y >= py) & (lo.z <= pz) & (hi.z >= pz);
        py = o.y + t1 * d.y;
        pz = o.z + t1 * d.z;
        hit |= (t1 >= 0.f) & (lo.y <= py) & (hi.y >= py) & (lo.z <= pz) & (hi.z >= pz);
        float nr = t0 < t1 ? t0 : t1;
        lb = nr > lb ? nr : lb;
    }
    if (d.y != 0.f) {
        float t0 = (lo.y - o.y) / d.y;
        float t1 = (hi.y - o.y) / d.y;
        float pz = o.z + t0 * d.z, px = o.x + t0 * d.x;
        hit |= (t0 >= 0.f) & (lo.z <= pz) & (hi.z >= pz) & (lo.x <= px) & (hi.x >= px);
        pz = o.z + t1 * d.z;
        px = o.x + t1 * d.x;
        hit |= (t1 >= 0.f) & (lo.z <= pz) & (hi.z >= pz) & (lo.x <= px) & (hi.x >= px);
        float nr = t0 < t1 ? t0 : t1;
        lb = nr > lb ? nr : lb;
    }
    if (d.z != 0.f) {
        float t0 = (lo.z - o.z) / d.z;
        float t1 = (hi.z - o.z) / d.z;
        float px = o.x + t0 * d.x, py = o.y + t0 * d.y;
        hit |= (t0 >= 0.f) & (lo.x <= px) & (hi.x >= px) & (lo.y <= py) & (hi.y >= py);
        px = o.x + t1 * d.x;
        py = o.y + t1 * d.y;
        hit |= (t1 >= 0.f) & (lo.x <= px) & (hi.x >= px) & (lo.y <= py) & (hi.y >= py);
        float nr = t0 < t1 ? t0 : t1;
        lb = nr > lb ? nr : lb;
    }
    return BoxHit{hit, lb};
}

// ---- slab classifier -------------------------------------------------------------------------
// The reference box test costs six correctly rounded divisions and thirty compares. In exact
// arithmetic it is the plain "ray meets closed box, t >= 0" predicate; in fp32 it can differ from
// that only when two of the slab parameters (three entry values, three exit values) or an entry /
// exit value and the box's other faces nearly coincide. slab_classify() evaluates the slab
// parameters with per-ray reciprocals and returns
//     BOX_MISS / BOX_HIT  when the reference's fp32 result is PROVABLY that (all deciding gaps
//                         exceed the rounding bound M derived in DESIGN.md "slab classifier"),
//     BOX_AMBIGUOUS       otherwise; the traversal then descends anyway and remembers that the path
//                         holds an unproven box: a triangle found below it is accepted only after the
//                         exact reference test has passed on every ancestor (rare).
// Signs need no margin: t = RN(P - o) * y has exactly the sign of the reference's quotient.
enum { BOX_MISS = 0, BOX_HIT = 1, BOX_AMBIGUOUS = 2 };

struct RayPre {
    f3 inv;       // RN(1/d) per axis
    float a_ray;  // absolute part of the margin: 16 eps * max_a (|o_a| + bound_a) * |inv_a|
    bool fast;    // the classifier's error model holds for this ray
    bool near;    // the origin lies within DevScene::far_o of the coordinate origin. The fast tree's single-triangle
                  // boxes are padded by the reach of the reference's edge-sign rounding (mirt.hip tri_fast_box), which
                  // grows with |v - o|: the pad is sized for origins within 4 x the scene's bound, so only such rays may
                  // take the fast route (k_intersect<FAST> skips the others, k_verify lists them for the exact re-run).
                  // The exact kernels do not look at it: the classifier's margin a_ray scales with |o| by itself.
};

__device__ __forceinline__ bool in_range(float v, float lo, float hi) {
    float a = fabsf(v);
    return a >= lo && a < hi;
}

__device__ __forceinline__ RayPre ray_prepare(const DevScene& sc, f3 o, f3 d) {
    RayPre p;
    const float lo_d = 9.094947e-13f /*2^-40*/, hi_d = 1.0995116e12f /*2^40*/;
    const float lo_o = 1.4551915e-11f /*2^-36*/, hi_o = 1.1529215e18f /*2^60*/;
    p.fast = sc.fast_ok != 0 && in_range(d.x, lo_d, hi_d) && in_range(d.y, lo_d, hi_d) &&
             in_range(d.z, lo_d, hi_d) && (o.x == 0.f || in_range(o.x, lo_o, hi_o)) &&
             (o.y == 0.f || in_range(o.y, lo_o, hi_o)) && (o.z == 0.f || in_range(o.z, lo_o, hi_o));
    p.near = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))) <= sc.far_o;  // (false for NaN)
    p.inv = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
    float ax = (fabsf(o.x) + sc.bound[0]) * fabsf(p.inv.x);
    float ay = (fabsf(o.y) + sc.bound[1]) * fabsf(p.inv.y);
    float az = (fabsf(o.z) + sc.bound[2]) * fabsf(p.inv.z);
    p.a_ray = 9.5367431640625e-07f /*16 * 2^-24*/ * fmaxf(ax, fmaxf(ay, az));
    return p;
}

// Absolute slack of the EXACT kernels' prune (k_intersect's while-while loop). The reference prunes nothing; these
// kernels skip a box that starts beyond the closest hit's t. The reference's edge signs (KDTree.cpp:267-289) accept a
// line that passes a triangle's edge on the wrong side by ~16 eps |v - o|, so an accepted hit can lie that far outside
// the triangle, hence outside the leaf boxes that list it; along an axis the ray barely moves on (|1/d_a| large) that
// is 16 eps |v - o| |1/d_a| in t: a ray with d_z = 3e-6 aimed at a triangle's corner had its hit 0.3 in front of the
// leaf's box (found by scripts/fuzz_campaign.py in the visibility query, whose limit sits right behind the hit). This
// is a_ray without the axes the ray does not move along at all (1/d_a infinite: there the box test is a containment
// test and no t depends on it), doubled because |v - o| is a norm where a_ray takes one axis.
__device__ __forceinline__ float exact_prune_slack(const DevScene& sc, f3 o, const RayPre& p) {
    const float INF = __builtin_inff();
    const float ix = fabsf(p.inv.x), iy = fabsf(p.inv.y), iz = fabsf(p.inv.z);
    const float ax = ix < INF ? (fabsf(o.x) + sc.bound[0]) * ix : 0.f;
    const float ay = iy < INF ? (fabsf(o.y) + sc.bound[1]) * iy : 0.f;
    const float az = iz < INF ? (fabsf(o.z) + sc.bound[2]) * iz : 0.f;
    return 1.9073486328125e-06f /*32 * 2^-24*/ * fmaxf(ax, fmaxf(ay, az));
}

__device__ __forceinline__ int slab_classify(const float4 lo, const float4 hi, bool flat, f3 o,
                                             const RayPre& p, float& lb) {
    const float t0x = (lo.x - o.x) * p.inv.x, t1x = (hi.x - o.x) * p.inv.x;
    const float t0y = (lo.y - o.y) * p.inv.y, t1y = (hi.y - o.y) * p.inv.y;
    const float t0z = (lo.z - o.z) * p.inv.z, t1z = (hi.z - o.z) * p.inv.z;
    const float nx = fminf(t0x, t1x), fx = fmaxf(t0x, t1x);
    const float ny = fminf(t0y, t1y), fy = fmaxf(t0y, t1y);
    const float nz = fminf(t0z, t1z), fz = fmaxf(t0z, t1z);
    const float tn = fmaxf(nx, fmaxf(ny, nz)), n2 = __builtin_amdgcn_fmed3f(nx, ny, nz);
    const float tf = fminf(fx, fminf(fy, fz)), f2 = __builtin_amdgcn_fmed3f(fx, fy, fz);
    // one fused multiply-add: never below the two-rounding value the derivation assumes
    const float M = __builtin_fmaf(1.9073486e-06f /*32 * 2^-24*/, fabsf(tn) + fabsf(tf), p.a_ray);
    const float gap = tf - tn, e_top = tn - n2, e_bot = f2 - tf;
    lb = fmaxf(tn, 0.f);
    // a box of zero extent along an axis (an axis-aligned wall): that axis yields the same entry and
    // exit value, bit for bit, so tn == tf means the other two slabs contain it and the two never cross
    const bool on_flat = flat & (tn == tf);
    const bool ok_gap = (gap > M) | on_flat;
    // the face that decides the hit: the exit face, or (origin outside) the entry face as well
    // (on a flat box the one face is entry and exit at once: it must be clear of the other axes' entries AND exits)
    const float e = on_flat ? fminf(e_top, e_bot) : (tn >= 0.f ? fmaxf(e_top, e_bot) : e_bot);
    const bool hit = ok_gap & (tf >= 0.f) & (e > M);
    const bool miss = (tf < -M) | (fminf(fminf(-gap, e_top), e_bot) > M) | (on_flat & (tf < 0.f));
    return hit ? BOX_HIT : (miss ? BOX_MISS : BOX_AMBIGUOUS);
}
// the same decision on a PairRec child (planes of one axis adjacent): identical operations and
// roundings, two at a time
typedef float v2f __attribute__((ext_vector_type(2)));
// Packed fp32 with one half of a register pair broadcast to both result halves (VOP3P op_sel / op_sel_hi). Written
// as instructions because this compiler does not fold a splat into op_sel: it builds the splat as a second register
// pair (two moves and two more live registers per value, which at 7 waves per SIMD end up in scratch). Each is one
// IEEE operation per half, rounded as the scalar instruction rounds (no contraction across them).
//   pk_sub_b<H>(a, b) = a - b[H]     pk_mul_b<H>(a, b) = a * b[H]     pk_fma_b<HB, HC>(a, b, c) = fma(a, b[HB], c[HC])
template <int H>
__device__ __forceinline__ v2f pk_sub_b(v2f a, v2f b) {
    v2f r;
    if constexpr (H == 0)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int H>
__device__ __forceinline__ v2f pk_mul_b(v2f a, v2f b) {
    v2f r;
    if constexpr (H == 0)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    else
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int HB, int HC>
__device__ __forceinline__ v2f pk_fma_b(v2f a, v2f b, v2f c) {
    v2f r;
    if constexpr (HB == 0 && HC == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if constexpr (HB == 1 && HC == 1)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if constexpr (HB == 0 && HC == 1)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// min / max of values that come out of the instructions above: written as instructions too, because the compiler
// puts a canonicalising v_max_f32 x, x in front of every fminf / fmaxf whose operand it cannot prove canonical
// (twelve per traversal step). No NaN reaches them: rays with a zero or denormal direction component are outside
// the classifier's error model (RayPre::fast) and never enter the fast kernels' traversal.
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
struct RayPre2 {
    v2f oxy, ixy, ozi;  // origin / reciprocal direction as register pairs (halves broadcast by op_sel); ozi = (o.z, 1/d.z)
    v2f cxy;            // slab_miss_pair: (-o.x / d.x, -o.y / d.y) as RN(-o * RN(1/d)), and dzc.y likewise for z
    v2f dxy, dzc;       // tri_inside: the direction; dzc = (d.z, -o.z / d.z)
};
__device__ __forceinline__ RayPre2 ray_pairs(f3 o, f3 d, const RayPre& p) {
    RayPre2 q;
    q.dxy = v2f{d.x, d.y};
    q.oxy = v2f{o.x, o.y};
    q.ixy = v2f{p.inv.x, p.inv.y};
    q.ozi = v2f{o.z, p.inv.z};
    q.cxy = v2f{-o.x * p.inv.x, -o.y * p.inv.y};
    q.dzc = v2f{d.z, -o.z * p.inv.z};
    return q;
}
__device__ __forceinline__ int slab_classify_pair(const float4 a, const float4 b, const RayPre2& q,
                                                  const RayPre& p, float& lb) {
    const v2f tx = (v2f{a.x, a.y} - q.oxy.xx) * q.ixy.xx;
    const v2f ty = (v2f{a.z, a.w} - q.oxy.yy) * q.ixy.yy;
    const v2f tz = (v2f{b.x, b.y} - q.ozi.xx) * q.ozi.yy;
    const float nx = fminf(tx.x, tx.y), fx = fmaxf(tx.x, tx.y);
    const float ny = fminf(ty.x, ty.y), fy = fmaxf(ty.x, ty.y);
    const float nz = fminf(tz.x, tz.y), fz = fmaxf(tz.x, tz.y);
    const float tn = fmaxf(nx, fmaxf(ny, nz)), n2 = __builtin_amdgcn_fmed3f(nx, ny, nz);
    const float tf = fminf(fx, fminf(fy, fz)), f2 = __builtin_amdgcn_fmed3f(fx, fy, fz);
    const float M = __builtin_fmaf(1.9073486e-06f /*32 * 2^-24*/, fabsf(tn) + fabsf(tf), p.a_ray);
    const float gap = tf - tn, e_top = tn - n2, e_bot = f2 - tf;
    lb = fmaxf(tn, 0.f);
    const bool flat = __float_as_int(b.w) != 0;
    const bool on_flat = flat & (tn == tf);
    const bool ok_gap = (gap > M) | on_flat;
    // (on a flat box the one face is entry and exit at once: it must be clear of the other axes' entries AND exits)
    const float e = on_flat ? fminf(e_top, e_bot) : (tn >= 0.f ? fmaxf(e_top, e_bot) : e_bot);
    const bool hit = ok_gap & (tf >= 0.f) & (e > M);
    const bool miss = (tf < -M) | (fminf(fminf(-gap, e_top), e_bot) > M) | (on_flat & (tf < 0.f));
    return hit ? BOX_HIT : (miss ? BOX_MISS : BOX_AMBIGUOUS);
}
// The fast tree's box test (k_intersect<FAST>): it only has to be CONSERVATIVE, i.e. report a miss only when the
// reference test (KDTree.cpp:34-52) provably rejects the box; everything else is entered and k_verify sorts it out.
// So no hit is proven here and the slab parameters come from one fused multiply-add per plane,
//     t~ = RN(P * y + c),  y = RN(1/d),  c = RN(-o * y):   |t~ - (P - o)/d| <= 3 eps (|P| + |o|) |y|,
// which the absolute part of the margin covers (a_ray = 16 eps max_a (|o_a| + bound_a) |y_a|, every |P| <= bound).
// MISS is reported when the box lies wholly behind the origin (tf < -M) or the three slab intervals are disjoint by
// more than M (tn - tf > M): a face the reference accepted would have its parameter t inside all three slabs up to
// the reference's own rounding (eps (|o_b| + |P_b|) / |d_b| + 2 eps |t| per compare, in t units), hence
// tn - err <= t <= tf + err, which tn - tf > M >= 2 err excludes. `unproven` tells the caller that the box was
// entered only because of the margin (|tf - tn| <= M on a box that is not flat): a ray for which that keeps
// happening (a nearly axis-parallel direction makes M enormous) is given up and left to the exact kernel.
template <bool WANT_UNPROVEN = true>
__device__ __forceinline__ bool slab_miss_pair(const float4 a, const float4 b, const RayPre2& q, const RayPre& p,
                                               float& lb, bool& unproven) {
    const v2f tx = pk_fma_b<0, 0>(v2f{a.x, a.y}, q.ixy, q.cxy);
    const v2f ty = pk_fma_b<1, 1>(v2f{a.z, a.w}, q.ixy, q.cxy);
    const v2f tz = pk_fma_b<1, 1>(v2f{b.x, b.y}, q.ozi, q.dzc);
    const float tn = vmax3(vmin(tx.x, tx.y), vmin(ty.x, ty.y), vmin(tz.x, tz.y));
    const float tf = vmin3(vmax(tx.x, tx.y), vmax(ty.x, ty.y), vmax(tz.x, tz.y));
    const float M = __builtin_fmaf(1.9073486e-06f /*32 * 2^-24*/, fabsf(tn) + fabsf(tf), p.a_ray);
    const float gap = tf - tn;
    lb = vmax(tn, 0.f);
    const bool miss = vmin(tf, gap) < -M;  // (tf < -M) | (gap < -M)
    unproven = WANT_UNPROVEN ? (!miss & (__float_as_int(b.w) == 0) & !(gap > M)) : false;
    return miss;
}
// k_verify's first look at a box of the winner's chain: a SUFFICIENT condition for "the reference test passes",
// cheaper than the full classifier. With the slab parameters of slab_miss_pair (one fused multiply-add per plane;
// c = RN(-o * RN(1/d)) per axis), the reference accepts the EXIT face when that face lies in front of the origin
// (tf > M), the box is entered before it is left by more than the margin (tf - tn > M: the exit point is inside the
// other two slabs on their entry side) and the exit is clear of the other two axes' exits (f2 - tf > M: inside them
// on their exit side). The rounding of these t~ is within 3/16 of a_ray each, so differences of two of them and the
// reference's own compare error still fit M. A box that fails this (flat boxes always do) goes to slab_classify.
// `axis`: which axis' exit plane that face is (the unique smallest exit value when the proof holds: f2 - tf > M > 0);
// the face itself is that axis' max plane for a positive direction component, its min plane otherwise.
__device__ __forceinline__ bool slab_exit_face_proven(const float4 lo, const float4 hi, f3 c, const RayPre& p, int& axis) {
    const float t0x = __builtin_fmaf(lo.x, p.inv.x, c.x), t1x = __builtin_fmaf(hi.x, p.inv.x, c.x);
    const float t0y = __builtin_fmaf(lo.y, p.inv.y, c.y), t1y = __builtin_fmaf(hi.y, p.inv.y, c.y);
    const float t0z = __builtin_fmaf(lo.z, p.inv.z, c.z), t1z = __builtin_fmaf(hi.z, p.inv.z, c.z);
    const float fx = fmaxf(t0x, t1x), fy = fmaxf(t0y, t1y), fz = fmaxf(t0z, t1z);
    const float tn = fmaxf(fminf(t0x, t1x), fmaxf(fminf(t0y, t1y), fminf(t0z, t1z)));
    const float tf = fminf(fx, fminf(fy, fz)), f2 = __builtin_amdgcn_fmed3f(fx, fy, fz);
    const float M = __builtin_fmaf(1.9073486e-06f /*32 * 2^-24*/, fabsf(tn) + fabsf(tf), p.a_ray);
    axis = fx == tf ? 0 : (fy == tf ? 1 : 2);
    return (tf > M) & (tf - tn > M) & (f2 - tf > M);
}
// The same with the flat boxes' case folded in (every axis-aligned wall triangle's leaf, and the node above two of
// them): a box of zero extent along an axis yields that axis' entry and exit parameter from the same plane by the same
// fused multiply-add, bit for bit, so tn == tf means that the one face is entry and exit at once and both other slabs
// contain it; the reference accepts that face when it lies in front of the origin (tf > M) and is clear of the other
// axes' entries (tn - n2 > M) and exits (f2 - tf > M): slab_classify's rule for flat boxes with this function's
// parameters. `via_flat`: the accepted plane is the box's min AND max plane on `axis`, so the run of ancestors that
// inherit the acceptance may be followed through either (k_verify takes the longer run).
__device__ __forceinline__ bool slab_face_proven(const float4 lo, const float4 hi, bool flat, f3 c, const RayPre& p,
                                                 int& axis, bool& via_flat) {
    const float t0x = __builtin_fmaf(lo.x, p.inv.x, c.x), t1x = __builtin_fmaf(hi.x, p.inv.x, c.x);
    const float t0y = __builtin_fmaf(lo.y, p.inv.y, c.y), t1y = __builtin_fmaf(hi.y, p.inv.y, c.y);
    const float t0z = __builtin_fmaf(lo.z, p.inv.z, c.z), t1z = __builtin_fmaf(hi.z, p.inv.z, c.z);
    const float fx = fmaxf(t0x, t1x), fy = fmaxf(t0y, t1y), fz = fmaxf(t0z, t1z);
    const float nx = fminf(t0x, t1x), ny = fminf(t0y, t1y), nz = fminf(t0z, t1z);
    const float tn = fmaxf(nx, fmaxf(ny, nz)), n2 = __builtin_amdgcn_fmed3f(nx, ny, nz);
    const float tf = fminf(fx, fminf(fy, fz)), f2 = __builtin_amdgcn_fmed3f(fx, fy, fz);
    const float M = __builtin_fmaf(1.9073486e-06f /*32 * 2^-24*/, fabsf(tn) + fabsf(tf), p.a_ray);
    axis = fx == tf ? 0 : (fy == tf ? 1 : 2);
    via_flat = flat & (tn == tf) & (tn - n2 > M);
    return (tf > M) & (f2 - tf > M) & ((tf - tn > M) | via_flat);
}
__device__ __forceinline__ bool box_is_flat(const float4 lo, const float4 hi) {
    return (lo.x == hi.x) | (lo.y == hi.y) | (lo.z == hi.z);
}

// ---- triangle test (KDTree.cpp:267-289) ----------------------------------------------------------
// In two parts, each the reference's operations in the reference's order (no contraction), so the pair is the
// reference test bit for bit: tri_inside() is the three edge signs (the ray's line passes through the triangle),
// tri_plane() the distance along the ray (one IEEE division) with the reference's rejections. The traversal
// kernels run the first per triangle and the second once per leaf pass, for the triangle a lane stopped at.
__device__ __forceinline__ f3 tri_v0(const TriRec& T) { return f3{T.x.x, T.y.x, T.z.x}; }
__device__ __forceinline__ f3 tri_v1(const TriRec& T) { return f3{T.x.y, T.y.y, T.z.y}; }
__device__ __forceinline__ f3 tri_v2(const TriRec& T) { return f3{T.x.w, T.y.w, T.z.w}; }
// (origin and direction come as the register pairs of RayPre2: a half broadcast to both edges is an op_sel of the
// packed instruction, where a pair built from one scalar would cost two moves and two more registers)
__device__ __forceinline__ bool tri_inside(const float4 X, const float4 Y, const float4 Z, const RayPre2& q) {
    // edges 0 and 1 in the two halves of packed registers: l = (v1 - v0 | v2 - v1), r = (v0 - o | v1 - o)
    const v2f ax = {X.x, X.y}, bx = {X.z, X.w}, ay = {Y.x, Y.y}, by = {Y.z, Y.w}, az = {Z.x, Z.y}, bz = {Z.z, Z.w};
    const v2f lx = bx - ax, ly = by - ay, lz = bz - az;
    const v2f rx = pk_sub_b<0>(ax, q.oxy), ry = pk_sub_b<1>(ay, q.oxy), rz = pk_sub_b<0>(az, q.ozi);
    const v2f cx = ly * rz - lz * ry, cy = lz * rx - lx * rz, cz = lx * ry - ly * rx;  // cross3(l, r)
    v2f s01 = pk_mul_b<0>(cx, q.dxy);                                                 // dot3(d, .)
    s01 = s01 + pk_mul_b<1>(cy, q.dxy);
    s01 = s01 + pk_mul_b<0>(cz, q.dzc);
    const f3 o = mk3(q.oxy.x, q.oxy.y, q.ozi.x), d = mk3(q.dxy.x, q.dxy.y, q.dzc.x);
    const f3 v0 = mk3(X.x, Y.x, Z.x), v2 = mk3(X.w, Y.w, Z.w);
    const float s2 = dot3(d, cross3(v0 - v2, v2 - o));
    const bool all_pos = (s01.x >= 0.f) & (s01.y >= 0.f) & (s2 >= 0.f);
    const bool all_neg = (s01.x <= 0.f) & (s01.y <= 0.f) & (s2 <= 0.f);
    return all_pos | all_neg;
}
__device__ __forceinline__ float tri_plane(const float4 N, f3 v2, f3 o, f3 d) {
    const f3 n = mk3(N.x, N.y, N.z);
    const float orthoDist = dot3(n, v2 - o);
    const float dd = dot3(n, d);
    const float q = orthoDist / dd;
    if (dd == 0.f || q < 0.f) return -1.f;
    return q;
}
__device__ __forceinline__ float tri_test(const TriRec& T, f3 o, f3 d) {
    RayPre2 q2;
    q2.oxy = v2f{o.x, o.y};
    q2.ozi = v2f{o.z, 0.f};
    q2.dxy = v2f{d.x, d.y};
    q2.dzc = v2f{d.z, 0.f};
    const float q = tri_plane(T.n, tri_v2(T), o, d);  // (NaN stays NaN, as in the reference's single expression)
    return tri_inside(T.x, T.y, T.z, q2) ? q : -1.f;
}

// ---- shading normal (SimpleShape.cpp:300-339 through SceneObject.cpp:250-254) ------------------
// the interpolated corner normals before normalisation
__device__ __forceinline__ f3 world_normal_raw(const TriRec& T, const VnRec& V, f3 pos) {
    f3 v0 = tri_v0(T), v1 = tri_v1(T), v2 = tri_v2(T);
    f3 b1 = v1 - v0, b2 = v2 - v0;
    f3 v = pos - v0;
    const float l1 = V.w.x, d12 = V.w.y, den = V.w.z, l1l1 = V.w.w;
    float vb1 = dot3(v, b1), vb2 = dot3(v, b2);
    float u2 = (vb1 * d12 - vb2 * l1 * l1) / den;
    float u1 = (vb1 - u2 * d12) / l1l1;
    f3 n0 = mk3(V.n0.x, V.n0.y, V.n0.z), n1 = mk3(V.n1.x, V.n1.y, V.n1.z),
       n2 = mk3(V.n2.x, V.n2.y, V.n2.z);
    return ((1.f - u1 - u2) * n0 + u1 * n1) + u2 * n2;
}
__device__ __forceinline__ f3 world_normal(const TriRec& T, const VnRec& V, bool has_vn, f3 pos) {
    if (!has_vn) return mk3(V.n0.x, V.n0.y, V.n0.z);  // normalize((v1-v0) x (v2-v0)), rounded on the host at upload
    return normalize3(world_normal_raw(T, V, pos));
}

// ---- textures (Texture.cpp:43-94, SimpleShape.cpp:274-298) --------------------------------------
// SimpleShape::getTexCoord: the barycentric weights of world_normal applied to the corner texture
// coordinates; (0,0,0) for a scene without any
__device__ __forceinline__ f3 tex_coord(const DevScene& sc, const TriRec& T, int slot, f3 pos) {
    if (!sc.vts) return mk3(0, 0, 0);
    const VtRec W = sc.vts[slot];
    f3 v0 = tri_v0(T), v1 = tri_v1(T), v2 = tri_v2(T);
    f3 b1 = v1 - v0, b2 = v2 - v0;
    f3 v = pos - v0;
    float d12 = dot3(b1, b2);
    float l1 = len3(b1), l2 = len3(b2);
    float vb1 = dot3(v, b1), vb2 = dot3(v, b2);
    float u2 = (vb1 * d12 - vb2 * l1 * l1) / (d12 * d12 - l1 * l1 * l2 * l2);
    float u1 = (vb1 - u2 * d12) / (l1 * l1);
    return ((1.f - u1 - u2) * mk3(W.t0.x, W.t0.y, W.t0.z) + u1 * mk3(W.t1.x, W.t1.y, W.t1.z)) +
           u2 * mk3(W.t2.x, W.t2.y, W.t2.z);
}
// rows 0..2 of M * (v, 1), accumulated left to right like nvMatrix.h's operator*
__device__ __forceinline__ f3 apply_rows(const float* m, f3 v) {
    return mk3(((v.x * m[0] + v.y * m[1]) + v.z * m[2]) + m[3], ((v.x * m[4] + v.y * m[5]) + v.z * m[6]) + m[7],
               ((v.x * m[8] + v.y * m[9]) + v.z * m[10]) + m[11]);
}
// (int)floor(x) as the reference's platform computes it: x86's cvttss2si / cvttsd2si return INT_MIN for a NaN and for
// anything outside of int's range, where this GPU's conversion saturates and turns a NaN into 0. A texture coordinate IS
// NaN, or astronomically large, on a degenerate or needle triangle (getTexCoord divides by ~0), and the texel the
// reference then reads decides a bump-mapped normal (scripts/fuzz_pt.py TEX=2, seed 7132).
__device__ __forceinline__ int floor_to_int_x86(float x) {
    const float f = floorf(x);
    return (f >= -2147483648.f && f < 2147483648.f) ? (int)f : (int)0x80000000;
}
// Texture::getColor: bilinear, wrapping, texel centres at +0.5; the gray branch reads its fourth texel
// at (top, LEFT) as the reference does (Texture.cpp:86)
__device__ __forceinline__ f3 tex_color(const DevTexture& X, f3 coord) {
    const f3 tc = apply_rows(X.coord, coord);
    float u = tc.x * (float)X.width - 0.5f;
    float v = tc.y * (float)X.height - 0.5f;
    int left = floor_to_int_x86(u), bottom = floor_to_int_x86(v);
    int right = left + 1, top = bottom + 1;
    u -= (float)left;
    v -= (float)bottom;
    left %= X.width;
    right %= X.width;
    top %= X.height;
    bottom %= X.height;
    if (left < 0) left += X.width;
    if (right < 0) right += X.width;
    if (top < 0) top += X.height;
    if (bottom < 0) bottom += X.height;
    const float wu = 1.f - u, wv = 1.f - v;
    f3 color;
    if (X.channels == 3) {
        auto at = [&](int y, int x) {
            const float* p = X.texels + 3 * ((size_t)y * X.width + x);
            return mk3(p[0], p[1], p[2]);
        };
        color = mk3(0, 0, 0) + (at(bottom, left) * wu) * wv;
        color = color + (at(bottom, right) * u) * wv;
        color = color + (at(top, left) * wu) * v;
        color = color + (at(top, right) * u) * v;
    } else {
        auto at = [&](int y, int x) { return X.texels[(size_t)y * X.width + x]; };
        float c = 0.f + (at(bottom, left) * wu) * wv;
        c = c + (at(bottom, right) * u) * wv;
        c = c + (at(top, left) * wu) * v;
        c = c + (at(top, left) * u) * v;
        color = mk3(c, c, c);
    }
    return apply_rows(X.color, color);
}

// Texture::getGrad (Texture.cpp:96-147): forward differences of the texel under the coordinate (first
// channel of an rgb texture), their length through the colour transform as .z, their direction through
// the inverse coordinate transform as .xy. `inv`: rows 0..2 of inverse(coordTransform), from the host.
struct DevBump {
    int texture;     // index into DevScene.textures
    float inv[12];
};
__device__ __forceinline__ f3 tex_grad(const DevTexture& X, const float* inv, f3 coord) {
    const f3 tc = apply_rows(X.coord, coord);
    const float u = tc.x * (float)X.width - 0.5f;
    const float v = tc.y * (float)X.height - 0.5f;
    int left = floor_to_int_x86(u), bottom = floor_to_int_x86(v);
    int right = left + 1, top = bottom + 1;
    left %= X.width;
    right %= X.width;
    top %= X.height;
    bottom %= X.height;
    if (left < 0) left += X.width;
    if (right < 0) right += X.width;
    if (top < 0) top += X.height;
    if (bottom < 0) bottom += X.height;
    const int ch = X.channels;
    auto at = [&](int y, int x) { return X.texels[ch * ((size_t)y * X.width + x)]; };
    const float lb = at(bottom, left);
    f3 g = mk3(at(bottom, right) - lb, at(top, left) - lb, 0.f);
    float var_n = len3(g);
    var_n = ((var_n * X.color[0] + var_n * X.color[1]) + var_n * X.color[2]) + 0.f * X.color[3];
    g = normalize3(g);
    f3 r = mk3(((g.x * inv[0] + g.y * inv[1]) + g.z * inv[2]) + 0.f * inv[3],
               ((g.x * inv[4] + g.y * inv[5]) + g.z * inv[6]) + 0.f * inv[7], var_n);
    return r;
}
// smallFuncs.h:11-24 (rotMat) applied to a direction
__device__ __forceinline__ f3 rot_about(f3 axis, float angle, f3 v) {
    const float c = cosf(angle), s = sinf(angle), ic = 1.f - c;
    const float x = axis.x, y = axis.y, z = axis.z;
    return mk3(v.x * (c + ic * x * x) + v.y * (ic * x * y - s * z) + v.z * (ic * x * z + s * y),
               v.x * (ic * x * y + s * z) + v.y * (c + ic * y * y) + v.z * (ic * y * z - s * x),
               v.x * (ic * x * z - s * y) + v.y * (ic * y * z + s * x) + v.z * (c + ic * z * z));
}
// SceneObject::getWorldNormal's bump branch (SceneObject.cpp:258-282): the interpolated normal `n` tilted
// towards the texture's gradient, expressed in the triangle's edge directions through the corner
// texture coordinates. (The libm calls - atan2, sin, cos - are the device's: not bit-identical to glibc's.)
__device__ __forceinline__ f3 bump_normal(const DevScene& sc, const TriRec& T, int slot, const DevBump& B, f3 n, f3 pos) {
    const VtRec W = sc.vts[slot];
    const f3 g = tex_grad(sc.textures[B.texture], B.inv, tex_coord(sc, T, slot, pos));
    f3 b1 = tri_v1(T) - tri_v0(T), b2 = tri_v2(T) - tri_v0(T);
    const f3 duv1 = mk3(W.t1.x, W.t1.y, W.t1.z) - mk3(W.t0.x, W.t0.y, W.t0.z);
    const f3 duv2 = mk3(W.t2.x, W.t2.y, W.t2.z) - mk3(W.t0.x, W.t0.y, W.t0.z);
    const float k2 = (g.x * duv1.y - g.y * duv1.x) / (duv1.y * duv2.x - duv1.x * duv2.y);
    const float k1 = (g.y - k2 * duv2.y) / duv1.y;
    b1 = normalize3(b1);
    b2 = normalize3(b2);
    const f3 dl = k1 * b1 + k2 * b2;
    const f3 dh = n * g.z;
    if (len3(dh) * 1000.f < len3(dl)) return n;
    const float angle = atan2f(len3(dh), len3(dl));
    const f3 axis = normalize3(cross3(dl, dh));
    return rot_about(axis, angle, n);
}
// SceneObject::getWorldNormal (SceneObject.cpp:250-283). VnRec.n2.w: bit 0 = the object interpolates vertex
// normals, bits 8.. = 1 + index into DevScene.bumps when the object has a bump texture AND texture
// coordinates. BUMP false: the scene has no bump textures (the instantiations every other scene runs).
template <bool BUMP>
__device__ __forceinline__ f3 shading_normal(const DevScene& sc, const TriRec& T, const VnRec& V, int slot, f3 pos) {
    const int flags = __float_as_int(V.n2.w);
    f3 n = world_normal(T, V, (flags & 1) != 0, pos);
    if (BUMP && (flags >> 8) != 0) n = bump_normal(sc, T, slot, sc.bumps[(flags >> 8) - 1], n, pos);
    return n;
}

// d . shading normal as the predicate needs it: only its SIGN (and whether it is zero) enters legal_noself. For a
// smooth triangle the reference normalises the interpolated normal nn first (a square root and three divisions);
// with S = sum d_a nn_a, the reference's value is S / |nn| up to 4u sum |d_a nn_a| / |nn| (u = 2^-24: one rounding
// per normalised component, product and sum) and the value computed here from nn itself is S up to 3u of the same
// sum, so when it exceeds 16u of that sum in magnitude its sign IS the reference's; otherwise (one candidate in
// ~10^6; NaN and underflow land here too) the reference's own operations run.
template <bool BUMP>
__device__ __forceinline__ float noself_dn(const DevScene& sc, const TriRec& T, const VnRec& V, int slot, f3 pos, f3 d) {
    const int flags = __float_as_int(V.n2.w);
    if ((flags & 1) == 0 || (BUMP && (flags >> 8) != 0)) return dot3(d, shading_normal<BUMP>(sc, T, V, slot, pos));
    const f3 nn = world_normal_raw(T, V, pos);
    float dn = dot3(d, nn);
    const float mag = (fabsf(d.x * nn.x) + fabsf(d.y * nn.y)) + fabsf(d.z * nn.z);
    if (!(fabsf(dn) > 9.5367431640625e-07f /*16 * 2^-24*/ * mag)) dn = dot3(d, normalize3(nn));
    return dn;
}

// NoSelfIntersectionCondition.cpp:12-27; ids are object ids, -1 = NULL
__device__ __forceinline__ bool legal_noself(float dn, int hitObj, int contact, int inside) {
    bool in = dn <= 0.f, out = dn >= 0.f;
    if (inside == contact) return !(in && contact == hitObj);
    return !(out && inside != hitObj);
}

__device__ __forceinline__ int unpack_contact(int aux) { return (int)((uint32_t)aux & 0xFFFFu) - 1; }
__device__ __forceinline__ int unpack_inside(int aux) {
    return (int)(((uint32_t)aux >> 16) & 0xFFFFu) - 1;
}

}  // namespace mirt
