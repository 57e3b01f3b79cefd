// pt_kernels.h — wavefront path tracing kernels: generate, shade (+ballot/prefix-sum compaction),
// accumulate. Replaces the per-pixel loop PathTracer.cpp:17-44 and MCRenderer::samplePath
// (MCRenderer.cpp:149-224); the materials follow DiffuseMaterial.cpp:4-76, GlossyMaterial.cpp:4-56,
// RefractiveMaterial.cpp:4-104 and SceneEmissiveObject.cpp:4-19. No path vertex list is stored:
// the throughput  prod_i color_i/(dirProb_i*originProb_i) * cos_i * decay_i  of PathTracer.cpp:28-40
// is carried forward and scored when the path ends on an emitter (PathTracer.cpp:25-26).
#pragma once
#include "mirt_dev.h"

#pragma clang fp contract(off)

#ifndef MIRT_TILE_MORTON
#define MIRT_TILE_MORTON 1
#endif

namespace mirt {

// Two fields of the path state carry a second value in their top bits (wavefront chaining, mirt.hip run_chained):
//   ctr: bits 0..23 the RNG draw counter (a path draws < 200 numbers), bits 24..31 the path's DEPTH = scatter events
//        behind it (k_shade reads it instead of a launch-wide depth: one launch may hold paths of different depths)
//   pid: bits 0..29 the path index inside its wavefront (<= 2^30 paths), bits 30..31 which of the (up to four)
//        radiance buffers of the L ring the path scores into
constexpr uint32_t CTR_MASK = 0x00FFFFFFu, PID_MASK = 0x3FFFFFFFu;
constexpr int CTR_DEPTH_SHIFT = 24, PID_GEN_SHIFT = 30;

// SoA path state; planes have `stride` elements. Hit fields (t, slot) are filled by k_intersect.
struct PathBuf {
    float* o;        // [3][stride] ray origin
    float* d;        // [3][stride] ray direction
    float* thr;      // [3][stride] throughput carried so far
    float* t;        // [stride]    hit distance
    int* slot;       // [stride]    hit triangle slot or -1
    uint32_t* key;   // [stride]    RNG stream key of (pixel, sample)
    uint32_t* ctr;   // [stride]    RNG draw counter
    int* ids;        // [stride]    MIRT_PACK_IDS(contact, inside) of the ray
    uint32_t* pid;   // [stride]    original path index = sample_local * W*H + pixel
    float* lpw;      // [stride]    PathTracerTest only: directionProb of the previous vertex, or < 0 when
                     //             that vertex was DEFINITE / there is none (lastSpecular, PathTracerTest.cpp:29-30)
    uint32_t stride;
};
// the same buffers seen from entry `off` on (two-ended compaction: the live paths of a bounce are [off, off + n))
__device__ __forceinline__ void shift_paths(PathBuf& b, uint32_t off) {
    b.o += off; b.d += off; b.thr += off; b.t += off; b.slot += off; b.key += off; b.ctr += off; b.ids += off;
    b.pid += off; b.lpw += off;
}

// Shadow rays of one bounce of PathTracerTest (colorByConnectingLights, PathTracerTest.cpp:85-135):
// what the visibility launch reads (o, d, ids, lim) and what the visibility query adds to L when unoccluded (ConnectArgs).
struct ShadowBuf {
    float* o;       // [3][stride]
    float* d;       // [3][stride]
    int* ids;       // [stride] MIRT_PACK_IDS(contact, inside) of the vertex
    float* lim;     // [stride] distance to the light sample
    float* c;       // [3][stride] throughput * contribution * MIS weight
    uint32_t* pid;  // [stride]
    unsigned char* vis;  // [stride] written by the visibility launch
    uint32_t stride;
};

struct DevCamera {
    f3 pos, front, right, up;  // up = orthoUp = right x front (Camera.cpp:6-10)
    int width, height;
    float sight;               // float(sightDist)
    double half_w, half_h;     // width/2.0, height/2.0 (double, Camera.cpp:21)
    // the pixel window this render covers (MirtTile, include/mirt.h; the whole image: 0, 0, width, height). Paths,
    // radiance slots and the caller's accumulator are indexed by the LOCAL pixel ly * tw + lx; the camera ray, the RNG
    // stream and the vignetting term use the pixel of the full image, so a tile's values are the full frame's.
    int tx0, ty0, tw, th;
};
// local pixel of the window -> pixel of the full image
__device__ __forceinline__ uint32_t tile_pixel(const DevCamera& c, uint32_t lp) {
    return ((uint32_t)c.ty0 + lp / (uint32_t)c.tw) * (uint32_t)c.width + (uint32_t)c.tx0 + lp % (uint32_t)c.tw;
}

// ---- counter RNG (DESIGN.md "RNG"): replaces the global MT19937 of RandGenerator.cpp:9 -------------
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t stream_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t h1 = lowbias32(sample + 0x9E3779B9U * (seed + 1U));
    return lowbias32(pixel ^ h1);
}
// 24-bit grid in [0,1) like RandGenerator.cpp:21-25
__device__ __forceinline__ float rng_float(uint32_t key, uint32_t& ctr) {
    uint32_t b = lowbias32(key + 0x9E3779B9U * (ctr++));
    return (float)(b >> 8) / 16777216.0f;
}

__device__ __forceinline__ f3 load3(const float* p, uint32_t stride, uint32_t i) {
    return mk3(p[i], p[stride + i], p[2 * stride + i]);
}
__device__ __forceinline__ void store3(float* p, uint32_t stride, uint32_t i, f3 v) {
    p[i] = v.x;
    p[stride + i] = v.y;
    p[2 * stride + i] = v.z;
}

// Camera.cpp:12-22
__device__ __forceinline__ f3 camera_dir(const DevCamera& c, uint32_t pixel, float u, float v) {
    int x = (int)(pixel % (uint32_t)c.width);
    int y = (int)(pixel / (uint32_t)c.width);
    float fx = (float)((double)((float)x + u) - c.half_w);
    float fy = (float)((c.half_h - (double)v) - (double)y);
    f3 dir = (c.front * c.sight + c.right * fx) + c.up * fy;
    return normalize3(dir);
}

#ifndef MIRT_PIXEL_MAJOR
#define MIRT_PIXEL_MAJOR 1
#endif
// Camera ray batch. ray i: sample spp_begin + i / (W*H), pixel i % (W*H).
__global__ void __launch_bounds__(256)
k_generate(DevCamera cam, uint32_t n, uint32_t spp_begin, uint32_t seed, float* __restrict__ O,
           float* __restrict__ D, uint32_t stride, PathBuf pb, int fill_state, uint32_t pid_gen = 0) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t wh = (uint32_t)cam.tw * (uint32_t)cam.th;
    uint32_t pixel = i % wh, sample_local = i / wh;  // (pixel of the window; of the full image from tile_pixel on)
#if MIRT_PIXEL_MAJOR
    // The wavefront path tracer's own pass (fill_state != 0) orders its batch PIXEL-major: the samples of one pixel are
    // neighbours in the batch, so the 64 rays a traversal wave takes differ by the jitter inside one pixel: they walk
    // the same nodes and test the same triangles, their hits have one material and one triangle record (k_verify,
    // k_shade at depth 0). Streams and radiance slots are keyed by (pixel, sample): the image does not depend on it.
    // 1024^2 x 1024 spp: 1 994 -> 2 021 Msamples/s (fog scene 1 841 -> 1 856).
    // (plain path tracing only: PathTracerTest adds a light connection to L[pid] at every bounce, and the connection's
    // read-modify-write of entries a whole image apart is slower than the traversal gains: 1 107 -> 971 Msamples/s)
    if (fill_state == 1) {
        const uint32_t s_cnt = n / wh;  // samples per pixel in this wavefront (n = wh * s_cnt)
        pixel = i / s_cnt;
        sample_local = i % s_cnt;
    }
#endif
    const uint32_t sample = spp_begin + sample_local;
    // The wavefront path tracer's own camera pass walks the image in 16 x 16 tiles (256 consecutive rays = one
    // tile = one ray chunk of a traversal wave, intersect_kernels.h): neighbours in the batch are neighbours on
    // the screen in both directions. Sample streams and radiance slots are keyed by (pixel, sample), so the
    // image does not depend on this order. The public ray generator (fill_state 0) keeps the documented one.
    if (fill_state && (cam.tw & 15) == 0 && (cam.th & 15) == 0) {
        const uint32_t tile = pixel >> 8, within = pixel & 255u, tiles_x = (uint32_t)cam.tw >> 4;
#if MIRT_TILE_MORTON
        // inside the tile along a Z curve: any 64 consecutive rays (one refill of a wave) cover an 8 x 8 block
        auto even_bits = [](uint32_t v) {
            v &= 0x55u;
            v = (v | (v >> 1)) & 0x33u;
            v = (v | (v >> 2)) & 0x0Fu;
            return v;
        };
        const uint32_t wx = even_bits(within), wy = even_bits(within >> 1);
#else
        const uint32_t wx = within & 15u, wy = within >> 4;
#endif
        pixel = ((tile / tiles_x) * 16u + wy) * (uint32_t)cam.tw + (tile % tiles_x) * 16u + wx;
    }
    const uint32_t local_pixel = pixel;
    pixel = tile_pixel(cam, local_pixel);
    uint32_t key = stream_key(seed, pixel, sample), ctr = 0;
    float u = rng_float(key, ctr);
    float v = rng_float(key, ctr);
    f3 dir = camera_dir(cam, pixel, u, v);
    store3(O, stride, i, cam.pos);
    store3(D, stride, i, dir);
    if (fill_state) {
        // vertex 0: color 1, probabilities 1, cosine term |front . dir| because the camera is the
        // contact object (Camera.cpp:25, Ray.cpp:71-78)
        float c0 = fabsf(dot3(cam.front, dir));
        c0 = c0 > 1e-6f ? c0 : 1e-6f;
        if (fill_state == 2) {  // PathTracerTest.cpp:26-30: throughput 1, lastSpecular, the loop starts at vertex 1
            c0 = 1.f;
            pb.lpw[i] = -1.f;
        }
        store3(pb.thr, pb.stride, i, mk3(c0, c0, c0));
        pb.key[i] = key;
        pb.ctr[i] = ctr;
        pb.ids[i] = MIRT_PACK_IDS(-1, -1);
        pb.pid[i] = (sample_local * wh + local_pixel) | pid_gen;  // sample_local * (window pixels) + local pixel, whatever the order of the batch
    }
}

// LocalFrame.h:14-28
struct Frame {
    f3 s, t, n;
};
__device__ __forceinline__ Frame frame_from_normal(f3 n) {
    Frame f;
    f.n = n;
    f3 tmp = fabsf(n.z) > 0.99f ? mk3(1, 0, 0) : mk3(0, 0, 1);
    f.s = normalize3(cross3(n, tmp));
    f.t = normalize3(cross3(f.s, n));
    return f;
}
__device__ __forceinline__ f3 to_world(const Frame& f, f3 a) {
    return (f.s * a.x + f.n * a.y) + f.t * a.z;
}
// RandGenerator.cpp:152-168
__device__ __forceinline__ f3 hemi_cos(float expTerm, float dx, float dy) {
    float u1 = (float)(6.28318530717958 * (double)dx);
    float u2 = powf(dy, 1.f / (expTerm + 1.f));
    float u3 = sqrtf(1.f - u2 * u2);
    return normalize3(mk3(cosf(u1) * u3, u2, sinf(u1) * u3));
}
// CosineSphericalSampler.h:29-34
__device__ __forceinline__ float cos_pdf(f3 n, f3 dir, float coeff) {
    float c = dot3(n, dir);
    c = c > 0.f ? c : 0.f;
    return (float)((double)(powf(c, coeff) * (coeff + 1.f)) / 6.28318530717958);
}
// *std::max_element(&c.x, &c.x + 3) (DiffuseMaterial.cpp:48,84,91, GlossyMaterial.cpp:30,75,84): the first component stays
// the largest unless a later one compares greater (fmaxf would drop a NaN component where this keeps or skips it)
__device__ __forceinline__ float max3f(f3 c) {
    float m = c.x;
    if (m < c.y) m = c.y;
    if (m < c.z) m = c.z;
    return m;
}
__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}
// smallFuncs.h:11-24 applied to a direction (mirt_dev.h rot_about)
__device__ __forceinline__ f3 rotate_about(f3 axis, float angle, f3 v) { return rot_about(axis, angle, v); }
// SceneObject.h:58 / SceneRefractiveObject.h:22: mirrors keep the default 1
__device__ __forceinline__ float refr_coeff(const DevObject& o) {
    return (o.type == MIRT_OBJ_REFRACTIVE && o.ior >= 0.f) || o.type == MIRT_OBJ_VPM ? o.ior : 1.f;
}

// Camera.cpp:60-78: divide by |dir(pixel centre) . front|
__device__ __forceinline__ float vignette_div(const DevCamera& c, uint32_t pixel) {
    int x = (int)(pixel % (uint32_t)c.width);
    int y = (int)(pixel / (uint32_t)c.width);
    float fx = (float)(((double)x + 0.5) - c.half_w);
    float fy = (float)((c.half_h - 0.5) - (double)y);
    f3 dir = normalize3((c.front * c.sight + c.right * fx) + c.up * fy);
    return fabsf(dot3(dir, c.front));
}

// ---- volumetric objects: Scene::findInsideObject (Scene.cpp:354-390) on the device ---------------
// Unconditional closest hit inside ONE object's own tree (Scene::objKDTrees[i].intersect): the reference box test
// (KDTree.cpp:34-52) on every node whose parent passed, in pre-order with escape indices instead of a stack. The
// reference visits both children of every node it enters and keeps the smallest t (later leaf on a tie, earlier
// triangle inside a leaf: KDTree.cpp:305, 319-320); the order of the visit is free, and a subtree whose box starts
// beyond the closest hit so far cannot change the answer (the exact kernels' pruning rule WITHOUT exact_prune_slack,
// mirt_dev.h: the ray's reciprocals it needs cost k_shade<VOL> two registers it does not have, 95 -> 97 VGPRs and a
// wave per SIMD. What that leaves open is a ray that barely moves along an axis and meets a corner of the object's
// surface to within 1e-6: its inside-object answer may differ from the reference's, inside the image tolerance).
__device__ inline float trace_object(const DevScene& sc, const VolRoot& R, f3 o, f3 d, int& slot_out) {
    int best_slot = -1;
    float best = -1.f, thr = __builtin_inff();
    const int end = R.first + R.count;
    for (int i = R.first; i < end;) {
        const VolNode N = sc.vnodes[i];
        const BoxHit b = box_test(N.lo, N.hi, o, d);
        if (!b.hit || b.lb > thr) {
            i = __float_as_int(N.lo.w);  // past the subtree
            continue;
        }
        for (int s = N.leaf_begin; s < N.leaf_end; s++) {
            const float t = tri_test(sc.tris[s], o, d);
            if (!(t >= 0.f)) continue;
            if (!(best_slot < 0 || t < best || (t == best && best_slot < N.leaf_begin))) continue;
            best = t;
            best_slot = s;
            thr = t * 1.0001f + 1e-5f;
        }
        i++;  // first child, or (a leaf) the next node in pre-order
    }
    slot_out = best_slot;
    return best_slot >= 0 ? best : -1.f;
}

// Scene::findInsideObject's test of ONE small object (VolRoot::small, e.g. the 12 triangles of a fog cube): is the
// object's own surface met from inside both backwards and forwards along the ray? The two unconditional closest-hit
// queries of the reference (objKDTrees[i].intersect along -d and along d), computed the other way round and in one
// pass: every triangle of the object is tested once - the three edge signs of the reference test are those of the
// ray's LINE (negating d negates every sign exactly, and "all >= 0 or all <= 0" does not care), and the plane distance
// along -d is exactly minus the one along d (n.(-d) = -(n.d), and IEEE division is odd) - so a triangle the line
// passes through is a candidate of the forward query when q >= 0 and of the backward one when -q >= 0. Only for a
// candidate that would become the closest hit of its query is the chain of boxes above it, its leaf up to the root,
// put to the reference box test (decided by the slab classifier where that is proven, by the reference arithmetic
// where not: the rule of k_verify). The reference's answer is the closest triangle among those whose whole chain
// passes, later leaf / earlier triangle on a tie; walking the tree instead costs ~14 box tests of six IEEE divisions
// each per query for a ray that starts inside a cube. (No distance pruning here, so the answer is the plain definition.)
__device__ inline bool inside_small(const DevScene& sc, const VolRoot& R, f3 o, f3 d, const RayPre& pre) {
    const f3 bd = neg(d);
    RayPre preb = pre;
    preb.inv = neg(pre.inv);  // RN(1 / -d) = -RN(1 / d)
    RayPre2 q2;
    q2.oxy = v2f{o.x, o.y};
    q2.ozi = v2f{o.z, 0.f};
    q2.dxy = v2f{d.x, d.y};
    q2.dzc = v2f{d.z, 0.f};
    float ft = -1.f, bt = -1.f;            // closest hit forwards / backwards
    int fs = -1, bs = -1, fl = -1, bl = -1;  // its slot and leaf node
    auto chain = [&](int leaf, f3 dir, const RayPre& p) {
        for (int n = leaf; n >= 0;) {
            const VolNode A = sc.vnodes[n];
            int code = BOX_AMBIGUOUS;
            float unused;
            if (p.fast) code = slab_classify(A.lo, A.hi, A.flat != 0, o, p, unused);
            if (code == BOX_MISS) return false;
            if (code == BOX_AMBIGUOUS && !box_test(A.lo, A.hi, o, dir).hit) return false;
            n = A.parent;
        }
        return true;
    };
    const int end = R.first + R.count;
    for (int i = R.first; i < end; i++) {
        const int lb = sc.vnodes[i].leaf_begin, le = sc.vnodes[i].leaf_end;  // (-1, -1 for an inner node)
        for (int s = lb; s < le; s++) {
            const TriRec T = sc.tris[s];
            if (!tri_inside(T.x, T.y, T.z, q2)) continue;
            // tri_plane (KDTree.cpp:283-288) for d; for -d the quotient changes its sign and nothing else
            const f3 n = mk3(T.n.x, T.n.y, T.n.z);
            const float dd = dot3(n, d);
            if (dd == 0.f) continue;
            const float q = dot3(n, tri_v2(T) - o) / dd, qb = -q;
            if (!(q < 0.f) && q >= 0.f && (fs < 0 || q < ft || (q == ft && fl != i)) && chain(i, d, pre)) {
                ft = q;
                fs = s;
                fl = i;
            }
            if (!(qb < 0.f) && qb >= 0.f && (bs < 0 || qb < bt || (qb == bt && bl != i)) && chain(i, bd, preb)) {
                bt = qb;
                bs = s;
                bl = i;
            }
        }
    }
    if (bs < 0 || bt < 1e-6f || fs < 0 || ft < 1e-6f) return false;
    VnRec V = sc.vns[bs];
    f3 nn = shading_normal<true>(sc, sc.tris[bs], V, bs, o + bd * bt);
    if (dot3(nn, bd) < 1e-6f) return false;
    V = sc.vns[fs];
    nn = shading_normal<true>(sc, sc.tris[fs], V, fs, o + d * ft);
    return !(dot3(nn, d) < 1e-6f);
}

// the last volumetric object (other than `current`) whose own surface is met from inside both
// backwards and forwards along the ray; -1 = none
#ifndef MIRT_X_FI_ATTR
#define MIRT_X_FI_ATTR inline
#endif
__device__ MIRT_X_FI_ATTR int find_inside(const DevScene& sc, f3 o, f3 d, int current) {
#ifdef MIRT_X_NOFIALL
    return current < 0 ? sc.vols[sc.n_vols - 1].object : -1;
#endif
    const f3 bd = neg(d);
    const RayPre pre = ray_prepare(sc, o, d);
    for (int k = sc.n_vols - 1; k >= 0; k--) {
        const VolRoot R = sc.vols[k];
        if (R.object == current) continue;
        if (R.small) {
            if (inside_small(sc, R, o, d, pre)) return R.object;
            continue;
        }
        int slot = -1;
        float dist = trace_object(sc, R, o, bd, slot);
        if (dist < 1e-6f) continue;
        VnRec V = sc.vns[slot];
        f3 n = shading_normal<true>(sc, sc.tris[slot], V, slot, o + bd * dist);
        if (dot3(n, bd) < 1e-6f) continue;
        dist = trace_object(sc, R, o, d, slot);
        if (dist < 1e-6f) continue;
        V = sc.vns[slot];
        n = shading_normal<true>(sc, sc.tris[slot], V, slot, o + d * dist);
        if (dot3(n, d) < 1e-6f) continue;
        return R.object;
    }
    return -1;
}

// Camera.cpp:35: a camera ray's insideObject is findInsideObject(ray), asked when the ray is made. One lane per
// camera ray of a wavefront, right after k_generate (scenes that hold a volumetric object only); the answer goes
// into the path's packed ids, where k_shade reads every other path's inside object too.
__global__ void __launch_bounds__(256)
k_inside0(DevScene sc, const float* __restrict__ O, const float* __restrict__ D, uint32_t stride, uint32_t n,
          int* __restrict__ ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ids[i] = MIRT_PACK_IDS(-1, find_inside(sc, load3(O, stride, i), load3(D, stride, i), -1));
}

// ---- homogeneous medium helpers, SceneVPMObject.h:22-47 ----------------------------------------
__device__ __forceinline__ float lum_y(f3 c) { return (0.212671f * c.x + 0.715160f * c.y) + 0.072169f * c.z; }
__device__ __forceinline__ f3 sig_t(const DevObject& o) { return mk3(o.sigma_t[0], o.sigma_t[1], o.sigma_t[2]); }
__device__ __forceinline__ f3 vpm_transmittance(const DevObject& o, float dist) {
    return mk3(expf(-o.sigma_t[0] * dist), expf(-o.sigma_t[1] * dist), expf(-o.sigma_t[2] * dist));
}
__device__ __forceinline__ float vpm_p_medium(const DevObject& o, float s) {
    return lum_y(mk3(o.sigma_t[0] * expf(-o.sigma_t[0] * s), o.sigma_t[1] * expf(-o.sigma_t[1] * s),
                     o.sigma_t[2] * expf(-o.sigma_t[2] * s)));
}
__device__ __forceinline__ float vpm_p_surface(const DevObject& o, float d) { return lum_y(vpm_transmittance(o, d)); }
// HGPhaseFunc.h:11-17 = HGPhaseSampler.h:63-69
__device__ __forceinline__ float hg_value(float g, float cosTheta) {
    const float temp = (1.f + g * g) - 2.f * g * cosTheta;
    return (float)(1.0 / (4.0 * 3.14159265358979) * (double)(1.f - g * g) / (double)(temp * sqrtf(temp)));
}

// what a scattering event hands to the common tail of k_shade
struct Scatter {
    f3 pos;            // the new vertex
    f3 nd;             // outgoing direction; zero = the path ends here
    f3 color;
    float dirProb, originProb;
    int contact, inside;  // of the outgoing ray
    bool cos_term;     // the new vertex has a cosine term against `nn` (Ray.cpp:71-78)
    bool emitted;      // the path ended on an emitter: score it
    bool random;       // directionSampleType == Ray::RANDOM (Ray.h:46 default; DEFINITE for the diffuse
                       // pass-through, DiffuseMaterial.cpp:42, and every refractive outcome, RefractiveMaterial.cpp:31)
    f3 nn;             // contact normal (when cos_term)
};

// Fresnel split shared by RefractiveMaterial.cpp:48-101 and SceneVPMObject.cpp:23-86, 151-213.
// `clamp_cos`: the refractive material clamps the cosine before acos, the medium does not;
// `unit_refl`: the medium normalises the mirror direction before use.
// Sets nd, dirProb, through and `factor`: on total reflection (total = true) the colour is DIVIDED
// by factor (the cosine term), otherwise MULTIPLIED by it, exactly as the reference rounds them.
// Draws one number unless totally reflected.
__device__ __forceinline__ void fresnel_split(f3 nn, f3 d, float cur_n, float next_n, bool clamp_cos, bool unit_refl,
                                              bool surface_cos, uint32_t key, uint32_t& ctr, f3& nd,
                                              float& factor, float& dirProb, bool& through, bool& total) {
    f3 refl = (nn * (-dot3(nn, d))) * 2.f + d;
    if (unit_refl) refl = normalize3(refl);
    float c = dot3(d, nn);
    if (clamp_cos) c = clampf(c, -1.f, 1.f);
    const float theta = acosf(c);
    const float sin_phi = cur_n / next_n * sinf(theta);
    through = false;
    dirProb = 1.f;
    auto cosine = [&](f3 dir) {  // Ray::getCosineTerm of the outgoing ray
        if (!surface_cos) return 1.f;
        float ct = fabsf(dot3(nn, dir));
        return ct > 1e-6f ? ct : 1e-6f;
    };
    total = sin_phi >= 1.f;
    if (total) {
        nd = refl;
        factor = cosine(nd);
    } else {
        float phi = asinf(sin_phi);
        if ((double)theta > 3.14159265358979 / 2) phi = (float)(3.14159265358979 - (double)phi);
        const f3 axis = normalize3(cross3(nn, d));
        const f3 rdir = normalize3(rotate_about(axis, phi, nn));
        const float cos_t = fabsf(cosf(theta)), cos_p = fabsf(cosf(phi));
        const float a1 = fabsf(cur_n * cos_t - next_n * cos_p) / (cur_n * cos_t + next_n * cos_p);
        const float a2 = fabsf(next_n * cos_t - cur_n * cos_p) / (next_n * cos_t + cur_n * cos_p);
        const float er = (a1 * a1 + a2 * a2) / 2.f;
        const float p = clampf(er, 0.f, 1.f);
        if (rng_float(key, ctr) < p) {
            nd = refl;
            factor = er / cosine(nd);
            dirProb = p;
        } else {
            nd = rdir;
            factor = (1.f - er) / cosine(nd);
            dirProb = 1.f - p;
            through = true;
        }
    }
    nd = normalize3(nd);
}

// What an object's materials read at a point: the constant of the object, or its texture at
// getTexCoord(triangle, at) (DiffuseMaterial.cpp:23,83,99; GlossyMaterial.cpp:11-12,66,74,78)
__device__ __forceinline__ f3 diffuse_color(const DevScene& sc, const DevObject& ob, const TriRec& T, int slot, f3 at) {
    if (ob.tex_diffuse >= 0) return tex_color(sc.textures[ob.tex_diffuse], tex_coord(sc, T, slot, at));
    return mk3(ob.color[0], ob.color[1], ob.color[2]);
}
__device__ __forceinline__ f3 glossy_color(const DevScene& sc, const DevObject& ob, const TriRec& T, int slot, f3 at) {
    if (ob.tex_gcolor >= 0) return tex_color(sc.textures[ob.tex_gcolor], tex_coord(sc, T, slot, at));
    return ob.type == MIRT_OBJ_PHONG ? mk3(ob.color2[0], ob.color2[1], ob.color2[2]) : mk3(ob.color[0], ob.color[1], ob.color[2]);
}
__device__ __forceinline__ float glossy_coeff(const DevScene& sc, const DevObject& ob, const TriRec& T, int slot, f3 at) {
    if (ob.tex_gcoeff >= 0) return tex_color(sc.textures[ob.tex_gcoeff], tex_coord(sc, T, slot, at)).x;
    return ob.coeff;
}

// Material scatter of the object that was HIT (DiffuseMaterial.cpp:4-76, GlossyMaterial.cpp:4-56,
// RefractiveMaterial.cpp:4-104, SceneEmissiveObject.cpp:4-19). `inside` = the incoming ray's inside
// object. VOL: the scene holds volumetric objects, so leaving a refractive object asks
// findInsideObject for the medium outside (RefractiveMaterial.cpp:52-53).
template <bool VOL>
__device__ __forceinline__ Scatter surface_scatter(const DevScene& sc, const DevObject& ob, int hobj, const TriRec& T,
                                                   int slot, f3 nn, f3 pos, f3 d, int inside, uint32_t key, uint32_t& ctr) {
    Scatter r;
    r.pos = pos;
    r.nd = mk3(0, 0, 0);
    r.color = mk3(0, 0, 0);
    r.dirProb = 1.f;
    r.originProb = 1.f;
    r.contact = hobj;
    r.inside = inside;
    r.cos_term = true;
    r.emitted = false;
    r.random = true;
    r.nn = nn;
    f3 col = mk3(ob.color[0], ob.color[1], ob.color[2]);
    float coeff = ob.coeff;
    int type = ob.type;
    if (type == MIRT_OBJ_PHONG) {
        // SceneObject.cpp:366-373 with ScenePhongObject.h:16-19: one of {glossy, diffuse}, equal chance, one
        // draw of the path's stream; the chance is not divided out
        type = rng_float(key, ctr) < 0.5f ? MIRT_OBJ_GLOSSY : MIRT_OBJ_DIFFUSE;
    }
    if (type == MIRT_OBJ_DIFFUSE) {
        col = diffuse_color(sc, ob, T, slot, pos);  // DiffuseMaterial.cpp:23: the texture at the hit point
    } else if (type == MIRT_OBJ_GLOSSY) {
        // GlossyMaterial.cpp:11-12 read the textures at getTexCoord(tid, outRay.origin) before outRay.origin is
        // set (:17): a default-constructed vec3f, (0,0,0)
        col = glossy_color(sc, ob, T, slot, mk3(0, 0, 0));
        coeff = glossy_coeff(sc, ob, T, slot, mk3(0, 0, 0));
    }
    if (type == MIRT_OBJ_EMISSIVE) {
        // SceneEmissiveObject.cpp:4-19: the path ends; radiance only from the front side
        r.color = dot3(d, nn) <= 0.f ? col : mk3(0, 0, 0);
        r.emitted = true;
        r.inside = -1;
    } else if (type == MIRT_OBJ_DIFFUSE) {
        if (dot3(d, nn) > 0.f) {  // DiffuseMaterial.cpp:35-44: back side passes through
            r.nd = d;
            r.color = mk3(1, 1, 1);
            r.random = false;
        } else {
            Frame lf = frame_from_normal(nn);
            float dx = rng_float(key, ctr), dy = rng_float(key, ctr);
            f3 dir = to_world(lf, hemi_cos(1.f, dx, dy));
            float p = max3f(col);
            if (rng_float(key, ctr) < p) {
                bool ok = dot3(nn, d) <= 0.f && dot3(nn, dir) >= 0.f;
                r.color = ok ? div3(col, 3.14159265358979f) : mk3(0, 0, 0);
                r.dirProb = p * cos_pdf(nn, dir, 1.f);
                r.nd = dir;
            }
        }
    } else if (type == MIRT_OBJ_GLOSSY) {
        f3 refl = (nn * (-dot3(nn, d))) * 2.f + d;
        Frame olf = frame_from_normal(refl);
        float dx = rng_float(key, ctr), dy = rng_float(key, ctr);
        f3 dir = to_world(olf, hemi_cos(coeff, dx, dy));
        f3 c = mk3(0, 0, 0);
        if (dot3(nn, d) <= 0.f && dot3(nn, dir) >= 0.f) {
            float rd = dot3(refl, dir);
            if (rd > 0.f) c = col * powf(rd, coeff);
        }
        if (len3(c) != 0.f) {
            float p = max3f(col);
            if (rng_float(key, ctr) < p) {
                r.color = c;
                r.dirProb = cos_pdf(refl, dir, coeff) * p;
                r.nd = dir;
            }
        }
    } else if (type == MIRT_OBJ_REFRACTIVE) {
        r.random = false;
        if (ob.ior < 0.f) {  // mirror
            f3 refl = (nn * (-dot3(nn, d))) * 2.f + d;
            r.nd = refl;
            float ct = fabsf(dot3(nn, r.nd));
            ct = ct > 1e-6f ? ct : 1e-6f;
            r.color = div3(col, ct);
        } else {
            // leaving our own boundary: the medium outside is the enclosing volumetric object
            int outside = hobj;
            if (inside == hobj) outside = VOL ? find_inside(sc, pos, d, hobj) : -1;
            float cur_n = inside >= 0 ? refr_coeff(sc.objects[inside]) : 1.f;
            float next_n = outside >= 0 ? refr_coeff(sc.objects[outside]) : 1.f;
            float factor;
            bool through, total;
            fresnel_split(nn, d, cur_n, next_n, true, false, true, key, ctr, r.nd, factor, r.dirProb, through, total);
            // (total reflection divides, the random branches multiply: RefractiveMaterial.cpp:67,90,96)
            r.color = total ? div3(col, factor) : col * factor;
            if (through) r.inside = outside;
        }
    }
    return r;
}


// SurfaceSampler::genSample / SceneObject::emit: first index whose running sum is >= rnd
// (std::lower_bound, Scene.cpp:94, SceneObject.cpp:152), clamped to the last
__device__ __forceinline__ int lower_bound_clamped(const float* v, int n, float rnd) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v[mid] < rnd) lo = mid + 1;
        else hi = mid;
    }
    return lo >= n ? n - 1 : lo;
}
__device__ __forceinline__ uint32_t connection_key(uint32_t key) { return lowbias32(key ^ 0x5bd1e995U); }

// colorByConnectingLights (PathTracerTest.cpp:85-135) up to, but not including, the visibility test:
// samples the emission sampler (Scene.cpp:91-98, SceneObject.cpp:137-160, SimpleShape.cpp:28-42) and
// returns contribution * MIS weight together with the shadow ray (direction, length). `hob`: the
// vertex's contact object; `d`: direction of the ray that arrived at the vertex (lastRay.direction).
// Six draws of the connection stream at 6 * vertexIndex: object, triangle, two for the point, two
// for the emitted direction that the connection overwrites. Returns false when nothing can arrive.
// `medium` >= 0: the vertex has NO contact object and lies inside that homogeneous medium (a scattering event,
// SceneVPMObject.cpp:91-137): Ray::getBSDF / getDirectionSampleProbDensity then ask the inside object (Ray.cpp:21-24,
// 35-38), i.e. SceneVPMObject::getBSDF = ds * phase(d, out) (SceneVPMObject.cpp:239-246) and albedo * phase pdf
// (:257-267), and there is no cosine at the vertex (PathTracerTest.cpp:112-116). As in the reference, the
// connection segment itself carries no transmittance.
__device__ __forceinline__ bool connect_light(const DevScene& sc, const DevObject& hob, const TriRec& T, int slot, f3 nn,
                                              f3 pos, f3 d, uint32_t key, uint32_t vertex, f3& out_dir, float& out_dist,
                                              f3& contrib, int medium = -1) {
    if (sc.n_lights <= 0) return false;
    const uint32_t ck = connection_key(key);
    uint32_t cc = 6u * vertex;
    float rnd = rng_float(ck, cc) * sc.light_total_weight;
    int li = 0;
    {  // few lights: running sums read through the records
        int lo = 0, hi = sc.n_lights;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sc.lights[mid].weight_sum < rnd) lo = mid + 1;
            else hi = mid;
        }
        li = lo >= sc.n_lights ? sc.n_lights - 1 : lo;
    }
    const DevLight Lg = sc.lights[li];
    rnd = rng_float(ck, cc) * Lg.total_area;
    const int ti = lower_bound_clamped(sc.light_area + Lg.first, Lg.count, rnd);
    const int lslot = sc.light_slot[Lg.first + ti];
    const TriRec LT = sc.tris[lslot];
    const float u1 = rng_float(ck, cc), u2 = rng_float(ck, cc);
    const float su1 = sqrtf(u1);
    const float u = 1.f - su1, v = u2 * su1;
    const f3 y = (tri_v0(LT) * u + tri_v1(LT) * v) + tri_v2(LT) * ((1.f - u) - v);
    const DevObject lob = sc.objects[Lg.object];
    f3 ldir = pos - y;  // lightRay.direction = ray.origin - lightRay.origin
    const float dist = len3(ldir);
    const float dist2 = dist * dist;
    ldir = normalize3(ldir);
    const f3 odir = neg(ldir);
    // lastRay.getBSDF(outRay): DiffuseMaterial.cpp:94-101, GlossyMaterial.cpp:69-79 (times max colour)
    // (the materials' textures at the vertex itself: DiffuseMaterial.cpp:83,99, GlossyMaterial.cpp:66,74,78)
    const f3 col = diffuse_color(sc, hob, T, slot, pos);
    f3 bsdf = mk3(0, 0, 0);
    float pdf = 0.f;
    const bool sides = dot3(nn, d) <= 0.f && dot3(nn, odir) >= 0.f;
    if (medium >= 0) {
        const DevObject mo = sc.objects[medium];
        const float ph = hg_value(mo.g, dot3(d, odir));
        const f3 ds = mk3(mo.sigma_s[0], mo.sigma_s[1], mo.sigma_s[2]);
        bsdf = ds * ph;
        pdf = (lum_y(ds) / lum_y(sig_t(mo))) * ph;
    } else if (hob.type == MIRT_OBJ_DIFFUSE) {
        if (sides) bsdf = div3(col, 3.14159265358979f);
        pdf = max3f(col) * cos_pdf(nn, odir, 1.f);  // DiffuseMaterial.cpp:78-86
    } else if (hob.type == MIRT_OBJ_GLOSSY || hob.type == MIRT_OBJ_PHONG) {
        const f3 gcol = glossy_color(sc, hob, T, slot, pos);
        const float gcoeff = glossy_coeff(sc, hob, T, slot, pos);
        const f3 refl = (nn * (-dot3(nn, d))) * 2.f + d;
        // (GlossyMaterial.cpp:79 multiplies by max(colour) whatever evaluate() returned: a colour that is NaN - the
        // texture read at the NaN coordinate of a degenerate triangle - makes the product NaN, not zero, and the
        // connection goes on with it: scripts/fuzz_pt.py TEX=2, seed 3243)
        f3 c = mk3(0, 0, 0);
        if (sides) {
            const float rd = dot3(refl, odir);
            if (rd > 0.f) c = gcol * powf(rd, gcoeff);
        }
        bsdf = c * max3f(gcol);
        pdf = cos_pdf(refl, odir, gcoeff);  // GlossyMaterial.cpp:58-67
        if (hob.type == MIRT_OBJ_PHONG) {  // SceneObject.cpp:227-248, 341-352: BSDFs add, densities average
            bsdf = (mk3(0, 0, 0) + bsdf) + (sides ? div3(col, 3.14159265358979f) : mk3(0, 0, 0));
            pdf = (pdf + max3f(col) * cos_pdf(nn, odir, 1.f)) / 2.f;
        }
    }
    if (lum_y(bsdf) < 1e-7f) return false;
    const VnRec LV = sc.vns[lslot];
    const f3 nl = shading_normal<true>(sc, LT, LV, lslot, y);
    const float cos_at_light = fminf(fmaxf(0.f, fabsf(dot3(nl, ldir))), 1.f);
    if (cos_at_light < 1e-6f) return false;
    const float cos_to_light = medium >= 0 ? 1.f : fminf(fmaxf(0.f, fabsf(dot3(nn, odir))), 1.f);
    const float dir_pdf_w = Lg.origin_prob * dist2 / cos_at_light;
    const float mis = dir_pdf_w / (dir_pdf_w + pdf);
    const f3 lcol = mk3(lob.color[0], lob.color[1], lob.color[2]);
    const f3 tmp = div3(((lcol * cos_at_light) * bsdf) * cos_to_light, Lg.origin_prob * dist2);
    contrib = tmp * mis;
    out_dir = odir;
    out_dist = dist;
    return true;
}

// One bounce for every live path: score emitter hits, scatter, Russian roulette, write the next ray
// of each surviving path to `out` at a compacted position (wave ballot + block prefix + one atomic
// per block). Results do not depend on the order of `out` (RNG and radiance slots are keyed by pid).
// VOL: the scene holds a homogeneous medium (SceneVPMObject.cpp:4-233): the scatter of a path that
// is inside one is the medium's (MCRenderer.cpp:167-170), camera rays look up their inside object
// (Camera.cpp:35); scenes without media run the VOL = false instantiation, which has none of it.
// NEE: PathTracerTest (PathTracerTest.cpp:20-78) instead of PathTracer: the throughput is the
// reference's `throughput` there (no originProb, no camera cosine, no vignetting compensation), an
// emitter hit is weighted against the light connection that could have found it, and every RANDOM
// vertex writes one shadow ray + its pending contribution to `sh` (compacted like the paths); the
// visibility launch finishes it (ConnectArgs). With VOL the estimator is the reference's inside media too: light
// connections from scattering events through the phase function, and NO division by originProb (PathTracerTest.cpp:70-74).
// Order of a block's survivors in the next bounce's buffer: 0 = as the waves hold them (each wave writes one contiguous
// run), 1 = by the octant of the new direction across the block, 2 = by octant inside each wave. Round 1 introduced 1
// (+1.6 % then). With round 3's kernels and wavefront widths the large k_shade launches run at HBM speed, where the
// scattered stores of 1 cost more than the ordering gives the traversal: 1024^2 x 1024 spp 2 002 (1) / 2 066 (2) / 2 089 (0)
// Msamples/s, 24 spp 1 671 / - / 1 722, PathTracerTest 1 106 / - / 1 132, time inside the traversal launches equal. A batch
// captured from the path tracer and handed to mirt_intersect_dev is traced 3 % faster with 1 (Cornell bounce batch
// 1.62 against 1.68 ms): an order a caller may give its own batches, not one the path tracer should pay for.
#ifndef MIRT_SHADE_OCTANT
#define MIRT_SHADE_OCTANT 0
#endif
#ifndef MIRT_SHADE_BIN
#define MIRT_SHADE_BIN 1
#endif
// threads per k_shade block (a power of two, 128 .. 1024): the paths of a block are regrouped by material before
// they are shaded, so the larger the block the fewer waves straddle two materials, and the longer every wave waits
// at the block's barriers. Measured, 1024^2 x 1024 spp on the Cornell scene: 128 -> 1 386, 256 -> 1 735, 512 -> 1 699,
// 1024 -> 1 633 Msamples/s (without the regrouping, MIRT_SHADE_BIN=0: 1 718).
#ifndef MIRT_SHADE_BLOCK
#define MIRT_SHADE_BLOCK 256
#endif
constexpr int SHADE_BLOCK = MIRT_SHADE_BLOCK, SHADE_WAVES = MIRT_SHADE_BLOCK / 64;
// TWO (run_chained): two-ended compaction. Every block ends on one atomic for its room in the output, and atomics on
// one 64-byte line are served one per 11.4 ns whoever asks (scripts/experiments/atomic_rate.hip): 143 000 blocks of a
// 32 Mi wavefront's first bounce = 1.6 ms, which is what that launch took (1.13 ms with half of the atomics left out).
// So blocks with an even index count upwards from the middle of the buffer (n_out[0]) and blocks with an odd index
// downwards (n_out[32], 128 B away): the survivors are the dense range [mid - n_out[32], mid + n_out[0]), whose start
// and length k_live hands to the consumers (off_in here, off_ptr of the traversal kernels).
template <bool VOL, bool NEE = false, bool TWO = false>
__global__ void __launch_bounds__(MIRT_SHADE_BLOCK)
k_shade(DevScene sc, DevCamera cam, PathBuf in, const uint32_t* __restrict__ n_in, uint32_t n_max,
        PathBuf out, uint32_t* __restrict__ n_out, float* __restrict__ L, uint32_t L_stride,
        uint32_t max_depth, ShadowBuf sh = ShadowBuf{}, uint32_t* __restrict__ n_sh = nullptr,
        const uint32_t* __restrict__ off_in = nullptr) {
    __shared__ uint32_t s_wcnt[SHADE_WAVES];
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_wcnt2[SHADE_WAVES];
    __shared__ uint32_t s_base2;
    __shared__ uint32_t s_off[SHADE_WAVES][8];  // start of every (wave, class) run inside the block, class-major
    const uint32_t n = n_in ? min(*n_in, n_max) : n_max;
    if (blockIdx.x * blockDim.x >= n) return;  // the grid is sized for n_max; whole block, before any barrier
    if (off_in) shift_paths(in, *off_in);
#if MIRT_SHADE_BIN
    // ---- which path this lane shades: the block's 256 paths, regrouped by the material that will run ------------
    // Neighbouring paths hit different objects, and every material is its own long branch (Lambert / Phong lobe /
    // mirror / Fresnel, each with its own sampling and transcendental calls): taken in batch order a wave executes
    // most of them one after the other (2 260 VALU instructions issued per wave at 28 of 64 lanes, measured). Results
    // are keyed by the path id, so the order inside a block is free: a counting sort by object type through LDS.
    uint32_t i;
    {
        __shared__ uint32_t s_tcnt[SHADE_WAVES][8];
        __shared__ uint16_t s_perm[SHADE_BLOCK];
        const uint32_t own = blockIdx.x * blockDim.x + threadIdx.x;
        int mtype = 7;  // no hit / beyond the count: last, does nothing
        if (own < n) {
            const int slot = in.slot[own];
            if (slot >= 0) mtype = sc.objects[__float_as_int(sc.vns[slot].n0.w)].type & 7;
        }
        const uint32_t ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
        uint32_t before = 0;
        for (int k = 0; k < 8; k++) {
            const unsigned long long mk = __ballot(mtype == k);
            if (ln == 0) s_tcnt[wv][k] = __popcll(mk);
            if (k == mtype) before = __popcll(mk & ((1ull << ln) - 1ull));
        }
        __syncthreads();
        if (threadIdx.x < SHADE_WAVES * 8) {  // one thread per (class, wave): everything that sorts before it
            const uint32_t k = threadIdx.x / SHADE_WAVES, w2 = threadIdx.x % SHADE_WAVES;
            uint32_t off = 0;
            for (uint32_t kk = 0; kk < k; kk++)
                for (uint32_t ww = 0; ww < SHADE_WAVES; ww++) off += s_tcnt[ww][kk];
            for (uint32_t ww = 0; ww < w2; ww++) off += s_tcnt[ww][k];
            s_off[w2][k] = off;
        }
        __syncthreads();
        s_perm[s_off[wv][mtype] + before] = (uint16_t)threadIdx.x;
        __syncthreads();
        i = blockIdx.x * blockDim.x + s_perm[threadIdx.x];
    }
#else
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
#endif

    bool alive = false;
    f3 no = mk3(0, 0, 0), nd = mk3(0, 0, 0), thr = mk3(0, 0, 0);
    uint32_t key = 0, ctr = 0, pid = 0, depth = 0;
    int nids = 0;
    float* Lg = L;  // the radiance buffer of this path's wavefront (L ring)
    bool shadow = false;  // NEE: this lane emits a shadow ray
    f3 sh_o = mk3(0, 0, 0), sh_d = mk3(0, 0, 0), sh_c = mk3(0, 0, 0);
    float sh_lim = 0.f, lpw = -1.f;
    int sh_ids = 0;

    if (i < n) {
        const int slot = in.slot[i];
        if (slot >= 0) {
            const f3 o = load3(in.o, in.stride, i), d = load3(in.d, in.stride, i);
            const float t = in.t[i];
            thr = load3(in.thr, in.stride, i);
            key = in.key[i];
            ctr = in.ctr[i];
            depth = ctr >> CTR_DEPTH_SHIFT;  // scatter events behind this path (0: a camera ray's hit)
            ctr &= CTR_MASK;
            pid = in.pid[i];
            Lg = L + (size_t)(pid >> PID_GEN_SHIFT) * 3u * (size_t)L_stride;
            int inside = unpack_inside(in.ids[i]);
            // (a camera ray's inside object, Camera.cpp:35, was put there by k_inside0)
            const TriRec T = sc.tris[slot];
            const VnRec V = sc.vns[slot];
            const int hobj = __float_as_int(V.n0.w);
            const DevObject ob = sc.objects[hobj];
            const f3 pos = o + d * t;
            const f3 nn = shading_normal<true>(sc, T, V, slot, pos);

            // whose scatter() runs (MCRenderer.cpp:167-174): the inside object's if there is one; a
            // refractive inside object hands a hit on another object to that object
            // (RefractiveMaterial.cpp:42-46)
            int sobj = inside >= 0 ? inside : hobj;
            if (VOL && sobj != hobj) {
                const DevObject so = sc.objects[sobj];
                if (so.type != MIRT_OBJ_VPM) sobj = hobj;
            }
            Scatter sr;
            if (VOL && sc.objects[sobj].type == MIRT_OBJ_VPM) {
                // ---- SceneVPMObject::scatter -------------------------------------------------------
                const DevObject vo = sc.objects[sobj];
                // HomoMediaDistSampler.h:9-11, drawn before the case split
                const float u = rng_float(key, ctr);
#ifdef MIRT_X_FLOATMED
                const float samp = -logf(1.f - u) / lum_y(sig_t(vo));
#else
                const float samp = (float)(-log(1.0 - (double)u) / (double)lum_y(sig_t(vo)));
#endif
                const bool go_in = (hobj == sobj) && (inside != sobj);
                const bool be_in = (inside == sobj);
                const bool out_of_vol = samp >= t;
                sr.pos = pos;
                sr.nd = mk3(0, 0, 0);
                sr.color = mk3(0, 0, 0);
                sr.dirProb = 1.f;
                sr.originProb = 1.f;
                sr.contact = -1;
                sr.inside = inside;
                sr.cos_term = false;
                sr.emitted = false;
                sr.random = true;  // SceneVPMObject.cpp:7; DEFINITE at the medium's boundary (:44, 73, 81, 171, 200, 209)
                sr.nn = nn;
                // The surface met inside this medium is ANOTHER medium's boundary (nested or overlapping media):
                // SceneVPMObject.cpp:215-228 hands the ray, now at the hit point with intersectDist 0, to
                // intersectObject->scatter(), which is SceneVPMObject::scatter of the object that was hit: it draws its
                // own free-flight distance (:8-9, unused) and finds intersectObject == this, insideObject != this,
                // i.e. "go in" (:16-86) from the medium the path is in; the caller then multiplies originProb by its
                // own P_surface (:227).
                const bool other_medium = be_in && out_of_vol && hobj != sobj && ob.type == MIRT_OBJ_VPM;
                if (go_in || other_medium) {
                    if (other_medium) (void)rng_float(key, ctr);
                    sr.random = false;
                    const float cur_n = inside >= 0 ? refr_coeff(sc.objects[inside]) : 1.f;
                    float factor;
                    bool through, total;
                    fresnel_split(nn, d, cur_n, refr_coeff(ob), false, true, false, key, ctr, sr.nd, factor,
                                  sr.dirProb, through, total);
                    sr.color = total ? mk3(1, 1, 1) : mk3(factor, factor, factor);  // cosine term is 1
                    sr.contact = hobj;
                    if (through) sr.inside = hobj;
                    if (other_medium) sr.originProb = vpm_p_surface(vo, t);
                } else if (be_in && !out_of_vol) {
                    Frame lf = frame_from_normal(d);
                    sr.pos = o + d * samp;
                    // HGPhaseSampler.h:13-34
                    const float e1 = rng_float(key, ctr), e2 = rng_float(key, ctr);
                    float cosTheta;
                    if (fabsf(vo.g) < 1e-5f) {
                        cosTheta = 1.f - 2.f * e1;
                    } else {
                        const float sqrTerm = (1.f - vo.g * vo.g) / ((1.f - vo.g) + 2.f * vo.g * e1);
                        cosTheta = ((1.f + vo.g * vo.g) - sqrTerm * sqrTerm) / (2.f * vo.g);
                    }
                    const float sinTheta = sqrtf(1.f - cosTheta * cosTheta);
#ifdef MIRT_X_FLOATMED
                    const float sinPhi = sinf(6.2831853f * e2);
                    const float cosPhi = cosf(6.2831853f * e2);
#else
                    const float sinPhi = (float)sin(2.0 * 3.14159265358979 * (double)e2);
                    const float cosPhi = (float)cos(2.0 * 3.14159265358979 * (double)e2);
#endif
                    const f3 dir = to_world(lf, mk3(sinTheta * cosPhi, cosTheta, sinTheta * sinPhi));
                    const float ph = hg_value(vo.g, dot3(d, dir));
                    const f3 ds = mk3(vo.sigma_s[0], vo.sigma_s[1], vo.sigma_s[2]);
                    const float albedo = lum_y(ds) / lum_y(sig_t(vo));
                    const float rander = rng_float(key, ctr);
                    sr.inside = sobj;
                    sr.originProb = vpm_p_medium(vo, samp);
                    if (rander < albedo) {
                        sr.nd = dir;
                        sr.dirProb = albedo * hg_value(vo.g, dot3(lf.n, dir));
                        sr.color = mk3(ph, ph, ph) * ds;
                    }
                } else if (be_in && out_of_vol) {
                    if (hobj == sobj) {  // leaving through the medium's own boundary
                        sr.random = false;
                        const int outside = find_inside(sc, pos, d, sobj);
                        const float next_n = outside >= 0 ? refr_coeff(sc.objects[outside]) : 1.f;
                        float factor;
                        bool through, total;
                        fresnel_split(nn, d, refr_coeff(vo), next_n, false, true, false, key, ctr, sr.nd, factor,
                                      sr.dirProb, through, total);
                        sr.color = total ? mk3(1, 1, 1) : mk3(factor, factor, factor);
                        sr.contact = sobj;
                        sr.inside = through ? outside : sobj;
                        sr.originProb = vpm_p_surface(vo, t);
                    } else {  // a surface inside the medium: that object's scatter, from the hit point
                        sr = surface_scatter<VOL>(sc, ob, hobj, T, slot, nn, pos, d, sobj, key, ctr);
                        sr.originProb = sr.originProb * vpm_p_surface(vo, t);
                    }
                }
            } else {
                sr = surface_scatter<VOL>(sc, ob, hobj, T, slot, nn, pos, d, inside, key, ctr);
            }

            if (NEE) {
                // vertex i = depth + 1 of the eye path, reached from x_{i-1} = o along d
                const bool same = depth >= 1 && sr.pos.x == o.x && sr.pos.y == o.y && sr.pos.z == o.z;  // PathTracerTest.cpp:65
                // the vertex is in the path when it ended there or the depth cap lets it continue
                // (MCRenderer.cpp:181-191: the zero-direction test comes before the cap)
                const bool counted = len3(sr.nd) < 0.5f || depth + 1 <= max_depth;
                if (!same && counted) {
                    if (sr.emitted) {  // PathTracerTest.cpp:34-57 (the colour whatever the side, getColor())
                        float mis = 1.f;
                        const float last = in.lpw[i];
                        if (depth >= 1 && last >= 0.f) {
                            const float cosine = dot3(nn, neg(d));
                            const float dist = len3(sr.pos - o);
                            const float dir_pdf_w = ob.emit_pdf_a * dist * dist / fabsf(cosine);
                            mis = last / (last + dir_pdf_w);
                        }
                        const f3 add = (thr * mk3(ob.color[0], ob.color[1], ob.color[2])) * mis;
                        const uint32_t px = pid & PID_MASK;
                        Lg[px] += add.x;
                        Lg[L_stride + px] += add.y;
                        Lg[2 * L_stride + px] += add.z;
                    } else {
                        if (sr.random) {  // PathTracerTest.cpp:59-63
                            f3 cdir, contrib;
                            float cdist;
                            // (a vertex without a contact object is a scattering event inside the medium sr.inside)
                            const int medium = (VOL && sr.contact < 0) ? sr.inside : -1;
                            if (connect_light(sc, ob, T, slot, nn, sr.pos, d, key, depth + 1, cdir, cdist, contrib, medium)) {
                                shadow = true;
                                sh_o = sr.pos;
                                sh_d = cdir;
                                sh_lim = cdist;
                                sh_c = thr * contrib;
                                sh_ids = MIRT_PACK_IDS(sr.contact, sr.inside);
                            }
                        }
                        if (!(len3(sr.nd) < 0.5f)) {  // PathTracerTest.cpp:68-76 (a NaN direction goes on, MCRenderer.cpp:181)
                            lpw = sr.random ? sr.dirProb : -1.f;
                            float ct = 1.f;  // Ray::getCosineTerm (Ray.cpp:71-78): 1 without a contact object / for a medium's boundary
                            if (!VOL || sr.cos_term) ct = fabsf(dot3(sr.nn, sr.nd));
                            ct = ct > 1e-6f ? ct : 1e-6f;
                            thr = thr * div3(sr.color * ct, sr.dirProb);
                            if (inside >= 0) {  // eyePath[i - 1].getRadianceDecay (PathTracerTest.cpp:72-74); originProb is never divided out there
                                const DevObject io = sc.objects[inside];
                                const float dist = fmaxf(len3(sr.pos - o), 1e-5f);
                                if (io.type == MIRT_OBJ_REFRACTIVE)
                                    thr = thr * mk3(powf(io.decay[0], dist), powf(io.decay[1], dist), powf(io.decay[2], dist));
                                else if (VOL && io.type == MIRT_OBJ_VPM)
                                    thr = thr * vpm_transmittance(io, dist);
                            }
                            alive = depth + 1 <= max_depth;
                            no = sr.pos;
                            nd = sr.nd;
                            nids = MIRT_PACK_IDS(sr.contact, sr.inside);
                        }
                    }
                }
            } else {
            // decay along the finished segment (Ray.cpp:66-69): RefractiveMaterial.h:15 or the
            // medium's transmittance (SceneVPMObject.cpp:235-237), over the distance between vertices
            if (inside >= 0) {
                const DevObject io = sc.objects[inside];
                if (io.type == MIRT_OBJ_REFRACTIVE) {
                    float dist = len3(sr.pos - o);
                    thr = thr * mk3(powf(io.decay[0], dist), powf(io.decay[1], dist), powf(io.decay[2], dist));
                } else if (VOL && io.type == MIRT_OBJ_VPM) {
                    thr = thr * vpm_transmittance(io, len3(sr.pos - o));
                }
            }
            // PathTracer.cpp:31: color / directionProb / originProb
            thr = thr * div3(div3(sr.color, sr.dirProb), sr.originProb);
            nd = sr.nd;
            if (sr.emitted) {
                const uint32_t px = pid & PID_MASK;
                float vd = vignette_div(cam, tile_pixel(cam, px % ((uint32_t)cam.tw * (uint32_t)cam.th)));
                f3 rad = div3(thr, vd);
                Lg[px] = rad.x;
                Lg[L_stride + px] = rad.y;
                Lg[2 * L_stride + px] = rad.z;
            } else if (!(len3(nd) < 0.5f)) {
                // (MCRenderer.cpp:181 ends the path when length < 0.5: a NaN direction, e.g. from a bump-mapped
                // triangle whose first edge runs along u (SceneObject.cpp:272 divides by duv1.y), is traced and misses)
                // Ray.cpp:71-78: cosine term of the new vertex against its contact normal
                float ct = 1.f;
                if (sr.cos_term) ct = fabsf(dot3(sr.nn, nd));
                ct = ct > 1e-6f ? ct : 1e-6f;
                thr = thr * ct;
                // MCRenderer.cpp:181-191
                alive = depth + 1 <= max_depth;
                no = sr.pos;
                nids = MIRT_PACK_IDS(sr.contact, sr.inside);
            }
            }  // !NEE
        }
    }

    // ---- compaction: ballot inside the wave, prefix over the block's waves, one atomic per block
    const unsigned long long m = __ballot(alive);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wcnt[w] = __popcll(m);
#if MIRT_SHADE_OCTANT == 1
    // survivors of a block leave ordered by the octant of their new direction: the next traversal's waves
    // (which take consecutive rays) then hold rays that agree on the order in which children are visited
    __shared__ uint32_t s_oct[SHADE_WAVES][8];
    const int oct = (nd.x < 0.f ? 1 : 0) | (nd.y < 0.f ? 2 : 0) | (nd.z < 0.f ? 4 : 0);
    uint32_t in_wave_before = 0;
    for (int k = 0; k < 8; k++) {
        const unsigned long long mk = __ballot(alive && oct == k);
        if (lane == 0) s_oct[w][k] = __popcll(mk);
        if (k == oct) in_wave_before = __popcll(mk & ((1ull << lane) - 1ull));
    }
#elif MIRT_SHADE_OCTANT == 2
    // the same inside each wave only: a wave's survivors stay one contiguous run (whole lines stored), ordered by octant
    {
        const int oct = (nd.x < 0.f ? 1 : 0) | (nd.y < 0.f ? 2 : 0) | (nd.z < 0.f ? 4 : 0);
        uint32_t before = 0, mine = 0;
        for (int k = 0; k < 8; k++) {
            const unsigned long long mk = __ballot(alive && oct == k);
            if (k < oct) before += __popcll(mk);
            if (k == oct) mine = __popcll(mk & ((1ull << lane) - 1ull));
        }
        rank = before + mine;
    }
#endif
    const unsigned long long m2 = NEE ? __ballot(shadow) : 0ull;
    if (NEE && lane == 0) s_wcnt2[w] = __popcll(m2);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0, tot2 = 0;
        for (int k = 0; k < SHADE_WAVES; k++) tot += s_wcnt[k];
        if (TWO) {
            const uint32_t mid = (gridDim.x >> 1) * (uint32_t)SHADE_BLOCK;  // (k_live is given the same value)
            if (!tot) s_base = 0u;
            else if (blockIdx.x & 1u) s_base = mid - (atomicAdd(n_out + 32, tot) + tot);
            else s_base = mid + atomicAdd(n_out, tot);
        } else {
            s_base = tot ? atomicAdd(n_out, tot) : 0u;
        }
        if (NEE) {
            for (int k = 0; k < SHADE_WAVES; k++) tot2 += s_wcnt2[k];
            s_base2 = tot2 ? atomicAdd(n_sh, tot2) : 0u;
        }
    }
#if MIRT_SHADE_OCTANT == 1
    // (s_off is free again: every thread read its entry of the material sort before the barriers above)
    if (threadIdx.x < SHADE_WAVES * 8) {
        const uint32_t k = threadIdx.x / SHADE_WAVES, w2 = threadIdx.x % SHADE_WAVES;
        uint32_t off = 0;
        for (uint32_t kk = 0; kk < k; kk++)
            for (uint32_t ww = 0; ww < SHADE_WAVES; ww++) off += s_oct[ww][kk];
        for (uint32_t ww = 0; ww < w2; ww++) off += s_oct[ww][k];
        s_off[w2][k] = off;
    }
#endif
    __syncthreads();
    if (NEE && shadow) {
        uint32_t wb = 0;
        for (uint32_t k = 0; k < w; k++) wb += s_wcnt2[k];
        const uint32_t j = s_base2 + wb + (uint32_t)__popcll(m2 & ((1ull << lane) - 1ull));
        store3(sh.o, sh.stride, j, sh_o);
        store3(sh.d, sh.stride, j, sh_d);
        store3(sh.c, sh.stride, j, sh_c);
        sh.lim[j] = sh_lim;
        sh.ids[j] = sh_ids;
        sh.pid[j] = pid;
    }
    if (alive) {
        uint32_t wb = 0;
#if MIRT_SHADE_OCTANT == 1
        wb = s_off[w][oct];
        rank = in_wave_before;
#else
        for (uint32_t k = 0; k < w; k++) wb += s_wcnt[k];
#endif
        const uint32_t j = s_base + wb + rank;
        store3(out.o, out.stride, j, no);
        store3(out.d, out.stride, j, nd);
        store3(out.thr, out.stride, j, thr);
        out.key[j] = key;
        out.ctr[j] = ctr | ((depth + 1u) << CTR_DEPTH_SHIFT);
        out.ids[j] = nids;
        out.pid[j] = pid;
        if (NEE) out.lpw[j] = lpw;
    }
}

// (The pending contribution of an unoccluded light connection is added to its path's radiance, PathTracerTest.cpp:62
// after MCRenderer::testVisibility and :97-98, by the visibility query's own kernels: ConnectArgs in mirt_dev.h. One
// shadow ray per path and bounce, so no two lanes touch the same L entry.)

// rgb_sum[p] = (..((rgb_sum[p] + L_0) + L_1) + ..): the launch's samples are added ONE BY ONE, in sample order, to what
// the accumulator already holds, so the image is the same fp32 sum however the samples are cut into wavefronts
// (spp_per_launch, the automatic width, a tile's smaller wavefronts): deterministic, and bit-equal across those choices.
__global__ void __launch_bounds__(256)
k_accumulate(const float* __restrict__ L, uint32_t L_stride, uint32_t wh, uint32_t n_samples,
             float* __restrict__ rgb_sum) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= wh) return;
    float r = rgb_sum[3 * p], g = rgb_sum[3 * p + 1], b = rgb_sum[3 * p + 2];
    for (uint32_t s = 0; s < n_samples; s++) {
        uint32_t k = s * wh + p;
        r += L[k];
        g += L[L_stride + k];
        b += L[2 * L_stride + k];
    }
    rgb_sum[3 * p] = r;
    rgb_sum[3 * p + 1] = g;
    rgb_sum[3 * p + 2] = b;
}

// Wavefront chaining (mirt.hip run_chained): moves the n = min(*n_src, n_max) intersected paths src[0, n) to
// dst[dst_off, dst_off + n) and leaves cnt_base + n in *cnt_out: the survivors of a wavefront's cut into the carry
// buffer, and from there behind the next wavefront's camera hits.
__global__ void __launch_bounds__(256)
k_carry(PathBuf src, const uint32_t* __restrict__ n_src, uint32_t n_max, PathBuf dst, uint32_t dst_off,
        uint32_t* __restrict__ cnt_out, uint32_t cnt_base, const uint32_t* __restrict__ off_src = nullptr) {
    const uint32_t n = min(*n_src, n_max);
    if (off_src) shift_paths(src, *off_src);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *cnt_out = cnt_base + n;
    if (i >= n) return;
    const uint32_t j = dst_off + i;
    store3(dst.o, dst.stride, j, load3(src.o, src.stride, i));
    store3(dst.d, dst.stride, j, load3(src.d, src.stride, i));
    store3(dst.thr, dst.stride, j, load3(src.thr, src.stride, i));
    dst.t[j] = src.t[i];
    dst.slot[j] = src.slot[i];
    dst.key[j] = src.key[i];
    dst.ctr[j] = src.ctr[i];
    dst.ids[j] = src.ids[i];
    dst.pid[j] = src.pid[i];
    if (src.lpw && dst.lpw) dst.lpw[j] = src.lpw[i];  // (PathTracerTest: the density of the last sampled direction)
}

// two-ended compaction (k_shade<.., TWO>): length and start of the range the two counters describe
__global__ void k_live(const uint32_t* __restrict__ ab, uint32_t mid, uint32_t* __restrict__ n_out,
                       uint32_t* __restrict__ off_out) {
    *n_out = ab[0] + ab[32];
    *off_out = mid - ab[32];
}

// a[i] += b[i]: the reduce step of mirt_render_pt_multi (context k's radiance sums into context 0's)
__global__ void __launch_bounds__(256) k_add(float* __restrict__ a, const float* __restrict__ b, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}

}  // namespace mirt
