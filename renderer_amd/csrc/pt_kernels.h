// pt_kernels.h — wavefront path tracing kernels: generate, shade (+ballot/prefix-sum compaction),
// accumulate. Replaces the per-pixel loop PathTracer.cpp:17-44 and MCRenderer::samplePath
// (MCRenderer.cpp:149-224); the materials follow DiffuseMaterial.cpp:4-76, GlossyMaterial.cpp:4-56,
// RefractiveMaterial.cpp:4-104 and SceneEmissiveObject.cpp:4-19. No path vertex list is stored:
// the throughput  prod_i color_i/(dirProb_i*originProb_i) * cos_i * decay_i  of PathTracer.cpp:28-40
// is carried forward and scored when the path ends on an emitter (PathTracer.cpp:25-26).
#pragma once
#include "mirt_dev.h"

#pragma clang fp contract(off)

namespace mirt {

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
    uint32_t stride;
};

struct DevCamera {
    f3 pos, front, right, up;  // up = orthoUp = right x front (Camera.cpp:6-10)
    int width, height;
    float sight;               // float(sightDist)
    double half_w, half_h;     // width/2.0, height/2.0 (double, Camera.cpp:21)
};

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

// Camera ray batch. ray i: sample spp_begin + i / (W*H), pixel i % (W*H).
__global__ void __launch_bounds__(256)
k_generate(DevCamera cam, uint32_t n, uint32_t spp_begin, uint32_t seed, float* __restrict__ O,
           float* __restrict__ D, uint32_t stride, PathBuf pb, int fill_state) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t wh = (uint32_t)cam.width * (uint32_t)cam.height;
    uint32_t pixel = i % wh, sample = spp_begin + i / wh;
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
        store3(pb.thr, pb.stride, i, mk3(c0, c0, c0));
        pb.key[i] = key;
        pb.ctr[i] = ctr;
        pb.ids[i] = MIRT_PACK_IDS(-1, -1);
        pb.pid[i] = i;
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
__device__ __forceinline__ float max3f(f3 c) { return fmaxf(c.x, fmaxf(c.y, c.z)); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}
// smallFuncs.h:11-24 applied to a direction
__device__ __forceinline__ f3 rotate_about(f3 axis, float angle, f3 v) {
    float c = cosf(angle), s = sinf(angle), ic = 1.f - c;
    float x = axis.x, y = axis.y, z = axis.z;
    return mk3(v.x * (c + ic * x * x) + v.y * (ic * x * y - s * z) + v.z * (ic * x * z + s * y),
               v.x * (ic * x * y + s * z) + v.y * (c + ic * y * y) + v.z * (ic * y * z - s * x),
               v.x * (ic * x * z - s * y) + v.y * (ic * y * z + s * x) + v.z * (c + ic * z * z));
}
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

// One bounce for every live path: score emitter hits, scatter, Russian roulette, write the next ray
// of each surviving path to `out` at a compacted position (wave ballot + block prefix + one atomic
// per block). Results do not depend on the order of `out` (RNG and radiance slots are keyed by pid).
__global__ void __launch_bounds__(256)
k_shade(DevScene sc, DevCamera cam, PathBuf in, const uint32_t* __restrict__ n_in, uint32_t n_max,
        PathBuf out, uint32_t* __restrict__ n_out, float* __restrict__ L, uint32_t L_stride,
        uint32_t depth, uint32_t max_depth) {
    __shared__ uint32_t s_wcnt[4];
    __shared__ uint32_t s_base;
    const uint32_t n = n_in ? min(*n_in, n_max) : n_max;
    if (blockIdx.x * blockDim.x >= n) return;  // the grid is sized for n_max; whole block, before any barrier
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;

    bool alive = false;
    f3 no = mk3(0, 0, 0), nd = mk3(0, 0, 0), thr = mk3(0, 0, 0);
    uint32_t key = 0, ctr = 0, pid = 0;
    int nids = 0;

    if (i < n) {
        const int slot = in.slot[i];
        if (slot >= 0) {
            const f3 o = load3(in.o, in.stride, i), d = load3(in.d, in.stride, i);
            const float t = in.t[i];
            thr = load3(in.thr, in.stride, i);
            key = in.key[i];
            ctr = in.ctr[i];
            pid = in.pid[i];
            const int inside = unpack_inside(in.ids[i]);
            const TriRec T = sc.tris[slot];
            const VnRec V = sc.vns[slot];
            const int hobj = __float_as_int(V.n0.w);
            const DevObject ob = sc.objects[hobj];
            const f3 pos = o + d * t;
            // decay along the finished segment, Ray.cpp:66-69 + RefractiveMaterial.h:15
            if (inside >= 0) {
                const DevObject io = sc.objects[inside];
                if (io.type == MIRT_OBJ_REFRACTIVE) {
                    float dist = len3(pos - o);
                    thr = thr * mk3(powf(io.decay[0], dist), powf(io.decay[1], dist),
                                    powf(io.decay[2], dist));
                }
            }
            const f3 nn = world_normal(T, V, ob.has_vn != 0, pos);
            const f3 col = mk3(ob.color[0], ob.color[1], ob.color[2]);
            f3 color = mk3(0, 0, 0);
            float dirProb = 1.f;
            int ninside = inside;
            bool term = true;
            nd = mk3(0, 0, 0);
            if (ob.type == MIRT_OBJ_EMISSIVE) {
                // SceneEmissiveObject.cpp:4-19: the path ends; radiance only from the front side
                color = dot3(d, nn) <= 0.f ? col : mk3(0, 0, 0);
                f3 w = div3(color, dirProb);
                thr = thr * w;
                float vd = vignette_div(cam, pid % ((uint32_t)cam.width * (uint32_t)cam.height));
                f3 rad = div3(thr, vd);
                L[pid] = rad.x;
                L[L_stride + pid] = rad.y;
                L[2 * L_stride + pid] = rad.z;
            } else if (ob.type == MIRT_OBJ_DIFFUSE) {
                if (dot3(d, nn) > 0.f) {  // DiffuseMaterial.cpp:35-44: back side passes through
                    nd = d;
                    color = mk3(1, 1, 1);
                    term = false;
                } else {
                    Frame lf = frame_from_normal(nn);
                    float dx = rng_float(key, ctr), dy = rng_float(key, ctr);
                    f3 dir = to_world(lf, hemi_cos(1.f, dx, dy));
                    float p = max3f(col);
                    if (rng_float(key, ctr) < p) {
                        bool ok = dot3(nn, d) <= 0.f && dot3(nn, dir) >= 0.f;
                        color = ok ? div3(col, 3.14159265358979f) : mk3(0, 0, 0);
                        dirProb = p * cos_pdf(nn, dir, 1.f);
                        nd = dir;
                        term = false;
                    }
                }
            } else if (ob.type == MIRT_OBJ_GLOSSY) {
                f3 refl = (nn * (-dot3(nn, d))) * 2.f + d;
                Frame olf = frame_from_normal(refl);
                float dx = rng_float(key, ctr), dy = rng_float(key, ctr);
                f3 dir = to_world(olf, hemi_cos(ob.coeff, dx, dy));
                f3 c = mk3(0, 0, 0);
                if (dot3(nn, d) <= 0.f && dot3(nn, dir) >= 0.f) {
                    float rd = dot3(refl, dir);
                    if (rd > 0.f) c = col * powf(rd, ob.coeff);
                }
                if (len3(c) != 0.f) {
                    float p = max3f(col);
                    if (rng_float(key, ctr) < p) {
                        color = c;
                        dirProb = cos_pdf(refl, dir, ob.coeff) * p;
                        nd = dir;
                        term = false;
                    }
                }
            } else if (ob.type == MIRT_OBJ_REFRACTIVE) {
                f3 refl = (nn * (-dot3(nn, d))) * 2.f + d;
                term = false;
                if (ob.ior < 0.f) {  // mirror
                    nd = refl;
                    float ct = fabsf(dot3(nn, nd));
                    ct = ct > 1e-6f ? ct : 1e-6f;
                    color = div3(col, ct);
                } else {
                    float theta = acosf(clampf(dot3(d, nn), -1.f, 1.f));
                    // leaving our own boundary: the medium outside is the enclosing volumetric object
                    // (Scene.cpp:354-390); only homogeneous media count and none are resident yet
                    int outside = (inside == hobj) ? -1 : hobj;
                    float cur_n = inside >= 0 ? refr_coeff(sc.objects[inside]) : 1.f;
                    float next_n = outside >= 0 ? refr_coeff(sc.objects[outside]) : 1.f;
                    float sin_phi = cur_n / next_n * sinf(theta);
                    if (sin_phi >= 1.f) {
                        nd = refl;
                        float ct = fabsf(dot3(nn, nd));
                        ct = ct > 1e-6f ? ct : 1e-6f;
                        color = div3(col, ct);
                    } else {
                        float phi = asinf(sin_phi);
                        if ((double)theta > 3.14159265358979 / 2)
                            phi = (float)(3.14159265358979 - (double)phi);
                        f3 axis = normalize3(cross3(nn, d));
                        f3 rdir = normalize3(rotate_about(axis, phi, nn));
                        float cos_t = fabsf(cosf(theta)), cos_p = fabsf(cosf(phi));
                        float a1 = fabsf(cur_n * cos_t - next_n * cos_p) / (cur_n * cos_t + next_n * cos_p);
                        float a2 = fabsf(next_n * cos_t - cur_n * cos_p) / (next_n * cos_t + cur_n * cos_p);
                        float er = (a1 * a1 + a2 * a2) / 2.f;
                        float p = clampf(er, 0.f, 1.f);
                        if (rng_float(key, ctr) < p) {
                            nd = refl;
                            float ct = fabsf(dot3(nn, nd));
                            ct = ct > 1e-6f ? ct : 1e-6f;
                            color = col * (er / ct);
                            dirProb = p;
                        } else {
                            nd = rdir;
                            float ct = fabsf(dot3(nn, nd));
                            ct = ct > 1e-6f ? ct : 1e-6f;
                            color = col * ((1.f - er) / ct);
                            dirProb = 1.f - p;
                            ninside = outside;
                        }
                    }
                    nd = normalize3(nd);
                }
            }
            if (!term && ob.type != MIRT_OBJ_EMISSIVE) {
                thr = thr * div3(color, dirProb);
                // Ray.cpp:71-78: cosine term of the new vertex against its contact normal
                float ct = 1.f;
                if (len3(nd) > 0.5f) ct = fabsf(dot3(nn, nd));
                ct = ct > 1e-6f ? ct : 1e-6f;
                thr = thr * ct;
                // MCRenderer.cpp:181-191
                alive = len3(nd) >= 0.5f && depth + 1 <= max_depth;
                no = pos;
                nids = MIRT_PACK_IDS(hobj, ninside);
            }
        }
    }

    // ---- compaction: ballot inside the wave, prefix over the block's waves, one atomic per block
    const unsigned long long m = __ballot(alive);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wcnt[w] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        s_base = tot ? atomicAdd(n_out, tot) : 0u;
    }
    __syncthreads();
    if (alive) {
        uint32_t wb = 0;
        for (uint32_t k = 0; k < w; k++) wb += s_wcnt[k];
        const uint32_t j = s_base + wb + rank;
        store3(out.o, out.stride, j, no);
        store3(out.d, out.stride, j, nd);
        store3(out.thr, out.stride, j, thr);
        out.key[j] = key;
        out.ctr[j] = ctr;
        out.ids[j] = nids;
        out.pid[j] = pid;
    }
}

// rgb_sum[p] += sum over the launch's samples of L, in sample order (deterministic).
__global__ void __launch_bounds__(256)
k_accumulate(const float* __restrict__ L, uint32_t L_stride, uint32_t wh, uint32_t n_samples,
             float* __restrict__ rgb_sum) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= wh) return;
    float r = 0.f, g = 0.f, b = 0.f;
    for (uint32_t s = 0; s < n_samples; s++) {
        uint32_t k = s * wh + p;
        r += L[k];
        g += L[L_stride + k];
        b += L[2 * L_stride + k];
    }
    rgb_sum[3 * p] += r;
    rgb_sum[3 * p + 1] += g;
    rgb_sum[3 * p + 2] += b;
}

}  // namespace mirt
