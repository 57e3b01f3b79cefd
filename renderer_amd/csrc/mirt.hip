// mirt.hip — C-ABI implementation of libmirt.so (see include/mirt.h). gfx950 / MI355X only.
// Host side: context, scene upload (flattened tree -> HBM records), launch logic of the
// generate -> intersect -> shade(+compact) wavefront loop. No CPU fallback exists: every compute
// entry point fails with MIRT_E_NO_DEVICE / MIRT_E_HIP when there is no GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "intersect_kernels.h"
#include "pt_kernels.h"
#include "../host/vecmath.h"  // matrix4::inverse (the scheme of nvMatrix.h:233-369) for Texture::getGrad

#pragma clang fp contract(off)

using namespace mirt;

struct MirtContext {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;
    std::string info;
    // scene
    PairRec* d_pairs = nullptr;
    TriRec* d_tris = nullptr;
    VnRec* d_vns = nullptr;
    DevObject* d_objs = nullptr;
    DevScene sc{};
    bool has_scene = false;
    uint64_t scene_serial = 0; // successful uploads so far (mirt_scene_serial)
    unsigned char* d_scratch = nullptr;  // this device's copy of `scratch` (hipGetSymbolAddress in mirt_create)
    bool has_media = false;    // scene holds a homogeneous medium (MIRT_OBJ_VPM)
    bool media_ready = true;   // ... and every such object came with its own tree
    VolRoot* d_vols = nullptr;
    VolNode* d_vnodes = nullptr;
    PairRec* d_fpairs = nullptr;    // fast tree (build_fast_tree) + what k_verify needs
    NodeRec* d_rnodes = nullptr;
    int* d_slot_node = nullptr;
    bool fast_tree = false;
    uint32_t* d_fb_list = nullptr;  // rays k_verify hands to the exact kernel
    int* d_vis_slot = nullptr;      // visibility through the fast tree: occluder slot per ray, for k_verify
    size_t vis_slot_cap = 0;
    size_t fb_cap = 0;
    std::vector<void*> retired;     // list buffers outgrown by a larger batch: freed at the next upload / destroy, so that
                                    // growing never has to wait for the launches that may still read the old one
    DevLight* d_lights = nullptr;   // emission sampler tables (PathTracerTest)
    VtRec* d_vts = nullptr;          // textures: corner texture coordinates per slot, texture records, texel pool
    DevTexture* d_textures = nullptr;
    float* d_texels = nullptr;
    DevBump* d_bumps = nullptr;
    bool has_bump = false;
    bool fast_all_modes = true;     // modes 0 / 1 through the fast tree too (MIRT_FAST_ALL_MODES=0: the reference tree)
    float* d_light_area = nullptr;
    int* d_light_slot = nullptr;
    int block = 256;
    size_t lds_main = 0;   // the first lds_levels stack levels live in LDS, deeper ones in d_spill
    uint32_t lds_levels = 0;
    // the same three for the fast-tree instantiations, which are built for more waves per SIMD
    size_t lds_main_fast = 0;
    uint32_t lds_levels_fast = 0;
    int max_blocks_fast = 0;
    uint2* d_spill = nullptr;
    // wavefront path state
    PathBuf pb[2]{};
    size_t pb_cap = 0;
    float* d_L = nullptr;
    int l_gens = 1;              // radiance buffers in the L ring (wavefront chaining: up to 4)
    PathBuf carry{};             // wavefront chaining: the paths a wavefront hands to the next one
    size_t carry_cap = 0;
    uint32_t* d_carry_cnt = nullptr;
    ShadowBuf shb{};             // PathTracerTest: shadow rays of the running bounce
    size_t shb_cap = 0;
    uint32_t* d_ns = nullptr;    // [cnt_cap] shadow rays written by bounce k
    uint32_t* d_wq2 = nullptr;   // [3 * cnt_cap] queue words of the visibility launches (queue head, list length, queue head)
    uint32_t* d_cnt = nullptr;   // [cnt_cap] live paths after k bounces of the running wavefront
    // two-ended compaction (run_chained, k_shade<.., TWO>): per bounce the two counters, 128 B apart ([64 * k] counts
    // upwards from the middle of the path buffer, [64 * k + 32] downwards), and where the live range starts
    uint32_t* d_ab = nullptr;    // [64 * cnt_cap]
    uint32_t* d_off = nullptr;   // [cnt_cap]
    const uint32_t* cur_off = nullptr;  // start of the batch the intersect launches being enqueued work on (device word), or null
    uint32_t* d_wq = nullptr;    // [cnt_cap] ray-queue heads, one per intersect launch of the wavefront
    uint32_t* h_cnt = nullptr;   // pinned, [cnt_cap]
    size_t cnt_cap = 0;
    unsigned long long* d_counters = nullptr;  // [8] audit counters of the instrumented kernel
    uint32_t* d_queue = nullptr;               // ray queue head of the running launch
    int n_cus = 256;
    int max_blocks = 1024;                     // resident blocks of the intersect kernel (grid cap)
    // staging for the host-pointer intersect
    float* st_o = nullptr;
    float* st_d = nullptr;
    int* st_aux = nullptr;
    float* st_t = nullptr;
    int* st_id = nullptr;
    size_t st_cap = 0;
    std::vector<hipEvent_t> events;
};

static thread_local std::string g_create_err;

// ---- static device scratch of mirt_intersect_noalloc_dev (include/mirt.h) ---------------------------
// Callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc) get the
// closest-hit call's three scratch buffers from this file-scope array instead of hipMalloc'd memory: every GPU the
// library runs on holds its own copy at its own address until the process ends. In host code the array's NAME
// is only a stand-in for hipGetSymbolAddress; mirt_create resolves the address once per context, with the
// context's device current, and everything below uses that pointer (kernel arguments, hipMemsetAsync).
//   [0, 256)                         queue head of the fast launch, list length, queue head of the exact launch
//   [256, 256 + 4 * MAX_RAYS)        ray indices k_verify hands to the exact re-run
//   [.., + NOALLOC_SPILL_BYTES)      stack levels beyond the LDS share (the grid is cut down until they fit)
constexpr size_t NOALLOC_QUEUE_BYTES = 256;
constexpr size_t NOALLOC_LIST_BYTES = (size_t)MIRT_NOALLOC_MAX_RAYS * sizeof(uint32_t);
constexpr size_t NOALLOC_SPILL_BYTES = (size_t)32 << 20;
constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_QUEUE_BYTES + NOALLOC_LIST_BYTES + NOALLOC_SPILL_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

#define HIPCHK(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                    \
            return MIRT_E_HIP;                                                                  \
        }                                                                                       \
    } while (0)

static int fail(MirtContext* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// ---- host copies of the exact-arithmetic helpers (fp32, no contraction) --------------------------
namespace {
struct h3 {
    float x, y, z;
};
inline h3 hsub(h3 a, h3 b) { return h3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline h3 hcross(h3 l, h3 r) {
    return h3{l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x};
}
inline float hdot(h3 a, h3 b) {
    float r = a.x * b.x;
    r = r + a.y * b.y;
    r = r + a.z * b.z;
    return r;
}
inline h3 hnormalize(h3 a) {
    float r = sqrtf(hdot(a, a));
    if (r > 0.f) return h3{a.x / r, a.y / r, a.z / r};
    return a;
}
}  // namespace

extern "C" {

const char* mirt_last_error(const MirtContext* ctx) {
    return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

int mirt_create(int device, MirtContext** out) {
    if (!out) {
        g_create_err = "mirt_create: out is NULL";
        return MIRT_E_INVALID;
    }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("mirt_create: no HIP device (") +
                       (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                       "); libmirt has no CPU path";
        return MIRT_E_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        g_create_err = "mirt_create: device index out of range";
        return MIRT_E_INVALID;
    }
    MirtContext* ctx = new MirtContext;
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_err = std::string("mirt_create: ") + hipGetErrorString(e);
        delete ctx;
        return MIRT_E_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s arch=%s CUs=%d LDS/block=%zu HBM=%.1f GiB", prop.name,
                 prop.gcnArchName, prop.multiProcessorCount, (size_t)prop.sharedMemPerBlock,
                 prop.totalGlobalMem / 1073741824.0);
        ctx->info = buf;
        ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // this device's address of the static scratch array (the array's name itself is no device address)
    if ((e = hipGetSymbolAddress((void**)&ctx->d_scratch, HIP_SYMBOL(scratch))) != hipSuccess || !ctx->d_scratch) {
        g_create_err = std::string("mirt_create: hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return MIRT_E_HIP;
    }
    if ((e = hipMalloc(&ctx->d_counters, 8 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMalloc(&ctx->d_queue, 64 * sizeof(uint32_t))) != hipSuccess) {
        g_create_err = std::string("mirt_create: ") + hipGetErrorString(e);
        delete ctx;
        return MIRT_E_HIP;
    }
    *out = ctx;
    return MIRT_OK;
}

static void free_retired(MirtContext* ctx) {  // (ensure_fb_list: buffers outgrown while launches might still read them)
    for (void* p : ctx->retired) hipFree(p);
    ctx->retired.clear();
}

static void free_scene(MirtContext* ctx) {
    hipFree(ctx->d_pairs);
    ctx->d_pairs = nullptr;
    hipFree(ctx->d_tris);
    hipFree(ctx->d_vns);
    hipFree(ctx->d_objs);
    hipFree(ctx->d_spill);
    ctx->d_spill = nullptr;
    hipFree(ctx->d_vols);
    hipFree(ctx->d_vnodes);
    ctx->d_vols = nullptr;
    ctx->d_vnodes = nullptr;
    hipFree(ctx->d_fpairs);
    hipFree(ctx->d_rnodes);
    hipFree(ctx->d_slot_node);
    ctx->d_fpairs = nullptr;
    ctx->d_rnodes = nullptr;
    ctx->d_slot_node = nullptr;
    ctx->fast_tree = false;
    hipFree(ctx->d_vts);
    hipFree(ctx->d_textures);
    hipFree(ctx->d_texels);
    hipFree(ctx->d_bumps);
    ctx->d_bumps = nullptr;
    ctx->has_bump = false;
    ctx->d_vts = nullptr;
    ctx->d_textures = nullptr;
    ctx->d_texels = nullptr;
    hipFree(ctx->d_lights);
    hipFree(ctx->d_light_area);
    hipFree(ctx->d_light_slot);
    ctx->d_lights = nullptr;
    ctx->d_light_area = nullptr;
    ctx->d_light_slot = nullptr;
    ctx->d_tris = nullptr;
    ctx->d_vns = nullptr;
    ctx->d_objs = nullptr;
    ctx->has_scene = false;
}

static void free_paths(MirtContext* ctx) {
    for (int k = 0; k < 2; k++) {
        PathBuf& p = ctx->pb[k];
        hipFree(p.o); hipFree(p.d); hipFree(p.thr); hipFree(p.t); hipFree(p.slot);
        hipFree(p.key); hipFree(p.ctr); hipFree(p.ids); hipFree(p.pid); hipFree(p.lpw);
        p = PathBuf{};
    }
    hipFree(ctx->d_L);
    ctx->d_L = nullptr;
    ctx->pb_cap = 0;
    ctx->l_gens = 1;
    {
        PathBuf& p = ctx->carry;
        hipFree(p.o); hipFree(p.d); hipFree(p.thr); hipFree(p.t); hipFree(p.slot);
        hipFree(p.key); hipFree(p.ctr); hipFree(p.ids); hipFree(p.pid); hipFree(p.lpw);
        p = PathBuf{};
        ctx->carry_cap = 0;
    }
    hipFree(ctx->d_carry_cnt);
    ctx->d_carry_cnt = nullptr;
    ShadowBuf& b = ctx->shb;
    hipFree(b.o); hipFree(b.d); hipFree(b.ids); hipFree(b.lim); hipFree(b.c); hipFree(b.pid); hipFree(b.vis);
    b = ShadowBuf{};
    ctx->shb_cap = 0;
}

static void free_staging(MirtContext* ctx) {
    hipFree(ctx->st_o); hipFree(ctx->st_d); hipFree(ctx->st_aux); hipFree(ctx->st_t);
    hipFree(ctx->st_id);
    ctx->st_o = ctx->st_d = ctx->st_t = nullptr;
    ctx->st_aux = ctx->st_id = nullptr;
    ctx->st_cap = 0;
}

int mirt_destroy(MirtContext* ctx) {
    if (!ctx) return MIRT_OK;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();  // (launches on the callers' own streams included)
    free_scene(ctx);
    free_paths(ctx);
    free_staging(ctx);
    for (hipEvent_t ev : ctx->events) hipEventDestroy(ev);
    hipFree(ctx->d_cnt);
    hipFree(ctx->d_ab);
    hipFree(ctx->d_off);
    hipFree(ctx->d_wq);
    hipFree(ctx->d_ns);
    hipFree(ctx->d_wq2);
    hipFree(ctx->d_counters);
    hipFree(ctx->d_queue);
    hipFree(ctx->d_fb_list);
    hipFree(ctx->d_vis_slot);
    free_retired(ctx);
    if (ctx->h_cnt) hipHostFree(ctx->h_cnt);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return MIRT_OK;
}

uint64_t mirt_scene_serial(const MirtContext* ctx) { return ctx && ctx->has_scene ? ctx->scene_serial : 0; }

int mirt_device_info(MirtContext* ctx, char* buf, size_t buflen) {
    if (!ctx || !buf || buflen == 0) return MIRT_E_INVALID;
    snprintf(buf, buflen, "%s", ctx->info.c_str());
    return MIRT_OK;
}

// A flattened pre-order tree as MirtSceneDesc / MirtObjTree describe it.
struct TreeView {
    int N, T;  // nodes, triangle slots
    const float* mn;
    const float* mx;
    const int32_t* right;
    const int32_t* lbeg;
    const int32_t* lend;
};

// Validates the flattened tree (every index a kernel will follow is range-checked here, on the
// host, so a malformed description cannot fault the GPU) and returns its depth.
static int validate_tree(const TreeView& s, int* depth_out, std::string* why) {
    const int N = s.N, T = s.T;
    // iterative pre-order walk: (node, depth)
    std::vector<std::pair<int, int>> st;
    st.push_back({0, 0});
    int maxd = 0;
    std::vector<char> seen(N, 0);
    long long slots_seen = 0;
    int expect = 0;  // pre-order: nodes must be visited in index order
    while (!st.empty()) {
        auto [i, d] = st.back();
        st.pop_back();
        if (i != expect) { *why = "nodes are not in pre-order"; return MIRT_E_INVALID; }
        expect++;
        if (i < 0 || i >= N || seen[i]) { *why = "node index out of range or revisited"; return MIRT_E_INVALID; }
        seen[i] = 1;
        maxd = std::max(maxd, d);
        int r = s.right[i];
        if (r < 0) {
            int b = s.lbeg[i], e = s.lend[i];
            if (b < 0 || e < b || e > T) { *why = "leaf slot range out of bounds"; return MIRT_E_INVALID; }
            if (b != slots_seen) { *why = "leaf slot ranges are not contiguous in pre-order"; return MIRT_E_INVALID; }
            slots_seen = e;
        } else {
            if (r <= i + 1 || r >= N) { *why = "right child index out of range"; return MIRT_E_INVALID; }
            st.push_back({r, d + 1});      // visited after the whole left subtree
            st.push_back({i + 1, d + 1});
        }
    }
    if (expect != N) { *why = "unreachable nodes"; return MIRT_E_INVALID; }
    if (slots_seen != T) { *why = "leaves do not cover all triangle slots"; return MIRT_E_INVALID; }
    *depth_out = maxd;
    return MIRT_OK;
}

// What the slab classifier's error model needs of a tree: every node bound finite and below 2^60, and
// the tree's extent per axis (largest |coordinate| of its root box, which is what the absolute part of
// the margin, RayPre::a_ray, is built from) zero or at least 2^-36. Individual bounds may be arbitrarily
// small (a tessellated sphere leaves coordinates like 6e-17 where a cosine should be zero): the rounding
// of the reference's quotients and products is relative to magnitudes that `extent` bounds, and where a
// tiny operand underflows the absolute error is below 2^-149, i.e. below 2^-149 * 2^40 in t units, far
// under a_ray >= 2^-20 * 2^-36 * 2^-40.
static bool bounds_in_model(const TreeView& s) {
    float extent[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < s.N; i++)
        for (int c = 0; c < 3; c++)
            for (float v : {s.mn[3 * i + c], s.mx[3 * i + c]}) {
                float a = std::fabs(v);
                if (!(a < 1.1529215e18f)) return false;  // also NaN/inf
                if (i == 0) extent[c] = std::max(extent[c], a);
            }
    for (int c = 0; c < 3; c++)
        if (!(extent[c] == 0.f || extent[c] >= 1.4551915e-11f)) return false;
    return true;
}

// ---- the fast tree ---------------------------------------------------------------------------------
// A second hierarchy over the scene's TRIANGLES (each in the device slot the reference tree's leaf order gives
// it, each a leaf of its own with the box of its three corners), built here with a binned surface-area
// heuristic. The reference's median split on vertex coordinates makes a ray walk several root-to-leaf chains
// of heavily overlapping boxes (112 box tests per ray on the 1 M-triangle scene) and test every triangle of
// every leaf it reaches (12 per ray); a hierarchy built by surface area over the triangles themselves roughly
// halves the first and divides the second by three. The traversal kernel walks THIS tree with a conservative
// box test and so tests a superset of the triangles the reference would reach (a triangle the reference test
// accepts is crossed by the ray inside its own box, up to a rounding the slab margin covers); k_verify then
// proves the winner's chain of REFERENCE boxes with the reference's own test, and ties between triangles
// follow the reference's order through DevScene::tie_rank (DESIGN.md "fast tree"). Output: a pre-order
// TreeView (left child = i + 1) whose leaves are single slots.
struct FastTree {
    std::vector<float> mn, mx;
    std::vector<int32_t> right, lbeg, lend;
    int depth = 0;
};

// slot_box(slot, mn, mx) widens mn / mx by the triangle in that device slot.
// The fast tree's box of one triangle: the box of its corners, widened by how far outside the triangle the
// reference's edge-sign test (KDTree.cpp:267-289) can still accept a ray. Each sign d.(l_i x r_i) carries a rounding
// error of <= 16 eps |d||l_i||r_i|, i.e. the ray may pass an edge's line on the wrong side by ~16 eps |r_i| (for a ray
// not nearly parallel to that edge); where two edges meet at an angle theta the accepted region reaches beyond the
// corner by that over sin(theta). A well-shaped triangle gets ~1e-5 of the scene's size, a needle (found by
// tests/test_gpu_fuzz.py: a sliver 1e-4 wide accepted 2e-4 beyond its tip) up to the whole reference leaf box, to
// which build_fast_tree clips. bound: largest coordinate magnitude of the scene; the pad is sized for rays that
// start within four times that, and only those take the fast route (DevScene::far_o, RayPre::near in mirt_dev.h).
static void tri_fast_box(const float* p, float bound, float* mn, float* mx) {
    double sin_min = 1.0;
    for (int c = 0; c < 3; c++) {
        const float* a = p + 3 * c;
        const float* b = p + 3 * ((c + 1) % 3);
        const float* e = p + 3 * ((c + 2) % 3);
        const double u[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]};
        const double w[3] = {(double)e[0] - a[0], (double)e[1] - a[1], (double)e[2] - a[2]};
        const double cr[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
        const double nu = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), nw = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const double sn = nu > 0 && nw > 0 ? std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) / (nu * nw) : 0.0;
        sin_min = std::min(sin_min, sn);
    }
    const double pad_d = 64.0 * 5.9604644775390625e-08 * 4.0 * (double)bound / std::max(sin_min, 1e-12);
    const float pad = (float)std::min(pad_d, 8.0 * (double)bound) * 1.0000002f;
    for (int v = 0; v < 3; v++)
        for (int k = 0; k < 3; k++) {
            mn[k] = std::min(mn[k], p[3 * v + k] - pad);
            mx[k] = std::max(mx[k], p[3 * v + k] + pad);
        }
}

#ifndef MIRT_SAH_BINS
#define MIRT_SAH_BINS 16
#endif
static void build_fast_tree(const TreeView& ref, FastTree& ft, const std::function<void(int, float*, float*)>& slot_box) {
    struct Leaf {
        float mn[3], mx[3], c[3];
        int b, e;
    };
    std::vector<Leaf> leaves;
    for (int i = 0; i < ref.N; i++)
        if (ref.right[i] < 0)
            for (int t = ref.lbeg[i]; t < ref.lend[i]; t++) {
                Leaf l;
                l.b = t;
                l.e = t + 1;
                for (int k = 0; k < 3; k++) l.mn[k] = INFINITY, l.mx[k] = -INFINITY;
                slot_box(t, l.mn, l.mx);
                // never larger than the reference leaf's own box: the reference tests the triangle only for rays
                // that pass that box (k_verify checks it), so nothing outside it can matter
                for (int k = 0; k < 3; k++) {
                    l.mn[k] = std::max(l.mn[k], ref.mn[3 * i + k]);
                    l.mx[k] = std::min(l.mx[k], ref.mx[3 * i + k]);
                }
                for (int k = 0; k < 3; k++) l.c[k] = 0.5f * (l.mn[k] + l.mx[k]);
                leaves.push_back(l);
            }
    if (leaves.empty()) return;
    auto area = [](const float* mn, const float* mx) {
        const double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    };
    const int BINS = MIRT_SAH_BINS;
    // iterative pre-order construction: a node's right index is patched when its right child is emitted
    struct Job {
        int b, e, depth, parent;  // parent: node whose `right` this job fills (-1: left child or root)
    };
    std::vector<Job> st;
    st.push_back({0, (int)leaves.size(), 0, -1});
    while (!st.empty()) {
        const Job j = st.back();
        st.pop_back();
        const int node = (int)ft.right.size();
        if (j.parent >= 0) ft.right[j.parent] = node;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int q = j.b; q < j.e; q++)
            for (int k = 0; k < 3; k++) {
                mn[k] = std::min(mn[k], leaves[q].mn[k]);
                mx[k] = std::max(mx[k], leaves[q].mx[k]);
                cmn[k] = std::min(cmn[k], leaves[q].c[k]);
                cmx[k] = std::max(cmx[k], leaves[q].c[k]);
            }
        for (int k = 0; k < 3; k++) {
            ft.mn.push_back(mn[k]);
            ft.mx.push_back(mx[k]);
        }
        ft.right.push_back(-1);
        ft.lbeg.push_back(-1);
        ft.lend.push_back(-1);
        ft.depth = std::max(ft.depth, j.depth);
        const int n = j.e - j.b;
        if (n == 1) {
            ft.lbeg[node] = leaves[j.b].b;
            ft.lend[node] = leaves[j.b].e;
            continue;
        }
        int mid = -1;
        // surface-area splits only while halving the range could still finish inside the depth the kernels' stacks
        // allow (a node of n triangles needs ceil(log2 n) more levels when halved): the depth is bounded whatever
        // the geometry, and never by more than MIRT_MAX_TREE_DEPTH
        int levels = 0;
        while ((1ll << levels) < n) levels++;
        if (j.depth < 40 && j.depth + 1 + levels < MIRT_MAX_TREE_DEPTH) {
            double best = INFINITY;
            int best_axis = -1, best_bin = -1;
            for (int k = 0; k < 3; k++) {
                const float ext = cmx[k] - cmn[k];
                if (!(ext > 0.f)) continue;
                int cnt[BINS] = {0};
                float bmn[BINS][3], bmx[BINS][3];
                for (int b = 0; b < BINS; b++)
                    for (int c = 0; c < 3; c++) bmn[b][c] = INFINITY, bmx[b][c] = -INFINITY;
                for (int q = j.b; q < j.e; q++) {
                    int b = std::min(BINS - 1, (int)((leaves[q].c[k] - cmn[k]) / ext * BINS));
                    cnt[b]++;
                    for (int c = 0; c < 3; c++) {
                        bmn[b][c] = std::min(bmn[b][c], leaves[q].mn[c]);
                        bmx[b][c] = std::max(bmx[b][c], leaves[q].mx[c]);
                    }
                }
                double rarea[BINS];
                int rcnt[BINS];
                float amn[3] = {INFINITY, INFINITY, INFINITY}, amx[3] = {-INFINITY, -INFINITY, -INFINITY};
                int acc = 0;
                for (int b = BINS - 1; b > 0; b--) {
                    for (int c = 0; c < 3; c++) amn[c] = std::min(amn[c], bmn[b][c]), amx[c] = std::max(amx[c], bmx[b][c]);
                    acc += cnt[b];
                    rarea[b] = acc ? area(amn, amx) : 0.0;
                    rcnt[b] = acc;
                }
                for (int c = 0; c < 3; c++) amn[c] = INFINITY, amx[c] = -INFINITY;
                acc = 0;
                for (int b = 0; b + 1 < BINS; b++) {
                    for (int c = 0; c < 3; c++) amn[c] = std::min(amn[c], bmn[b][c]), amx[c] = std::max(amx[c], bmx[b][c]);
                    acc += cnt[b];
                    if (acc == 0 || rcnt[b + 1] == 0) continue;
                    const double cost = area(amn, amx) * acc + rarea[b + 1] * rcnt[b + 1];
                    if (cost < best) best = cost, best_axis = k, best_bin = b;
                }
            }
            if (best_axis >= 0) {
                const float ext = cmx[best_axis] - cmn[best_axis];
                auto it = std::partition(leaves.begin() + j.b, leaves.begin() + j.e, [&](const Leaf& l) {
                    return std::min(BINS - 1, (int)((l.c[best_axis] - cmn[best_axis]) / ext * BINS)) <= best_bin;
                });
                mid = (int)(it - leaves.begin());
                if (mid == j.b || mid == j.e) mid = -1;
            }
        }
        if (mid < 0) mid = j.b + n / 2;
        st.push_back({mid, j.e, j.depth + 1, node});  // right child: emitted after the whole left subtree
        st.push_back({j.b, mid, j.depth + 1, -1});
    }
}

// A leaf link holds at most 127 triangles (LEAF_LINK_MAX); append_pairs hangs a larger reference leaf
// [b, e) under synthetic pairs by halving the range. The kernels break distance ties by slot position:
// the LATER leaf wins, inside one leaf link the EARLIER slot (KDTree.cpp:319-320, 305). The pieces of one
// reference leaf must therefore rank the other way round, the piece with the leaf's first triangles
// last: inside a reference leaf the earlier triangle wins (KDTree.cpp:305), and a triangle edge shared
// across a seam between two pieces produces exactly such a tie. This returns slot -> position in the
// reference's slot order with the pieces of every oversized leaf laid out in reverse.
constexpr int LEAF_LINK_MAX = 127;
static void leaf_pieces(int b, int e, std::vector<std::pair<int, int>>& out) {
    if (e - b <= LEAF_LINK_MAX) {
        out.push_back({b, e});
        return;
    }
    const int mid = b + (e - b) / 2;
    leaf_pieces(b, mid, out);
    leaf_pieces(mid, e, out);
}
static std::vector<int> seam_order(const TreeView& v) {
    std::vector<int> src(v.T);
    for (int k = 0; k < v.T; k++) src[k] = k;
    std::vector<std::pair<int, int>> pieces;
    for (int i = 0; i < v.N; i++) {
        if (v.right[i] >= 0 || v.lend[i] - v.lbeg[i] <= LEAF_LINK_MAX) continue;
        pieces.clear();
        leaf_pieces(v.lbeg[i], v.lend[i], pieces);
        int pos = v.lbeg[i];
        for (size_t j = pieces.size(); j-- > 0;)
            for (int k = pieces[j].first; k < pieces[j].second; k++) src[k] = pos++;
    }
    return src;
}

// Appends the tree's child-pair records (breadth-first over its inner nodes) to `pairs`; leaf
// links address slots slot_base + [begin, end). Returns the root's own record (box + link) and
// adds the levels that oversized leaves needed to *depth.
static void append_pairs(const TreeView& s, int slot_base, std::vector<PairRec>& pairs, float4* root_lo,
                         float4* root_hi, int* depth) {
    const int N = s.N;
    const size_t base = pairs.size();
    std::vector<int> order;  // inner nodes in BFS order; pair id = base + position
    std::vector<int> pair_of(N, -1);
    if (s.right[0] >= 0) {
        order.push_back(0);
        pair_of[0] = (int)base;
    }
    for (size_t q = 0; q < order.size(); q++) {
        int i = order[q];
        for (int c : {i + 1, s.right[i]})
            if (s.right[c] >= 0) {
                pair_of[c] = (int)(base + order.size());
                order.push_back(c);
            }
    }
    pairs.resize(base + order.size(), PairRec{});
    // A leaf link holds at most 127 triangles. The reference's build leaves larger leaves where no
    // split separates the triangles (a fan around one vertex). Such a leaf becomes a small subtree
    // of synthetic pairs whose children all carry the leaf's own box: the box test on them repeats
    // the leaf's test bit for bit, and seam_order() arranges the triangles inside the pieces so that the
    // tie rule is unchanged too (same halving as leaf_pieces).
    int extra_depth = 0;
    std::function<int(int, int, int, int)> range_link = [&](int c, int b, int e, int level) -> int {
        if (e - b <= LEAF_LINK_MAX) {
            extra_depth = std::max(extra_depth, level);
            return ~(int)((uint32_t)(slot_base + b) | ((uint32_t)(e - b) << 24));
        }
        const int mid = b + (e - b) / 2;
        const size_t q = pairs.size();
        pairs.push_back(PairRec{});
        const int ll = range_link(c, b, mid, level + 1), lr = range_link(c, mid, e, level + 1);
        const float* mn = s.mn + 3 * c;
        const float* mx = s.mx + 3 * c;
        int flat = (mn[0] == mx[0]) | (mn[1] == mx[1]) | (mn[2] == mx[2]);
        float fl, fr, fn;
        memcpy(&fl, &ll, 4);
        memcpy(&fr, &lr, 4);
        memcpy(&fn, &flat, 4);
        pair_pack(make_float4(mn[0], mn[1], mn[2], fl), make_float4(mx[0], mx[1], mx[2], fn), &pairs[q].la, &pairs[q].lb);
        pair_pack(make_float4(mn[0], mn[1], mn[2], fr), make_float4(mx[0], mx[1], mx[2], fn), &pairs[q].ra, &pairs[q].rb);
        return (int)q;
    };
    auto child_rec = [&](int c, float4* lo, float4* hi) {
        int lk;
        if (s.right[c] >= 0) lk = pair_of[c];
        else lk = range_link(c, s.lbeg[c], s.lend[c], 0);
        const float* mn = s.mn + 3 * c;
        const float* mx = s.mx + 3 * c;
        int flat = (mn[0] == mx[0]) | (mn[1] == mx[1]) | (mn[2] == mx[2]);
        float fl, fn;
        memcpy(&fl, &lk, 4);
        memcpy(&fn, &flat, 4);
        *lo = make_float4(mn[0], mn[1], mn[2], fl);
        *hi = make_float4(mx[0], mx[1], mx[2], fn);
    };
    for (size_t q = 0; q < order.size(); q++) {
        int i = order[q];
        float4 a0, a1, b0, b1;  // (pairs may grow inside child_rec: no references into it here)
        child_rec(i + 1, &a0, &a1);
        child_rec(s.right[i], &b0, &b1);
        pair_pack(a0, a1, &pairs[base + q].la, &pairs[base + q].lb);
        pair_pack(b0, b1, &pairs[base + q].ra, &pairs[base + q].rb);
    }
    child_rec(0, root_lo, root_hi);
    *depth += extra_depth;
}

int mirt_upload_scene(MirtContext* ctx, const MirtSceneDesc* s) {
    if (!ctx) return MIRT_E_INVALID;
    // struct_size versions the descriptor: the forms without per-object trees / without textures are still accepted
    const size_t size_v1 = offsetof(MirtSceneDesc, n_obj_trees), size_v2 = offsetof(MirtSceneDesc, tri_vt);
    if (!s || (s->struct_size != sizeof(MirtSceneDesc) && s->struct_size != size_v1 && s->struct_size != size_v2))
        return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: bad descriptor / struct_size");
    const int n_obj_trees = s->struct_size >= size_v2 ? s->n_obj_trees : 0;
    const bool tex_fields = s->struct_size == sizeof(MirtSceneDesc);
    const int n_textures = tex_fields ? s->n_textures : 0;
    const float* tri_vt = tex_fields ? s->tri_vt : nullptr;
    const MirtObjectTex* object_tex = tex_fields && n_textures > 0 ? s->object_tex : nullptr;
    if (n_textures < 0 || (n_textures > 0 && !s->textures))
        return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: n_textures without textures");
    size_t texel_total = 0;
    for (int k = 0; k < n_textures; k++) {
        const MirtTexture& X = s->textures[k];
        if (X.width <= 0 || X.height <= 0 || (X.channels != 1 && X.channels != 3) || !X.texels ||
            (long long)X.width * X.height > (1ll << 28))
            return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: bad texture (size, channels or NULL texels)");
        texel_total += (size_t)X.width * X.height * X.channels;
    }
    if (object_tex)
        for (int o = 0; o < s->n_objects; o++)
            for (int k : {object_tex[o].diffuse, object_tex[o].glossy_color, object_tex[o].glossy_coeff, object_tex[o].bump})
                if (k < -1 || k >= n_textures)
                    return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: object_tex index out of range");
    // bump textures (SceneObject.cpp:250-283): one DevBump per object that has one; needs texture coordinates
    std::vector<DevBump> bumps;
    std::vector<int> obj_bump(std::max(s->n_objects, 0), 0);  // 1 + index into bumps, 0 = none
    if (object_tex)
        for (int o = 0; o < s->n_objects; o++) {
            const int k = object_tex[o].bump;
            if (k < 0) continue;
            if (!tri_vt) return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: a bump texture needs tri_vt");
            const MirtTexture& X = s->textures[k];
            mirt_host::matrix4 m;
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) m(r, c) = X.coord_transform[4 * r + c];
            const mirt_host::matrix4 inv = m.inverse();
            DevBump B;
            B.texture = k;
            for (int q = 0; q < 12; q++) B.inv[q] = inv(q / 4, q % 4);
            bumps.push_back(B);
            obj_bump[o] = (int)bumps.size();
        }
    const MirtObjTree* obj_trees = n_obj_trees > 0 ? s->obj_trees : nullptr;
    if (s->n_nodes <= 0 || s->n_tris <= 0 || s->n_objects <= 0 || !s->node_min || !s->node_max ||
        !s->node_right || !s->node_leaf_begin || !s->node_leaf_end || !s->slot_tid || !s->tri_pos ||
        !s->tri_obj || !s->tri_local || !s->objects || n_obj_trees < 0 || (n_obj_trees > 0 && !obj_trees))
        return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: empty scene or NULL array");
    if (s->n_objects > MIRT_MAX_OBJECTS)
        return fail(ctx, MIRT_E_LIMIT, "mirt_upload_scene: more than 65533 objects");
    const TreeView main_tree{s->n_nodes, s->n_tris, s->node_min, s->node_max, s->node_right,
                             s->node_leaf_begin, s->node_leaf_end};
    int depth = 0;
    std::string why;
    if (validate_tree(main_tree, &depth, &why) != MIRT_OK)
        return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: " + why);
    const int N = s->n_nodes, T = s->n_tris;
    std::vector<char> tid_seen(T, 0);
    bool any_vn = false;
    int n_vpm = 0;
    for (int o = 0; o < s->n_objects; o++) {
        any_vn |= s->objects[o].has_vn != 0;
        int ty = s->objects[o].type;
        if (ty < MIRT_OBJ_DIFFUSE || ty > MIRT_OBJ_PHONG)
            return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: unknown object type");
        n_vpm += ty == MIRT_OBJ_VPM;
        if (s->objects[o].tri_begin < 0 || s->objects[o].tri_count < 0 ||
            (long long)s->objects[o].tri_begin + s->objects[o].tri_count > T)
            return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: object triangle range out of bounds");
    }
    if (any_vn && !s->tri_vn)
        return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: has_vn set but tri_vn is NULL");
    for (int k = 0; k < T; k++) {
        int g = s->slot_tid[k];
        if (g < 0 || g >= T || tid_seen[g])
            return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: slot_tid is not a permutation");
        tid_seen[g] = 1;
        if (s->tri_obj[g] < 0 || s->tri_obj[g] >= s->n_objects)
            return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: tri_obj out of range");
    }
    // per-object trees (volumetric objects): validated like the scene tree
    long long extra_slots = 0;
    std::vector<int> obj_depth(n_obj_trees, 0);
    std::vector<TreeView> obj_view(n_obj_trees);
    int prev_obj = -1;
    for (int k = 0; k < n_obj_trees; k++) {
        const MirtObjTree& ot = obj_trees[k];
        if (ot.object <= prev_obj || ot.object >= s->n_objects)
            return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: obj_trees must be ordered by distinct object index");
        prev_obj = ot.object;
        const MirtObject& ob = s->objects[ot.object];
        if (ot.n_nodes <= 0 || !ot.node_min || !ot.node_max || !ot.node_right || !ot.node_leaf_begin ||
            !ot.node_leaf_end || !ot.slot_tid || ob.tri_count <= 0)
            return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: empty object tree or NULL array");
        obj_view[k] = TreeView{ot.n_nodes, ob.tri_count, ot.node_min, ot.node_max, ot.node_right,
                               ot.node_leaf_begin, ot.node_leaf_end};
        if (validate_tree(obj_view[k], &obj_depth[k], &why) != MIRT_OK)
            return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: object tree: " + why);
        std::vector<char> seen(ob.tri_count, 0);
        for (int q = 0; q < ob.tri_count; q++) {
            int l = ot.slot_tid[q];
            if (l < 0 || l >= ob.tri_count || seen[l])
                return fail(ctx, MIRT_E_INVALID, "mirt_upload_scene: object tree slot_tid is not a permutation");
            seen[l] = 1;
        }
        extra_slots += ob.tri_count;
    }
    // rendering a scene with a homogeneous medium needs that object's own tree (findInsideObject)
    bool media_ready = true;
    for (int o = 0; o < s->n_objects; o++)
        if (s->objects[o].type == MIRT_OBJ_VPM) {
            bool found = false;
            for (int k = 0; k < n_obj_trees; k++) found |= obj_trees[k].object == o;
            media_ready &= found;
        }

    // the facts the slab classifier's error model needs
    float bound[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < 3; c++)
        for (float v : {s->node_min[c], s->node_max[c]}) bound[c] = std::max(bound[c], std::fabs(v));
    int fast_ok = bounds_in_model(main_tree) ? 1 : 0;
    // (a NaN corner drops out of the min / max that make the node boxes, so the boxes alone do not show it: a scene with
    // a corner that is not finite, or beyond the model's 2^60, takes the exact kernels, whose arithmetic on such
    // values is the reference's; the fast tree's builder is not asked to bin them)
    for (size_t k = 0; fast_ok && k < 9 * (size_t)T; k++)
        if (!(std::fabs(s->tri_pos[k]) < 1.1529215e18f)) fast_ok = 0;
    if ((long long)T + extra_slots > (1 << 24))
        return fail(ctx, MIRT_E_LIMIT, "mirt_upload_scene: more than 2^24 triangle slots");
    std::vector<PairRec> pairs;
    float4 root_lo, root_hi;
    append_pairs(main_tree, 0, pairs, &root_lo, &root_hi, &depth);
    if (pairs.empty()) pairs.push_back(PairRec{});
    // object trees (volumetric objects, for findInsideObject): pre-order nodes with escape indices (VolNode); their
    // triangles take the slots behind the scene's, in the object tree's own leaf order
    std::vector<VolRoot> vols(n_obj_trees);
    std::vector<VolNode> vnodes;
    {
        int slot_base = T;
        for (int k = 0; k < n_obj_trees; k++) {
            const TreeView& v = obj_view[k];
            vols[k].object = obj_trees[k].object;
            vols[k].first = (int)vnodes.size();
            vols[k].count = v.N;
            // (MIRT_VOL_SMALL=0 in the environment: every object tree is walked, for A/B runs and the equality test)
            const bool small_on = !getenv("MIRT_VOL_SMALL") || atoi(getenv("MIRT_VOL_SMALL")) != 0;
            vols[k].small = small_on && s->objects[obj_trees[k].object].tri_count <= MIRT_VOL_SMALL ? 1 : 0;
            std::vector<int> par(v.N, -1);
            for (int i = 0; i < v.N; i++)
                if (v.right[i] >= 0) par[i + 1] = par[v.right[i]] = vols[k].first + i;
            std::vector<int> after(v.N);  // the node that follows node i's subtree in pre-order (right[i] > i)
            for (int i = v.N - 1; i >= 0; i--) after[i] = v.right[i] < 0 ? i + 1 : after[v.right[i]];
            for (int i = 0; i < v.N; i++) {
                VolNode nd;
                const int esc = vols[k].first + after[i];
                float fe;
                memcpy(&fe, &esc, 4);
                nd.lo = make_float4(v.mn[3 * i], v.mn[3 * i + 1], v.mn[3 * i + 2], fe);
                nd.hi = make_float4(v.mx[3 * i], v.mx[3 * i + 1], v.mx[3 * i + 2], 0.f);
                nd.leaf_begin = v.right[i] < 0 ? slot_base + v.lbeg[i] : -1;
                nd.leaf_end = v.right[i] < 0 ? slot_base + v.lend[i] : -1;
                nd.parent = par[i];
                nd.flat = (v.mn[3 * i] == v.mx[3 * i] || v.mn[3 * i + 1] == v.mx[3 * i + 1] ||
                           v.mn[3 * i + 2] == v.mx[3 * i + 2]) ? 1 : 0;
                vnodes.push_back(nd);
            }
            slot_base += s->objects[obj_trees[k].object].tri_count;
        }
    }
    const std::vector<int> main_src = seam_order(main_tree);  // device slot -> position in the reference's slot order
    // fast tree over the reference leaves + the reference nodes with parent links for k_verify
    std::vector<PairRec> fpairs;
    float4 froot_lo = root_lo, froot_hi = root_hi;
    std::vector<NodeRec> rnodes(N);
    std::vector<int> slot_node(2 * (size_t)std::max(T, 1), 0);  // [0, T): leaf node of a slot; [T, 2T): its tie rank
    bool use_fast = true;
    if (const char* e = getenv("MIRT_FAST_TREE")) use_fast = atoi(e) != 0;  // A/B switch
    {
        std::vector<int> parent(N, -1);
        for (int i = 0; i < N; i++)
            if (s->node_right[i] >= 0) {
                parent[i + 1] = i;
                parent[s->node_right[i]] = i;
            } else {
                for (int k = s->node_leaf_begin[i]; k < s->node_leaf_end[i]; k++) slot_node[k] = i;
            }
        {
            // DevScene::tie_rank, per DEVICE slot (main_src: device slot -> position in the reference's slot order)
            std::vector<int> leaf_b(std::max(T, 1), 0), leaf_e(std::max(T, 1), 0);
            for (int i = 0; i < N; i++)
                if (s->node_right[i] < 0)
                    for (int k = s->node_leaf_begin[i]; k < s->node_leaf_end[i]; k++)
                        leaf_b[k] = s->node_leaf_begin[i], leaf_e[k] = s->node_leaf_end[i];
            for (int k = 0; k < T; k++) {
                const int r = main_src[k];
                slot_node[(size_t)std::max(T, 1) + k] = leaf_b[r] + (leaf_e[r] - 1 - r);
            }
        }
        for (int i = 0; i < N; i++) {
            const float* mn = s->node_min + 3 * i;
            const float* mx = s->node_max + 3 * i;
            const int flat = (mn[0] == mx[0]) | (mn[1] == mx[1]) | (mn[2] == mx[2]);
            float fp, ff;
            memcpy(&fp, &parent[i], 4);
            memcpy(&ff, &flat, 4);
            rnodes[i].lo = make_float4(mn[0], mn[1], mn[2], fp);
            rnodes[i].hi = make_float4(mx[0], mx[1], mx[2], ff);
            rnodes[i].pad[0] = rnodes[i].pad[1] = 0;
        }
        // NodeRec::jump (mirt_dev.h). Pre-order puts a parent before its children, so the parent's entries are final
        // when a child asks: the child's chain for a face runs on through the parent's exactly when the parent
        // inherits the face (same plane bits, other two axes no narrower; both CHECKED on the caller's boxes), and
        // inheriting is transitive along such a run (same plane again, containment of containment).
        for (int i = 0; i < N; i++)
            for (int f = 0; f < 6; f++) {
                const int p = parent[i];
                int j = p;
                if (p >= 0) {
                    const int a = f >> 1, b1 = (a + 1) % 3, b2 = (a + 2) % 3;
                    const float* cn = s->node_min + 3 * i;
                    const float* cx = s->node_max + 3 * i;
                    const float* pn = s->node_min + 3 * p;
                    const float* px = s->node_max + 3 * p;
                    const float cplane = (f & 1) ? cx[a] : cn[a], pplane = (f & 1) ? px[a] : pn[a];
                    const bool same = memcmp(&cplane, &pplane, 4) == 0;
                    const bool contains = pn[b1] <= cn[b1] && px[b1] >= cx[b1] && pn[b2] <= cn[b2] && px[b2] >= cx[b2];
                    if (same && contains) j = rnodes[p].jump[f];
                }
                rnodes[i].jump[f] = j;
            }
        if (!fast_ok) use_fast = false;
        if (use_fast) {
            FastTree ft;
            const float bmax = std::max(bound[0], std::max(bound[1], bound[2]));
            build_fast_tree(main_tree, ft, [&](int slot, float* mn, float* mx) {
                tri_fast_box(s->tri_pos + 9 * (size_t)s->slot_tid[main_src[slot]], bmax, mn, mx);
            });
            if (ft.right.empty()) {
                use_fast = false;
            } else {
                const TreeView fv{(int)ft.right.size(), T, ft.mn.data(), ft.mx.data(), ft.right.data(), ft.lbeg.data(),
                                  ft.lend.data()};
                int fdepth = ft.depth;
                append_pairs(fv, 0, fpairs, &froot_lo, &froot_hi, &fdepth);
                depth = std::max(depth, fdepth);
            }
        }
        if (fpairs.empty()) fpairs.push_back(PairRec{});
    }
    if (depth > MIRT_MAX_TREE_DEPTH)
        return fail(ctx, MIRT_E_LIMIT, "mirt_upload_scene: tree deeper than MIRT_MAX_TREE_DEPTH");
    const size_t TT = (size_t)T + (size_t)extra_slots;
    std::vector<TriRec> tris(TT);
    std::vector<VnRec> vns(TT);
    auto fill_slot = [&](size_t k, int g) {  // slot k holds global triangle g
        const float* p = s->tri_pos + 9 * (size_t)g;
        h3 v0{p[0], p[1], p[2]}, v1{p[3], p[4], p[5]}, v2{p[6], p[7], p[8]};
        // KDTree.cpp:283-284: n = normalize(lines[0] x lines[1])
        h3 n = hnormalize(hcross(hsub(v1, v0), hsub(v2, v1)));
        const float nn[3] = {n.x, n.y, n.z};
        tris[k] = tri_pack(p, p + 3, p + 6, nn);
        float fo, fg;
        int ob = s->tri_obj[g];
        memcpy(&fo, &ob, 4);
        memcpy(&fg, &g, 4);
        const int hv = s->objects[ob].has_vn != 0;
        const float* vn = hv ? s->tri_vn + 9 * (size_t)g : nullptr;
        // without vertex normals the shading normal is the face normal (SimpleShape.cpp:300-339)
        h3 fnrm = hnormalize(hcross(hsub(v1, v0), hsub(v2, v0)));
        vns[k].n0 = make_float4(vn ? vn[0] : fnrm.x, vn ? vn[1] : fnrm.y, vn ? vn[2] : fnrm.z, fo);
        vns[k].n1 = make_float4(vn ? vn[3] : 0.f, vn ? vn[4] : 0.f, vn ? vn[5] : 0.f, fg);
        float fh;
        const int flags = hv | (obj_bump[ob] << 8);  // shading_normal (mirt_dev.h)
        memcpy(&fh, &flags, 4);
        vns[k].n2 = make_float4(vn ? vn[6] : 0.f, vn ? vn[7] : 0.f, vn ? vn[8] : 0.f, fh);
        // the triangle-only terms of the barycentric weights (world_normal_raw, mirt_dev.h; SimpleShape.cpp:300-339)
        const h3 b1 = hsub(v1, v0), b2 = hsub(v2, v0);
        const float d12 = hdot(b1, b2), l1 = sqrtf(hdot(b1, b1)), l2 = sqrtf(hdot(b2, b2));
        vns[k].w = make_float4(l1, d12, d12 * d12 - l1 * l1 * l2 * l2, l1 * l1);
    };
    for (int k = 0; k < T; k++) fill_slot((size_t)k, s->slot_tid[main_src[k]]);
    std::vector<VtRec> vts(tri_vt ? TT : 0);
    auto fill_vt = [&](size_t k, int g) {
        const float* p = tri_vt + 9 * (size_t)g;
        vts[k].t0 = make_float4(p[0], p[1], p[2], 0.f);
        vts[k].t1 = make_float4(p[3], p[4], p[5], 0.f);
        vts[k].t2 = make_float4(p[6], p[7], p[8], 0.f);
    };
    if (tri_vt)
        for (int k = 0; k < T; k++) fill_vt((size_t)k, s->slot_tid[main_src[k]]);
    {
        size_t slot = (size_t)T;
        for (int k = 0; k < n_obj_trees; k++) {
            const MirtObject& ob = s->objects[obj_trees[k].object];
            for (int q = 0; q < ob.tri_count; q++) {  // (leaf ranges are kept whole: no seam order here)
                if (tri_vt) fill_vt(slot, ob.tri_begin + obj_trees[k].slot_tid[q]);
                fill_slot(slot++, ob.tri_begin + obj_trees[k].slot_tid[q]);
            }
        }
    }
    std::vector<DevObject> objs(s->n_objects);
    for (int o = 0; o < s->n_objects; o++) {
        const MirtObject& m = s->objects[o];
        DevObject& d = objs[o];
        d.type = m.type;
        d.has_vn = m.has_vn;
        d.volumetric = m.volumetric;
        d.pad = 0;
        for (int c = 0; c < 3; c++) {
            d.color[c] = m.color[c];
            d.color2[c] = m.color2[c];
            d.decay[c] = m.decay[c];
        }
        d.coeff = m.coeff;
        d.ior = m.ior;
        for (int c = 0; c < 3; c++) {
            d.sigma_s[c] = m.sigma_s[c];
            d.sigma_t[c] = m.sigma_s[c] + m.sigma_a[c];  // SceneVPMObject.h:82-86: dt = ds + da
        }
        d.g = m.g;
        d.emit_pdf_a = 0.f;
        d.tex_diffuse = object_tex ? object_tex[o].diffuse : -1;
        d.tex_gcolor = object_tex ? object_tex[o].glossy_color : -1;
        d.tex_gcoeff = object_tex ? object_tex[o].glossy_coeff : -1;
        d.tex_pad = 0;
    }
    // Emission sampler (Scene::preprocessEmissionSampler, Scene.cpp:193-204 -> SurfaceSampler::preprocess,
    // Scene.cpp:19-33 -> SceneObject::preprocessEmissionSampler, SceneObject.cpp:11-21): per emissive
    // object the running triangle areas (SimpleShape::getTriangleArea, SimpleShape.cpp:14-26: |l0 x l1| / 2,
    // fp32, accumulated in triangle order), weight = totalArea, normalised by the sum over the lights.
    std::vector<DevLight> lights;
    std::vector<float> light_area;
    std::vector<int> light_slot;
    float light_total_weight = 0.f;
    {
        std::vector<int> slot_of(T, -1);
        for (int k = 0; k < T; k++) slot_of[s->slot_tid[main_src[k]]] = k;
        for (int o = 0; o < s->n_objects; o++) {
            const MirtObject& m = s->objects[o];
            if (m.type != MIRT_OBJ_EMISSIVE) continue;
            DevLight L{};
            L.first = (int)light_area.size();
            L.count = m.tri_count;
            L.object = o;
            float total = 0.f;
            for (int k = 0; k < m.tri_count; k++) {
                const float* p = s->tri_pos + 9 * (size_t)(m.tri_begin + k);
                h3 v0{p[0], p[1], p[2]}, v1{p[3], p[4], p[5]}, v2{p[6], p[7], p[8]};
                h3 c = hcross(hsub(v1, v0), hsub(v2, v1));
                float r = 0.f;  // nvVector.h:501-507: r = 0; r += x*x; ...
                r += c.x * c.x;
                r += c.y * c.y;
                r += c.z * c.z;
                total += sqrtf(r) / 2;
                light_area.push_back(total);
                light_slot.push_back(slot_of[m.tri_begin + k]);
            }
            L.total_area = total;
            light_total_weight += total;
            L.weight_sum = light_total_weight;
            lights.push_back(L);
        }
        for (DevLight& L : lights) {
            const float weight = L.total_area / light_total_weight;  // normalizeEmissionWeight
            L.origin_prob = weight / L.total_area;
            objs[L.object].emit_pdf_a = L.origin_prob;
        }
    }

    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());  // (callers' streams included: the scene they read is replaced below)
    free_scene(ctx);
    free_retired(ctx);
    HIPCHK(ctx, hipMalloc(&ctx->d_pairs, sizeof(PairRec) * pairs.size()));
    HIPCHK(ctx, hipMemcpy(ctx->d_pairs, pairs.data(), sizeof(PairRec) * pairs.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMalloc(&ctx->d_tris, sizeof(TriRec) * TT));
    HIPCHK(ctx, hipMalloc(&ctx->d_vns, sizeof(VnRec) * TT));
    HIPCHK(ctx, hipMalloc(&ctx->d_objs, sizeof(DevObject) * (size_t)s->n_objects));
    HIPCHK(ctx, hipMemcpy(ctx->d_tris, tris.data(), sizeof(TriRec) * TT, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_vns, vns.data(), sizeof(VnRec) * TT, hipMemcpyHostToDevice));
    if (n_obj_trees > 0) {
        HIPCHK(ctx, hipMalloc(&ctx->d_vols, sizeof(VolRoot) * vols.size()));
        HIPCHK(ctx, hipMemcpy(ctx->d_vols, vols.data(), sizeof(VolRoot) * vols.size(), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMalloc(&ctx->d_vnodes, sizeof(VolNode) * vnodes.size()));
        HIPCHK(ctx, hipMemcpy(ctx->d_vnodes, vnodes.data(), sizeof(VolNode) * vnodes.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(ctx, hipMalloc(&ctx->d_fpairs, sizeof(PairRec) * fpairs.size()));
    HIPCHK(ctx, hipMemcpy(ctx->d_fpairs, fpairs.data(), sizeof(PairRec) * fpairs.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMalloc(&ctx->d_rnodes, sizeof(NodeRec) * rnodes.size()));
    HIPCHK(ctx, hipMemcpy(ctx->d_rnodes, rnodes.data(), sizeof(NodeRec) * rnodes.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMalloc(&ctx->d_slot_node, sizeof(int) * slot_node.size()));
    HIPCHK(ctx, hipMemcpy(ctx->d_slot_node, slot_node.data(), sizeof(int) * slot_node.size(), hipMemcpyHostToDevice));
    ctx->sc.fpairs = ctx->d_fpairs;
    ctx->sc.froot_lo = froot_lo;
    ctx->sc.froot_hi = froot_hi;
    ctx->sc.rnodes = ctx->d_rnodes;
    ctx->sc.slot_node = ctx->d_slot_node;
    ctx->sc.tie_rank = ctx->d_slot_node + std::max(T, 1);
    // (a scene outside the classifier's model, fast_ok = 0, would have every ray skipped by the fast walk, listed
    // by k_verify and re-run exactly: it stays on the reference tree alone)
    ctx->fast_tree = use_fast && fast_ok;
    // Modes 0 / 1 (coherent camera rays, as a rule) take the fast route as well: 4 Mi camera rays at 1024^2 take
    // 0.40 ms against 0.66 .. 1.54 ms on the reference tree on every scene of Data/ (scripts/camtest.py). Until the
    // fast tree's leaves became single triangles the two routes were close on small scenes and the first large call
    // timed both; that timing, its events and its stream synchronisation are gone.
    ctx->fast_all_modes = true;
    if (const char* e = getenv("MIRT_FAST_ALL_MODES")) ctx->fast_all_modes = atoi(e) != 0;  // A/B switch
    ctx->sc.vols = ctx->d_vols;
    ctx->sc.vnodes = ctx->d_vnodes;
    ctx->sc.n_vols = n_obj_trees;
    if (!lights.empty()) {
        HIPCHK(ctx, hipMalloc(&ctx->d_lights, sizeof(DevLight) * lights.size()));
        HIPCHK(ctx, hipMemcpy(ctx->d_lights, lights.data(), sizeof(DevLight) * lights.size(), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMalloc(&ctx->d_light_area, sizeof(float) * std::max<size_t>(1, light_area.size())));
        HIPCHK(ctx, hipMalloc(&ctx->d_light_slot, sizeof(int) * std::max<size_t>(1, light_slot.size())));
        HIPCHK(ctx, hipMemcpy(ctx->d_light_area, light_area.data(), sizeof(float) * light_area.size(), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(ctx->d_light_slot, light_slot.data(), sizeof(int) * light_slot.size(), hipMemcpyHostToDevice));
    }
    if (!vts.empty()) {
        HIPCHK(ctx, hipMalloc(&ctx->d_vts, sizeof(VtRec) * vts.size()));
        HIPCHK(ctx, hipMemcpy(ctx->d_vts, vts.data(), sizeof(VtRec) * vts.size(), hipMemcpyHostToDevice));
    }
    if (n_textures > 0) {
        HIPCHK(ctx, hipMalloc(&ctx->d_texels, sizeof(float) * texel_total));
        HIPCHK(ctx, hipMalloc(&ctx->d_textures, sizeof(DevTexture) * (size_t)n_textures));
        std::vector<DevTexture> dt(n_textures);
        size_t off = 0;
        for (int k = 0; k < n_textures; k++) {
            const MirtTexture& X = s->textures[k];
            const size_t cnt = (size_t)X.width * X.height * X.channels;
            HIPCHK(ctx, hipMemcpy(ctx->d_texels + off, X.texels, sizeof(float) * cnt, hipMemcpyHostToDevice));
            dt[k].texels = ctx->d_texels + off;
            dt[k].width = X.width, dt[k].height = X.height, dt[k].channels = X.channels, dt[k].pad = 0;
            for (int q = 0; q < 12; q++) dt[k].coord[q] = X.coord_transform[q], dt[k].color[q] = X.color_transform[q];
            off += cnt;
        }
        HIPCHK(ctx, hipMemcpy(ctx->d_textures, dt.data(), sizeof(DevTexture) * dt.size(), hipMemcpyHostToDevice));
    }
    if (!bumps.empty()) {
        HIPCHK(ctx, hipMalloc(&ctx->d_bumps, sizeof(DevBump) * bumps.size()));
        HIPCHK(ctx, hipMemcpy(ctx->d_bumps, bumps.data(), sizeof(DevBump) * bumps.size(), hipMemcpyHostToDevice));
    }
    ctx->has_bump = !bumps.empty();
    ctx->sc.bumps = ctx->d_bumps;
    ctx->sc.has_bump = ctx->has_bump ? 1 : 0;
    if (ctx->has_bump) ctx->fast_tree = false;  // k_verify / k_rerun are built without the bump branch
    ctx->sc.vts = ctx->d_vts;
    ctx->sc.textures = ctx->d_textures;
    ctx->sc.lights = ctx->d_lights;
    ctx->sc.light_area = ctx->d_light_area;
    ctx->sc.light_slot = ctx->d_light_slot;
    ctx->sc.n_lights = (int)lights.size();
    ctx->sc.light_total_weight = light_total_weight;
    HIPCHK(ctx, hipMemcpy(ctx->d_objs, objs.data(), sizeof(DevObject) * (size_t)s->n_objects, hipMemcpyHostToDevice));
    for (int c = 0; c < 3; c++) ctx->sc.bound[c] = bound[c];
    ctx->sc.fast_ok = fast_ok;
    ctx->sc.far_o = 4.f * std::max(bound[0], std::max(bound[1], bound[2]));  // (tri_fast_box's assumption, enforced: RayPre::near)
    ctx->sc.pairs = ctx->d_pairs;
    ctx->sc.root_lo = root_lo;
    ctx->sc.root_hi = root_hi;
    ctx->sc.tris = ctx->d_tris;
    ctx->sc.vns = ctx->d_vns;
    ctx->sc.objects = ctx->d_objs;
    ctx->sc.n_nodes = N;
    ctx->sc.n_tris = T;
    ctx->sc.n_objects = s->n_objects;
    ctx->sc.depth = depth;
    // per-lane stack: (depth+1) levels x (child link, entry distance) x block lanes. Only as many
    // levels live in LDS as still let the register-limited number of blocks stay resident; deeper
    // pushes (rare: the stack holds the far siblings of the current path, a handful) go to a global
    // spill area.
    ctx->block = MIRT_BLOCK;
    const int waves_per_block = std::max(1, ctx->block / 64);
    auto size_launch = [&](const void* fn, size_t* lds_main, uint32_t* lds_levels, int* max_blocks) {
        hipFuncAttributes fa{};
        int by_regs = 4;
        if (hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.numRegs > 0)
            by_regs = std::max(1, std::min(8, 512 / ((fa.numRegs + 7) / 8 * 8)));  // waves per SIMD
        const int want_blocks = std::max(1, by_regs * 4 / waves_per_block);
        const size_t lds_budget = (size_t)160 * 1024 / want_blocks;
        uint32_t levels = (uint32_t)std::min<size_t>((size_t)depth + 1, std::max<size_t>(4, lds_budget / (8 * (size_t)ctx->block)));
        if (const char* e = getenv("MIRT_LDS_LEVELS")) levels = (uint32_t)std::min<long>(depth + 1, std::max<long>(1, atol(e)));
        *lds_levels = levels;
        *lds_main = (size_t)levels * 8 * ctx->block;
        // persistent grid: as many blocks as stay resident (a larger grid would only queue; smaller
        // batches launch fewer). The occupancy answer is advisory, nothing waits on co-residency.
        *max_blocks = want_blocks * ctx->n_cus;  // (refined by the occupancy query below)
    };
    size_launch(reinterpret_cast<const void*>(&k_intersect<MIRT_MODE_NO_SELF, false, false>), &ctx->lds_main,
                &ctx->lds_levels, &ctx->max_blocks);
    size_launch(reinterpret_cast<const void*>(&k_intersect<MIRT_MODE_NO_SELF, false, false, false, true>),
                &ctx->lds_main_fast, &ctx->lds_levels_fast, &ctx->max_blocks_fast);
    {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_intersect<MIRT_MODE_NO_SELF, false, false>, ctx->block,
                                                         ctx->lds_main) == hipSuccess && occ >= 1)
            ctx->max_blocks = occ * ctx->n_cus;
        occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_intersect<MIRT_MODE_NO_SELF, false, false, false, true>,
                                                         ctx->block, ctx->lds_main_fast) == hipSuccess && occ >= 1)
            ctx->max_blocks_fast = occ * ctx->n_cus;
    }
    if (const char* e = getenv("MIRT_GRID_SCALE")) {  // experiment: leave wave slots free for another context's launches
        const double f = atof(e);
        if (f > 0.05 && f <= 1.0) {
            ctx->max_blocks = std::max(1, (int)(ctx->max_blocks * f));
            ctx->max_blocks_fast = std::max(1, (int)(ctx->max_blocks_fast * f));
        }
    }
    // one spill area serves both configurations (launches on a handle do not overlap)
    const size_t spill_a = ((size_t)depth + 1 - ctx->lds_levels) * (size_t)ctx->max_blocks;
    const size_t spill_b = ((size_t)depth + 1 - ctx->lds_levels_fast) * (size_t)ctx->max_blocks_fast;
    if (std::max(spill_a, spill_b))
        HIPCHK(ctx, hipMalloc(&ctx->d_spill, std::max(spill_a, spill_b) * ctx->block * sizeof(uint2)));
    ctx->has_scene = true;
    ctx->scene_serial++;
    ctx->has_media = n_vpm > 0;
    ctx->media_ready = media_ready;
    return MIRT_OK;
}

}  // extern "C"

// ---- launches -------------------------------------------------------------------------------------
// Scratch buffers of one closest-hit call: the context's own (hipMalloc'd, grown on demand) or, for
// mirt_intersect_noalloc_dev, slices of the static device array.
struct LaunchBufs {
    uint32_t* list;            // ray indices k_verify hands to the exact re-run (one per ray of the batch)
    uint2* spill;              // stack levels beyond the LDS share
    uint32_t max_blocks;       // grid caps the spill area is sized for: exact kernels / fast-tree kernels
    uint32_t max_blocks_fast;
};

template <int MODE, bool COUNT, bool SLOT_OUT, bool FAST>
static void launch_pair(MirtContext* ctx, const LaunchBufs& lb, hipStream_t st, dim3 grid, dim3 block, const float* O, const float* D,
                        const int* aux, uint32_t n, const uint32_t* n_ptr, uint32_t stride, float* t_out,
                        int* id_out, uint32_t* queue, const uint32_t* ray_list, uint32_t list_min = 0) {
    if constexpr (MODE == MIRT_MODE_NO_SELF && !COUNT && !FAST) {
        if (ctx->has_bump) {  // the predicate reads the bumped normal (SceneObject.cpp:250-283)
            hipLaunchKernelGGL((k_intersect<MODE, false, SLOT_OUT, false, false, true>), grid, block, ctx->lds_main, st,
                               ctx->sc, O, D, aux, n, n_ptr, stride, t_out, id_out, ctx->d_counters, queue,
                               ctx->lds_levels, lb.spill, (const float*)nullptr, (unsigned char*)nullptr, ray_list, list_min, ctx->cur_off);
            return;
        }
    }
    if constexpr (FAST) {  // built for more waves per SIMD: its own grid and LDS share
        const dim3 g(std::min<uint32_t>((n + block.x - 1) / block.x, lb.max_blocks_fast));
        hipLaunchKernelGGL((k_intersect<MODE, COUNT, SLOT_OUT, false, true>), g, block, ctx->lds_main_fast, st, ctx->sc, O,
                           D, aux, n, n_ptr, stride, t_out, id_out, ctx->d_counters, queue, ctx->lds_levels_fast,
                           lb.spill, (const float*)nullptr, (unsigned char*)nullptr, ray_list, list_min, ctx->cur_off);
        return;
    }
    hipLaunchKernelGGL((k_intersect<MODE, COUNT, SLOT_OUT, false, FAST>), grid, block, ctx->lds_main, st, ctx->sc, O, D,
                       aux, n, n_ptr, stride, t_out, id_out, ctx->d_counters, queue, ctx->lds_levels, lb.spill,
                       (const float*)nullptr, (unsigned char*)nullptr, ray_list, list_min, ctx->cur_off);
}

template <bool COUNT, bool SLOT_OUT, bool FAST>
static int launch_mode(MirtContext* ctx, const LaunchBufs& lb, hipStream_t st, dim3 grid, dim3 block, int mode, const float* O,
                       const float* D, const int* aux, uint32_t n, const uint32_t* n_ptr, uint32_t stride,
                       float* t_out, int* id_out, uint32_t* queue, const uint32_t* ray_list, uint32_t list_min = 0) {
    switch (mode) {
        case MIRT_MODE_UNCONDITIONAL:
            launch_pair<MIRT_MODE_UNCONDITIONAL, COUNT, SLOT_OUT, FAST>(ctx, lb, st, grid, block, O, D, aux, n, n_ptr, stride, t_out, id_out, queue, ray_list, list_min);
            break;
        case MIRT_MODE_LAST_TID:
            launch_pair<MIRT_MODE_LAST_TID, COUNT, SLOT_OUT, FAST>(ctx, lb, st, grid, block, O, D, aux, n, n_ptr, stride, t_out, id_out, queue, ray_list, list_min);
            break;
        case MIRT_MODE_NO_SELF:
            if (COUNT && ctx->has_bump) return fail(ctx, MIRT_E_INVALID, "intersect: scenes with bump textures cannot be instrumented");
            launch_pair<MIRT_MODE_NO_SELF, COUNT, SLOT_OUT, FAST>(ctx, lb, st, grid, block, O, D, aux, n, n_ptr, stride, t_out, id_out, queue, ray_list, list_min);
            break;
        default:
            return fail(ctx, MIRT_E_INVALID, "intersect: unknown mode");
    }
    HIPCHK(ctx, hipGetLastError());
    return MIRT_OK;
}

// list buffer of the fast path: one entry per ray of the largest batch so far. A larger batch than any before gets
// a new buffer; the old one is parked (launches already enqueued on any stream may still use it) and freed at the
// next upload / at destroy, so the call neither synchronises nor stalls the caller's pipeline.
static int ensure_fb_list(MirtContext* ctx, hipStream_t st, uint32_t n) {
    (void)st;
    if (n <= ctx->fb_cap) return MIRT_OK;
    if (ctx->d_fb_list) ctx->retired.push_back(ctx->d_fb_list);
    ctx->d_fb_list = nullptr;
    ctx->fb_cap = 0;
    const size_t want = std::max<size_t>(n, 1u << 20);  // (small batches first: do not grow in many small steps)
    HIPCHK(ctx, hipMalloc(&ctx->d_fb_list, want * sizeof(uint32_t)));
    ctx->fb_cap = want;
    return MIRT_OK;
}

// The context's own scratch buffers (the list grows to the largest batch so far: may allocate and synchronise).
static int context_bufs(MirtContext* ctx, hipStream_t st, uint32_t n, bool need_list, LaunchBufs* lb) {
    if (need_list && ensure_fb_list(ctx, st, n) != MIRT_OK) return MIRT_E_HIP;
    lb->list = ctx->d_fb_list;
    lb->spill = ctx->d_spill;
    lb->max_blocks = (uint32_t)ctx->max_blocks;
    lb->max_blocks_fast = (uint32_t)ctx->max_blocks_fast;
    return MIRT_OK;
}

// The reference tree alone: one launch of the exact kernel.
template <bool COUNT, bool SLOT_OUT>
static int route_exact(MirtContext* ctx, const LaunchBufs& lb, hipStream_t st, int mode, const float* O, const float* D,
                       const int* aux, uint32_t n, uint32_t stride, float* t_out, int* id_out, const uint32_t* n_ptr,
                       uint32_t* qwords) {
    const uint32_t want = (n + ctx->block - 1) / ctx->block;
    const dim3 block(ctx->block), grid(std::min<uint32_t>(want, lb.max_blocks));
    return launch_mode<COUNT, SLOT_OUT, false>(ctx, lb, st, grid, block, mode, O, D, aux, n, n_ptr, stride, t_out, id_out,
                                               qwords, nullptr);
}

// The fast route: k_intersect<FAST> over the fast tree, k_verify (the winner's chain of reference boxes, by
// the reference's test), and the exact re-run on the reference tree of the rays k_verify listed (few: ~1e-5 of a
// batch, plus rays with zero / extreme components).
template <bool COUNT, bool SLOT_OUT>
static int route_fast(MirtContext* ctx, const LaunchBufs& lb, hipStream_t st, int mode, const float* O, const float* D,
                      const int* aux, uint32_t n, uint32_t stride, float* t_out, int* id_out, const uint32_t* n_ptr,
                      uint32_t* qwords) {
    const uint32_t want = (n + ctx->block - 1) / ctx->block;
    const dim3 block(ctx->block), grid(std::min<uint32_t>(want, lb.max_blocks));
    int rc = launch_mode<COUNT, SLOT_OUT, true>(ctx, lb, st, grid, block, mode, O, D, aux, n, n_ptr, stride, t_out, id_out,
                                                qwords, nullptr);
    if (rc != MIRT_OK) return rc;
    hipLaunchKernelGGL((k_verify<SLOT_OUT, COUNT>), dim3((n + 255) / 256), dim3(256), 0, st, ctx->sc, O, D, n, n_ptr, stride,
                       (const float*)t_out, id_out, lb.list, qwords + 1, ctx->d_counters, ctx->cur_off);
    HIPCHK(ctx, hipGetLastError());
    // the listed rays, by the reference arithmetic on the reference tree: one wave per ray (k_rerun); the
    // instrumented call keeps the one-lane-per-ray kernel, whose counters the audit reads
    static const bool rerun_wide = !getenv("MIRT_RERUN_WIDE") || atoi(getenv("MIRT_RERUN_WIDE")) != 0;  // A/B switch
    if (!COUNT && rerun_wide) {
        const dim3 rg(std::min<uint32_t>(n, 1024)), rb(64);
        const uint32_t* list = lb.list;
        const uint32_t* count = qwords + 1;
        if (mode == MIRT_MODE_NO_SELF)
            hipLaunchKernelGGL((k_rerun<SLOT_OUT, MIRT_MODE_NO_SELF>), rg, rb, 0, st, ctx->sc, O, D, aux, stride, t_out, id_out, list, count, RERUN_WIDE_MAX, ctx->cur_off);
        else if (mode == MIRT_MODE_LAST_TID)
            hipLaunchKernelGGL((k_rerun<SLOT_OUT, MIRT_MODE_LAST_TID>), rg, rb, 0, st, ctx->sc, O, D, aux, stride, t_out, id_out, list, count, RERUN_WIDE_MAX, ctx->cur_off);
        else
            hipLaunchKernelGGL((k_rerun<SLOT_OUT, MIRT_MODE_UNCONDITIONAL>), rg, rb, 0, st, ctx->sc, O, D, aux, stride, t_out, id_out, list, count, RERUN_WIDE_MAX, ctx->cur_off);
        HIPCHK(ctx, hipGetLastError());
        // a list longer than that (whole batches outside the classifier's model: axis-parallel rays) is re-run with one
        // lane per ray by the persistent exact kernel; each of the two launches returns at once when the list is the other's
        rc = launch_mode<false, SLOT_OUT, false>(ctx, lb, st, grid, block, mode, O, D, aux, n, qwords + 1, stride, t_out, id_out,
                                                 qwords + 2, lb.list, RERUN_WIDE_MAX);
    } else {
        rc = launch_mode<COUNT, SLOT_OUT, false>(ctx, lb, st, grid, block, mode, O, D, aux, n, qwords + 1, stride, t_out,
                                                 id_out, qwords + 2, lb.list);
    }
    static const bool debug_list = getenv("MIRT_DEBUG_LIST") != nullptr;
    if (debug_list && rc == MIRT_OK) {  // diagnostics only: synchronises the stream
        uint32_t h[3] = {0, 0, 0};
        hipStreamSynchronize(st);
        hipMemcpy(h, qwords, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[mirt] fast path: %u rays, %u re-run on the reference tree\n", n, h[1]);
    }
    return rc;
}

// Closest hit of n rays (n_ptr != nullptr: the count is read on the device, at most n). `qwords`: three zeroed
// words owned by this call (queue head of the fast launch, list length, queue head of the exact launch).
// exact_only: the reference tree alone (the audit entry point, and scenes without a fast tree).
// bufs: the scratch buffers to use instead of the context's own; with them the call allocates nothing, creates no
// events and never synchronises (mirt_intersect_noalloc_dev).
template <bool COUNT, bool SLOT_OUT>
static int launch_intersect(MirtContext* ctx, hipStream_t st, int mode, const float* O, const float* D,
                            const int* aux, uint32_t n, uint32_t stride, float* t_out, int* id_out,
                            const uint32_t* n_ptr = nullptr, uint32_t* qwords = nullptr, bool exact_only = false,
                            const LaunchBufs* bufs = nullptr) {
    if (n == 0) return MIRT_OK;
    // the fast route serves every mode of a scene that has a fast tree (ctx->fast_all_modes: A/B switch for modes 0 / 1)
    const bool use_fast = !exact_only && ctx->fast_tree && (mode == MIRT_MODE_NO_SELF || ctx->fast_all_modes);
    LaunchBufs own;
    if (!bufs) {
        const int rc = context_bufs(ctx, st, n, use_fast, &own);
        if (rc != MIRT_OK) return rc;
        bufs = &own;
    }
    if (!qwords) {
        qwords = ctx->d_queue;
        HIPCHK(ctx, hipMemsetAsync(qwords, 0, 3 * sizeof(uint32_t), st));
    }
    if (!use_fast) return route_exact<COUNT, SLOT_OUT>(ctx, *bufs, st, mode, O, D, aux, n, stride, t_out, id_out, n_ptr, qwords);
    return route_fast<COUNT, SLOT_OUT>(ctx, *bufs, st, mode, O, D, aux, n, stride, t_out, id_out, n_ptr, qwords);
}

// visibility variant: vis_out[i] = 1 unless a legal hit lies nearer than limit[i]. `queue`: THREE zeroed words
// owned by this call (as in launch_intersect). In the predicate mode with a fast tree: any-hit walk of the
// fast tree, k_verify on the occluders it found, exact any-hit kernel for the rays k_verify listed.
// connect != nullptr (PathTracerTest): the pending light connections of the rays that turn out visible are added to
// their paths' radiance before this returns (by the query's own kernels on the fast route, by k_connect_all otherwise).
static int launch_visibility(MirtContext* ctx, hipStream_t st, int mode, const float* O, const float* D,
                             const int* aux, const float* limit, uint32_t n, uint32_t stride,
                             unsigned char* vis_out, const uint32_t* n_ptr = nullptr, uint32_t* queue = nullptr,
                             const ConnectArgs* connect = nullptr) {
    if (n == 0) return MIRT_OK;
    const uint32_t want = (n + ctx->block - 1) / ctx->block;
    dim3 block(ctx->block), grid(std::min<uint32_t>(want, (uint32_t)ctx->max_blocks));
    if (!queue) {
        queue = ctx->d_queue;
        HIPCHK(ctx, hipMemsetAsync(queue, 0, 3 * sizeof(uint32_t), st));
    }
    static const bool vis_fast = !getenv("MIRT_VIS_FAST") || atoi(getenv("MIRT_VIS_FAST")) != 0;  // A/B switch
    if (mode == MIRT_MODE_NO_SELF && ctx->fast_tree && vis_fast) {
        int rc = ensure_fb_list(ctx, st, n);
        if (rc != MIRT_OK) return rc;
        if (n > ctx->vis_slot_cap) {  // (as ensure_fb_list: the outgrown buffer is parked, nothing waits)
            if (ctx->d_vis_slot) ctx->retired.push_back(ctx->d_vis_slot);
            ctx->d_vis_slot = nullptr;
            ctx->vis_slot_cap = 0;
            const size_t want = std::max<size_t>(n, 1u << 20);
            HIPCHK(ctx, hipMalloc(&ctx->d_vis_slot, want * sizeof(int)));
            ctx->vis_slot_cap = want;
        }
        const dim3 gfast(std::min<uint32_t>(want, (uint32_t)ctx->max_blocks_fast));
        hipLaunchKernelGGL((k_intersect<MIRT_MODE_NO_SELF, false, false, true, true>), gfast, block, ctx->lds_main_fast, st,
                           ctx->sc, O, D, aux, n, n_ptr, stride, (float*)nullptr, ctx->d_vis_slot, ctx->d_counters, queue,
                           ctx->lds_levels_fast, ctx->d_spill, limit, vis_out);
        HIPCHK(ctx, hipGetLastError());
        if (connect)
            hipLaunchKernelGGL((k_verify<true, false, true>), dim3((n + 255) / 256), dim3(256), 0, st, ctx->sc, O, D, n, n_ptr, stride,
                               (const float*)nullptr, ctx->d_vis_slot, ctx->d_fb_list, queue + 1, ctx->d_counters,
                               (const uint32_t*)nullptr, *connect);
        else
            hipLaunchKernelGGL((k_verify<true, false>), dim3((n + 255) / 256), dim3(256), 0, st, ctx->sc, O, D, n, n_ptr, stride,
                               (const float*)nullptr, ctx->d_vis_slot, ctx->d_fb_list, queue + 1, ctx->d_counters);
        HIPCHK(ctx, hipGetLastError());
        // the listed rays on the reference tree: one wave per ray for a short list (k_rerun, as for the closest hit: the
        // lane-per-ray kernel lasts as long as its slowest ray's ~200 dependent exact box tests, 83 us per bounce of a
        // PathTracerTest render against 25), the lane-per-ray exact kernel for a long one; each returns at once when
        // the list is the other's
        static const bool rerun_wide = !getenv("MIRT_RERUN_WIDE") || atoi(getenv("MIRT_RERUN_WIDE")) != 0;  // A/B switch
        if (rerun_wide) {
            hipLaunchKernelGGL((k_rerun<false, MIRT_MODE_NO_SELF, true>), dim3(std::min<uint32_t>(n, 1024)), dim3(64), 0, st,
                               ctx->sc, O, D, aux, stride, (float*)nullptr, (int*)nullptr, (const uint32_t*)ctx->d_fb_list,
                               (const uint32_t*)(queue + 1), RERUN_WIDE_MAX, (const uint32_t*)nullptr, limit, vis_out);
            HIPCHK(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL((k_intersect<MIRT_MODE_NO_SELF, false, false, true>), grid, block, ctx->lds_main, st, ctx->sc,
                           O, D, aux, n, (const uint32_t*)(queue + 1), stride, (float*)nullptr, (int*)nullptr,
                           ctx->d_counters, queue + 2, ctx->lds_levels, ctx->d_spill, limit, vis_out,
                           (const uint32_t*)ctx->d_fb_list, rerun_wide ? RERUN_WIDE_MAX : 0u);
        HIPCHK(ctx, hipGetLastError());
        if (connect) {  // the re-run rays' connections (a short list; a long one is walked by the same few blocks)
            hipLaunchKernelGGL(k_connect_list, dim3(64), dim3(256), 0, st, *connect, (const unsigned char*)vis_out,
                               (const uint32_t*)ctx->d_fb_list, (const uint32_t*)(queue + 1));
            HIPCHK(ctx, hipGetLastError());
        }
        return MIRT_OK;
    }
#define MIRT_VIS(MODE)                                                                                        \
    hipLaunchKernelGGL((k_intersect<MODE, false, false, true>), grid, block, ctx->lds_main, st, ctx->sc, O, D, aux, \
                       n, n_ptr, stride, (float*)nullptr, (int*)nullptr, ctx->d_counters,                        \
                       queue, ctx->lds_levels, ctx->d_spill, limit, vis_out)
    switch (mode) {
        case MIRT_MODE_UNCONDITIONAL: MIRT_VIS(MIRT_MODE_UNCONDITIONAL); break;
        case MIRT_MODE_LAST_TID: MIRT_VIS(MIRT_MODE_LAST_TID); break;
        case MIRT_MODE_NO_SELF:
            if (ctx->has_bump)
                hipLaunchKernelGGL((k_intersect<MIRT_MODE_NO_SELF, false, false, true, false, true>), grid, block, ctx->lds_main,
                                   st, ctx->sc, O, D, aux, n, n_ptr, stride, (float*)nullptr, (int*)nullptr, ctx->d_counters,
                                   queue, ctx->lds_levels, ctx->d_spill, limit, vis_out);
            else
                MIRT_VIS(MIRT_MODE_NO_SELF);
            break;
        default: return fail(ctx, MIRT_E_INVALID, "visibility: unknown mode");
    }
#undef MIRT_VIS
    HIPCHK(ctx, hipGetLastError());
    if (connect) {
        hipLaunchKernelGGL(k_connect_all, dim3((n + 255) / 256), dim3(256), 0, st, *connect, (const unsigned char*)vis_out, n_ptr, n);
        HIPCHK(ctx, hipGetLastError());
    }
    return MIRT_OK;
}

static int check_intersect_args(MirtContext* ctx, const float* o, const float* d, const int32_t* aux,
                                size_t n, int mode, float* t_out, int32_t* tid_out) {
    if (!ctx) return MIRT_E_INVALID;
    if (!ctx->has_scene) return fail(ctx, MIRT_E_NO_SCENE, "intersect: no scene uploaded");
    if (mode < 0 || mode > 2) return fail(ctx, MIRT_E_INVALID, "intersect: unknown mode");
    if (n == 0) return MIRT_OK;
    if (!o || !d || !t_out || !tid_out || (mode != MIRT_MODE_UNCONDITIONAL && !aux))
        return fail(ctx, MIRT_E_INVALID, "intersect: NULL buffer");
    if (n > 0x7FFFFFFFull) return fail(ctx, MIRT_E_LIMIT, "intersect: more than 2^31-1 rays in one call");
    return MIRT_OK;
}

extern "C" {

int mirt_fast_tree(const MirtSceneDesc* s, int32_t capacity_nodes, int32_t* n_nodes_out, float* node_min,
                   float* node_max, int32_t* node_right, int32_t* leaf_begin, int32_t* leaf_end) {
    if (!s || !n_nodes_out || s->n_nodes <= 0 || s->n_tris < 0 || !s->node_min || !s->node_max || !s->node_right ||
        !s->node_leaf_begin || !s->node_leaf_end)
        return MIRT_E_INVALID;
    const TreeView view{s->n_nodes, s->n_tris, s->node_min, s->node_max, s->node_right, s->node_leaf_begin,
                        s->node_leaf_end};
    int depth = 0;
    std::string why;
    if (validate_tree(view, &depth, &why) != MIRT_OK) {
        g_create_err = "mirt_fast_tree: " + why;
        return MIRT_E_INVALID;
    }
    if (s->n_tris > 0 && (!s->tri_pos || !s->slot_tid)) return MIRT_E_INVALID;
    {  // (as mirt_upload_scene: no fast tree for a scene outside the classifier's model)
        bool in_model = bounds_in_model(view);
        for (size_t k = 0; in_model && k < 9 * (size_t)s->n_tris; k++) in_model = std::fabs(s->tri_pos[k]) < 1.1529215e18f;
        if (!in_model) {
            g_create_err = "mirt_fast_tree: coordinates that are not finite or beyond 2^60: such a scene has no fast tree";
            return MIRT_E_INVALID;
        }
    }
    FastTree ft;
    const std::vector<int> src = seam_order(view);  // device slot -> position in the reference's slot order
    float bmax = 0.f;
    for (int c = 0; c < 3; c++)
        for (float x : {s->node_min[c], s->node_max[c]}) bmax = std::max(bmax, std::fabs(x));
    build_fast_tree(view, ft, [&](int slot, float* mn, float* mx) {
        tri_fast_box(s->tri_pos + 9 * (size_t)s->slot_tid[src[slot]], bmax, mn, mx);
    });
    const int32_t n = (int32_t)ft.right.size();
    *n_nodes_out = n;
    if (n > capacity_nodes) return MIRT_E_LIMIT;
    if (n && (!node_min || !node_max || !node_right || !leaf_begin || !leaf_end)) return MIRT_E_INVALID;
    for (int32_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            node_min[3 * i + c] = ft.mn[3 * i + c];
            node_max[3 * i + c] = ft.mx[3 * i + c];
        }
        node_right[i] = ft.right[i];
        // a leaf is one triangle, given by its position in the reference's slot order
        leaf_begin[i] = ft.lbeg[i] >= 0 ? src[ft.lbeg[i]] : -1;
        leaf_end[i] = ft.lbeg[i] >= 0 ? src[ft.lbeg[i]] + 1 : -1;
    }
    return MIRT_OK;
}

int mirt_intersect_dev(MirtContext* ctx, const float* origins, const float* directions,
                       const int32_t* aux, size_t n, int mode, float* t_out, int32_t* tid_out,
                       void* stream) {
    int rc = check_intersect_args(ctx, origins, directions, aux, n, mode, t_out, tid_out);
    if (rc != MIRT_OK || n == 0) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    return launch_intersect<false, false>(ctx, st, mode, origins, directions, aux, (uint32_t)n,
                                          (uint32_t)n, t_out, tid_out);
}

// mirt_intersect_dev without a single allocation: scratch from this device's copy of the static array.
int mirt_intersect_noalloc_dev(MirtContext* ctx, const float* origins, const float* directions,
                               const int32_t* aux, size_t n, int mode, float* t_out, int32_t* tid_out,
                               void* stream) {
    int rc = check_intersect_args(ctx, origins, directions, aux, n, mode, t_out, tid_out);
    if (rc != MIRT_OK || n == 0) return rc;
    if (n > MIRT_NOALLOC_MAX_RAYS)
        return fail(ctx, MIRT_E_LIMIT, "intersect_noalloc: more than MIRT_NOALLOC_MAX_RAYS (4 Mi) rays in one call");
    if (!ctx->d_scratch) return fail(ctx, MIRT_E_HIP, "intersect_noalloc: the static scratch array was not resolved");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    LaunchBufs lb;
    uint32_t* qwords = reinterpret_cast<uint32_t*>(ctx->d_scratch);
    lb.list = reinterpret_cast<uint32_t*>(ctx->d_scratch + NOALLOC_QUEUE_BYTES);
    lb.spill = reinterpret_cast<uint2*>(ctx->d_scratch + NOALLOC_QUEUE_BYTES + NOALLOC_LIST_BYTES);
    // the persistent grid, cut down until the stack levels beyond the LDS share fit the spill slice
    auto cap = [&](int max_blocks, uint32_t lds_levels) -> uint32_t {
        const size_t deep = (size_t)ctx->sc.depth + 1 > lds_levels ? (size_t)ctx->sc.depth + 1 - lds_levels : 0;
        if (deep == 0) return (uint32_t)max_blocks;
        return (uint32_t)std::min<size_t>((size_t)max_blocks, NOALLOC_SPILL_BYTES / (deep * (size_t)ctx->block * sizeof(uint2)));
    };
    lb.max_blocks = cap(ctx->max_blocks, ctx->lds_levels);
    lb.max_blocks_fast = cap(ctx->max_blocks_fast, ctx->lds_levels_fast);
    if (lb.max_blocks == 0 || lb.max_blocks_fast == 0)
        return fail(ctx, MIRT_E_LIMIT, "intersect_noalloc: the tree is too deep for the static spill area");
    // the array keeps the last call's contents: the three queue / count words must start at zero
    HIPCHK(ctx, hipMemsetAsync(qwords, 0, 3 * sizeof(uint32_t), st));
    return launch_intersect<false, false>(ctx, st, mode, origins, directions, aux, (uint32_t)n, (uint32_t)n, t_out, tid_out,
                                          nullptr, qwords, false, &lb);
}

int mirt_intersect_audit_dev(MirtContext* ctx, const float* origins, const float* directions,
                             const int32_t* aux, size_t n, int mode, float* t_out, int32_t* tid_out,
                             uint64_t audit[8]) {
    int rc = check_intersect_args(ctx, origins, directions, aux, n, mode, t_out, tid_out);
    if (rc != MIRT_OK) return rc;
    if (!audit) return fail(ctx, MIRT_E_INVALID, "intersect_audit: NULL counter array");
    for (int k = 0; k < 8; k++) audit[k] = 0;
    if (n == 0) return MIRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_counters, 0, 8 * sizeof(unsigned long long), ctx->stream));
    rc = launch_intersect<true, false>(ctx, ctx->stream, mode, origins, directions, aux,
                                       (uint32_t)n, (uint32_t)n, t_out, tid_out, nullptr, nullptr,
                                       getenv("MIRT_AUDIT_FAST") == nullptr);
    if (rc != MIRT_OK) return rc;
    unsigned long long h[8];
    HIPCHK(ctx, hipMemcpyAsync(h, ctx->d_counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 8; k++) audit[k] = h[k];
    return MIRT_OK;
}

int mirt_intersect_count_dev(MirtContext* ctx, const float* origins, const float* directions,
                             const int32_t* aux, size_t n, int mode, float* t_out, int32_t* tid_out,
                             uint64_t* n_box_tests, uint64_t* n_tri_tests) {
    if (ctx && (!n_box_tests || !n_tri_tests)) return fail(ctx, MIRT_E_INVALID, "intersect_count: NULL counter");
    int rc = check_intersect_args(ctx, origins, directions, aux, n, mode, t_out, tid_out);
    if (rc != MIRT_OK) return rc;
    *n_box_tests = *n_tri_tests = 0;
    if (n == 0) return MIRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_counters, 0, 8 * sizeof(unsigned long long), ctx->stream));
    // the product's own path, instrumented: fast-tree traversal + chain verification + exact re-runs
    rc = launch_intersect<true, false>(ctx, ctx->stream, mode, origins, directions, aux, (uint32_t)n, (uint32_t)n,
                                       t_out, tid_out);
    if (rc != MIRT_OK) return rc;
    unsigned long long h[8];
    HIPCHK(ctx, hipMemcpyAsync(h, ctx->d_counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *n_box_tests = h[0];
    *n_tri_tests = h[1];
    return MIRT_OK;
}

int mirt_intersect(MirtContext* ctx, const float* origins, const float* directions, const int32_t* aux,
                   size_t n, int mode, float* t_out, int32_t* tid_out) {
    int rc = check_intersect_args(ctx, origins, directions, aux, n, mode, t_out, tid_out);
    if (rc != MIRT_OK || n == 0) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t CH = 1u << 24;  // rays per staged chunk
    size_t cap = std::min(n, CH);
    if (cap > ctx->st_cap) {
        free_staging(ctx);
        HIPCHK(ctx, hipMalloc(&ctx->st_o, 3 * cap * sizeof(float)));
        HIPCHK(ctx, hipMalloc(&ctx->st_d, 3 * cap * sizeof(float)));
        HIPCHK(ctx, hipMalloc(&ctx->st_aux, cap * sizeof(int)));
        HIPCHK(ctx, hipMalloc(&ctx->st_t, cap * sizeof(float)));
        HIPCHK(ctx, hipMalloc(&ctx->st_id, cap * sizeof(int)));
        ctx->st_cap = cap;
    }
    hipStream_t st = ctx->stream;
    for (size_t b = 0; b < n; b += CH) {
        size_t m = std::min(CH, n - b);
        for (int c = 0; c < 3; c++) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->st_o + c * m, origins + c * n + b, m * sizeof(float), hipMemcpyHostToDevice, st));
            HIPCHK(ctx, hipMemcpyAsync(ctx->st_d + c * m, directions + c * n + b, m * sizeof(float), hipMemcpyHostToDevice, st));
        }
        if (mode != MIRT_MODE_UNCONDITIONAL)
            HIPCHK(ctx, hipMemcpyAsync(ctx->st_aux, aux + b, m * sizeof(int), hipMemcpyHostToDevice, st));
        rc = launch_intersect<false, false>(ctx, st, mode, ctx->st_o, ctx->st_d, ctx->st_aux,
                                            (uint32_t)m, (uint32_t)m, ctx->st_t, ctx->st_id);
        if (rc != MIRT_OK) return rc;
        HIPCHK(ctx, hipMemcpyAsync(t_out + b, ctx->st_t, m * sizeof(float), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(tid_out + b, ctx->st_id, m * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    return MIRT_OK;
}

int mirt_visibility_dev(MirtContext* ctx, const float* origins, const float* directions, const int32_t* aux,
                        const float* max_dist, size_t n, int mode, uint8_t* visible, void* stream) {
    float dummy_t;
    int32_t dummy_i;
    int rc = check_intersect_args(ctx, origins, directions, aux, n, mode, &dummy_t, &dummy_i);
    if (rc != MIRT_OK) return rc;
    if (n && (!max_dist || !visible)) return fail(ctx, MIRT_E_INVALID, "visibility: NULL max_dist / visible");
    if (n == 0) return MIRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    return launch_visibility(ctx, st, mode, origins, directions, aux, max_dist, (uint32_t)n, (uint32_t)n, visible);
}

int mirt_visibility(MirtContext* ctx, const float* origins, const float* directions, const int32_t* aux,
                    const float* max_dist, size_t n, int mode, uint8_t* visible) {
    float dummy_t;
    int32_t dummy_i;
    int rc = check_intersect_args(ctx, origins, directions, aux, n, mode, &dummy_t, &dummy_i);
    if (rc != MIRT_OK) return rc;
    if (n && (!max_dist || !visible)) return fail(ctx, MIRT_E_INVALID, "visibility: NULL max_dist / visible");
    if (n == 0) return MIRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t CH = 1u << 24;  // rays per staged chunk
    size_t cap = std::min(n, CH);
    if (cap > ctx->st_cap) {
        free_staging(ctx);
        HIPCHK(ctx, hipMalloc(&ctx->st_o, 3 * cap * sizeof(float)));
        HIPCHK(ctx, hipMalloc(&ctx->st_d, 3 * cap * sizeof(float)));
        HIPCHK(ctx, hipMalloc(&ctx->st_aux, cap * sizeof(int)));
        HIPCHK(ctx, hipMalloc(&ctx->st_t, cap * sizeof(float)));
        HIPCHK(ctx, hipMalloc(&ctx->st_id, cap * sizeof(int)));
        ctx->st_cap = cap;
    }
    hipStream_t st = ctx->stream;
    for (size_t b = 0; b < n; b += CH) {
        size_t m = std::min(CH, n - b);
        for (int c = 0; c < 3; c++) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->st_o + c * m, origins + c * n + b, m * sizeof(float), hipMemcpyHostToDevice, st));
            HIPCHK(ctx, hipMemcpyAsync(ctx->st_d + c * m, directions + c * n + b, m * sizeof(float), hipMemcpyHostToDevice, st));
        }
        if (mode != MIRT_MODE_UNCONDITIONAL)
            HIPCHK(ctx, hipMemcpyAsync(ctx->st_aux, aux + b, m * sizeof(int), hipMemcpyHostToDevice, st));
        // limits travel in the t staging buffer, flags come back in the id staging buffer
        HIPCHK(ctx, hipMemcpyAsync(ctx->st_t, max_dist + b, m * sizeof(float), hipMemcpyHostToDevice, st));
        rc = launch_visibility(ctx, st, mode, ctx->st_o, ctx->st_d, ctx->st_aux, ctx->st_t, (uint32_t)m, (uint32_t)m,
                               (unsigned char*)ctx->st_id);
        if (rc != MIRT_OK) return rc;
        HIPCHK(ctx, hipMemcpyAsync(visible + b, ctx->st_id, m, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    return MIRT_OK;
}

}  // extern "C"

// ---- camera + wavefront loop ------------------------------------------------------------------------
static int make_camera(MirtContext* ctx, const MirtCamera* c, DevCamera* out, const MirtTile* tile = nullptr) {
    if (!c || c->width <= 0 || c->height <= 0)
        return fail(ctx, MIRT_E_INVALID, "camera: width/height must be positive");
    if ((uint64_t)c->width * (uint64_t)c->height > (1ull << 28))
        return fail(ctx, MIRT_E_LIMIT, "camera: more than 2^28 pixels");
    if (tile && (tile->x0 < 0 || tile->y0 < 0 || tile->width <= 0 || tile->height <= 0 ||
                 (int64_t)tile->x0 + tile->width > c->width || (int64_t)tile->y0 + tile->height > c->height))
        return fail(ctx, MIRT_E_INVALID, "tile: the pixel window must be non-empty and lie inside the image");
    out->tx0 = tile ? tile->x0 : 0;
    out->ty0 = tile ? tile->y0 : 0;
    out->tw = tile ? tile->width : c->width;
    out->th = tile ? tile->height : c->height;
    // Camera.cpp:6-10
    h3 pos{c->position[0], c->position[1], c->position[2]};
    h3 front = hnormalize(hsub(h3{c->focus[0], c->focus[1], c->focus[2]}, pos));
    h3 right = hnormalize(hcross(front, h3{c->up[0], c->up[1], c->up[2]}));
    h3 oup = hcross(right, front);
    out->pos = f3{pos.x, pos.y, pos.z};
    out->front = f3{front.x, front.y, front.z};
    out->right = f3{right.x, right.y, right.z};
    out->up = f3{oup.x, oup.y, oup.z};
    out->width = c->width;
    out->height = c->height;
    out->sight = (float)c->sight_dist;
    out->half_w = c->width / 2.0;
    out->half_h = c->height / 2.0;
    return MIRT_OK;
}

static int alloc_pathbuf(MirtContext* ctx, PathBuf& p, size_t cap) {
    cap += 256;  // two-ended compaction fills [mid - B, mid + A) with mid a multiple of the block size: up to 255 entries beyond the bound
    HIPCHK(ctx, hipMalloc(&p.o, 3 * cap * sizeof(float)));
    HIPCHK(ctx, hipMalloc(&p.d, 3 * cap * sizeof(float)));
    HIPCHK(ctx, hipMalloc(&p.thr, 3 * cap * sizeof(float)));
    HIPCHK(ctx, hipMalloc(&p.t, cap * sizeof(float)));
    HIPCHK(ctx, hipMalloc(&p.slot, cap * sizeof(int)));
    HIPCHK(ctx, hipMalloc(&p.key, cap * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&p.ctr, cap * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&p.ids, cap * sizeof(int)));
    HIPCHK(ctx, hipMalloc(&p.pid, cap * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&p.lpw, cap * sizeof(float)));
    p.stride = (uint32_t)cap;
    return MIRT_OK;
}

// path state for wavefronts of up to `cap` paths, `gens` radiance buffers in the L ring, and a carry buffer of
// `carry_cap` paths (0: none)
static int ensure_paths(MirtContext* ctx, size_t cap, int gens = 1, size_t carry_cap = 0) {
    if (cap <= ctx->pb_cap && gens <= ctx->l_gens && carry_cap <= ctx->carry_cap) return MIRT_OK;
    cap = std::max(cap, ctx->pb_cap);
    gens = std::max(gens, ctx->l_gens);
    carry_cap = std::max(carry_cap, ctx->carry_cap);
    free_paths(ctx);
    for (int k = 0; k < 2; k++) {
        const int rc = alloc_pathbuf(ctx, ctx->pb[k], cap);
        if (rc != MIRT_OK) return rc;
    }
    HIPCHK(ctx, hipMalloc(&ctx->d_L, (size_t)gens * 3 * cap * sizeof(float)));
    if (carry_cap) {
        const int rc = alloc_pathbuf(ctx, ctx->carry, carry_cap);
        if (rc != MIRT_OK) return rc;
        HIPCHK(ctx, hipMalloc(&ctx->d_carry_cnt, sizeof(uint32_t)));
    }
    ctx->pb_cap = cap;
    ctx->l_gens = gens;
    ctx->carry_cap = carry_cap;
    return MIRT_OK;
}

// shadow-ray buffers of PathTracerTest: at most one light connection per live path and bounce
static int ensure_shadow(MirtContext* ctx, size_t cap) {
    if (cap <= ctx->shb_cap) return MIRT_OK;
    ShadowBuf& b = ctx->shb;
    hipFree(b.o); hipFree(b.d); hipFree(b.ids); hipFree(b.lim); hipFree(b.c); hipFree(b.pid); hipFree(b.vis);
    b = ShadowBuf{};
    ctx->shb_cap = 0;
    HIPCHK(ctx, hipMalloc(&b.o, 3 * cap * sizeof(float)));
    HIPCHK(ctx, hipMalloc(&b.d, 3 * cap * sizeof(float)));
    HIPCHK(ctx, hipMalloc(&b.c, 3 * cap * sizeof(float)));
    HIPCHK(ctx, hipMalloc(&b.ids, cap * sizeof(int)));
    HIPCHK(ctx, hipMalloc(&b.lim, cap * sizeof(float)));
    HIPCHK(ctx, hipMalloc(&b.pid, cap * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&b.vis, cap));
    b.stride = (uint32_t)cap;
    ctx->shb_cap = cap;
    return MIRT_OK;
}

struct Capture {
    uint32_t depth_lo, depth_hi;
    float* o;
    float* d;
    int* aux;
    size_t capacity, count;
};

static hipEvent_t get_event(MirtContext* ctx, size_t k) {
    while (ctx->events.size() <= k) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        ctx->events.push_back(ev);
    }
    return ctx->events[k];
}

// per-bounce count / queue words of a wavefront (device) and their pinned host mirror
static int ensure_counts(MirtContext* ctx, hipStream_t st, size_t words) {
    if (words <= ctx->cnt_cap) return MIRT_OK;
    HIPCHK(ctx, hipStreamSynchronize(st));
    hipFree(ctx->d_cnt);
    hipFree(ctx->d_ab);
    hipFree(ctx->d_off);
    hipFree(ctx->d_wq);
    hipFree(ctx->d_ns);
    hipFree(ctx->d_wq2);
    hipHostFree(ctx->h_cnt);
    ctx->d_cnt = ctx->d_ab = ctx->d_off = ctx->d_wq = ctx->d_ns = ctx->d_wq2 = ctx->h_cnt = nullptr;
    ctx->cnt_cap = 0;
    HIPCHK(ctx, hipMalloc(&ctx->d_cnt, words * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->d_ab, 64 * words * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->d_off, words * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->d_wq, 3 * words * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->d_ns, words * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->d_wq2, 3 * words * sizeof(uint32_t)));
    HIPCHK(ctx, hipHostMalloc(&ctx->h_cnt, words * sizeof(uint32_t)));
    ctx->cnt_cap = words;
    return MIRT_OK;
}

// One wavefront of `S` samples per pixel starting at global sample index s0.
// The live-path count of every bounce stays on the device (d_cnt[depth]); the host enqueues the
// whole bounce sequence and looks at a count only now and then (to stop early) or when rays are
// being captured. *n_rays_dev accumulates the rays traced.
static int run_wavefront(MirtContext* ctx, const DevCamera& cam, const MirtRenderParams* prm,
                         uint32_t s0, uint32_t S, float* rgb_sum, Capture* cap, uint64_t* n_rays,
                         size_t* ev_used, bool nee = false, uint64_t* n_shadow = nullptr) {
    const uint32_t wh = (uint32_t)cam.tw * (uint32_t)cam.th;  // pixels of the window being rendered
    const uint32_t P = wh * S;
    hipStream_t st = ctx->stream;
    const int B = 256;
    PathBuf* cur = &ctx->pb[0];
    PathBuf* nxt = &ctx->pb[1];
    const size_t words = (size_t)prm->max_depth + 4;
    {
        const int rc = ensure_counts(ctx, st, words);
        if (rc != MIRT_OK) return rc;
    }
    if (nee) {
        HIPCHK(ctx, hipMemsetAsync(ctx->d_ns, 0, words * sizeof(uint32_t), st));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_wq2, 0, 3 * words * sizeof(uint32_t), st));
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->d_L, 0, 3 * ctx->pb_cap * sizeof(float), st));  // (generation 0 of the L ring)
    HIPCHK(ctx, hipMemsetAsync(ctx->d_cnt, 0, words * sizeof(uint32_t), st));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_ab, 0, 64 * words * sizeof(uint32_t), st));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_wq, 0, 3 * words * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_generate, dim3((P + B - 1) / B), dim3(B), 0, st, cam, P, s0, prm->seed, cur->o,
                       cur->d, cur->stride, *cur, nee ? 2 : 1);
    HIPCHK(ctx, hipGetLastError());
    if (ctx->sc.n_vols > 0) {  // Camera.cpp:35: the camera rays' inside object
        hipLaunchKernelGGL(k_inside0, dim3((P + 255) / 256), dim3(256), 0, st, ctx->sc, cur->o, cur->d, cur->stride, P, cur->ids);
        HIPCHK(ctx, hipGetLastError());
    }
    auto timed_intersect = [&](int mode, PathBuf* pb, uint32_t n, const uint32_t* n_ptr, uint32_t* queue,
                               const uint32_t* off = nullptr) -> int {
        hipEvent_t e0 = get_event(ctx, *ev_used), e1 = get_event(ctx, *ev_used + 1);
        if (!e0 || !e1) return fail(ctx, MIRT_E_HIP, "hipEventCreate failed");
        *ev_used += 2;
        HIPCHK(ctx, hipEventRecord(e0, st));
        ctx->cur_off = off;  // (where the batch starts in pb's arrays: two-ended compaction)
        int rc = launch_intersect<false, true>(ctx, st, mode, pb->o, pb->d, pb->ids, n, pb->stride,
                                               pb->t, pb->slot, n_ptr, queue);
        ctx->cur_off = nullptr;
        if (rc != MIRT_OK) return rc;
        HIPCHK(ctx, hipEventRecord(e1, st));
        return MIRT_OK;
    };
    // d_cnt[k] = paths alive after k bounces (k >= 1); the camera pass has P
    int rc = timed_intersect(MIRT_MODE_UNCONDITIONAL, cur, P, nullptr, ctx->d_wq);  // Camera.cpp:27-33: no predicate
    if (rc != MIRT_OK) return rc;
    uint32_t bound = P;  // host-side upper bound of the live paths
    uint32_t depths_run = 0;
    for (uint32_t depth = 0; depth <= prm->max_depth; depth++) {
        const uint32_t* n_in = depth ? ctx->d_cnt + depth : nullptr;
        const uint32_t* off_in = depth ? ctx->d_off + depth : nullptr;  // (camera hits lie at [0, P))
        uint32_t* n_out = ctx->d_cnt + depth + 1;
        uint32_t* off_out = ctx->d_off + depth + 1;
        uint32_t* ab = ctx->d_ab + 64 * (size_t)(depth + 1);
        // two-ended compaction (pt_kernels.h, k_shade<.., TWO>): survivors land in [mid - ab[32], mid + ab[0]); k_live
        // turns the two counters into the range's length (n_out) and start (off_out) for the next launches
        const uint32_t sgrid = (bound + SHADE_BLOCK - 1) / SHADE_BLOCK, mid = (sgrid >> 1) * (uint32_t)SHADE_BLOCK;
        if (nee) {
            // PathTracerTest: shade writes one shadow ray per RANDOM vertex; the visibility launch and
            // its kernels add the unoccluded contributions before the next bounce touches L again
            uint32_t* n_sh = ctx->d_ns + depth;
            if (ctx->has_media)
                hipLaunchKernelGGL((k_shade<true, true, true>), dim3(sgrid), dim3(SHADE_BLOCK), 0, st, ctx->sc, cam, *cur,
                                   n_in, bound, *nxt, ab, ctx->d_L, (uint32_t)ctx->pb_cap, prm->max_depth,
                                   ctx->shb, n_sh, off_in);
            else
                hipLaunchKernelGGL((k_shade<false, true, true>), dim3(sgrid), dim3(SHADE_BLOCK), 0, st, ctx->sc, cam, *cur,
                                   n_in, bound, *nxt, ab, ctx->d_L, (uint32_t)ctx->pb_cap, prm->max_depth,
                                   ctx->shb, n_sh, off_in);
            HIPCHK(ctx, hipGetLastError());
            const ConnectArgs ca{ctx->shb.c, ctx->shb.stride, ctx->shb.pid, ctx->d_L, (uint32_t)ctx->pb_cap, PID_MASK, (uint32_t)PID_GEN_SHIFT};
            int rcv = launch_visibility(ctx, st, MIRT_MODE_NO_SELF, ctx->shb.o, ctx->shb.d, ctx->shb.ids, ctx->shb.lim,
                                        bound, ctx->shb.stride, ctx->shb.vis, n_sh, ctx->d_wq2 + 3 * depth, &ca);
            if (rcv != MIRT_OK) return rcv;
        } else if (ctx->has_media)
            hipLaunchKernelGGL((k_shade<true, false, true>), dim3(sgrid), dim3(SHADE_BLOCK), 0, st, ctx->sc, cam, *cur, n_in,
                               bound, *nxt, ab, ctx->d_L, (uint32_t)ctx->pb_cap, prm->max_depth, ShadowBuf{},
                               (uint32_t*)nullptr, off_in);
        else
            hipLaunchKernelGGL((k_shade<false, false, true>), dim3(sgrid), dim3(SHADE_BLOCK), 0, st, ctx->sc, cam, *cur, n_in,
                               bound, *nxt, ab, ctx->d_L, (uint32_t)ctx->pb_cap, prm->max_depth, ShadowBuf{},
                               (uint32_t*)nullptr, off_in);
        HIPCHK(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_live, dim3(1), dim3(1), 0, st, (const uint32_t*)ab, mid, n_out, off_out);
        HIPCHK(ctx, hipGetLastError());
        depths_run = depth + 1;
        if (depth == prm->max_depth) break;  // nothing survives the depth cap (MCRenderer.cpp:187-191)
        std::swap(cur, nxt);
        // look at the count when capturing, and at depths 4, 8, 16, ... to stop early / shrink grids
        const bool look = cap || (depth >= 3 && ((depth + 1) & depth) == 0);
        uint32_t off_host = 0;
        if (look) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_cnt, n_out, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_cnt + 1, off_out, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            const uint32_t n = ctx->h_cnt[0];
            off_host = ctx->h_cnt[1];
            if (n > bound || (size_t)off_host + n > (size_t)cur->stride)
                return fail(ctx, MIRT_E_HIP, "wavefront: live-path counter corrupted");
            bound = n;
            if (n == 0) break;
        }
        if (cap && depth + 1 >= cap->depth_lo && depth + 1 <= cap->depth_hi && cap->count < cap->capacity) {
            size_t m = std::min((size_t)bound, cap->capacity - cap->count);
            for (int c = 0; c < 3; c++) {
                HIPCHK(ctx, hipMemcpyAsync(cap->o + c * cap->capacity + cap->count, cur->o + (size_t)c * cur->stride + off_host,
                                           m * sizeof(float), hipMemcpyDeviceToDevice, st));
                HIPCHK(ctx, hipMemcpyAsync(cap->d + c * cap->capacity + cap->count, cur->d + (size_t)c * cur->stride + off_host,
                                           m * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
            HIPCHK(ctx, hipMemcpyAsync(cap->aux + cap->count, cur->ids + off_host, m * sizeof(int), hipMemcpyDeviceToDevice, st));
            cap->count += m;
        }
        rc = timed_intersect(MIRT_MODE_NO_SELF, cur, bound, n_out, ctx->d_wq + 3 * (depth + 1), off_out);  // MCRenderer.cpp:193-197
        if (rc != MIRT_OK) return rc;
    }
    if (rgb_sum) {
        hipLaunchKernelGGL(k_accumulate, dim3((wh + B - 1) / B), dim3(B), 0, st, ctx->d_L,
                           (uint32_t)ctx->pb_cap, wh, S, rgb_sum);
        HIPCHK(ctx, hipGetLastError());
    }
    // rays traced by this wavefront = P + sum of the live counts (read after the stream drains)
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_cnt, ctx->d_cnt, (depths_run + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    *n_rays += P;
    for (uint32_t k = 1; k < depths_run + 1 && k <= prm->max_depth; k++) *n_rays += ctx->h_cnt[k];
    if (nee && n_shadow) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_cnt, ctx->d_ns, depths_run * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        for (uint32_t k = 0; k < depths_run; k++) *n_shadow += ctx->h_cnt[k];
    }
    return MIRT_OK;
}

// Wavefront chaining (plain path tracing, several wavefronts per call). A wavefront's last bounces hold a few
// thousand paths each and cost ~0.18 ms apiece whatever their size (one ray's ~30 dependent traversal phases): 13 %
// of a 32 Mi-path wavefront went into its last thirteen bounces. Here a wavefront is CUT after `cut` bounces: the
// survivors, already intersected, move to a carry buffer and join the NEXT wavefront behind its camera hits, so every
// launch of every wavefront but the last stays large. That works because k_shade reads each path's depth from the
// path (top byte of its RNG counter word) and scores into the radiance buffer of the path's own wavefront (top bits
// of its path id select one of `gens` buffers of the L ring); buffer g is added to the image, in wavefront order,
// just before wavefront w + gens reuses it, when every path of wavefront w has ended (a path lives through at most
// max_depth / cut + 1 wavefronts). Results are bit-identical to unchained wavefronts: same streams, same radiance
// values, same order of accumulation. A cut whose survivors would not fit the carry buffer (a scene that absorbs
// nothing) is skipped: that wavefront runs to its end.
static uint32_t chain_cut(uint32_t max_depth) {
    static const int forced = getenv("MIRT_CHAIN_CUT") ? atoi(getenv("MIRT_CHAIN_CUT")) : 0;  // experiments
    if (forced > 0) return (uint32_t)forced;
    // swept on the Cornell scene (32 Mi paths, Msamples/s): cut 6 1 293, 7 1 283, 8 1 275, 11 1 244 (unchained 1 187);
    // the L ring holds max_depth / cut + 1 <= 4 buffers (two bits of the path id)
    return std::max<uint32_t>(6u, max_depth / 4u + 1u);
}

static int run_chained(MirtContext* ctx, const DevCamera& cam, const MirtRenderParams* prm, uint32_t S, float* rgb_sum,
                       uint64_t* n_rays, size_t* ev_used, MirtRenderStats* stats, bool nee = false,
                       uint64_t* n_shadow = nullptr) {
    const uint32_t wh = (uint32_t)cam.tw * (uint32_t)cam.th;  // pixels of the window being rendered
    const uint32_t cut = chain_cut(prm->max_depth);
    const int gens = (int)(prm->max_depth / cut) + 1;
    const uint32_t n_wf = (prm->spp_count + S - 1) / S;
    const size_t cap = ctx->pb_cap;  // (ensure_paths was called by render_common with the carry share included)
    // this call's share of the carry buffer: a wavefront holds at most wh * S + carry_limit <= cap paths (the
    // buffers themselves may be larger, left by an earlier call)
    const size_t carry_limit = (size_t)wh * S / 4;
    hipStream_t st = ctx->stream;
    const int B = 256;
    const size_t words = (size_t)prm->max_depth + 4;
    int rc = ensure_counts(ctx, st, words);
    if (rc != MIRT_OK) return rc;
    auto samples_of = [&](uint32_t w) { return std::min(S, prm->spp_count - w * S); };
    auto accumulate = [&](uint32_t w) -> int {  // wavefront w's radiance buffer into the image
        hipLaunchKernelGGL(k_accumulate, dim3((wh + B - 1) / B), dim3(B), 0, st,
                           (const float*)(ctx->d_L + (size_t)(w % gens) * 3 * cap), (uint32_t)cap, wh, samples_of(w), rgb_sum);
        HIPCHK(ctx, hipGetLastError());
        return MIRT_OK;
    };
    uint32_t carried = 0;  // host-side upper bound of the paths waiting in the carry buffer
    for (uint32_t w = 0; w < n_wf; w++) {
        const uint32_t s = samples_of(w), P = wh * s, g = w % (uint32_t)gens;
        const bool last = w + 1 == n_wf;
        PathBuf* cur = &ctx->pb[0];
        PathBuf* nxt = &ctx->pb[1];
        if (w >= (uint32_t)gens && (rc = accumulate(w - gens)) != MIRT_OK) return rc;  // its buffer is reused now
        HIPCHK(ctx, hipMemsetAsync(ctx->d_L + (size_t)g * 3 * cap, 0, 3 * cap * sizeof(float), st));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_cnt, 0, words * sizeof(uint32_t), st));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_ab, 0, 64 * words * sizeof(uint32_t), st));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_wq, 0, 3 * words * sizeof(uint32_t), st));
        if (nee) {
            HIPCHK(ctx, hipMemsetAsync(ctx->d_ns, 0, words * sizeof(uint32_t), st));
            HIPCHK(ctx, hipMemsetAsync(ctx->d_wq2, 0, 3 * words * sizeof(uint32_t), st));
        }
        hipLaunchKernelGGL(k_generate, dim3((P + B - 1) / B), dim3(B), 0, st, cam, P, prm->spp_begin + w * S, prm->seed, cur->o,
                           cur->d, cur->stride, *cur, nee ? 2 : 1, g << PID_GEN_SHIFT);
        HIPCHK(ctx, hipGetLastError());
        if (ctx->sc.n_vols > 0) {  // Camera.cpp:35: the camera rays' inside object
            hipLaunchKernelGGL(k_inside0, dim3((P + 255) / 256), dim3(256), 0, st, ctx->sc, cur->o, cur->d, cur->stride, P, cur->ids);
            HIPCHK(ctx, hipGetLastError());
        }
        auto timed_intersect = [&](int mode, PathBuf* pb, uint32_t n, const uint32_t* n_ptr, uint32_t* queue,
                                   const uint32_t* off = nullptr) -> int {
            hipEvent_t e0 = get_event(ctx, *ev_used), e1 = get_event(ctx, *ev_used + 1);
            if (!e0 || !e1) return fail(ctx, MIRT_E_HIP, "hipEventCreate failed");
            *ev_used += 2;
            HIPCHK(ctx, hipEventRecord(e0, st));
            ctx->cur_off = off;  // (where the batch starts in pb's arrays: two-ended compaction)
            const int r = launch_intersect<false, true>(ctx, st, mode, pb->o, pb->d, pb->ids, n, pb->stride, pb->t, pb->slot,
                                                        n_ptr, queue);
            ctx->cur_off = nullptr;
            if (r != MIRT_OK) return r;
            HIPCHK(ctx, hipEventRecord(e1, st));
            return MIRT_OK;
        };
        rc = timed_intersect(MIRT_MODE_UNCONDITIONAL, cur, P, nullptr, ctx->d_wq);  // Camera.cpp:27-33: no predicate
        if (rc != MIRT_OK) return rc;
        if (carried) {  // the previous wavefront's survivors, already intersected, behind the camera hits
            hipLaunchKernelGGL(k_carry, dim3((carried + B - 1) / B), dim3(B), 0, st, ctx->carry, (const uint32_t*)ctx->d_carry_cnt,
                               carried, *cur, P, ctx->d_cnt, P);
            HIPCHK(ctx, hipGetLastError());
        }
        uint32_t bound = P + carried, depths_run = 0;
        const bool counted_start = carried != 0;  // d_cnt[0] holds P + the carried count
        carried = 0;
        for (uint32_t it = 0; it <= prm->max_depth; it++) {
            const uint32_t* n_in = (it || counted_start) ? ctx->d_cnt + it : nullptr;
            const uint32_t* off_in = it ? ctx->d_off + it : nullptr;  // (camera hits and carried paths lie at [0, n))
            uint32_t* n_out = ctx->d_cnt + it + 1;
            uint32_t* off_out = ctx->d_off + it + 1;
            uint32_t* ab = ctx->d_ab + 64 * (size_t)(it + 1);
            // two-ended compaction (pt_kernels.h, k_shade<.., TWO>): survivors land in [mid - ab[32], mid + ab[0])
            const uint32_t sgrid = (bound + SHADE_BLOCK - 1) / SHADE_BLOCK, mid = (sgrid >> 1) * (uint32_t)SHADE_BLOCK;
            if (nee) {
                // PathTracerTest (as in run_wavefront): one shadow ray per RANDOM vertex, the visibility launch and
                // its connections before the next bounce touches L; a carried path scores into its own wavefront's buffer
                uint32_t* n_sh = ctx->d_ns + it;
                if (ctx->has_media)
                    hipLaunchKernelGGL((k_shade<true, true, true>), dim3(sgrid), dim3(SHADE_BLOCK), 0, st, ctx->sc, cam, *cur, n_in,
                                       bound, *nxt, ab, ctx->d_L, (uint32_t)cap, prm->max_depth, ctx->shb, n_sh, off_in);
                else
                    hipLaunchKernelGGL((k_shade<false, true, true>), dim3(sgrid), dim3(SHADE_BLOCK), 0, st, ctx->sc, cam, *cur, n_in,
                                       bound, *nxt, ab, ctx->d_L, (uint32_t)cap, prm->max_depth, ctx->shb, n_sh, off_in);
                HIPCHK(ctx, hipGetLastError());
                const ConnectArgs ca{ctx->shb.c, ctx->shb.stride, ctx->shb.pid, ctx->d_L, (uint32_t)cap, PID_MASK, (uint32_t)PID_GEN_SHIFT};
                const int rcv = launch_visibility(ctx, st, MIRT_MODE_NO_SELF, ctx->shb.o, ctx->shb.d, ctx->shb.ids, ctx->shb.lim,
                                                  bound, ctx->shb.stride, ctx->shb.vis, n_sh, ctx->d_wq2 + 3 * it, &ca);
                if (rcv != MIRT_OK) return rcv;
            } else if (ctx->has_media)
                hipLaunchKernelGGL((k_shade<true, false, true>), dim3(sgrid), dim3(SHADE_BLOCK), 0, st, ctx->sc, cam, *cur, n_in, bound, *nxt,
                                   ab, ctx->d_L, (uint32_t)cap, prm->max_depth, ShadowBuf{}, (uint32_t*)nullptr, off_in);
            else
                hipLaunchKernelGGL((k_shade<false, false, true>), dim3(sgrid), dim3(SHADE_BLOCK), 0, st, ctx->sc, cam, *cur, n_in, bound, *nxt,
                                   ab, ctx->d_L, (uint32_t)cap, prm->max_depth, ShadowBuf{}, (uint32_t*)nullptr, off_in);
            HIPCHK(ctx, hipGetLastError());
            hipLaunchKernelGGL(k_live, dim3(1), dim3(1), 0, st, (const uint32_t*)ab, mid, n_out, off_out);
            HIPCHK(ctx, hipGetLastError());
            depths_run = it + 1;
            if (it == prm->max_depth) break;  // (camera paths end here; carried ones ended earlier, by their own depth)
            std::swap(cur, nxt);
            const bool at_cut = !last && it + 1 == cut;
            if (at_cut || (it >= 3 && ((it + 1) & it) == 0)) {  // look at the count: stop early / shrink grids / decide the cut
                HIPCHK(ctx, hipMemcpyAsync(ctx->h_cnt, n_out, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                HIPCHK(ctx, hipStreamSynchronize(st));
                const uint32_t n = ctx->h_cnt[0];
                if (n > bound) return fail(ctx, MIRT_E_HIP, "wavefront: live-path counter corrupted");
                bound = n;
                if (n == 0) break;
            }
            rc = timed_intersect(MIRT_MODE_NO_SELF, cur, bound, n_out, ctx->d_wq + 3 * (it + 1), off_out);  // MCRenderer.cpp:193-197
            if (rc != MIRT_OK) return rc;
            if (at_cut && bound <= carry_limit) {
                hipLaunchKernelGGL(k_carry, dim3((bound + B - 1) / B), dim3(B), 0, st, *cur, (const uint32_t*)n_out, bound, ctx->carry, 0u,
                                   ctx->d_carry_cnt, 0u, (const uint32_t*)off_out);
                HIPCHK(ctx, hipGetLastError());
                carried = bound;
                break;
            }
        }
        // rays traced by this wavefront = P + the live counts of its bounces (read after the stream drains)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_cnt, ctx->d_cnt, (depths_run + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        *n_rays += P;
        for (uint32_t k = 1; k < depths_run + 1 && k <= prm->max_depth; k++) *n_rays += ctx->h_cnt[k];
        if (nee && n_shadow) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_cnt, ctx->d_ns, depths_run * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            for (uint32_t k = 0; k < depths_run; k++) *n_shadow += ctx->h_cnt[k];
        }
        if (*ev_used > 4096) {  // fold the event pool so long renders do not grow it without bound
            if (stats)
                for (size_t k = 2; k + 1 < *ev_used; k += 2) {
                    float ms = 0;
                    hipEventElapsedTime(&ms, ctx->events[k], ctx->events[k + 1]);
                    stats->intersect_ms += ms;
                }
            *ev_used = 2;
        }
    }
    for (uint32_t w = n_wf > (uint32_t)gens ? n_wf - gens : 0; w < n_wf; w++)
        if ((rc = accumulate(w)) != MIRT_OK) return rc;
    return MIRT_OK;
}

static int render_common(MirtContext* ctx, const MirtRenderParams* prm, float* rgb_sum_dev, Capture* cap,
                         MirtRenderStats* stats, bool nee = false, const MirtTile* tile = nullptr) {
    if (!ctx) return MIRT_E_INVALID;
    if (!ctx->has_scene) return fail(ctx, MIRT_E_NO_SCENE, "render: no scene uploaded");
    if (!prm) return fail(ctx, MIRT_E_INVALID, "render: params is NULL");
    if (ctx->has_media && !ctx->media_ready)
        return fail(ctx, MIRT_E_INVALID,
                    "render: the scene holds a homogeneous medium (MIRT_OBJ_VPM) whose own tree was not "
                    "uploaded (MirtSceneDesc.obj_trees)");
    // a path carries its depth in the top byte of its RNG counter word (pt_kernels.h, CTR_DEPTH_SHIFT)
    if (prm->max_depth > 250) return fail(ctx, MIRT_E_LIMIT, "render: max_depth above 250");
    DevCamera cam;
    int rc = make_camera(ctx, &prm->camera, &cam, tile);
    if (rc != MIRT_OK) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t wh = (uint64_t)cam.tw * cam.th;  // pixels of the window being rendered
    uint32_t S = prm->spp_per_launch;
    // plain path tracing over several wavefronts: chained (run_chained); MIRT_CHAIN=0 switches it off (A/B)
    static const bool chain_on = !getenv("MIRT_CHAIN") || atoi(getenv("MIRT_CHAIN")) != 0;
    const bool can_chain = chain_on && !cap && rgb_sum_dev && chain_cut(prm->max_depth) < prm->max_depth &&
                           prm->max_depth / chain_cut(prm->max_depth) + 1 <= 4;
    // Each bounce is one launch and the late bounces of a wavefront are latency-bound (a few thousand
    // paths, ~0.2 ms each), so an unchained wavefront wants to be wide: 16 Mi -> 128 Mi paths is +25 %
    // samples/s on the Cornell scene, at ~18 GB of path state + L. A chained render has no such tail
    // (the survivors of the cut ride along with the next wavefront), so its default is 32 Mi paths
    // (~6 GB with the L ring and the carry buffer): 1305 Msamples/s against 1385 at 128 Mi.
    // Round 3, the final kernels, 1024^2 x 1024 spp chained with 32 / 64 / 128 Mi paths in flight: 1 880 / 1 941 / 1 978
    // Msamples/s (fog scene 1 753 / 1 808 / 1 842): fewer, larger launches and a quarter as many cuts. 128 Mi paths are
    // ~30 GB of path state, carry buffer and L ring, a tenth of this GPU's memory, so a render long enough to fill at
    // least four such wavefronts takes them, a shorter one 64 Mi, and 32 Mi below that.
    if (S == 0) {
        const uint64_t wide = std::max<uint64_t>(1, (1ull << 27) / wh), mid = std::max<uint64_t>(1, (1ull << 26) / wh),
                       narrow = std::max<uint64_t>(1, (1ull << 25) / wh);
        if (!can_chain || prm->spp_count <= narrow) S = (uint32_t)wide;  // (a single wavefront, or none can be chained)
        else if (prm->spp_count >= 4 * wide) S = (uint32_t)wide;
        else if (prm->spp_count >= 4 * mid) S = (uint32_t)mid;
        else S = (uint32_t)narrow;
    }
    S = std::min<uint32_t>(S, std::max<uint32_t>(prm->spp_count, 1));
    if (wh * S > (1ull << 30)) return fail(ctx, MIRT_E_LIMIT, "render: more than 2^30 paths per wavefront");
    // PathTracerTest (shadow-ray buffers beside the path state): at most 64 Mi paths per wavefront; a render that fits
    // one such wavefront stays one (1024^2 x 64 spp: 1 097 Msamples/s against 1 069 as two chained wavefronts of 32 Mi),
    // a longer one is chained at that width (512 spp: 1 145 against ~1 100 unchained)
    if (nee) {
        const uint32_t nee_cap = (uint32_t)std::max<uint64_t>(1, (1ull << 26) / wh);
        S = prm->spp_per_launch ? std::min(S, nee_cap) : std::min<uint32_t>(nee_cap, std::max<uint32_t>(prm->spp_count, 1));
    }
    const bool chained = can_chain && prm->spp_count > S;
    if (chained) {
        // (a width the library chose itself is halved, down to 32 Mi paths, when the device cannot hold it: another
        // context's buffers, a smaller GPU)
        const uint32_t narrow = (uint32_t)std::max<uint64_t>(1, (1ull << 25) / wh);
        for (;;) {
            const size_t carry_cap = (size_t)(wh * S) / 4;
            rc = ensure_paths(ctx, (size_t)(wh * S) + carry_cap, (int)(prm->max_depth / chain_cut(prm->max_depth)) + 1, carry_cap);
            if (rc == MIRT_OK || prm->spp_per_launch != 0 || S <= narrow) break;
            (void)hipGetLastError();
            S = std::max(narrow, S / 2);
        }
    } else {
        rc = ensure_paths(ctx, (size_t)(wh * S));
    }
    if (rc != MIRT_OK) return rc;
    if (nee) {  // (a chained wavefront also shades the paths carried over from the one before)
        rc = ensure_shadow(ctx, (size_t)(wh * S) + (chained ? (size_t)(wh * S) / 4 : 0));
        if (rc != MIRT_OK) return rc;
    }
    uint64_t n_shadow = 0;
    hipEvent_t t0 = get_event(ctx, 0), t1 = get_event(ctx, 1);
    if (!t0 || !t1) return fail(ctx, MIRT_E_HIP, "hipEventCreate failed");
    size_t ev_used = 2;
    uint64_t n_rays = 0;
    HIPCHK(ctx, hipEventRecord(t0, ctx->stream));
    if (chained) {
        rc = run_chained(ctx, cam, prm, S, rgb_sum_dev, &n_rays, &ev_used, stats, nee, &n_shadow);
        if (rc != MIRT_OK) return rc;
    } else
    for (uint32_t done = 0; done < prm->spp_count; done += S) {
        uint32_t s = std::min(S, prm->spp_count - done);
        rc = run_wavefront(ctx, cam, prm, prm->spp_begin + done, s, rgb_sum_dev, cap, &n_rays, &ev_used, nee, &n_shadow);
        if (rc != MIRT_OK) return rc;
        if (ev_used > 4096) {  // fold the event pool so long renders do not grow it without bound
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (stats) {
                for (size_t k = 2; k + 1 < ev_used; k += 2) {
                    float ms = 0;
                    hipEventElapsedTime(&ms, ctx->events[k], ctx->events[k + 1]);
                    stats->intersect_ms += ms;
                }
            }
            ev_used = 2;
        }
    }
    HIPCHK(ctx, hipEventRecord(t1, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (stats) {
        float ms = 0;
        hipEventElapsedTime(&ms, t0, t1);
        stats->gpu_ms = ms;
        for (size_t k = 2; k + 1 < ev_used; k += 2) {
            float m2 = 0;
            hipEventElapsedTime(&m2, ctx->events[k], ctx->events[k + 1]);
            stats->intersect_ms += m2;
        }
        stats->n_rays = n_rays;
        stats->n_shadow_rays = n_shadow;
        stats->n_samples = wh * prm->spp_count;
    }
    return MIRT_OK;
}

extern "C" {

int mirt_generate_rays_dev(MirtContext* ctx, const MirtCamera* camera, uint32_t spp_begin,
                           uint32_t spp_count, uint32_t seed, float* origins, float* directions) {
    if (!ctx) return MIRT_E_INVALID;
    if (!origins || !directions) return fail(ctx, MIRT_E_INVALID, "generate_rays: NULL buffer");
    DevCamera cam;
    int rc = make_camera(ctx, camera, &cam);
    if (rc != MIRT_OK) return rc;
    uint64_t n = (uint64_t)cam.width * cam.height * spp_count;
    if (n == 0) return MIRT_OK;
    if (n > 0x7FFFFFFFull) return fail(ctx, MIRT_E_LIMIT, "generate_rays: more than 2^31-1 rays");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PathBuf none{};
    hipLaunchKernelGGL(k_generate, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, cam,
                       (uint32_t)n, spp_begin, seed, origins, directions, (uint32_t)n, none, 0);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MIRT_OK;
}

int mirt_render_pt_dev(MirtContext* ctx, const MirtRenderParams* params, float* rgb_sum,
                       MirtRenderStats* stats) {
    if (stats) *stats = MirtRenderStats{};
    if (ctx && !rgb_sum) return fail(ctx, MIRT_E_INVALID, "render: rgb_sum is NULL");
    return render_common(ctx, params, rgb_sum, nullptr, stats);
}

int mirt_render_pt(MirtContext* ctx, const MirtRenderParams* params, float* rgb_sum,
                   MirtRenderStats* stats) {
    if (stats) *stats = MirtRenderStats{};
    if (!ctx) return MIRT_E_INVALID;
    if (!rgb_sum || !params) return fail(ctx, MIRT_E_INVALID, "render: NULL argument");
    if (!ctx->has_scene) return fail(ctx, MIRT_E_NO_SCENE, "render: no scene uploaded");
    if (params->camera.width <= 0 || params->camera.height <= 0)
        return fail(ctx, MIRT_E_INVALID, "camera: width/height must be positive");
    size_t bytes = (size_t)params->camera.width * params->camera.height * 3 * sizeof(float);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    float* d_rgb = nullptr;
    HIPCHK(ctx, hipMalloc(&d_rgb, bytes));
    hipError_t e = hipMemsetAsync(d_rgb, 0, bytes, ctx->stream);
    int rc = e == hipSuccess ? render_common(ctx, params, d_rgb, nullptr, stats)
                             : fail(ctx, MIRT_E_HIP, hipGetErrorString(e));
    if (rc == MIRT_OK) {
        e = hipMemcpy(rgb_sum, d_rgb, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, MIRT_E_HIP, hipGetErrorString(e));
    }
    hipFree(d_rgb);
    return rc;
}

int mirt_render_pttest_dev(MirtContext* ctx, const MirtRenderParams* params, float* rgb_sum,
                           MirtRenderStats* stats) {
    if (stats) *stats = MirtRenderStats{};
    if (ctx && !rgb_sum) return fail(ctx, MIRT_E_INVALID, "render_pttest: rgb_sum is NULL");
    return render_common(ctx, params, rgb_sum, nullptr, stats, true);
}

int mirt_render_pttest(MirtContext* ctx, const MirtRenderParams* params, float* rgb_sum,
                       MirtRenderStats* stats) {
    if (stats) *stats = MirtRenderStats{};
    if (!ctx) return MIRT_E_INVALID;
    if (!rgb_sum || !params) return fail(ctx, MIRT_E_INVALID, "render_pttest: NULL argument");
    if (!ctx->has_scene) return fail(ctx, MIRT_E_NO_SCENE, "render_pttest: no scene uploaded");
    if (params->camera.width <= 0 || params->camera.height <= 0)
        return fail(ctx, MIRT_E_INVALID, "camera: width/height must be positive");
    size_t bytes = (size_t)params->camera.width * params->camera.height * 3 * sizeof(float);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    float* d_rgb = nullptr;
    HIPCHK(ctx, hipMalloc(&d_rgb, bytes));
    hipError_t e = hipMemsetAsync(d_rgb, 0, bytes, ctx->stream);
    int rc = e == hipSuccess ? render_common(ctx, params, d_rgb, nullptr, stats, true)
                             : fail(ctx, MIRT_E_HIP, hipGetErrorString(e));
    if (rc == MIRT_OK) {
        e = hipMemcpy(rgb_sum, d_rgb, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, MIRT_E_HIP, hipGetErrorString(e));
    }
    hipFree(d_rgb);
    return rc;
}

int mirt_render_pt_tile_dev(MirtContext* ctx, const MirtRenderParams* params, const MirtTile* tile, float* rgb_sum,
                            MirtRenderStats* stats) {
    if (stats) *stats = MirtRenderStats{};
    if (ctx && !rgb_sum) return fail(ctx, MIRT_E_INVALID, "render: rgb_sum is NULL");
    return render_common(ctx, params, rgb_sum, nullptr, stats, false, tile);
}

int mirt_render_pttest_tile_dev(MirtContext* ctx, const MirtRenderParams* params, const MirtTile* tile, float* rgb_sum,
                                MirtRenderStats* stats) {
    if (stats) *stats = MirtRenderStats{};
    if (ctx && !rgb_sum) return fail(ctx, MIRT_E_INVALID, "render_pttest: rgb_sum is NULL");
    return render_common(ctx, params, rgb_sum, nullptr, stats, true, tile);
}

// [begin, begin + count) of part `idx` of `len` cut into `parts`: in steps of 16 where len allows it
static void cut_axis(int len, int parts, int idx, int* begin, int* count) {
    const int step = (len % 16 == 0 && len / 16 >= parts) ? 16 : 1;
    const int units = len / step, base = units / parts, rem = units % parts;
    *begin = (idx * base + std::min(idx, rem)) * step;
    *count = (base + (idx < rem ? 1 : 0)) * step;
}

int mirt_shard_tile(int32_t width, int32_t height, int32_t n_shards, int32_t k, MirtTile* out) {
    if (!out || width <= 0 || height <= 0 || n_shards <= 0 || k < 0 || k >= n_shards) return MIRT_E_INVALID;
    int gy = 1;
    for (int d = 1; d * d <= n_shards; d++)
        if (n_shards % d == 0) gy = d;
    const int gx = n_shards / gy;
    int x0, w, y0, h;
    cut_axis(width, gx, k % gx, &x0, &w);
    cut_axis(height, gy, k / gx, &y0, &h);
    *out = MirtTile{x0, y0, w, h};
    return MIRT_OK;
}

int mirt_context_device(const MirtContext* ctx) { return ctx ? ctx->device : -1; }

int mirt_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

// Sample shards over several contexts + one reduce (include/mirt.h). One host thread per context: a render call
// blocks on its own stream now and then (live-path counts), and the contexts' devices run independently.
static int render_multi(MirtContext* const* ctxs, int n_ctx, const MirtRenderParams* params, float* rgb_sum,
                        MirtRenderStats* stats, bool nee, int shard = MIRT_SHARD_SAMPLES) {
    if (stats) *stats = MirtRenderStats{};
    if (!ctxs || n_ctx <= 0 || !ctxs[0]) return MIRT_E_INVALID;
    MirtContext* c0 = ctxs[0];
    if (!params || !rgb_sum) return fail(c0, MIRT_E_INVALID, "render_multi: NULL argument");
    if (shard != MIRT_SHARD_SAMPLES && shard != MIRT_SHARD_TILES) return fail(c0, MIRT_E_INVALID, "render_multi: unknown shard mode");
    const bool tiles = shard == MIRT_SHARD_TILES;
    for (int k = 0; k < n_ctx; k++) {
        if (!ctxs[k]) return fail(c0, MIRT_E_INVALID, "render_multi: NULL context");
        for (int j = 0; j < k; j++)
            if (ctxs[j] == ctxs[k]) return fail(c0, MIRT_E_INVALID, "render_multi: the same context twice");
        if (!ctxs[k]->has_scene) return fail(c0, MIRT_E_NO_SCENE, "render_multi: a context holds no scene");
    }
    if (params->camera.width <= 0 || params->camera.height <= 0)
        return fail(c0, MIRT_E_INVALID, "camera: width/height must be positive");
    const size_t count = (size_t)params->camera.width * params->camera.height * 3, bytes = count * sizeof(float);
    if (count > 0x7FFFFFFFull) return fail(c0, MIRT_E_LIMIT, "render_multi: image too large");
    struct Shard {
        MirtRenderParams prm;
        MirtRenderStats st{};
        MirtTile tile{};
        float* acc = nullptr;
        size_t bytes = 0;
        int rc = MIRT_OK;
    };
    std::vector<Shard> sh(n_ctx);
    const uint32_t base = params->spp_count / (uint32_t)n_ctx, rem = params->spp_count % (uint32_t)n_ctx;
    uint32_t begin = params->spp_begin;
    for (int k = 0; k < n_ctx; k++) {
        sh[k].prm = *params;
        sh[k].bytes = bytes;
        if (tiles) {  // every context: all samples of its own tile
            mirt_shard_tile(params->camera.width, params->camera.height, n_ctx, k, &sh[k].tile);
            sh[k].bytes = (size_t)sh[k].tile.width * sh[k].tile.height * 3 * sizeof(float);
            continue;
        }
        // contiguous sample shards; the same split as renderer_amd.shard_samples
        sh[k].prm.spp_begin = begin;
        sh[k].prm.spp_count = base + ((uint32_t)k < rem ? 1u : 0u);
        begin += sh[k].prm.spp_count;
    }
    auto work = [&](int k) {
        MirtContext* ctx = ctxs[k];
        Shard& S = sh[k];
        if (S.bytes == 0) return;  // more contexts than pixels: an empty tile
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipMalloc(&S.acc, S.bytes);
        if (e == hipSuccess) e = hipMemsetAsync(S.acc, 0, S.bytes, ctx->stream);
        if (e != hipSuccess) {
            S.rc = fail(ctx, MIRT_E_HIP, std::string("render_multi: ") + hipGetErrorString(e));
            return;
        }
        if (S.prm.spp_count == 0) {  // more contexts than samples: an empty shard
            e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) S.rc = fail(ctx, MIRT_E_HIP, hipGetErrorString(e));
            return;
        }
        S.rc = render_common(ctx, &S.prm, S.acc, nullptr, &S.st, nee, tiles ? &S.tile : nullptr);
    };
    std::vector<std::thread> threads;
    for (int k = 1; k < n_ctx; k++) threads.emplace_back(work, k);
    work(0);
    for (std::thread& t : threads) t.join();
    int rc = MIRT_OK;
    for (int k = 0; k < n_ctx && rc == MIRT_OK; k++)
        if (sh[k].rc != MIRT_OK) {
            rc = sh[k].rc;
            if (k) c0->err = "render_multi: context " + std::to_string(k) + ": " + ctxs[k]->err;
        }
    float* tmp = nullptr;
    hipError_t e = hipSuccess;
    if (tiles && rc == MIRT_OK) {
        // gather: every tile's rows into their place in the host image (no arithmetic: the image is bit-equal to the
        // one-context render)
        std::vector<float> stage;
        const size_t W = (size_t)params->camera.width;
        for (int k = 0; k < n_ctx && e == hipSuccess; k++) {
            const Shard& S = sh[k];
            if (!S.acc) continue;
            stage.resize(S.bytes / sizeof(float));
            e = hipSetDevice(ctxs[k]->device);
            if (e == hipSuccess) e = hipMemcpy(stage.data(), S.acc, S.bytes, hipMemcpyDeviceToHost);
            if (e != hipSuccess) break;
            const size_t row = (size_t)S.tile.width * 3;
            for (int y = 0; y < S.tile.height; y++)
                memcpy(rgb_sum + ((size_t)(S.tile.y0 + y) * W + (size_t)S.tile.x0) * 3, stage.data() + (size_t)y * row, row * sizeof(float));
        }
        if (e != hipSuccess) rc = fail(c0, MIRT_E_HIP, std::string("render_multi: gather: ") + hipGetErrorString(e));
    }
    // reduce on context 0's device, in context order (deterministic): peer copy into a staging buffer, then add
    if (!tiles) e = hipSetDevice(c0->device);
    if (!tiles && rc == MIRT_OK && n_ctx > 1 && e == hipSuccess) e = hipMalloc(&tmp, bytes);
    for (int k = 1; !tiles && k < n_ctx && rc == MIRT_OK && e == hipSuccess; k++) {
        e = hipMemcpyPeer(tmp, c0->device, sh[k].acc, ctxs[k]->device, bytes);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_add, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, c0->stream, sh[0].acc, (const float*)tmp,
                           (uint32_t)count);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c0->stream);  // tmp is reused by the next context
    }
    if (!tiles && rc == MIRT_OK && e == hipSuccess) e = hipMemcpy(rgb_sum, sh[0].acc, bytes, hipMemcpyDeviceToHost);
    if (!tiles && rc == MIRT_OK && e != hipSuccess) rc = fail(c0, MIRT_E_HIP, std::string("render_multi: reduce: ") + hipGetErrorString(e));
    hipFree(tmp);
    for (int k = 0; k < n_ctx; k++) {
        if (!sh[k].acc) continue;
        hipSetDevice(ctxs[k]->device);
        hipFree(sh[k].acc);
    }
    if (stats && rc == MIRT_OK)
        for (int k = 0; k < n_ctx; k++) {
            stats->n_samples += sh[k].st.n_samples;
            stats->n_rays += sh[k].st.n_rays;
            stats->n_shadow_rays += sh[k].st.n_shadow_rays;
            stats->intersect_ms += sh[k].st.intersect_ms;
            stats->gpu_ms = std::max(stats->gpu_ms, sh[k].st.gpu_ms);
        }
    return rc;
}

int mirt_render_pt_multi(MirtContext* const* ctxs, int n_ctx, const MirtRenderParams* params, float* rgb_sum,
                         MirtRenderStats* stats) {
    return render_multi(ctxs, n_ctx, params, rgb_sum, stats, false);
}

int mirt_render_pttest_multi(MirtContext* const* ctxs, int n_ctx, const MirtRenderParams* params, float* rgb_sum,
                             MirtRenderStats* stats) {
    return render_multi(ctxs, n_ctx, params, rgb_sum, stats, true);
}

int mirt_render_pt_multi_shard(MirtContext* const* ctxs, int n_ctx, const MirtRenderParams* params, int shard,
                               float* rgb_sum, MirtRenderStats* stats) {
    return render_multi(ctxs, n_ctx, params, rgb_sum, stats, false, shard);
}

int mirt_render_pttest_multi_shard(MirtContext* const* ctxs, int n_ctx, const MirtRenderParams* params, int shard,
                                   float* rgb_sum, MirtRenderStats* stats) {
    return render_multi(ctxs, n_ctx, params, rgb_sum, stats, true, shard);
}

int mirt_capture_rays_dev(MirtContext* ctx, const MirtRenderParams* params, uint32_t depth_lo,
                          uint32_t depth_hi, float* origins, float* directions, int32_t* aux,
                          size_t capacity, size_t* n_captured) {
    if (!ctx) return MIRT_E_INVALID;
    if (!origins || !directions || !aux || !n_captured || capacity == 0)
        return fail(ctx, MIRT_E_INVALID, "capture_rays: NULL buffer or zero capacity");
    if (depth_lo < 1 || depth_hi < depth_lo)
        return fail(ctx, MIRT_E_INVALID, "capture_rays: need 1 <= depth_lo <= depth_hi");
    Capture cap{depth_lo, depth_hi, origins, directions, aux, capacity, 0};
    int rc = render_common(ctx, params, nullptr, &cap, nullptr);
    *n_captured = cap.count;
    return rc;
}

}  // extern "C"
