// main.cpp — command-line shell in the shape of the reference's main.cpp:10-17 / Renderer.cpp:52-62:
// load a config, render, write the PFM. Usage: mirt_render [Data/Config.xml] [--spp N] [--seed S] [--out file.pfm]
//        [--output-iter K] (periodic dumps, PathTracer.cpp:47-55) [--resume acc.bin] [--save-state acc.bin]
//        [--gpus N] (devices 0..N-1: sample shards + one reduce, mirt_render_pt_multi; default 1)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

#include "host.h"

using namespace mirt_host;

int main(int argc, char** argv) {
    std::string config = "Data/Config.xml", out, resume, save_state;
    unsigned spp = 0, seed = 0;
    int output_iter = 0, gpus = 1;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--spp") && i + 1 < argc) spp = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "--output-iter") && i + 1 < argc) output_iter = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--resume") && i + 1 < argc) resume = argv[++i];
        else if (!strcmp(argv[i], "--save-state") && i + 1 < argc) save_state = argv[++i];
        else config = argv[i];
    }
    try {
        if (gpus > 1) {
            if (gpus > mirt_device_count()) throw std::runtime_error("--gpus: more than the HIP devices present");
            std::vector<int> devices;
            for (int k = 0; k < gpus; k++) devices.push_back(k);
            IntersectionHIP::setDevices(devices);  // before the scene is loaded: the upload then goes to all of them
        }
        Renderer renderer;
        renderer.loadConfig(config);
        PathTracerHIP* pt = dynamic_cast<PathTracerHIP*>(renderer.mcRenderer);
        if (pt && spp) pt->spp = spp;
        if (pt && seed) pt->seed = seed;
        if (!out.empty()) renderer.mcRenderer->setSavePath(out);
        if (pt && output_iter > 0) {
            pt->progressive = true;
            pt->setOutputIter(output_iter);
        }
        if (pt && !resume.empty()) pt->loadState(resume, renderer.camera);
        std::vector<vec3f> px = renderer.render();
        if (pt && !save_state.empty()) pt->saveState(save_state, renderer.camera);
        if (pt) {
            const MirtRenderStats& st = pt->lastStats;
            printf("samples %llu rays %llu gpu %.3f ms (intersect %.3f ms) -> %.2f Msamples/s %.2f Mrays/s\n",
                   (unsigned long long)st.n_samples, (unsigned long long)st.n_rays, st.gpu_ms, st.intersect_ms,
                   st.n_samples / st.gpu_ms * 1e-3, st.n_rays / st.gpu_ms * 1e-3);
        }
        IntersectionHIP::shutdown();
    } catch (const std::exception& e) {
        fprintf(stderr, "mirt_render: %s\n", e.what());
        return 1;
    }
    return 0;
}
