// Times d3h_sdf_mlp_fwd of several builds of csrc/sdf_mlp.hip (tools/probe/sdf_variants.sh compiles them with D3H_PROBE_* switches that
// remove one ingredient at a time: epilogue, weight staging, the per-chunk barrier) -- where does the time of a round go?
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef long long (*i64fn)(void);
typedef long long (*actfn)(long long);
// d3h_sdf_mlp_fwd as of ABI version 4 (include/d3h.h): `int max_cus` before the stream
typedef int (*fwdfn)(const float*, const float*, float, const float*, float*, float*, float*, long long, int, void*);
typedef int (*verfn)(void);
int main(int argc, char** argv) {
    const long long n = argc > 2 ? atoll(argv[2]) : 262144;
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    i64fn wf = (i64fn)dlsym(h, "d3h_sdf_mlp_wpack_floats");
    actfn af = (actfn)dlsym(h, "d3h_sdf_mlp_act_floats");
    fwdfn fwd = (fwdfn)dlsym(h, "d3h_sdf_mlp_fwd");
    verfn ver = (verfn)dlsym(h, "d3h_abi_version");
    if (!ver || ver() < 4) { printf("%s: ABI version %d, this probe needs >= 4 (max_cus argument)\n", argv[1], ver ? ver() : 0); return 1; }
    const long long nw = wf(), na = af(n);
    std::vector<float> hw(nw), hx(3 * n);
    srand(1);
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.5f;
    float *w, *x, *sdf, *act;
    hipMalloc(&w, nw * 4); hipMalloc(&x, 3 * n * 4); hipMalloc(&sdf, n * 4); hipMalloc(&act, na * 4);
    hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), 3 * n * 4, hipMemcpyHostToDevice);
    for (int with_act = 0; with_act < 2; ++with_act) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int r = 0; r < 3; ++r) fwd(x, nullptr, 0.f, w, sdf, nullptr, with_act ? act : nullptr, n, 0, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) fwd(x, nullptr, 0.f, w, sdf, nullptr, with_act ? act : nullptr, n, 0, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s n=%lld act=%d  %.3f ms  %.1f TFLOP/s\n", argv[1], n, with_act, ms / 10, 826880.0 * n / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
