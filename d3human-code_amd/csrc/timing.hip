// timing.hip -- per-launch HIP-event timing of selected kernels, ON THE STREAM THEY ARE LAUNCHED ON (bench.py's roofline entries).
//
// The step runs on three streams (main, the eikonal chain, the texture-table scatter); an event pair recorded by the host on "the
// current stream" around a Python-level op sees neither the other streams nor the individual kernels of a multi-kernel entry point.
// The instrumented entry points therefore bracket their kernels themselves: d3h_ktime_begin / d3h_ktime_end record a hipEvent pair on
// the launch stream when timing is enabled (two host calls of ~1 us; nothing when disabled, which is the default).
#include "d3h_common.h"
// the ABI version this library was built with (D3H_ABI_VERSION of include/d3h.h); callers compare it with the header they compiled against
extern "C" int d3h_abi_version(void) { return D3H_ABI_VERSION; }
#ifdef D3H_EMULATED      // host emulation of the kernels (tests): no events, the entry points exist and report nothing
extern "C" int d3h_timing_enable(int) { return D3H_OK; }
extern "C" int d3h_timing_select(unsigned long long) { return D3H_OK; }
extern "C" int d3h_timing_reserve(int64_t) { return D3H_OK; }
extern "C" int64_t d3h_timing_read(int*, int64_t*, float*, int64_t) { return 0; }
#else
#include <vector>

namespace {
struct Rec { int id; long long units; hipEvent_t a, b; };
bool g_on = false;
unsigned long long g_mask = ~0ull;       // bit id: kernel id `id` is timed while timing is on (d3h_timing_select)
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t take() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

int d3h_ktime_begin(int id, long long units, hipStream_t s) {
    if (!g_on || !((g_mask >> (id & 63)) & 1ull)) return -1;
    Rec r{id, units, take(), take()};
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}

void d3h_ktime_end(int handle, hipStream_t s) {
    if (handle < 0 || handle >= (int)g_recs.size()) return;
    (void)hipEventRecord(g_recs[handle].b, s);
}

// 1: start collecting (drops earlier records), 0: stop.  Single host thread, as every other entry point.
extern "C" int d3h_timing_enable(int on) {
    if (on) {
        for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
        g_recs.clear();
    }
    g_on = on != 0;
    return D3H_OK;
}

// Which kernel ids (D3H_KT_* in d3h_common.h, bit per id) are timed while timing is enabled; default all.  Every event pair is two host calls
// on the launch path: a benchmark times the kernels its headline needs inside the timed region and the rest in a pass of its own.
extern "C" int d3h_timing_select(unsigned long long mask) {
    g_mask = mask;
    return D3H_OK;
}

// Pre-create events for `n_records` kernel records, so that the timed region of a benchmark creates none (hipEventCreate inside the
// region is host time on the launch path).
extern "C" int d3h_timing_reserve(int64_t n_records) {
    while ((int64_t)g_pool.size() < 2 * n_records) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return D3H_ERR_ARG;
        g_pool.push_back(e);
    }
    g_recs.reserve((size_t)n_records);
    return D3H_OK;
}

// Number of records collected since d3h_timing_enable(1); with ids / units / ms (HOST arrays of `cap` entries, may be NULL) also copies
// the first min(cap, count) records: kernel id (D3H_KT_* in d3h_common.h), the work units passed at the launch, milliseconds between
// the two events (waits for each end event).
extern "C" int64_t d3h_timing_read(int* ids, int64_t* units, float* ms, int64_t cap) {
    const int64_t n = (int64_t)g_recs.size();
    if (!ids || !units || !ms) return n;
    for (int64_t i = 0; i < n && i < cap; ++i) {
        (void)hipEventSynchronize(g_recs[i].b);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_recs[i].a, g_recs[i].b);
        ids[i] = g_recs[i].id; units[i] = g_recs[i].units; ms[i] = t;
    }
    return n;
}
#endif
