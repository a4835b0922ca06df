// In-library kernel timing with HIP events (for bench.py's `roofline` object): when enabled, every tracked launcher
// brackets its launch with two events on the launch stream; m3l_prof_end() resolves them into per-class totals.
// A class = kernel kind + problem shape, e.g. "gemm_nt[49152x768x192]".  Disabled (the default) it costs one branch.
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/m3l_amd.h"
#include "common.cuh"
#include "kernels.h"

namespace {
struct Rec { int cls; hipEvent_t a, b; };
struct Cls { std::string name; double ms = 0, work = 0, bytes = 0; long launches = 0; double work_per_launch = 0, bytes_per_launch = 0; };
bool g_on = false;
int g_stride = 1;
long g_seen = 0;
std::string g_filter;
std::vector<Rec> g_recs;
std::vector<Cls> g_cls;
std::map<std::string, int> g_index;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
}  // namespace

ProfScope::ProfScope(const char* kind, long a, long b, long c, double work, hipStream_t st, double bytes) : idx(-1), stream(st) {
    if (!g_on) return;
    char name[96];
    snprintf(name, sizeof(name), "%s[%ldx%ldx%ld]", kind, a, b, c);
    if (!g_filter.empty() && strstr(name, g_filter.c_str()) == nullptr) return;
    if ((g_seen++ % g_stride) != 0) return;          // sample every g_stride-th matching launch
    auto it = g_index.find(name);
    int cls;
    if (it == g_index.end()) {
        cls = (int)g_cls.size();
        g_index[name] = cls;
        Cls cl;
        cl.name = name;
        cl.work_per_launch = work;
        cl.bytes_per_launch = bytes;
        g_cls.push_back(cl);
    } else {
        cls = it->second;
    }
    Rec r;
    r.cls = cls;
    r.a = get_event();
    r.b = get_event();
    hipEventRecord(r.a, st);
    idx = (int)g_recs.size();
    g_recs.push_back(r);
}
ProfScope::~ProfScope() {
    if (idx >= 0) hipEventRecord(g_recs[idx].b, stream);
}

extern "C" {
void m3l_prof_begin(const char* filter, int stride) {
    g_stride = stride > 0 ? stride : 1;
    g_seen = 0;
    g_recs.clear();
    g_cls.clear();
    g_index.clear();
    g_filter = filter ? filter : "";
    g_on = true;
}
void m3l_prof_end(void) {
    g_on = false;
    hipDeviceSynchronize();
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_cls[r.cls].ms += ms;
            g_cls[r.cls].launches += 1;
            g_cls[r.cls].work += g_cls[r.cls].work_per_launch;
            g_cls[r.cls].bytes += g_cls[r.cls].bytes_per_launch;
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
}
// cost of one empty event bracket on `stream` (record, record, nothing between), median of n: the fixed part of every
// bracketed launch that is not kernel time (subtract it before comparing with rocprofv3's in-kernel durations)
double m3l_prof_event_overhead_us(void* stream, int n) {
    hipStream_t st = (hipStream_t)stream;
    if (n < 1) n = 1;
    std::vector<hipEvent_t> ev(2 * n);
    for (auto& e : ev) e = get_event();
    for (int i = 0; i < n; ++i) {
        hipEventRecord(ev[2 * i], st);
        hipEventRecord(ev[2 * i + 1], st);
    }
    hipStreamSynchronize(st);
    std::vector<float> ms(n, 0.f);
    for (int i = 0; i < n; ++i) hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1]);
    for (auto& e : ev) g_pool.push_back(e);
    std::sort(ms.begin(), ms.end());
    return 1e3 * ms[n / 2];
}
int m3l_prof_count(void) { return (int)g_cls.size(); }
int m3l_prof_get(int i, char* name, size_t n, double* ms_total, long* launches, double* work_total, double* bytes_total) {
    if (i < 0 || i >= (int)g_cls.size()) return 1;
    if (name && n) {
        strncpy(name, g_cls[i].name.c_str(), n - 1);
        name[n - 1] = 0;
    }
    if (ms_total) *ms_total = g_cls[i].ms;
    if (launches) *launches = g_cls[i].launches;
    if (work_total) *work_total = g_cls[i].work;
    if (bytes_total) *bytes_total = g_cls[i].bytes;
    return 0;
}
}
