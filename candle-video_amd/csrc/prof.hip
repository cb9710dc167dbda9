// Optional per-kernel-class timing with HIP events on the launch stream (used by bench.py for the
// `roofline` object).  Disabled by default: zero overhead on the product path.
#include <cstdlib>
#include <mutex>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "options.h"
#include "../../include/ltxhip.h"

namespace {
struct Rec { int kind; hipEvent_t a, b; double work; int kernel; hipEvent_t ka, kb; int nk; };
std::mutex g_mu;
std::vector<Rec> g_recs;
bool g_on = false;
double g_ms[LTX_PROF_NKINDS], g_work[LTX_PROF_NKINDS];
long long g_cnt[LTX_PROF_NKINDS];
double g_kms[LTX_PROF_NKINDS][LTX_PROFK_N], g_kwork[LTX_PROF_NKINDS][LTX_PROFK_N];
long long g_kcnt[LTX_PROF_NKINDS][LTX_PROFK_N];
thread_local int t_kernel = 0;
thread_local Rec* t_cur = nullptr;
bool g_kernel_events = true;            // LTX_PROF_KERNEL_EVENTS=0: stream-level brackets only (the rounds 1-3 measurement)
}  // namespace

bool ltx_prof_begin(int kind, double work, hipStream_t s, void** token) {
    *token = nullptr;
    if (!g_on) return false;
    Rec* r = new Rec{kind, nullptr, nullptr, work, 0, nullptr, nullptr, 0};
    t_kernel = 0;
    if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) { delete r; return false; }
    (void)hipEventRecord(r->a, s);
    *token = r;
    t_cur = r;
    return true;
}
bool ltx_prof_kernel_events(hipEvent_t* a, hipEvent_t* b) {
    Rec* r = t_cur;
    if (!r || !g_kernel_events) return false;
    if (r->nk++ != 0) return false;                      // a second kernel inside one timed launch: the stream bracket stands
    if (hipEventCreate(&r->ka) != hipSuccess || hipEventCreate(&r->kb) != hipSuccess) { r->nk = 2; return false; }
    *a = r->ka; *b = r->kb;
    return true;
}
void ltx_prof_kernel(int which) { if (which >= 0 && which < LTX_PROFK_N) t_kernel = which; }
void ltx_prof_end(void* token, hipStream_t s) {
    if (!token) return;
    Rec* r = reinterpret_cast<Rec*>(token);
    (void)hipEventRecord(r->b, s);
    r->kernel = t_kernel;
    t_cur = nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back(*r);
    delete r;
}

extern "C" int ltx_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    g_kernel_events = ltx_opt().prof_kernel_events != 0;
    for (auto& r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); if (r.ka) (void)hipEventDestroy(r.ka); if (r.kb) (void)hipEventDestroy(r.kb); }
    g_recs.clear();
    for (int i = 0; i < LTX_PROF_NKINDS; ++i) {
        g_ms[i] = 0; g_work[i] = 0; g_cnt[i] = 0;
        for (int k = 0; k < LTX_PROFK_N; ++k) { g_kms[i][k] = 0; g_kwork[i][k] = 0; g_kcnt[i][k] = 0; }
    }
    return LTX_OK;
}
namespace {
int fold() {     // synchronises the device and folds all recorded launches into the per-kind / per-kernel totals
    HIP_TRY(hipDeviceSynchronize());
    for (auto& r : g_recs) {
        float ms = 0.f;
        hipError_t e = hipErrorUnknown;
        if (r.nk == 1 && r.ka && r.kb) e = hipEventElapsedTime(&ms, r.ka, r.kb);          // the kernel's own start -> end
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipEventElapsedTime(&ms, r.a, r.b); }
        if (r.ka) (void)hipEventDestroy(r.ka);
        if (r.kb) (void)hipEventDestroy(r.kb);
        if (e == hipSuccess) {
            g_ms[r.kind] += ms; g_work[r.kind] += r.work; g_cnt[r.kind] += 1;
            g_kms[r.kind][r.kernel] += ms; g_kwork[r.kind][r.kernel] += r.work; g_kcnt[r.kind][r.kernel] += 1;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    return LTX_OK;
}
}  // namespace
// One (class, kernel) cell: the launches of class `kind` that `kernel` (LTX_PROFK_*) served.
extern "C" int ltx_prof_report_kernel(int kind, int kernel, double* total_ms, double* total_work, long long* count) {
    if (kind < 0 || kind >= LTX_PROF_NKINDS || kernel < 0 || kernel >= LTX_PROFK_N) LTX_FAIL(LTX_ERR_ARG, "ltx_prof_report_kernel: bad kind / kernel");
    std::lock_guard<std::mutex> lk(g_mu);
    LTX_TRY(fold());
    if (total_ms) *total_ms = g_kms[kind][kernel];
    if (total_work) *total_work = g_kwork[kind][kernel];
    if (count) *count = g_kcnt[kind][kernel];
    return LTX_OK;
}
// Folds all recorded launches into per-kind totals and returns one kind.
extern "C" int ltx_prof_report(int kind, double* total_ms, double* total_work, long long* count) {
    if (kind < 0 || kind >= LTX_PROF_NKINDS) LTX_FAIL(LTX_ERR_ARG, "ltx_prof_report: bad kind");
    std::lock_guard<std::mutex> lk(g_mu);
    LTX_TRY(fold());
    if (total_ms) *total_ms = g_ms[kind];
    if (total_work) *total_work = g_work[kind];
    if (count) *count = g_cnt[kind];
    return LTX_OK;
}
