// prof.hip — optional per-kernel timing with HIP events (bench.py's live roofline measurement).
#include "common.hpp"
#include <cstring>
#include <map>
#include <string>
#include <mutex>
#include <vector>

namespace mdh {

static int g_prof_on = 0; // 0 off, 1 every range, 2 the k_neighbor range only (an event pair costs ~8 us of stream time per range)
static std::mutex g_prof_mu;
struct Rec { const char *name; hipEvent_t a, b; };
static std::vector<Rec> g_recs;
static std::vector<hipEvent_t> g_free;

bool prof_enabled() { return g_prof_on != 0; }

static hipEvent_t get_event()
{
    if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

ProfRange::ProfRange(const char *name, hipStream_t st) : name_(name), st_(st), a_(nullptr)
{
    if (!g_prof_on || (g_prof_on == 2 && std::strcmp(name, "k_neighbor") != 0)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    a_ = get_event();
    if (a_) (void)hipEventRecord(a_, st_);
}

ProfRange::~ProfRange()
{
    if (!a_) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t b = get_event();
    if (b) (void)hipEventRecord(b, st_);
    g_recs.push_back(Rec{name_, a_, b});
}

} // namespace mdh

using namespace mdh;

extern "C" {

int mdh_prof_enable(int on)
{
    g_prof_on = on == 2 ? 2 : (on != 0 ? 1 : 0);
    return MDH_OK;
}

int mdh_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_recs) { if (r.a) g_free.push_back(r.a); if (r.b) g_free.push_back(r.b); }
    g_recs.clear();
    return MDH_OK;
}

int mdh_prof_report(char *buf, int buflen)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::map<std::string, std::pair<int, double>> acc;
    for (auto &r : g_recs) {
        if (!r.a || !r.b) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        auto &e = acc[r.name];
        e.first += 1;
        e.second += (double)ms;
    }
    std::string out;
    for (auto &kv : acc)
        out += kv.first + " " + std::to_string(kv.second.first) + " " + std::to_string(kv.second.second) + "\n";
    if ((int)out.size() + 1 > buflen) { set_error("mdh_prof_report: buffer too small"); return MDH_ERR_ARG; }
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}
}

MDH_WARM_UNIT(prof)
