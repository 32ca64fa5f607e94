#include "lo_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
static thread_local char g_err[512] = "";
void lo_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
const char* lo_get_error() { return g_err; }
int lo_check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return LO_OK;
  lo_set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return LO_ERR_HIP;
}

// ---- HIP-event profiler -------------------------------------------------------------------------
#include <string>
#include <vector>
bool g_lo_prof_on = false;
thread_local hipEvent_t g_lo_stop_event = nullptr;   // see LO_LAUNCH_STOP (lo_common.h)
bool g_lo_prof_layers = false;     // per-layer record names (lo_prof_enable(2), or LO_PROF_LAYERS in the environment)
const char* g_lo_prof_tag = nullptr;
namespace {
struct Rec { const char* name; double flops, bytes; hipEvent_t e0, e1; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_used = 0;
hipEvent_t take_event() {
  if (g_pool_used == g_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_pool.push_back(e);
  }
  return g_pool[g_pool_used++];
}
}  // namespace
void lo_prof_begin(const char* name, double flops, double bytes, hipStream_t st) {
  Rec r{name, flops, bytes, take_event(), take_event()};
  if (r.e0) (void)hipEventRecord(r.e0, st);
  g_recs.push_back(r);
}
void lo_prof_end(hipStream_t st) {
  if (!g_recs.empty() && g_recs.back().e1) (void)hipEventRecord(g_recs.back().e1, st);
}
// Under LO_PROF_LAYERS a conv-like launch is named "<kernel> <Cin>><Cout> <Hin>><Hout> T<taps>": one interned string per distinct
// (kernel, geometry), since the records keep the pointer.  Without it (or with the profiler off) the kernel name itself.
const char* lo_prof_geom_name(const char* base, const LoGeom& g) {
  if (!g_lo_prof_layers || !g_lo_prof_on) return base;
  static std::vector<std::string*> table;
  char text[96];
  snprintf(text, sizeof(text), "%s %d>%d %d>%d T%d", base, g.Cin, g.Cout, g.Hin, g.Hout, g.T[0] * (g.n_phase > 1 ? -g.n_phase : 1));
  for (std::string* e : table)
    if (*e == text) return e->c_str();
  table.push_back(new std::string(text));
  return table.back()->c_str();
}
// a stable copy of a launcher's formatted kernel name (the records keep the pointer, a static buffer would be overwritten)
const char* lo_prof_intern(const char* text) {
  if (!g_lo_prof_on) return text;
  static std::vector<std::string*> table;
  for (std::string* e : table)
    if (*e == text) return e->c_str();
  table.push_back(new std::string(text));
  return table.back()->c_str();
}
extern "C" void lo_prof_enable(int on) {
  g_lo_prof_on = on != 0;
  g_lo_prof_layers = on == 2 || getenv("LO_PROF_LAYERS") != nullptr;
  if (on) { g_recs.clear(); g_pool_used = 0; }
}
extern "C" int lo_prof_count(void) { return (int)g_recs.size(); }
// call after the stream has been synchronised
extern "C" int lo_prof_get(int i, char* name, int name_cap, double* ms, double* flops, double* bytes) {
  if (i < 0 || i >= (int)g_recs.size()) return LO_ERR_ARG;
  const Rec& r = g_recs[i];
  float t = 0.f;
  if (!r.e0 || !r.e1 || hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) t = -1.f;
  snprintf(name, name_cap, "%s", r.name);
  *ms = t; *flops = r.flops; *bytes = r.bytes;
  return LO_OK;
}
