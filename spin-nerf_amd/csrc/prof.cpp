#include "prof.h"

#include <mutex>
#include <vector>

#include "../../include/spinnerf_hip.h"

namespace snr {
namespace {
struct Rec { int id; hipEvent_t a, b; };
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open_a = nullptr;
int g_open_id = -1;
const char* kNames[K_COUNT] = {"mlp_pack", "mlp_fwd", "mlp_dgrad", "mlp_wgrad", "mlp_wgrad_reduce", "sample_coarse",
                               "composite_fwd", "composite_bwd", "sample_fine", "make_rays", "adam", "hg_pack", "hg_fwd",
                               "hg_bwd", "mlp_wgrad_pair", "composite_train", "composite_train_reg", "composite_train_sample",
                               "pack_rays_sample", "pack_rays", "adam_pack", "wgrad_post"};
hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void prof_begin(int id, hipStream_t s) {
  if (!g_on) return;
  std::lock_guard<std::mutex> l(g_mu);
  g_open_a = get_event();
  g_open_id = id;
  (void)hipEventRecord(g_open_a, s);
}
void prof_end(hipStream_t s) {
  if (!g_on || g_open_id < 0) return;
  std::lock_guard<std::mutex> l(g_mu);
  hipEvent_t b = get_event();
  (void)hipEventRecord(b, s);
  g_recs.push_back(Rec{g_open_id, g_open_a, b});
  g_open_id = -1;
}
}  // namespace snr

using namespace snr;

extern "C" int snr_prof_enable(int on) {
  std::lock_guard<std::mutex> l(g_mu);
  g_on = on != 0;
  return SNR_OK;
}
extern "C" int snr_prof_kernel_count(void) { return K_COUNT; }
extern "C" const char* snr_prof_kernel_name(int id) { return id >= 0 && id < K_COUNT ? kNames[id] : "?"; }

// Sums the elapsed time of every recorded launch per kernel id (waits for the events), then clears.
extern "C" int snr_prof_read(double* total_ms, int64_t* launches) {
  if (!total_ms || !launches) return SNR_ERR_NULL;
  std::lock_guard<std::mutex> l(g_mu);
  for (int i = 0; i < K_COUNT; ++i) { total_ms[i] = 0; launches[i] = 0; }
  for (const Rec& r : g_recs) {
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return (int)e;
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e != hipSuccess) return (int)e;
    total_ms[r.id] += ms;
    launches[r.id] += 1;
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return SNR_OK;
}
