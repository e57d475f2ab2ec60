// api_core.cpp -- error text, ABI version and the hipEvent kernel-class timers bench.py reads (C ABI: include/rgfm.h).
#include "rgfm_host.h"

extern "C" const char* rgfm_last_error(void) { return g_err.c_str(); }
extern "C" int rgfm_abi_version(void) { return RGFM_ABI_VERSION; }

extern "C" int rgfm_profile_enable(int enable) {
  g_prof.on = enable != 0;
  return RGFM_OK;
}
extern "C" int rgfm_profile_reset(void) {
  int rc = prof_collect();
  for (int k = 0; k < RGFM_KCLASS_COUNT; ++k) {
    g_prof.flops[k] = g_prof.sum_ms[k] = 0, g_prof.launches[k] = 0;
    g_prof.iv[k].clear();
  }
  g_prof.have_base = false;
  return rc;
}
extern "C" int rgfm_profile_read(int kclass, double* busy_ms, double* sum_ms, int64_t* launches, double* flops) {
  if (kclass < 0 || kclass >= RGFM_KCLASS_COUNT) return fail(RGFM_EINVAL, "bad kernel class %d", kclass);
  int rc = prof_collect();
  if (rc) return rc;
  if (busy_ms) *busy_ms = prof_union_ms(kclass);
  if (sum_ms) *sum_ms = g_prof.sum_ms[kclass];
  if (launches) *launches = g_prof.launches[kclass];
  if (flops) *flops = g_prof.flops[kclass];
  return RGFM_OK;
}

extern "C" int rgfm_profile_reserve(int64_t launches) {
  if (launches < 0) return fail(RGFM_EINVAL, "negative launch count");
  const size_t want = (size_t)launches * 2;
  while (g_prof.ev.size() < want) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    g_prof.ev.push_back(e);
  }
  g_prof.cls.resize(g_prof.ev.size() / 2);
  if (!g_prof.base) HIP_TRY(hipEventCreate(&g_prof.base));
  return RGFM_OK;
}
