// Host-only probe: the direct device layer opened, read and closed; between the steps the allocator walks its free
// lists (malloc_trim consolidates them), so a heap that one step damaged fails at the check behind that step.
// usage: smi_loop [rounds] [status-reads per round]
#include <malloc.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gpumon.h"

static void churn(const char* where) {
  fprintf(stderr, "check after %s\n", where);
  std::vector<void*> v;
  for (int i = 0; i < 2000; i++) v.push_back(malloc(16 + (i * 37) % 3000));
  for (size_t i = 0; i < v.size(); i += 2) free(v[i]);
  for (size_t i = 1; i < v.size(); i += 2) v[i] = realloc(v[i], 5000);
  for (size_t i = 1; i < v.size(); i += 2) free(v[i]);
  malloc_trim(0);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1, reads = argc > 2 ? atoi(argv[2]) : 1;
  for (int i = 0; i < n; i++) {
    gm_status_t st = gm_smi_init("amdsmi", nullptr);
    if (st != GM_ST_OK) {
      fprintf(stderr, "init -> %d\n", (int)st);
      return 2;
    }
    churn("init");
    for (int r = 0; r < reads; r++) {
      gm_device_status_t d;
      memset(&d, 0, sizeof d);
      d.version = GM_DEVICE_STATUS_VERSION;
      st = gm_smi_device_status(0, &d);
      if (st != GM_ST_OK) fprintf(stderr, "status -> %d\n", (int)st);
    }
    churn("device_status");
    gm_smi_shutdown();
    churn("shutdown");
  }
  printf("%d rounds done\n", n);
  return 0;
}
