// Per-launch timing with HIP events on the launch stream (used by bench.py to price the dominant kernel live, in the
// same process and on the same shapes as the timed run).  Disabled by default; never active during graph capture.
#include "cd_common.h"

#include <cstdio>
#include <map>
#include <vector>

namespace cd {
namespace prof {

namespace {
struct Record {
  std::string cat;
  hipEvent_t a = nullptr, b = nullptr;
  double flops = 0, bytes = 0;
};
bool g_enabled = false;
std::vector<Record> g_records;
}  // namespace

bool enabled() { return g_enabled; }

void begin() {
  for (auto& r : g_records) {
    if (r.a) hipEventDestroy(r.a);
    if (r.b) hipEventDestroy(r.b);
  }
  g_records.clear();
  g_enabled = true;
}

Scope::Scope(const char* category, hipStream_t s, double flops, double bytes) : idx(-1), stream(s) {
  if (!g_enabled) return;
  Record r;
  r.cat = category;
  r.flops = flops;
  r.bytes = bytes;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  hipEventRecord(r.a, s);
  g_records.push_back(r);
  idx = (int)g_records.size() - 1;
}

Scope::~Scope() {
  if (idx >= 0) hipEventRecord(g_records[idx].b, stream);
}

int end(char* buf, int cap) {
  g_enabled = false;
  hipDeviceSynchronize();
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0, flops_total = 0, bytes_total = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : g_records) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      Agg& a = agg[r.cat];
      a.n += 1; a.ms += ms; a.flops = r.flops; a.bytes = r.bytes;
      a.flops_total += r.flops; a.bytes_total += r.bytes;  // (a category may mix shapes: the GroupNorm backward of every level)
    }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  g_records.clear();
  std::string out = "{";
  bool first = true;
  for (auto& kv : agg) {
    char line[512];
    std::snprintf(line, sizeof line,
                  "%s\"%s\": {\"launches\": %ld, \"ms\": %.6f, \"flops\": %.0f, \"bytes\": %.0f, \"flops_total\": %.0f, \"bytes_total\": %.0f}",
                  first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes, kv.second.flops_total,
                  kv.second.bytes_total);
    out += line;
    first = false;
  }
  out += "}";
  if ((int)out.size() + 1 > cap) return -1;
  std::snprintf(buf, cap, "%s", out.c_str());
  return (int)out.size();
}

}  // namespace prof
}  // namespace cd
