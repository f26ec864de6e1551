// Which CUs does a CU-masked stream run on?  hipcc --offload-arch=gfx950 tools/cu_mask_probe.hip -o tools/cu_mask_probe
// Prints, for a few masks, the set of (XCC_ID, SE, SH/CU) ids the workgroups of a 1024-block launch landed on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <map>
#include <vector>
#include <cstring>

__global__ void probe(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));      // HW_REG_XCC_ID
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));        // HW_REG_HW_ID
    out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw;
  }
  // stay resident long enough that the blocks spread over every allowed CU
  long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 200000) { }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
  const int nb = 1024;
  unsigned* d; hipMalloc(&d, nb * 2 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(nb), dim3(1024), 65536, s, d);       // 64 KB LDS + 1024 threads: at most 2 blocks per CU
  hipStreamSynchronize(s);
  std::vector<unsigned> h(nb * 2);
  hipMemcpy(h.data(), d, nb * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per;
  for (int b = 0; b < nb; ++b) {
    unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
    per[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  int total = 0;
  printf("%s:", name);
  for (auto& kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  total CUs %d\n", total);
  if (total <= 16) {
    for (auto& kv : per) { printf("    xcc%u:", kv.first); for (unsigned v : kv.second) printf(" se%u.sh%u.cu%u", v >> 8, (v >> 4) & 1, v & 15); printf("\n"); }
  }
  hipFree(d); hipStreamDestroy(s);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d\n", p.multiProcessorCount);
  std::vector<uint32_t> all(8, 0xffffffffu);
  run("all 256", all);
  std::vector<uint32_t> m(8, 0);
  m[0] = 0xff; run("bits 0-7", m);
  m.assign(8, 0); m[0] = 0x1; run("bit 0", m);
  m.assign(8, 0); m[0] = 0x3; run("bits 0-1", m);
  m.assign(8, 0); m[0] = 0x101; run("bits 0,8", m);
  m.assign(8, 0); m[7] = 0xff000000u; run("bits 248-255", m);
  m.assign(8, 0xffffffffu); m[0] = 0xffffff00u; run("all but 0-7", m);
  m.assign(8, 0xffffffffu); m[0] = 0xfffffffcu; run("all but 0-1", m);
  return 0;
}
