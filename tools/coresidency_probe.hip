// Can small kernels start while a one-workgroup-per-CU kernel is resident?  hipcc --offload-arch=gfx950 -O2 tools/coresidency_probe.hip
// Stream 1: `big` = G workgroups x 512 threads x 110 KB LDS spinning ~1.5 ms.  Stream 2 (launched 100 us later): a 1-workgroup
// 1024-thread 88 KB kernel and a 100-workgroup 256-thread kernel; their latency says whether they waited for `big` to drain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <chrono>
#include <thread>

__global__ void big(unsigned* xcc, long long cycles, int skipmode) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xf;
  // skipmode 1: three workgroups of the XCD that serves workgroup 0 leave at once; 2: additionally one per XCD
  const int b = blockIdx.x;
  if (skipmode >= 1 && (b == 232 || b == 240 || b == 248)) return;
  if (skipmode == 2 && b >= 248) return;
  long long t0 = __builtin_readcyclecounter();
  double a = threadIdx.x;
  while (__builtin_readcyclecounter() - t0 < cycles) { a = a * 1.0000001 + 1e-9; lds[threadIdx.x] = a; }
  if (a == 12345.0) xcc[0] = 0;
}
__global__ void small_one(double* out) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) out[1] = (double)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xf);
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 20000) { }
  if (threadIdx.x == 0) out[0] = lds[5];
}
__global__ void small_many(double* out) {
  long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 2000) { }
  if (threadIdx.x == 0) out[blockIdx.x] = 1.0;
}

int main() {
  hipStream_t s1, s2, s3;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
  unsigned* xcc; double* out;
  hipMalloc(&xcc, 1024 * sizeof(unsigned)); hipMalloc(&out, 1024 * sizeof(double));
  hipFuncSetAttribute((const void*)big, hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024);
  hipFuncSetAttribute((const void*)small_one, hipFuncAttributeMaxDynamicSharedMemorySize, 88 * 1024);
  hipEvent_t e[8];
  for (auto& x : e) hipEventCreate(&x);
  const int grids[] = {256, 255, 248, 240, 224};
  for (int rep = 0; rep < 2; ++rep)
  for (int G : grids) {
    hipDeviceSynchronize();
    hipEventRecord(e[0], s1);
    hipLaunchKernelGGL(big, dim3(G), dim3(512), 110 * 1024, s1, xcc, 3000000LL, 0);      // ~1.4 ms at 2.1 GHz
    hipEventRecord(e[1], s1);
    std::this_thread::sleep_for(std::chrono::microseconds(200));
    hipEventRecord(e[2], s2);
    hipLaunchKernelGGL(small_one, dim3(1), dim3(1024), 88 * 1024, s2, out);
    hipEventRecord(e[3], s2);
    hipLaunchKernelGGL(small_one, dim3(1), dim3(1024), 88 * 1024, s2, out);
    hipEventRecord(e[4], s2);
    hipEventRecord(e[5], s3);
    hipLaunchKernelGGL(small_many, dim3(100), dim3(256), 0, s3, out);
    hipEventRecord(e[6], s3);
    hipDeviceSynchronize();
    float tb, t1, t2, t3, o2, o3;
    hipEventElapsedTime(&tb, e[0], e[1]); hipEventElapsedTime(&t1, e[2], e[3]); hipEventElapsedTime(&t2, e[3], e[4]);
    hipEventElapsedTime(&t3, e[5], e[6]); hipEventElapsedTime(&o2, e[0], e[2]); hipEventElapsedTime(&o3, e[0], e[5]);
    std::vector<unsigned> h(G);
    hipMemcpy(h.data(), xcc, G * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, int> cnt;
    for (int b = 0; b < G; ++b) cnt[h[b]]++;
    printf("G=%3d big %.0f us | one-wg kernels (start +%.0f us): %.0f, %.0f us | 100-wg kernel (start +%.0f us): %.0f us | per XCD:", G, tb * 1e3, o2 * 1e3,
           t1 * 1e3, t2 * 1e3, o3 * 1e3, t3 * 1e3);
    for (auto& kv : cnt) printf(" %d", kv.second);
    printf("\n");
  }
  // ---- three concurrent one-workgroup kernels (three streams) + a 100-workgroup kernel against a 256-workgroup launch
  //      in which some workgroups leave at once
  hipStream_t s4; hipStreamCreateWithFlags(&s4, hipStreamNonBlocking);
  double* out2; hipMalloc(&out2, 1024 * sizeof(double));
  double* out3; hipMalloc(&out3, 1024 * sizeof(double));
  for (int rep = 0; rep < 2; ++rep)
  for (int mode = 0; mode <= 2; ++mode) {
    hipDeviceSynchronize();
    hipEventRecord(e[0], s1);
    hipLaunchKernelGGL(big, dim3(256), dim3(512), 110 * 1024, s1, xcc, 3000000LL, mode);
    std::this_thread::sleep_for(std::chrono::microseconds(200));
    hipEventRecord(e[1], s2); hipLaunchKernelGGL(small_one, dim3(1), dim3(1024), 88 * 1024, s2, out); hipEventRecord(e[2], s2);
    hipEventRecord(e[3], s3); hipLaunchKernelGGL(small_one, dim3(1), dim3(1024), 88 * 1024, s3, out2); hipEventRecord(e[4], s3);
    hipEventRecord(e[5], s4); hipLaunchKernelGGL(small_one, dim3(1), dim3(1024), 88 * 1024, s4, out3);
    hipLaunchKernelGGL(small_many, dim3(100), dim3(256), 0, s4, out3 + 8); hipEventRecord(e[6], s4);
    hipDeviceSynchronize();
    float t1, t2, t3;
    hipEventElapsedTime(&t1, e[1], e[2]); hipEventElapsedTime(&t2, e[3], e[4]); hipEventElapsedTime(&t3, e[5], e[6]);
    double x1[2], x2[2], x3[2]; unsigned b0;
    hipMemcpy(x1, out, 16, hipMemcpyDeviceToHost); hipMemcpy(x2, out2, 16, hipMemcpyDeviceToHost); hipMemcpy(x3, out3, 16, hipMemcpyDeviceToHost);
    hipMemcpy(&b0, xcc, 4, hipMemcpyDeviceToHost);
    printf("skipmode %d: three one-wg kernels %.0f, %.0f, %.0f(+100-wg) us on XCC %d %d %d; big's workgroup 0 on XCC %u\n", mode, t1 * 1e3, t2 * 1e3,
           t3 * 1e3, (int)x1[1], (int)x2[1], (int)x3[1], b0);
  }
  return 0;
}
