// mfma_probe.hip -> tools/probe/libdfn_probe.so — a bench-only helper library, NOT part of libdfnet_hip.so (bench.py and
// tools/gpu_power.sh load it by path): the dense f16 MFMA rate this GPU SUSTAINS (power management
// included) for a loop of nothing but independent v_mfma_f32_32x32x16_f16, with all-zero operands and with operands that toggle
// like real data.  bench.py reports it next to roofline.peak: on MI355X the nominal 2.5 PFLOP/s holds for zero operands only —
// with random operands the clock settles near 1.6 GHz (tools/ubench/mfma_power.hip is the stand-alone form, fp32 MFMA included).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>
#include <cstdio>

namespace dfn {
namespace {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_probe_kernel(const float* in, float* out, int iters) {
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (_Float16)in[(threadIdx.x * 8 + i * 4096 + j) & 32767];
      b[i][j] = (_Float16)in[(threadIdx.x * 8 + i * 4096 + j + 16384) & 32767];
    }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x16{0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(n + r) & 3], b[n], acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
}  // namespace
}  // namespace dfn

namespace {
thread_local char g_probe_err[256] = "";
int fail(int code, const char* what, hipError_t e = hipSuccess) {
  snprintf(g_probe_err, sizeof(g_probe_err), "%s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
  return code;
}
constexpr int DFN_OK = 0, DFN_ERR_ARG = -1, DFN_ERR_HIP = -2;
}  // namespace

// Text of the last failure of this library on the calling thread.
extern "C" const char* dfn_probe_last_error(void) { return g_probe_err; }

// Dense-f16 MFMA rate in TFLOP/s over `seconds` (0 < seconds <= 5) of back-to-back launches on `stream`; random_operands: operands
// that toggle like real data (0: all zero).  Assumes one 512-thread block per CU (grid = CU count, 8 resident waves each).
// Returns 0, -1 (bad argument) or -2 (a HIP call failed); dfn_probe_last_error() has the text.
extern "C" int dfn_probe_mfma_rate(int random_operands, double seconds, double* tflops, void* stream) {
  using namespace dfn;
  if (!tflops || !(seconds > 0.0) || seconds > 5.0) return fail(DFN_ERR_ARG, "dfn_probe_mfma_rate: need tflops != NULL and 0 < seconds <= 5");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  std::vector<float> host(32768, 0.f);
  if (random_operands) {
    uint32_t x = 12345u;
    for (float& v : host) { x = x * 1664525u + 1013904223u; v = float(x >> 8) * (2.f / 16777216.f) - 1.f; }   // uniform [-1, 1)
  }
  float *in = nullptr, *out = nullptr;
  if (hipMalloc(&in, host.size() * 4) != hipSuccess) return fail(DFN_ERR_HIP, "dfn_probe_mfma_rate: hipMalloc");
  if (hipMalloc(&out, size_t(cus) * 512 * 4) != hipSuccess) { (void)hipFree(in); return fail(DFN_ERR_HIP, "dfn_probe_mfma_rate: hipMalloc"); }
  hipEvent_t e0, e1;
  int rc = DFN_OK;
  if (hipMemcpy(in, host.data(), host.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    (void)hipFree(in); (void)hipFree(out);
    return fail(DFN_ERR_HIP, "dfn_probe_mfma_rate: upload / event creation failed");
  }
  // 16 MFMAs per iteration and wave, 8 waves per CU; one launch ~ 0.1 s at the nominal rate, the clock settles within the first
  const int iters = 300000;
  const int launches = seconds < 0.2 ? 2 : int(seconds / 0.1 + 0.5);
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(cus), dim3(512), 0, s, in, out, iters);   // settle
  (void)hipEventRecord(e0, s);
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(mfma_probe_kernel, dim3(cus), dim3(512), 0, s, in, out, iters);
  const hipError_t le = hipGetLastError();
  (void)hipEventRecord(e1, s);
  float ms = 0.f;
  if (le != hipSuccess) rc = fail(DFN_ERR_HIP, "dfn_probe_mfma_rate: kernel launch", le);
  else if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) rc = fail(DFN_ERR_HIP, "dfn_probe_mfma_rate: timing the launches failed");
  else *tflops = double(launches) * iters * 16.0 * 8.0 * cus * (2.0 * 32 * 32 * 16) / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(in); (void)hipFree(out);
  return rc;
}
