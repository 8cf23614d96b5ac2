// handoff.hip - what a dependent phase costs on MI355X: as a launch of its own in a hipGraph chain, or as a phase of ONE launch whose blocks
// hand over through write-through (sc1) stores, an agent-scope arrival counter and cache-bypassing (sc1) loads.
// Every phase: NB blocks x 256 threads; block b reads the WHOLE previous vector (V floats), adds its block id, writes its V / NB slice.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/handoff.hip -o tools/micro/handoff && tools/micro/handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int V = 2560, NT = 256;

__device__ __forceinline__ float ldb(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stw(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float phase_body(const float* in, float* out, int b, int nb, bool coherent) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < V; i += NT) s += coherent ? ldb(in + i) : in[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  const int per = V / nb;
  for (int i = threadIdx.x; i < per; i += NT) {
    const float v = s * 1e-4f + (float)(b * per + i);
    if (coherent) stw(out + b * per + i, v); else out[b * per + i] = v;
  }
  __syncthreads();
  return s;
}

__global__ void one_phase(const float* in, float* out, int nb) { phase_body(in, out, blockIdx.x, nb, false); }

// P phases in one launch; buf[p] is phase p's output; cnt[p] its arrivals (zero before the launch; reset by block 0 at the end)
__global__ void chained(float* bufs, unsigned* cnt, int P, int nb) {
  const int b = blockIdx.x;
  for (int p = 0; p < P; ++p) {
    if (p > 0) {
      if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt + (p - 1) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nb && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
    }
    phase_body(bufs + (size_t)p * V, bufs + (size_t)(p + 1) * V, b, nb, p > 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt + p * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (b == 0) {
    if (threadIdx.x == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(cnt + (P - 1) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nb && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
      for (int p = 0; p < P; ++p) __hip_atomic_store(cnt + p * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main() {
  float* bufs; unsigned* cnt;
  const int PMAX = 8;
  CK(hipMalloc(&bufs, (size_t)(PMAX + 1) * V * 4));
  CK(hipMalloc(&cnt, 32 * PMAX * 4));
  CK(hipMemset(bufs, 0, (size_t)(PMAX + 1) * V * 4));
  CK(hipMemset(cnt, 0, 32 * PMAX * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nb : {8, 32, 128}) {
    for (int P : {1, 2, 5}) {
      // (a) a hipGraph chain of P launches
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int p = 0; p < P; ++p) hipLaunchKernelGGL(one_phase, dim3(nb), dim3(NT), 0, st, bufs + (size_t)p * V, bufs + (size_t)(p + 1) * V, nb);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e0, st));
      const int REP = 200;
      for (int i = 0; i < REP; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms_a; CK(hipEventElapsedTime(&ms_a, e0, e1));
      // (b) ONE launch with P phases, also replayed as a graph
      hipGraph_t g2; hipGraphExec_t ge2;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      hipLaunchKernelGGL(chained, dim3(nb), dim3(NT), 0, st, bufs, cnt, P, nb);
      CK(hipStreamEndCapture(st, &g2));
      CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
      for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge2, st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < REP; ++i) CK(hipGraphLaunch(ge2, st));
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms_b; CK(hipEventElapsedTime(&ms_b, e0, e1));
      std::vector<float> h(V);
      CK(hipMemcpy(h.data(), bufs + (size_t)P * V, V * 4, hipMemcpyDeviceToHost));
      printf("blocks %3d phases %d : %d launches %.2f us | one launch %.2f us   (check %.3f)\n", nb, P, P, ms_a * 1e3 / REP, ms_b * 1e3 / REP, h[V - 1]);
    }
  }
  // a 5-phase chain where every graph replay is one step of a longer chain (back-to-back graph launches hide nothing: each waits for the last)
  return 0;
}
